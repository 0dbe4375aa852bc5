#!/bin/bash
# rocprofv3 kernel trace of one bench workload -> profiles/<tag>_<workload>_kernel_stats.csv (last 10 steps).
# Usage (on the MI355X box, from the repo root): tools/profile_step.sh <tag> [workload] [extra bench args]
set -e
TAG=$1; WL=${2:-detector_train}; shift; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_${WL}
rm -rf "$OUT"; mkdir -p "$OUT" profiles
rocprofv3 --kernel-trace --stats -d "$OUT" -- python3 bench.py --steps 10 --warmup 4 --workload "$WL" --no-cpu-baseline --no-extra "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
DB=$(find "$OUT" -name '*.db' | head -1)
python3 tools/rocpd_kernel_stats.py "$DB" 10 > "profiles/${TAG}_${WL}_kernel_stats.csv" 2> "$OUT/window.txt"
cat "$OUT/window.txt"; tail -c 400 "$OUT/bench.json" | head -c 400; echo
cp "profiles/${TAG}_${WL}_kernel_stats.csv" gpurun_out/
find "$OUT" -type f ! -name "bench.json" ! -name "bench.err" ! -name "window.txt" -delete   # the trace database is too big to travel back
