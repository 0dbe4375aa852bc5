#!/bin/bash
# Re-creates pdanet_amd/tuning/tunableop_mi355x_once16k_b2.csv on an MI355X: PyTorch TunableOp benchmarks every
# GEMM shape of one training step once (about 12 minutes) and records the fastest hipBLASLt / rocBLAS solution.
# bench.py only LOADS the committed file (tuning off).  Usage: tools/tune_gemms.sh [workload] [extra bench args]
set -e
cd "$(dirname "$0")/.."
WL=${1:-backbone}; shift || true
OUT=${PDA_TUNE_OUT:-gpurun_out/tunableop_results.csv}
mkdir -p "$(dirname "$OUT")"
export PDA_NO_TUNED_GEMMS=1 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$PWD/$OUT" \
       PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_VERBOSE=0
( python bench.py --steps 3 --warmup 1 --workload "$WL" --no-cpu-baseline "$@" > "${OUT%.csv}.log" 2>&1; echo "tuning run exit $?" ) &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 50; echo "tuning..."; done     # keeps a watchdog-ed runner informed
ls -la "${OUT%.csv}"*.csv
echo "copy the result (suffix 0 = device 0) over pdanet_amd/tuning/tunableop_mi355x_once16k_b2.csv to adopt it"
