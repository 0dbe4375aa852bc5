#!/bin/bash
# End-of-round measurement batch on one MI355X box (repo root): bench lines, kernel-trace summaries, counter passes.
# Usage: tools/round_measure.sh <tag>        e.g. r03_z  ->  profiles/<tag>_*  (copies under gpurun_out/profiles/)
set -u
TAG=${1:-r03_z}
export TMPDIR=/tmp
mkdir -p profiles gpurun_out/profiles
run() { echo "== $*" >&2; "$@"; }
run python bench.py > profiles/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
run tools/profile_step.sh ${TAG} detector_train > gpurun_out/${TAG}_prof_train.log 2>&1
python3 tools/kernel_categories.py profiles/${TAG}_detector_train_kernel_stats.csv > profiles/${TAG}_detector_train_categories.txt
run tools/profile_step.sh ${TAG} backbone_infer > gpurun_out/${TAG}_prof_infer.log 2>&1
python3 tools/kernel_categories.py profiles/${TAG}_backbone_infer_kernel_stats.csv > profiles/${TAG}_backbone_infer_categories.txt
PDA_SPLIT_GEMM=0 run python bench.py --no-extra --no-cpu-baseline > profiles/${TAG}_bench_split_gemm_off.json 2>> gpurun_out/${TAG}_bench.err
run python bench.py --points 60000 --steps 10 --warmup 6 --no-extra --no-cpu-baseline > profiles/${TAG}_bench_once_60000pt.json 2>> gpurun_out/${TAG}_bench.err
run python bench.py --points 65536 --batch 8 --steps 10 --warmup 6 --no-extra --no-cpu-baseline > profiles/${TAG}_c5_detector_train_b8_65536.json 2>> gpurun_out/${TAG}_bench.err
run tools/pmc_passes.sh ${TAG%%_*} > gpurun_out/${TAG}_pmc.log 2>&1
tools/pmc_ss.sh > profiles/${TAG}_sa_small_sq_counters.txt 2>/dev/null
cp -r profiles/${TAG}_* gpurun_out/profiles/ 2>/dev/null
for f in profiles/${TAG}_bench.json profiles/${TAG}_bench_split_gemm_off.json profiles/${TAG}_bench_once_60000pt.json profiles/${TAG}_c5_detector_train_b8_65536.json; do
  python3 -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['ms_per_step'],3), round(d['value'],2), d['config'].get('graph_tail'))" || echo "$f: no line"
done
