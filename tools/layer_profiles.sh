#!/bin/bash
# Per-layer kernel-trace summaries (training form, one SA layer in isolation): tools/layer_profiles.sh <tag> [layers...]
set -u
TAG=${1:-r03_lp}; shift
LAYERS=${*:-1 2 5}
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles
for L in $LAYERS; do
  out=gpurun_out/lp_${TAG}_L$L
  rm -rf $out
  rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 tools/sa_layer_bench.py $L 20 > gpurun_out/lp_${TAG}_L$L.log 2>&1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f gpurun_out/profiles/${TAG}_L${L}_kernel_stats.csv
  grep "SA layer" gpurun_out/lp_${TAG}_L$L.log
done
