#!/bin/bash
# HBM-traffic and MFMA-utilisation counter passes for the bench's roofline kernels (MI355X_MICROARCH.md, rocprofv3 PMC
# slots: FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are collected WITHOUT any other trace domain).
# Usage (MI355X box, repo root): tools/pmc_passes.sh r02      -> profiles/r02_pmc/{traffic,mfma_util}.json + raw rows
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
DST=profiles/${TAG}_pmc
rm -rf "$OUT"; mkdir -p "$OUT" "$DST"
run() {   # name, counters..., target
    local name=$1; shift; local target=${@: -1}; set -- "${@:1:$(($#-1))}"
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/pmc_targets.py "$target" 5 > "$OUT/$name.log" 2>&1
    local f=$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)
    grep -E '^"Correlation_Id"|pda::' "$f" > "$DST/$name.csv"
    echo "$name: $(wc -l < "$DST/$name.csv") rows"
}
for t in fps ball_query wgrad; do
    run fetch_size_$t FETCH_SIZE $t
    run write_size_$t WRITE_SIZE $t
done
for t in wgrad sa_mlp lin_cols lin_split gemm_split sa_small; do
    run mfma_busy_$t SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE $t
done
python3 tools/pmc_summarize.py "$DST"
rm -rf "$OUT"; mkdir -p gpurun_out/profiles; cp -r "$DST" gpurun_out/profiles/
