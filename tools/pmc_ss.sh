#!/bin/bash
# SQ counters of the narrow-SA training passes (csrc/sa_train_small.hip) on the isolated layer-0 bench.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ss
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 tools/sa_layer_bench.py 0 3 > $OUT.log 2>&1
f=$(find $OUT -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "ss_" not in k: continue
    k = k.split("(")[0].replace("void pda::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in sorted(agg.items()):
    n = max(cnt[k], 1)
    wc = c["SQ_WAVE_CYCLES"]
    print("%-28s launches %2d | wave quad-cycles %.3g | parked %.2f | issue-stall %.2f | active %.2f | VALU insts/wave-launch %.0f | LDS insts %.0f | MFMA busy cycles %.3g" % (
        k, n, wc / n, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_INSTS_VALU"] / n, c["SQ_INSTS_LDS"] / n, c["SQ_VALU_MFMA_BUSY_CYCLES"] / n))
PY
find $OUT -type f -delete
