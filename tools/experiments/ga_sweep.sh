#!/bin/bash
for cfg in "1 1 20" "1 1 24" "1 1 28" "1 1 36" "1 1 40"; do
  set -- $cfg
  export PDA_GA_ROWS=$1 PDA_GA_MINW=$2 PDA_GA_FLOOR=$3
  bash tools/layer_profiles.sh r03_ga 1 2 > /dev/null
  python3 - <<PY
import csv
tot=0; parts=[]
for L in (1,2):
    for r in csv.DictReader(open(f"gpurun_out/profiles/r03_ga_L{L}_kernel_stats.csv")):
        if "group_attention" in r["Name"]:
            t=float(r["AverageNs"])/1e3; tot+=t; parts.append("%s=%.0f"%(r["Name"][33:48].replace(" ",""),t))
print("$1 $2 $3", "total %.0f"%tot, " ".join(parts))
PY
done
