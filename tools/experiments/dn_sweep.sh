#!/bin/bash
for b in 128 64 32 16; do
  export PDA_DN_UNIQUE_BLOCKS=$b
  bash tools/profile_step.sh r03_dn detector_train > gpurun_out/dn_prof.log 2>&1
  python3 tools/kernel_categories.py profiles/r03_dn_detector_train_kernel_stats.csv | grep densitynet | sed "s/^/blocks $b: /"
  grep densitynet profiles/r03_dn_detector_train_kernel_stats.csv | awk -F, '{printf "   %s %s %.1f\n", substr($1,1,40), $2, $4/1000}' | sort | head -12
done
