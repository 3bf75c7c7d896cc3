#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "0 0 single" "0 1 single" "0 0 group" "0 1 group"; do
  set -- $cfg
  echo "=== RGBDFE_GRAPHS=$1 RGBDFE_RANSAC_SPLIT=$2 $3"
  RGBDFE_GRAPHS=$1 RGBDFE_RANSAC_SPLIT=$2 timeout 120 python tools/r04_hang_probe.py 55 $3 2>&1 | grep -E "^ok|PROBE_DONE|Timeout|frontend.py|Error" | tail -12
done > gpurun_out/r04_hang_probe2.log 2>&1
cat gpurun_out/r04_hang_probe2.log
SIFT1_VARIANTS="base:-DRGBDFE_SIFT1_NV=11 burst:-DRGBDFE_SIFT1_BURST=1 bursttree:-DRGBDFE_SIFT1_BURST=1_-DRGBDFE_SIFT1_TREE=1" bash tools/sweep_sift_onepass.sh run > gpurun_out/r04_sift_sweep2.log 2>&1; cat gpurun_out/r04_sift_sweep2.log | cut -c1-260
