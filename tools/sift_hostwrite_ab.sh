#!/bin/bash
# GPU box: the SIFT extraction batch entry point with the batch's read-backs written by the kernels themselves (default) against
# the copies behind them (RGBDFE_SIFT_HOSTWRITE=0), alternating in one call; then the kernel trace of either.
cd $GRAFT_REPO_ROOT; R=$PWD; O=$R/gpurun_out/sift_hostwrite; mkdir -p $O; export TMPDIR=/tmp
for hw in 1 0; do RGBDFE_SIFT_HOSTWRITE=$hw timeout 600 python -m pytest tests/test_gpu_sift_extract.py tests/test_gpu_sift_e2e.py -x -q 2>&1 | tail -1 | sed "s/^/hostwrite=$hw tests: /"; done
for rep in 1 2 3 4; do
  for hw in 1 0; do
    echo -n "hostwrite=$hw rep $rep: "
    RGBDFE_SIFT_HOSTWRITE=$hw timeout 300 python tools/bench_sift_batch.py 640 480 32 9 2>/dev/null | tail -1 | cut -c1-300
  done
done
for hw in 1 0; do
  cd /tmp; rm -rf $O/t$hw
  RGBDFE_SIFT_HOSTWRITE=$hw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$hw -o trace -- python $R/tools/detect_workload.py sift_batch 640 480 0 32 4 > /dev/null 2>&1
  cd $R
  python - $O/t$hw/trace_kernel_stats.csv $hw <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("hostwrite=%s kernels us/frame: total %.1f" % (sys.argv[2], tot / 128e3))
for r in rows[:8]: print("   %-60s calls %4s  us/frame %.2f" % (r["Name"].split("::")[-1][:60], r["Calls"], float(r["TotalDurationNs"]) / 128e3))
PY
  find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -delete
done
