# GPU box: tests/test_gpu_orb.py with the fused pyramid kernel (the default), then a kernel trace of the 640x480 detect workload
# with RGBDFE_ORB_PYRAMID=fused and =levels (one call: boxes differ)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 250 python -m pytest tests/test_gpu_orb.py -q -x -p no:cacheprovider --timeout 200 > gpurun_out/pyr_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/pyr_tests.log
cd /tmp && export TMPDIR=/tmp
for m in fused levels; do
RGBDFE_ORB_PYRAMID=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pyr_trace_$m -o trace -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 56 3 > $GRAFT_REPO_ROOT/gpurun_out/pyr_trace_$m.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/pyr_trace_$m -name "*.db" -delete
echo "== $m"; head -6 $(find $GRAFT_REPO_ROOT/gpurun_out/pyr_trace_$m -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-230
done
