# GPU box: tests/test_gpu_orb.py, then a kernel trace of single 7-frame calls (one chain in flight) with the fused pyramid
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_orb.py -q -x -p no:cacheprovider --timeout 150 > gpurun_out/pyr_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/pyr_tests.log
cd /tmp && export TMPDIR=/tmp
for m in fused; do
RGBDFE_ORB_PYRAMID=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m -o trace -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 7 24 > $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m -name "*.db" -delete
done
