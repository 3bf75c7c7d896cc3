# GPU box: kernel trace of single super-frame calls (7 frames per call, one launch chain in flight: kernel durations without
# the overlap of the batch pipeline), fused pyramid vs one launch per level, one call
cd /tmp && export TMPDIR=/tmp
for m in fused levels; do
RGBDFE_ORB_PYRAMID=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m -o trace -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 7 24 > $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/serial_trace_$m -name "*.db" -delete
done
echo done
