#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_async.py -m gpu -q -p no:cacheprovider --timeout 300 > gpurun_out/r04_gputests12.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r04_gputests12.log
{
for which in single group; do
  echo "=== library defaults, $which"
  timeout 100 python tools/r04_hang_probe.py 40 $which 2>&1 | grep -v "amdgpu.ids" | tail -3
done
} > gpurun_out/r04_default_probe.log 2>&1
cat gpurun_out/r04_default_probe.log
