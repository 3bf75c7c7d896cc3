#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sift.py -x -q -m gpu > gpurun_out/sift_tests.log 2>&1; echo "sift tests rc=$?"
tail -3 gpurun_out/sift_tests.log
for f in ${FAST_LIST:-1}; do
RGBDFE_SIFT_FAST_KEYS=$f timeout 300 python bench.py --config sift --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'])"
done
