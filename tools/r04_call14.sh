#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider --timeout 200 > gpurun_out/r04_gputests14.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_gputests14.log
timeout 600 bash tools/profile_r03.sh r04d detect_640x480_orb1000 detect_1280x960_orb4000 2>&1 | tail -3
