#!/bin/bash
# GPU box: one-pass SIFT kernel variants, ORB detect after the score-plane change, thread stress
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/sweep_sift_onepass.sh run > gpurun_out/r04_sift_sweep.log 2>&1; cat gpurun_out/r04_sift_sweep.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_sift_e2e.py -m gpu -q -p no:cacheprovider --timeout 300 > gpurun_out/r04_gputests4.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r04_gputests4.log
timeout 500 python tools/stress_threads.py 10 > gpurun_out/r04_stress_threads.log 2>&1; echo "stress rc $?"; tail -6 gpurun_out/r04_stress_threads.log
