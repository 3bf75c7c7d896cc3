#!/bin/bash
# GPU box: the whole GPU suite (no -x: every failure is wanted), the bench line, the thread stress, the SIFT end-to-end report
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputests2.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r04_gputests2.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err; echo "bench rc $?"; tail -3 gpurun_out/r04_bench2.err
timeout 600 python tools/stress_threads.py 50 > gpurun_out/r04_stress_threads.log 2>&1; echo "stress rc $?"; tail -4 gpurun_out/r04_stress_threads.log
timeout 300 python tools/sift_e2e.py > gpurun_out/r04_sift_e2e.json 2> gpurun_out/r04_sift_e2e.err; echo "e2e rc $?"
