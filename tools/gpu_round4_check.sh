#!/bin/bash
# GPU box: the whole GPU suite, the driver's bench line, the r04 counter passes of the pair path and of the SIFT matcher
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 > gpurun_out/r04_gputests_final.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r04_gputests_final.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc $?"; tail -2 gpurun_out/r04_bench_final.err
timeout 600 bash tools/profile_r03.sh r04 orb sift
