#!/bin/bash
# GPU box: GPU suite, the driver's bench line and the r04 counter passes of the pair path (orb + heavy)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04_gputests.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "bench rc $?"
timeout 900 bash tools/profile_r03.sh r04 orb heavy
