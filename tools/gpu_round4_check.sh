#!/bin/bash
# GPU box: the whole GPU suite, the driver's bench line, and the detect counter passes (kernel trace + FETCH / WRITE)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 120 -x > gpurun_out/r04_gputests_final.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r04_gputests_final.log
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc $?"; tail -2 gpurun_out/r04_bench_final.err
timeout 150 bash tools/profile_round.sh r04e detect_640x480_orb1000 detect_1280x960_orb4000
