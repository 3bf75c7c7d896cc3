cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 bash tools/profile_round.sh r05 > gpurun_out/profile_r05.log 2>&1
tail -3 gpurun_out/profile_r05.log
timeout 400 python bench.py > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err
tail -c 600 gpurun_out/r05_bench_full.json
