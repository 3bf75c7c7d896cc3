#!/bin/bash
# GPU box: the pair-path test modules with the recording stage forced either way (the split path for every batch: its
# fallback machinery sees small batches too; the one-kernel stage for every batch), smoke(), torchrun with one rank
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MODS="tests/test_gpu_pairs.py tests/test_gpu_async.py tests/test_gpu_multi.py tests/test_gpu_sift.py tests/test_gpu_flann.py tests/test_gpu_g2o.py tests/test_gpu_dist_two_ranks.py"
for M in 1 0; do
  RGBDFE_RANSAC_SPLIT=$M timeout 700 python -m pytest $MODS -m gpu -q -p no:cacheprovider --timeout 300 > gpurun_out/r04_gputests_split$M.log 2>&1; echo "RGBDFE_RANSAC_SPLIT=$M tests rc $?"; tail -3 gpurun_out/r04_gputests_split$M.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> gpurun_out/r04_torchrun.err | tail -1 | cut -c1-300
