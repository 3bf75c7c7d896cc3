#!/bin/bash
# Runs on the GPU box (via gpurun): the whole -m gpu suite, smoke(), the bench line, and bench.py under
# torch.distributed.run with one rank.  Logs under gpurun_out/full_check/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/full_check; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_check/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'])
print('roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','traffic','avg_launch_ms')})
print('match', d.get('match_roofline')); print('issue', d.get('issue_roofline'))
print('sift', d.get('sift')); print('detect', d.get('detect')); print('cpu', d.get('cpu_baseline')); print('ref', d.get('cpu_baseline_reference_code'))
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2> $O/torchrun.err | tail -1 | cut -c1-200
