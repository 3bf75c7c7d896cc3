#!/bin/bash
# GPU box: the -m gpu suite on the split RANSAC path, then bench A/B (RGBDFE_RANSAC_SPLIT=1|0) at both depth-noise levels.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04_ab}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for noise in 0.01 0.002; do
for split in 1 0; do
  RGBDFE_RANSAC_SPLIT=$split timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --depth-noise $noise > $O/bench_n${noise}_s$split.json 2> $O/bench_n${noise}_s$split.err
  echo "noise $noise split $split rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_n${noise}_s$split.json').read().strip().splitlines()[-1])
    print('  pairs/s %.0f  ms/step %.4f  serial %s  parity %s' % (d['value'], d['ms_per_step'], d['timing'].get('serial_stage_ms'), d.get('parity_check')))
except Exception as e:
    print('  no line', e); print(open('$O/bench_n${noise}_s$split.err').read()[-1500:])
PY
done
done
