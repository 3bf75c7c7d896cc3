#!/bin/bash
# GPU box: bench.py (pair path only) under several environment settings of the library, at both depth-noise levels, alternating
# in ONE call.   tools/ab_env.sh <out-tag> "VAR=a VAR2=b" "VAR=c" ...     ("-" = no setting)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
NOISES=${NOISES:-"0.01 0.002"}
for rep in 1 2 3; do
i=0
for setting in "$@"; do
  i=$((i+1)); [ "$setting" = "-" ] && setting=""
  for noise in $NOISES; do
    env $setting timeout 300 python bench.py --steps 20 --warmup 25 --no-extras --no-cpu-baseline --depth-noise $noise > $O/s${i}_n${noise}_$rep.json 2> $O/s${i}_n${noise}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open('$O/s${i}_n${noise}_$rep.json').read().strip().splitlines()[-1])
    t=d.get('timing',{})
    print('%-34s noise %-6s rep $rep  pairs/s %9.0f  ms/step %.4f  serial %s  parity %s' % ('[$setting]', '$noise', d['value'], d['ms_per_step'], t.get('serial_stage_ms'), (d.get('parity_check') or {}).get('ok')))
except Exception as e:
    print('[$setting] $noise: no line', e); print(open('$O/s${i}_n${noise}_$rep.err').read()[-800:])
PY
  done
done
done
