#!/bin/bash
# GPU box: bench.py (pair path only) with each of the given library variants, at both depth-noise levels, in ONE call --
# the numbers of different gpurun boxes differ by a few percent, the variants of one call are comparable.
#   tools/ab_libs.sh <out-tag> <variant-tag>...      ("base" = rgbdslam_v2_amd/librgbdfe.so, else librgbdfe_<tag>.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
NOISES=${NOISES:-"0.01 0.002"}
for rep in 1 2; do
for tag in "$@"; do
  LIBF=$PWD/rgbdslam_v2_amd/librgbdfe_$tag.so; [ "$tag" = base ] && LIBF=$PWD/rgbdslam_v2_amd/librgbdfe.so
  for noise in $NOISES; do
    RGBDFE_LIB=$LIBF timeout 300 python bench.py --steps 20 --warmup 25 --no-extras --no-cpu-baseline --depth-noise $noise > $O/${tag}_n${noise}_$rep.json 2> $O/${tag}_n${noise}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open('$O/${tag}_n${noise}_$rep.json').read().strip().splitlines()[-1])
    t=d.get('timing',{})
    print('%-8s noise %-6s rep $rep  pairs/s %9.0f  ms/step %.4f  serial %s  parity %s' % ('$tag', '$noise', d['value'], d['ms_per_step'], t.get('serial_stage_ms'), (d.get('parity_check') or {}).get('ok')))
except Exception as e:
    print('$tag $noise: no line', e); print(open('$O/${tag}_n${noise}_$rep.err').read()[-800:])
PY
  done
done
done
