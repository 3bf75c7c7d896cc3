#!/bin/bash
# GPU box: the detection batch entry point with the pass read-back written by the measure kernel itself (default) against the
# copy behind the kernels (RGBDFE_DETECT_HOSTWRITE=0), alternating in one call; then the kernel trace of either.
cd $GRAFT_REPO_ROOT
O=gpurun_out/detect_hostwrite; mkdir -p $O
for rep in 1 2 3 4; do
  for hw in 1 0; do
    for cfg in "640 480 1000 112" "1280 960 4000 56"; do
      echo -n "hostwrite=$hw rep $rep: "
      RGBDFE_DETECT_HOSTWRITE=$hw timeout 300 python tools/bench_detect_batch.py $cfg 9 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['width'], d['frames'], 'ms/frame median', d['ms_per_frame_median'], 'min', d['ms_per_frame_min'], 'fps', d['frames_per_s_median'])"
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for hw in 1 0; do
  RGBDFE_DETECT_HOSTWRITE=$hw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_hw$hw -o t -- python $GRAFT_REPO_ROOT/tools/bench_detect_batch.py 640 480 1000 168 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/trace_hw$hw -name "*kernel_stats.csv" | head -1)
  echo "== hostwrite=$hw kernel stats (168 + 14 frames)"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:12]: print('  %-50s calls %4s total %8.1f us avg %7.1f' % (r['Name'][:50], r['Calls'], int(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
print('  all kernels: %.1f us = %.2f us per frame (182 frames)' % (tot/1e3, tot/1e3/182))
PY
done
