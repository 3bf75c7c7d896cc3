#!/bin/bash
# Here (not on the GPU box): gpurun_out/prof_<tag>/ -> profiles/<tag>/ (tracked), without the per-dispatch kernel traces
# (kernel_stats.csv keeps the per-kernel averages) and empty logs; then profiles/<tag>_pmc_summary.json is rebuilt from
# the tracked copy, so every figure bench.py quotes is reproducible from files in profiles/.
set -eu
TAG=${1:-r03}
SRC=gpurun_out/prof_$TAG
DST=profiles/$TAG
rm -rf $DST
mkdir -p $DST
(cd $SRC && find . -type f ! -name "*kernel_trace.csv" ! -name "*.db" ! -size 0 | while read f; do mkdir -p "../../$DST/$(dirname "$f")"; cp "$f" "../../$DST/$f"; done)
python tools/make_pmc_summary.py $TAG --from profiles | tail -1
du -sh $DST
