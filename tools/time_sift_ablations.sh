#!/bin/bash
# Times the builds of tools/sift_ablation.sh on the GPU box (match stage of bench.py --config sift); results are wrong by construction.
cd $GRAFT_REPO_ROOT
for n in ${ABL_LIST:-1 2 6 7 8}; do
echo -n "abl$n: "
RGBDFE_LIB=$PWD/rgbdslam_v2_amd/librgbdfe_abl$n.so timeout 300 python bench.py --config sift --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'])"
done
