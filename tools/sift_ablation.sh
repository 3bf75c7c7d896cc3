#!/bin/bash
# Builds timing-ablation variants of sift_top2_fast_kernel (results are WRONG by construction) into
# rgbdslam_v2_amd/librgbdfe_abl<N>.so; run with RGBDFE_LIB=<that> python bench.py --config sift
set -e
cd "$(dirname "$0")/../rgbdslam_v2_amd/csrc"
OBJS=$(ls *.o | grep -v -e sift_match -e select_ransac_prof)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function \
     -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DRGBDFE_SIFT_ABL=$n -c sift_match.hip -o /tmp/sift_match_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_abl$n.so $OBJS /tmp/sift_match_abl$n.o
done
