#!/bin/bash
# CPU (cross-compile): registers, LDS, scratch and occupancy of every kernel of librgbdfe.so as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage), with the flags of csrc/Makefile.   Usage: tools/kernel_resources.sh > profiles/<tag>/kernel_resources.txt
cd "$(dirname "$0")/../rgbdslam_v2_amd/csrc"
for f in hamming_nn hamming_mfma place_recognition l2_knn edges select_ransac ransac_split sift_match project3d emm orb_kernels orb_host sift_extract api_batches api_context api_pairs api_detect api_frame api_group rgbdfe_api; do
  extra=""
  case $f in
    sift_match) extra="-mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize";;
    hamming_mfma) extra="-mllvm -amdgpu-mfma-vgpr-form";;
    select_ransac|ransac_split) extra="-fno-slp-vectorize";;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wno-unused-function $extra \
    --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $f.hip -o /dev/null 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk -v file=$f.hip '
    /remark: Function Name:/ {name=$NF}
    /remark: +TotalSGPRs:/ {s=$NF} /remark: +VGPRs:/ {v=$NF} /remark: +AGPRs:/ {a=$NF}
    /remark: +ScratchSize/ {sc=$NF} /remark: +Occupancy/ {oc=$NF} /remark: +VGPRs Spill:/ {vs=$NF}
    /remark: +LDS Size/ {lds=$NF; printf "%-16s VGPR %3s AGPR %3s SGPR %3s  scratch %4s B/lane  spilled VGPR %3s  LDS %6s B  occupancy %s waves/SIMD  %s\n", file, v, a, s, sc, vs, lds, oc, name}' | c++filt | sed 's/(anonymous namespace):://; s/^\(.*waves\/SIMD  \)void /\1/; s/(.*//'
done
