#!/bin/bash
# Builds a variant of ONE translation unit of librgbdfe.so with extra compiler flags into rgbdslam_v2_amd/librgbdfe_<tag>.so
# (the other objects are the regular build's): A/B timing of kernel forms in one gpurun call, RGBDFE_LIB=<that> selects it.
#   tools/build_variant.sh sift_extract c16 -DRGBDFE_SIFT_DESC_COPIES=16
set -e
UNIT=$1; TAG=$2; shift 2
cd "$(dirname "$0")/../rgbdslam_v2_amd/csrc"
make -s >/dev/null
EXTRA=$(make -s -p -n 2>/dev/null | grep "^FLAGS_$UNIT :=" | sed 's/^[^=]*=//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function \
  $EXTRA "$@" -c $UNIT.hip -o /tmp/${UNIT}_$TAG.o
OBJS=$(ls *.o | grep -v -e "^$UNIT.o" -e _prof -e _stats)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_$TAG.so $OBJS /tmp/${UNIT}_$TAG.o 2>/dev/null
echo built rgbdslam_v2_amd/librgbdfe_$TAG.so
