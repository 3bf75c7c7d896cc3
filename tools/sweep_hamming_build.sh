# Builds librgbdfe variants of the fp4 Hamming kernel (query tiles per wave x train tiles per LDS stage) next to the product
# library and times each with tools/bench_hamming_modes.py (gpurun): RGBDFE_LIB selects the library file.
set -e
cd rgbdslam_v2_amd/csrc
for cfg in "2 4" "2 2" "2 8" "4 4" "1 4" "4 2"; do set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function \
    -mllvm -amdgpu-mfma-vgpr-form -DRGBDFE_HAMMING_QT=$1 -DRGBDFE_HAMMING_STAGE=$2 -c hamming_mfma.hip -o /tmp/hm_$1_$2.o
  OBJS=$(ls *.o | grep -v "^hamming_mfma.o$" | grep -v prof | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_hm_$1_$2.so $OBJS /tmp/hm_$1_$2.o
done
ls -la ../librgbdfe_hm_*.so | wc -l
