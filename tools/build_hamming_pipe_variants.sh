# Builds librgbdfe_hp_<name>.so variants of the pipelined Hamming kernel next to the product library (CPU, cross-compiled).
set -e
cd rgbdslam_v2_amd/csrc
build() {  # name, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function \
    -mllvm -amdgpu-mfma-vgpr-form $2 -c hamming_mfma.hip -o /tmp/hp_$1.o
  OBJS=$(ls *.o | grep -v "^hamming_mfma.o$" | grep -v prof | grep -v _wd | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_hp_$1.so $OBJS /tmp/hp_$1.o
}
build w2 "-DRGBDFE_HAMMING_PIPE_WAVES=2"
build burst "-DRGBDFE_HAMMING_PIPE_BURST=1"
build alt4 "-DRGBDFE_HAMMING_PIPE_ALT4=1 -DRGBDFE_HAMMING_PIPE_WAVES=2"
# timing-only builds (keys forced to "no match"): what the stream costs without its reductions / loads + barriers
build d1 "-DRGBDFE_HAMMING_PIPE_DIAG=1"
build d4 "-DRGBDFE_HAMMING_PIPE_DIAG=4"
build alt4d1 "-DRGBDFE_HAMMING_PIPE_ALT4=1 -DRGBDFE_HAMMING_PIPE_WAVES=2 -DRGBDFE_HAMMING_PIPE_DIAG=1"
ls -la ../librgbdfe_hp_*.so
