# Variant builds of select_ransac.hip (compiler scheduling / unrolling switches) next to the product library; time each with
# bench.py on both depth-noise regimes (gpurun): RGBDFE_LIB selects the library file.
set -e
cd rgbdslam_v2_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function -fno-slp-vectorize"
i=0
while read -r FL; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $BASE $FL -c select_ransac.hip -o /tmp/sr_v$i.o 2>/tmp/sr_v$i.err || { echo "variant $i failed: $FL"; continue; }
  OBJS=$(ls *.o | grep -v "^select_ransac.o$" | grep -v prof | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_sr_v$i.so $OBJS /tmp/sr_v$i.o
  echo "v$i: $FL"
done <<'LIST'
-mllvm -amdgpu-enable-max-ilp-scheduling-strategy
-mllvm -amdgpu-schedule-metric-bias=0
-fno-unroll-loops
-mllvm -enable-post-misched=0
-mllvm -amdgpu-use-divergent-register-indexing
-Os
LIST
