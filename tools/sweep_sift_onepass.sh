#!/bin/bash
# Build-time variants of the one-pass SIFT kernel (sift_top2_onepass_kernel) as librgbdfe_s1_<tag>.so (CPU side: hipcc
# cross-compiles), then -- on the GPU box -- the sift sub-record of bench.py with each of them (RGBDFE_LIB picks the
# library; every variant passes the sub-record's oracle-constant check or aborts).
#   tools/sweep_sift_onepass.sh build        (here)
#   tools/sweep_sift_onepass.sh run          (GPU box, via gpurun)
# Timing-only decomposition (results void, no parity check): SIFT1_VARIANTS="base:-DRGBDFE_SIFT1_NV=11 d1:-DRGBDFE_SIFT1_DIAG=1
# d2:-DRGBDFE_SIFT1_DIAG=2 d4:-DRGBDFE_SIFT1_DIAG=4" SIFT1_NO_PARITY=1 -- no digest / row side only / column side only: what the
# MFMA + LDS + barrier stream of the one-pass kernel costs without (part of) its VALU work (prepared at the end of round 4,
# not yet run: DESIGN.md section 7).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
V=${SIFT1_VARIANTS:-"base:-DRGBDFE_SIFT1_NV=11 burst:-DRGBDFE_SIFT1_BURST=1 bursttree:-DRGBDFE_SIFT1_BURST=1_-DRGBDFE_SIFT1_TREE=1"}
if [ "${1:-build}" = build ]; then
  cd rgbdslam_v2_amd/csrc && make -s
  REST=$(ls *.o | grep -v '^sift_match' | grep -v '_prof.o' | grep -v '^sift_match_')
  for v in $V; do
    tag=${v%%:*}; defs=$(echo ${v#*:} | tr '_' ' ' | sed 's/RGBDFE SIFT1 /RGBDFE_SIFT1_/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function \
      -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize $defs -c sift_match.hip -o sift_match_$tag.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librgbdfe_s1_$tag.so $REST sift_match_$tag.o || exit 1
    echo "built librgbdfe_s1_$tag.so ($defs)"
  done
  rm -f sift_match_*.o
else
  cat > /tmp/sift_ab.py <<'PY'
import json, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
seq, _, _ = bench.orb_workload(1)
r = bench.sift_subrecord(seq, 0, os.environ.get("SIFT1_NO_PARITY") != "1")
print(json.dumps({"value": r["value"], "ms_per_step": r["ms_per_step"], "serial_stage_ms": r["serial_stage_ms"], "frac": r["roofline"]["frac"], "parity": (r.get("parity_check") or {}).get("ok")}))
PY
  for v in $V; do
    tag=${v%%:*}
    echo -n "$tag: "; RGBDFE_LIB=$ROOT/rgbdslam_v2_amd/librgbdfe_s1_$tag.so timeout 200 python /tmp/sift_ab.py 2>&1 | tail -1
  done
fi
