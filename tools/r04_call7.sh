#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for which in single group; do
  echo "=== bounded spins + fallback, RGBDFE_RANSAC_SPLIT=1, $which"
  RGBDFE_GRAPHS=0 RGBDFE_RANSAC_SPLIT=1 timeout 120 python tools/r04_hang_probe.py 50 $which 2>&1 | grep -v "amdgpu.ids" | tail -8
done
echo "=== large batches (4000 pairs), split, 40 s"
RGBDFE_RANSAC_SPLIT=1 timeout 100 python - <<'PY'
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench
from rgbdslam_v2_amd._lib import RESULT_DTYPE, LIB_PATH
from rgbdslam_v2_amd.frontend import FrontEnd
seq, pq, pt = bench.orb_workload(1)
fe = FrontEnd(max_nodes=200, max_keypoints=1024, max_pairs_per_batch=4096, seed=bench.SEED)
for f in range(200): fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(4)]
t0 = time.time(); n = 0; bad = 0; tk = []
exp = bench.expected("orb", 0.01, 1)
while time.time() - t0 < 40:
    if len(tk) >= 2:
        t, b = tk.pop(0); fe.wait_ticket(t, None)
        if n % 50 == 0:
            r = np.frombuffer(b.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
            bad += bench.pair_aggregates(r) != exp
    tk.append((fe.submit_pair_list(pq, pt, bufs[n % 4].data_ptr()), bufs[n % 4])); n += 1
fe.synchronize()
print("large batches: %d batches of 4000 pairs in %.0f s, %d sampled aggregate mismatches, refinement waves that gave up: %d" % (
    n, time.time() - t0, bad, C.CDLL(LIB_PATH).rgbdfe_debug_split_gave_up()))
PY
} > gpurun_out/r04_fallback_probe.log 2>&1
cat gpurun_out/r04_fallback_probe.log
