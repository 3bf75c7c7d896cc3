"""PCIe-inclusive rate of the pair path: the synchronous host-buffer entry point
rgbdfe_match_pair_list (pair ids in, result PODs out over PCIe), same workload as bench.py."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 200, 1000
NOISE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01   # sigma_z = 0.01 z^2
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=NOISE)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
t0 = time.perf_counter()
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
t_up = time.perf_counter() - t0
fe.match_pair_list(pq, pt)
t0 = time.perf_counter()
K = 10
for _ in range(K):
    out = fe.match_pair_list(pq, pt)
dt = (time.perf_counter() - t0) / K
print(json.dumps({"pairs_per_s_host_buffers_sync": round(len(pq) / dt, 1), "ms_per_4000_pairs": round(dt * 1e3, 3),
                  "node_upload_us": round(t_up / F * 1e6, 1), "result_bytes_per_call": int(out.nbytes)}))
