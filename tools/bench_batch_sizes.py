"""GPU box: one synchronous pair batch of n pairs, results left in HBM, n = 64 ... 4000 (ms per batch, 20 repetitions each).
    python tools/bench_batch_sizes.py [depth_noise=0.01]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=noise)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
import torch
buf = torch.zeros(4096 * 1744, dtype=torch.uint8, device="cuda")
out = {}
for n in (64, 256, 300, 512, 768, 1024, 1536, 2048, 3000, 4000):
    fe.wait_ticket(fe.submit_pair_list(pq[:n], pt[:n], buf.data_ptr()), None)
    reps = 20
    fe.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fe.wait_ticket(fe.submit_pair_list(pq[:n], pt[:n], buf.data_ptr()), None)
    fe.synchronize()
    out[n] = round((time.perf_counter() - t0) / reps * 1e3, 3)
print(json.dumps(out))
