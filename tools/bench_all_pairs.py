"""All-pairs loop-closure search (the shape of BASELINE configs[4]): every frame of a trajectory against every earlier
frame.  Unlike the bench's 20 neighbours per frame most pairs do not overlap, so the `min_matches` gate
(node.cpp:1319) ends them after the Hamming stage; reports pairs/s, the fraction of pairs that reach RANSAC and the
fraction that become edges."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd, RESULT_DTYPE

F = int(sys.argv[1]) if len(sys.argv) > 1 else 90
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
# F frames in places of 10: frames of one place see the same world points, different places share nothing (their
# descriptors are unrelated, so the nearest neighbours are chance matches that RANSAC has to reject)
places = [synth.make_sequence(n_frames=10, n_kp=N, seed=1000 + p) for p in range((F + 9) // 10)]
seq = {"desc": [pl["desc"][i] for pl in places for i in range(10)][:F],
       "xyz1": [pl["xyz1"][i] for pl in places for i in range(10)][:F]}
pq = np.array([q for q in range(F) for t in range(q)], np.int32)
pt = np.array([t for q in range(F) for t in range(q)], np.int32)
fe = FrontEnd(max_nodes=F, max_keypoints=max(1024, N), max_pairs_per_batch=len(pq))
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(4)]
def run(steps):
    tk = []
    for s in range(steps):
        if len(tk) >= 4:
            fe.wait_ticket(tk.pop(0), None)
        tk.append(fe.submit_pair_list(pq, pt, bufs[s % 4].data_ptr()))
    for t in tk:
        fe.wait_ticket(t, None)
    fe.synchronize()
run(3)
t0 = time.perf_counter()
steps = 12
run(steps)
dt = time.perf_counter() - t0
res = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[: len(pq)]
print(json.dumps({"frames": F, "keypoints": N, "pairs_per_step": int(len(pq)), "pairs_per_s": round(len(pq) * steps / dt, 1),
                  "ms_per_step": round(dt / steps * 1e3, 3),
                  "pairs_reaching_ransac": round(float((res["real_iterations"] > 0).mean()), 4),
                  "edge_fraction": round(float((res["id1"] >= 0).mean()), 4)}))
