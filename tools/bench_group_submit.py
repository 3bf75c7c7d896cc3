"""Multi-device handle on ONE GPU (device 0 listed G times): host submission time of a sharded batch from the single
submitting thread, and the batch's wall time, for G = 1, 2, 4, 8 (VERDICT r2 #6a).  Weak scaling: 4000 pairs per listed
device.  With G devices on G real GPUs the wall time is that of one shard; here the shards share one chip, so only the
submission figures carry over."""
import os
os.environ.setdefault("RGBDFE_GRAPHS", "1")   # the cached launch chains are what this tool measures (off by default in the library)
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import COMPACT_DTYPE
from rgbdslam_v2_amd.frontend import FrontEnd

F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
out = {}
for G in (1, 2, 4, 8):
    pq, pt = synth.candidate_pairs(F, per_frame=min(20 * G, F - 1))
    fe = FrontEnd(device_id=0, max_nodes=F, max_keypoints=1024, max_pairs_per_batch=(len(pq) + G - 1) // G, device_ids=[0] * G)
    for f in range(F):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    per = (len(pq) + G - 1) // G
    bufs = [torch.zeros(G * per * COMPACT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0") for _ in range(G)]
    ptrs = [b.data_ptr() for b in bufs]
    for _ in range(6):   # every ring slot of every device sees the batch shape once: graphs cached
        fe.match_pair_list_allgather_compact(pq, pt, ptrs)
    sub, wall = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        fe.match_pair_list_allgather_compact(pq, pt, ptrs)
        wall.append((time.perf_counter() - t0) * 1e3)
        sub.append(fe.group_submit_us())
    out["G%d" % G] = {"pairs": int(len(pq)), "submit_us_median": round(float(np.median(sub)), 1),
                      "submit_us_per_device": round(float(np.median(sub)) / G, 1), "wall_ms_median": round(float(np.median(wall)), 3),
                      "transport": fe.gather_transport()}
    fe.close()
print(json.dumps(out))
