"""Throughput of the synchronous pair op over the batch size, one-wave-per-pair kernel vs record / replay path."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
out = {}
for n in (1, 20, 64, 128, 256, 512, 768, 1024, 1536, 2048, 4000):
    row = {}
    for name, limit, chunk in (("one_wave", 0, 0), ("record_replay_auto", 1 << 30, 0), ("record_replay_7", 1 << 30, 7),
                               ("record_replay_14", 1 << 30, 14), ("record_replay_28", 1 << 30, 28)):
        fe.set_latency_mode(limit, chunk)
        fe.match_pair_list(pq[:n], pt[:n])
        reps = 5 if n >= 512 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            fe.match_pair_list(pq[:n], pt[:n])
        row[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    out[n] = row
    print(n, row, flush=True)
print(json.dumps(out))
