"""Latency of the live-SLAM call shape: one new node against its 20 candidates (GraphManager::nodeComparisons per
frame, graph_manager.cpp:531-583) through the synchronous rgbdfe_match_node_pairs, host buffers in and out."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 60, 1000
NOISE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01   # sigma_z = 0.01 z^2
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=NOISE)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=64)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
out = {}
for n_cand in (1, 3, 20):
    ts = []
    for f in range(25, F):
        cand = np.arange(f - n_cand, f, dtype=np.int32)
        t0 = time.perf_counter()
        r = fe.match_node_pairs(f, cand)
        ts.append(time.perf_counter() - t0)
    out["ms_%d_candidates" % n_cand] = round(float(np.median(ts)) * 1e3, 3)
    out["edges_%d" % n_cand] = int((r["id1"] >= 0).sum())
t0 = time.perf_counter()
fe.upload_node(0, seq["desc"][0], seq["xyz1"][0])
out["upload_node_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
print(json.dumps(out))
