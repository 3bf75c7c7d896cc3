import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 60, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=64)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
for chunk in (7, 5, 4, 3, 2):
    fe.set_latency_mode((1 << 31) - 1, chunk)
    out = {}
    for n_cand in (1, 20):
        ts = []
        for f in range(25, F):
            cand = np.arange(f - n_cand, f, dtype=np.int32)
            t0 = time.perf_counter()
            r = fe.match_node_pairs(f, cand)
            ts.append(time.perf_counter() - t0)
        out["ms_%d" % n_cand] = round(float(np.median(ts)) * 1e3, 3)
    print(chunk, out, flush=True)
# stage split for 20 candidates with the default
fe.set_latency_mode((1 << 31) - 1, 0)
fe.set_profiling(True); fe.reset_kernel_time()
for f in range(25, F):
    fe.match_node_pairs(f, np.arange(f - 20, f, dtype=np.int32))
fe.set_profiling(False)
for k, nm in ((0, "hamming"), (1, "ransac stage")):
    ms, n, _ = fe.kernel_time(k)
    print(nm, "ms per call %.3f" % (ms / max(n, 1)))
