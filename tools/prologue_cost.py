"""select+RANSAC stage time over the iteration count (one-wave schedule, 4000 pairs, one batch in flight): the
intercept is the per-wave prologue (match selection, 3-D gather, covariance records) + result write that every
recording / replay wave of the record / replay schedule repeats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
KERNEL_RANSAC = 1
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
for mode, limit in (("one_wave", 0), ("record_replay", (1 << 31) - 1)):
    fe.set_latency_mode(limit, 0)
    for iters in (0, 1, 7, 14, 28, 56, 100, 200):
        fe.set_params(ransac_iterations=iters)
        fe.match_pair_list(pq, pt)
        fe.set_profiling(True)
        fe.reset_kernel_time()
        for _ in range(3):
            tk = fe.submit_pair_list(pq, pt, 0) if False else None
            fe.match_pair_list(pq, pt)
        fe.set_profiling(False)
        ms, n, _ = fe.kernel_time(KERNEL_RANSAC)
        print(mode, iters, "stage ms per 2000-pair piece: %.3f" % (ms / max(n, 1)), flush=True)
