"""A batch too large for the record / replay scratch (pairs x iterations > 2^24 records) runs as several record / replay
pieces: its results must equal those of the same pairs submitted in bench-sized batches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 200, 600
seq = synth.make_sequence(n_frames=F, n_kp=N)
rng = np.random.default_rng(1)
n = 24000
pq = rng.integers(1, F, n).astype(np.int32)
pt = (pq - rng.integers(1, 12, n)).clip(0).astype(np.int32)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=n)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
big = fe.match_pair_list(pq, pt)            # 2 pieces of 12000 pairs x 200 iterations = 2.4M records each: record / replay
fe.set_params(ransac_iterations=1500)       # ransac_iterations = 1500: 12000 x 1500 = 18M > 2^24 records: pieces of 11184 pairs
big400 = fe.match_pair_list(pq, pt)
small400 = np.concatenate([fe.match_pair_list(pq[a:a + 4000], pt[a:a + 4000]) for a in range(0, n, 4000)])
fe.set_params(ransac_iterations=200)
small = np.concatenate([fe.match_pair_list(pq[a:a + 4000], pt[a:a + 4000]) for a in range(0, n, 4000)])
print("200 iterations: big == small:", big.tobytes() == small.tobytes())
print("1500 iterations: big == small:", big400.tobytes() == small400.tobytes(), "edges", int((big400["id1"] >= 0).sum()))
