import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import test_gpu_pairs as T
from oracle import pyoracle as po
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
bad = 0
fe2 = FrontEnd(device_id=0, max_nodes=12, max_keypoints=1536, max_pairs_per_batch=64)
if len(sys.argv) > 1:  # one_wave | latency (single recording phase) | phased (four phases, forced for small batches)
    {"one_wave": lambda: fe2.set_latency_mode(0, 0), "latency": lambda: fe2.set_latency_mode(64, 7),
     "phased": lambda: fe2.set_latency_mode(64, -5)}[sys.argv[1]]()
first, count = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 40)  # fuzz_pairs.py <mode> <first> <count>
for master in range(first, first + count):
    rng = np.random.default_rng(9000 + master)
    F = 8
    sizes = [int(rng.choice([0, 1, 2, 3, 4, 5, 20, 21, 22, 64, 65, 300, 301, 777, 1000, 1536])) for _ in range(F)]
    seq = synth.make_sequence(n_frames=F, n_kp=1536, n_world=int(rng.choice([1600, 4000])), seed=500 + master,
                              nan_fraction=float(rng.choice([0.0, 0.05, 0.3, 0.9])))
    nodes = []
    for f in range(F):
        d, x = seq["desc"][f][: sizes[f]].copy(), seq["xyz1"][f][: sizes[f]].copy()
        if sizes[f] > 10 and rng.random() < 0.3: x[rng.random(sizes[f]) < 0.1, 2] = 0.0
        if sizes[f] > 10 and rng.random() < 0.2: d[:] = rng.integers(0, 256, d.shape, dtype=np.uint8)
        if sizes[f] > 10 and rng.random() < 0.2: d[1::2] = d[0::2][: len(d[1::2])]
        if sizes[f] > 10 and rng.random() < 0.1: x[:, :3] *= np.float32(rng.choice([1e-3, 1e3]))
        nodes.append((d, x)); fe2.upload_node(f, d, x)
    kw = dict(max_matches=int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 319, 320])),
              min_matches=int(rng.choice([0, 1, 3, 4, 5, 20, 50, 400])),
              ransac_iterations=int(rng.choice([0, 1, 6, 7, 8, 9, 50, 200, 300])),
              max_dist_for_inliers=float(rng.choice([0.1, 0.5, 2.0, 3.0, 100.0])),
              depth_cov=float(rng.choice([1e-4, 2.5e-5, 1e-3, 1e-8])), seed=int(rng.integers(0, 2**31)))
    fe2.set_params(**kw)
    pq = rng.integers(0, F, 32).astype(np.int32); pt = rng.integers(0, F, 32).astype(np.int32)
    out = fe2.match_pair_list(pq, pt)
    prm = po.default_params(**kw)
    for rec, q, t in zip(out, pq, pt):
        ref = po.match_node_pair(nodes[q][0], nodes[q][1], int(q), nodes[t][0], nodes[t][1], int(t), prm)
        try:
            T.check_against_oracle(rec, ref)
        except AssertionError as e:
            bad += 1
            print("MISMATCH master", master, "pair", q, t, "sizes", sizes[q], sizes[t], kw, str(e)[:200])
    for f in range(F): fe2.release_node(f)
print("fuzz done, mismatches:", bad)
