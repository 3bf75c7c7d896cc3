"""One-off fuzz of the SIFT pair op (GPU) against the oracle: ragged node sizes, duplicates, NaN depths, parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd, inlier_indices
bad = 0
fe = FrontEnd(device_id=0, max_nodes=12, max_keypoints=1536, max_pairs_per_batch=64)
for master in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    rng = np.random.default_rng(7000 + master)
    F = 6
    sizes = [int(rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 640, 1000, 1023, 1024, 1536])) for _ in range(F)]
    seq = synth.make_sequence(n_frames=F, n_kp=1536, n_world=4000, seed=300 + master, nan_fraction=float(rng.choice([0.0, 0.1])))
    sd = synth.sift_descriptors_like(seq["desc"], seed=master)
    nodes = []
    for f in range(F):
        d, x = sd[f][: sizes[f]].copy(), seq["xyz1"][f][: sizes[f]].copy()
        if sizes[f] > 10 and rng.random() < 0.3: d[1::2] = d[0::2][: len(d[1::2])]   # exact duplicates: tie rules
        if sizes[f] > 10 and rng.random() < 0.2: d[rng.random(sizes[f]) < 0.2] = 0.0
        nodes.append((d, x)); fe.upload_sift_node(f, d, x)
    kw = dict(max_matches=int(rng.choice([1, 5, 64, 65, 300, 320])), min_matches=int(rng.choice([0, 4, 20])),
              ransac_iterations=int(rng.choice([0, 8, 200])), seed=int(rng.integers(0, 2**31)))
    fe.set_params(**kw)
    pq = rng.integers(0, F, 16).astype(np.int32); pt = rng.integers(0, F, 16).astype(np.int32)
    out, dist = fe.match_sift_pair_list(pq, pt)
    prm = po.default_params(**kw)
    for rec, dd, q, t in zip(out, dist, pq, pt):
        ref = po.match_sift_node_pair(nodes[q][0], nodes[q][1], int(q), nodes[t][0], nodes[t][1], int(t), prm)
        n = ref["n_all"]
        T = np.array(rec["trafo"], np.float32).reshape(4, 4).T
        ok = (rec["n_all"] == n and np.array_equal(rec["all_q"][:n], ref["all_q"]) and np.array_equal(rec["all_t"][:n], ref["all_t"])
              and np.array_equal(dd[:n], ref["all_dist"]) and (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
              and rec["n_inl"] == ref["n_inl"] and rec["real_iterations"] == ref["real_iterations"]
              and np.array_equal(inlier_indices(rec), ref["inl_idx"]) and np.array_equal(T, ref["T"]))
        if not ok:
            bad += 1
            print("MISMATCH", master, q, t, sizes[q], sizes[t], kw, rec["n_all"], n)
    for f in range(F): fe.release_node(f)
print("sift fuzz done, mismatches:", bad)
