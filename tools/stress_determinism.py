"""Stress: the same pair batches many times (big batch, small batches, interleaved contexts) must give
byte-identical results every time and match the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
from oracle import pyoracle as po

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_golden.npz"))
seq = synth.make_sequence(n_frames=16, n_kp=600, n_world=2400, seed=21)
pq, pt = synth.candidate_pairs(16, per_frame=8, seed=21)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    fe = FrontEnd(max_nodes=32, max_keypoints=1024, max_pairs_per_batch=64 if rep % 2 else 1000)
    fe.set_params(seed=int(g["seed"]), depth_cov=float(g["depth_cov"]))
    for f in range(4):
        fe.upload_node(f, g["desc"][f], g["xyz1"][f])
    out = fe.match_pair_list(g["pairs"][:, 0], g["pairs"][:, 1])
    for k, rec in enumerate(out):
        if rec["n_all"] != int(g[f"p{k}_n_all"]) or rec["n_inl"] != int(g[f"p{k}_n_inl"]):
            bad += 1
            print("rep", rep, "golden mismatch pair", k, rec["n_all"], rec["n_inl"], int(g[f"p{k}_n_all"]), int(g[f"p{k}_n_inl"]),
                  rec["real_iterations"], int(g[f"p{k}_real_iterations"]))
    for f in range(4):
        fe.release_node(f)
    fe.set_params(seed=20260923, depth_cov=1e-4)
    for f in range(16):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    a = fe.match_pair_list(pq, pt)
    if rep == 0:
        ref = a.copy()
    if a.tobytes() != ref.tobytes():
        bad += 1
        d = np.flatnonzero([x.tobytes() != y.tobytes() for x, y in zip(a, ref)])
        print("rep", rep, "nondeterministic pairs", d[:10], a["n_inl"][d[:5]], ref["n_inl"][d[:5]], a["n_all"][d[:5]], ref["n_all"][d[:5]])
    # small batches take the split-train (atomicMin) path of the Hamming kernel
    for j in range(0, 64, 4):
        b = fe.match_pair_list(pq[j:j + 4], pt[j:j + 4])
        if b.tobytes() != ref[j:j + 4].tobytes():
            bad += 1
            print("rep", rep, "small batch", j, "differs", b["n_all"], ref["n_all"][j:j + 4], b["n_inl"], ref["n_inl"][j:j + 4])
    fe.close()
print("bad", bad)
