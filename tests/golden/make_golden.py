"""Generates the committed golden fixtures.  Run in the BUILD container (needs /root/reference
for the executable reference pin):

    python tests/golden/make_golden.py

hamming_golden.npz  -- inputs + outputs of the REFERENCE's own bruteForceSearchORB
                       (src/features.cpp:163-182 compiled into oracle/_ref/libref_bforb.so).
pair_golden.npz     -- frozen outputs of the oracle's full pair path on a small seeded
                       sequence (guards the oracle against drift; "parity unpinned" parts).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from rgbdslam_v2_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hamming_cases():
    rng = np.random.Generator(np.random.PCG64(20260923))
    cases = {}
    # random
    cases["rand"] = (rng.integers(0, 256, (96, 32), dtype=np.uint8),
                     rng.integers(0, 256, (80, 32), dtype=np.uint8))
    # near duplicates + exact duplicates (ties -> first index wins), exact match in LAST row
    t = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t[7] = t[3]
    t[21] = t[3]
    q = t[rng.integers(0, 40, 64)].copy()
    flip = rng.integers(0, 32, 64)
    q[np.arange(64), flip] ^= (1 << rng.integers(0, 8, 64)).astype(np.uint8)
    q[0] = t[39]  # only exact partner is the last train row, which is never searched
    q[1] = t[3]
    cases["ties"] = (q, t)
    # tiny train sets: size 1 and 2
    cases["nt1"] = (rng.integers(0, 256, (8, 32), dtype=np.uint8),
                    rng.integers(0, 256, (1, 32), dtype=np.uint8))
    cases["nt2"] = (rng.integers(0, 256, (8, 32), dtype=np.uint8),
                    rng.integers(0, 256, (2, 32), dtype=np.uint8))
    # all-zero / all-one descriptors (hd 0 and 256)
    q = np.zeros((4, 32), np.uint8)
    t = np.full((5, 32), 255, np.uint8)
    t[2] = 0
    cases["extremes"] = (q, t)
    return cases


def main():
    assert po.ref_lib() is not None, "reference pin not built (needs /root/reference)"
    out = {}
    for name, (q, t) in hamming_cases().items():
        hd = np.empty(q.shape[0], np.int32)
        idx = np.empty(q.shape[0], np.int32)
        for i in range(q.shape[0]):
            hd[i], idx[i] = po.ref_hamming_nn(q[i], t)
        out[name + "_q"], out[name + "_t"], out[name + "_hd"], out[name + "_idx"] = q, t, hd, idx
    np.savez_compressed(os.path.join(HERE, "hamming_golden.npz"), **out)

    seq = synth.make_sequence(n_frames=4, n_kp=256, n_world=900, seed=11)
    prm = po.default_params()
    pairs = [(1, 0), (2, 0), (3, 1), (3, 2)]
    g = dict(desc=seq["desc"], xyz1=seq["xyz1"], pairs=np.array(pairs, np.int32),
             seed=np.uint32(prm.seed), depth_cov=np.float64(prm.depth_cov))
    for k, (q, t) in enumerate(pairs):
        r = po.match_node_pair(seq["desc"][q], seq["xyz1"][q], q, seq["desc"][t], seq["xyz1"][t], t, prm)
        for key in ("id1", "id2", "n_all", "n_inl", "rmse", "T", "info_scale", "valid_iterations",
                    "real_iterations", "all_q", "all_t", "all_hd", "inl_idx"):
            g[f"p{k}_{key}"] = np.asarray(r[key])
    np.savez_compressed(os.path.join(HERE, "pair_golden.npz"), **g)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
