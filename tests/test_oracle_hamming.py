"""Pins the oracle's Hamming NN (A.1/A.2) on the reference's own function and golden vectors."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hamming_golden.npz")
CASES = ["rand", "ties", "nt1", "nt2", "extremes"]


def popcount_rows(a, b):
    return np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(axis=2)


def numpy_nn(q, t):
    """Independent numpy restatement of features.cpp:168-182 (last row skipped, first min wins)."""
    nq = q.shape[0]
    if t.shape[0] <= 1:
        return np.full(nq, 257, np.int32), np.full(nq, -1, np.int32)
    d = popcount_rows(q, t[:-1])
    idx = d.argmin(axis=1).astype(np.int32)  # argmin returns the first minimum
    return d[np.arange(nq), idx].astype(np.int32), idx


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden(case):
    g = np.load(GOLD)
    hd, idx = po.hamming_nn_batch(g[case + "_q"], g[case + "_t"])
    assert np.array_equal(hd, g[case + "_hd"])
    assert np.array_equal(idx, g[case + "_idx"])


def test_golden_documents_the_size_minus_one_quirk():
    g = np.load(GOLD)
    # query 0 of "ties" equals the LAST train row exactly, yet the reference does not find hd=0
    assert g["ties_hd"][0] > 0
    # with a single train row nothing is searched at all: (257, -1)
    assert np.all(g["nt1_hd"] == 257) and np.all(g["nt1_idx"] == -1)
    # ties: first index wins (rows 3, 7, 21 are identical)
    assert g["ties_idx"][1] == 3 and g["ties_hd"][1] == 0


@pytest.mark.skipif(po.ref_lib() is None, reason="reference pin (oracle/_ref) not built")
def test_oracle_matches_live_reference_function():
    rng = np.random.default_rng(5)
    for nq, nt in [(50, 37), (3, 2), (17, 300), (5, 1)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        hd, idx = po.hamming_nn_batch(q, t)
        for i in range(nq):
            d, j = po.ref_hamming_nn(q[i], t)
            assert (d, j) == (hd[i], idx[i])


def test_oracle_matches_numpy_restatement():
    rng = np.random.default_rng(6)
    for nq, nt in [(64, 64), (100, 999), (7, 3), (9, 0)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        hd, idx = po.hamming_nn_batch(q, t)
        hd2, idx2 = numpy_nn(q, t)
        assert np.array_equal(hd, hd2) and np.array_equal(idx, idx2)


def test_feature_matching_gate_truncation_and_order():
    rng = np.random.default_rng(7)
    t = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    q = t[rng.permutation(400)[:350]].copy()
    # flip a controlled number of bits so that many hd ties exist
    for i in range(350):
        bits = rng.choice(256, size=i % 6, replace=False)
        for b in bits:
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    q = np.concatenate([q, rng.integers(0, 256, (50, 32), dtype=np.uint8)])
    hd, idx = po.hamming_nn_batch(q, t)
    mq, mt, mhd = po.feature_matching_orb(q, t, max_matches=300)
    assert len(mq) == 300
    # every emitted match obeys the hd < 128 gate (node.cpp:572) and is the NN result
    assert np.all(mhd < 128)
    assert np.array_equal(mhd, hd[mq]) and np.array_equal(mt, idx[mq])
    # order = (hd, queryIdx) ascending; set = the 300 smallest keys
    keys = mhd.astype(np.int64) * 65536 + mq
    assert np.all(np.diff(keys) > 0)
    allk = np.sort(hd[hd < 128].astype(np.int64) * 65536 + np.flatnonzero(hd < 128))
    assert np.array_equal(keys, allk[:300])
    # fewer than max_matches survivors: nothing is dropped
    mq2, _, _ = po.feature_matching_orb(q[:100], t, max_matches=300)
    assert len(mq2) == int((hd[:100] < 128).sum())
