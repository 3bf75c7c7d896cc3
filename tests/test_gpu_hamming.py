"""-m gpu: the HIP Hamming-NN kernel (through the C ABI) vs the oracle / reference golden vectors.
Bit-exact: integer work."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hamming_golden.npz")


@pytest.fixture(scope="module")
def fe():
    from rgbdslam_v2_amd.frontend import FrontEnd
    f = FrontEnd(device_id=0, max_nodes=48, max_keypoints=4096, max_pairs_per_batch=1024)
    yield f
    f.close()


@pytest.mark.parametrize("case", ["rand", "ties", "nt1", "nt2", "extremes"])
def test_kernel_matches_reference_golden(fe, case):
    g = np.load(GOLD)
    hd, idx = fe.bruteForceSearchORB_batch(g[case + "_q"], g[case + "_t"])
    assert np.array_equal(hd, g[case + "_hd"])
    assert np.array_equal(idx, g[case + "_idx"])


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (64, 3), (65, 5), (511, 64), (512, 65), (513, 999),
                                   (1000, 1000), (1500, 1499), (4000, 4000), (7, 0), (0, 9)])
def test_kernel_matches_oracle_random(fe, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    hd, idx = fe.bruteForceSearchORB_batch(q, t)
    hd2, idx2 = po.hamming_nn_batch(q, t)
    assert np.array_equal(hd, hd2) and np.array_equal(idx, idx2)


def test_kernel_ties_first_minimum_wins_across_splits(fe):
    # many exact duplicates spread over the whole train set: small batches split the train rows
    # over blocks (atomicMin combine) and must still return the FIRST row of minimal distance.
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
    t = base[rng.integers(0, 8, 2000)]
    q = base[rng.integers(0, 8, 700)]
    hd, idx = fe.bruteForceSearchORB_batch(q, t)
    hd2, idx2 = po.hamming_nn_batch(q, t)
    assert np.array_equal(hd, hd2) and np.array_equal(idx, idx2)
    assert np.all(hd == 0)


def test_resident_nodes_and_large_batch_path(fe):
    # The unsplit (large batch) launch path: many pairs at once through match_pair_list, checked on
    # the match lists (hd, q, t) which are a pure function of the NN keys.
    rng = np.random.default_rng(11)
    n_nodes, N = 40, 300
    descs = rng.integers(0, 256, (n_nodes, N, 32), dtype=np.uint8)
    # make nodes related so that hd<128 matches exist
    for k in range(1, n_nodes):
        keep = rng.random(N) < 0.6
        descs[k][keep] = descs[0][rng.permutation(N)[: keep.sum()]]
        flips = rng.random((N, 256)) < 0.05
        descs[k] ^= np.packbits(flips, axis=1, bitorder="little")
    xyz = np.concatenate([rng.uniform(-1, 1, (n_nodes, N, 2)), rng.uniform(1, 3, (n_nodes, N, 1)),
                          np.ones((n_nodes, N, 1))], 2).astype(np.float32)
    for k in range(n_nodes):
        fe.upload_node(100 + k, descs[k], xyz[k])
    pq = np.repeat(np.arange(n_nodes), n_nodes - 1)
    pt = np.array([c for f in range(n_nodes) for c in range(n_nodes) if c != f])
    assert len(pq) >= 1024
    pq, pt = pq[:1024], pt[:1024]
    out = fe.match_pair_list(pq + 100, pt + 100)
    for rec, q, t in zip(out[::37], pq[::37], pt[::37]):
        mq, mt, mhd = po.feature_matching_orb(descs[q], descs[t], 300)
        n = len(mq)
        assert rec["n_all"] == n
        assert np.array_equal(rec["all_q"][:n], mq) and np.array_equal(rec["all_t"][:n], mt)
        assert np.array_equal(rec["all_hd"][:n], mhd)
    hd, idx = fe.hamming_nn_nodes(100, 101)
    hd2, idx2 = po.hamming_nn_batch(descs[0], descs[1])
    assert np.array_equal(hd, hd2) and np.array_equal(idx, idx2)
    for k in range(n_nodes):
        fe.release_node(100 + k)
