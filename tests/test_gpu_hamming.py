"""-m gpu: the HIP Hamming-NN kernel (through the C ABI) vs the oracle / reference golden vectors.
Bit-exact: integer work."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hamming_golden.npz")


# every Hamming kernel of the library must return the reference's keys: 1 = fp4 MFMA contraction with the row term in
# the accumulator's initial value (the default), 2 = the same with the row term added by the VALU, 0 = xor + popcount
# 3 = the MFMA contraction as a software pipeline inside every wave
@pytest.fixture(scope="module", params=[1, 3, 2, 0], ids=["mfma", "mfma_pipelined", "mfma_valu_row", "popcount"])
def fe(request):
    from rgbdslam_v2_amd.frontend import FrontEnd
    f = FrontEnd(device_id=0, max_nodes=48, max_keypoints=4096, max_pairs_per_batch=1024)
    f.set_hamming_mode(request.param)
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


def test_all_hamming_kernels_agree_on_a_full_batch():
    """4000-keypoint nodes, many pairs, ragged row counts: the three kernels write identical match lists."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(5)
    n_nodes = 12
    rows = [4000, 3999, 3969, 4001 - 64, 33, 32, 31, 1, 2, 2500, 1024, 1000]
    descs = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in rows]
    for k in range(1, n_nodes):  # related nodes, so that hd < 128 matches (and ties) exist
        m = min(rows[k], rows[0])
        take = rng.permutation(m)[: m // 2]
        descs[k][take] = descs[0][take]
        flips = rng.random((len(take), 256)) < 0.04
        descs[k][take] ^= np.packbits(flips, axis=1, bitorder="little")
    pq = np.array([a for a in range(n_nodes) for b in range(n_nodes) if a != b], np.int32)
    pt = np.array([b for a in range(n_nodes) for b in range(n_nodes) if a != b], np.int32)
    outs = []
    for mode in (0, 1, 2, 3):
        fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=4032, max_pairs_per_batch=256)
        fe.set_hamming_mode(mode)
        for k in range(n_nodes):
            xyz = np.concatenate([rng.uniform(-1, 1, (rows[k], 2)), rng.uniform(1, 3, (rows[k], 1)),
                                  np.ones((rows[k], 1))], 1).astype(np.float32)
            fe.upload_node(k, descs[k], xyz)
        keys = []
        for a, b in zip(pq[::7], pt[::7]):
            hd, idx = fe.hamming_nn_nodes(int(a), int(b))
            keys.append((hd, idx))
        out = fe.match_pair_list(pq, pt)
        outs.append((keys, out["n_all"].copy(), out["all_q"].copy(), out["all_t"].copy(), out["all_hd"].copy()))
        fe.close()
    for k, (a, b) in enumerate(zip(pq[::7], pt[::7])):
        hd_ref, idx_ref = po.hamming_nn_batch(descs[a], descs[b])
        for mode in range(4):
            assert np.array_equal(outs[mode][0][k][0], hd_ref), (mode, a, b)
            assert np.array_equal(outs[mode][0][k][1], idx_ref), (mode, a, b)
    for mode in (1, 2, 3):
        for f in range(1, 5):
            assert np.array_equal(outs[0][f], outs[mode][f])
