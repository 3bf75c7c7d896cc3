"""-m gpu: the bf16-MFMA SIFT matcher (SiftGPUWrapper::match semantics) and the SIFT pair op vs
the oracle.  The dot products are integers computed exactly on the matrix cores, so match lists
are compared bit-exactly; distances and poses are float and asserted both within tolerance and
bit-equal (same operation order)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu


# both RANSAC paths: one wave per pair, and the record / replay latency path small batches take by default
@pytest.fixture(scope="module", params=[0, 1 << 20], ids=["one_wave_per_pair", "record_replay"])
def fe(request):
    from rgbdslam_v2_amd.frontend import FrontEnd
    f = FrontEnd(device_id=0, max_nodes=24, max_keypoints=1280, max_pairs_per_batch=256)
    f.set_latency_mode(request.param, 0)
    yield f
    f.close()


def _xyz(rng, n):
    return np.concatenate([rng.uniform(-1, 1, (n, 2)), rng.uniform(1, 3, (n, 1)), np.ones((n, 1))], 1).astype(np.float32)


def _rand_sift(rng, n):
    v = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = np.minimum(v, 0.2)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


@pytest.mark.parametrize("n1,n2", [(1000, 1000), (300, 700), (129, 127), (128, 128), (257, 33), (1, 1), (40, 1), (1, 40)])
def test_sift_match_nodes_vs_oracle(fe, n1, n2):
    rng = np.random.default_rng(n1 * 31 + n2)
    d2 = _rand_sift(rng, n2)
    d1 = _rand_sift(rng, n1)
    k = min(n1, n2) * 2 // 3
    src = rng.permutation(n2)[:k]
    d1[:k] = d2[src] + rng.normal(0, 0.01, (k, 128)).astype(np.float32)
    d1 = np.abs(d1)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    fe.upload_sift_node(1, d1, _xyz(rng, n1))
    fe.upload_sift_node(2, d2, _xyz(rng, n2))
    mq, mt, md = fe.sift_match_nodes(1, 2)
    oq, ot, od = po.sift_match(d1, d2)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot)
    assert np.allclose(md, od, rtol=1e-6, atol=0)
    assert np.array_equal(md, od)
    if min(n1, n2) > 100:
        assert len(mq) > k // 2
    fe.release_node(1)
    fe.release_node(2)


def _check_nodes(fe, d1, d2, rng):
    fe.upload_sift_node(1, d1, _xyz(rng, len(d1)))
    fe.upload_sift_node(2, d2, _xyz(rng, len(d2)))
    mq, mt, md = fe.sift_match_nodes(1, 2)
    oq, ot, od = po.sift_match(d1, d2)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)
    fe.release_node(1)
    fe.release_node(2)
    return len(mq)


def test_sift_key_paths(fe):
    """The dot-product kernel has two key formats (sift_match.hip): float keys  dot + (31 - seq) / 32  when both nodes
    hold <= 1024 rows and every quantised squared norm is < 2^19, integer keys otherwise.  Same answers from both:
    nodes above 1024 rows, descriptors that are not unit length (one node or both), norms right under the 2^19 limit
    with duplicated rows (the tie-break bits sit 24 bits under the leading bit there), and saturated u8 values."""
    rng = np.random.default_rng(77)
    base = _rand_sift(rng, 1200)
    noisy = np.abs(base + rng.normal(0, 0.01, base.shape).astype(np.float32))
    noisy /= np.linalg.norm(noisy, axis=1, keepdims=True)
    perm = rng.permutation(1200)
    assert _check_nodes(fe, noisy[perm][:1000], base[:1000], rng) > 300      # float keys
    assert _check_nodes(fe, noisy[perm][:1100], base[:900], rng) > 300       # > 1024 rows on one side: integer keys
    assert _check_nodes(fe, noisy[perm][:900], base[:1100], rng) > 300
    assert _check_nodes(fe, noisy[perm][:1000] * 1.6, base[:1000], rng) > 300   # |d|^2 = 2.56 * 2^18: integer keys
    _check_nodes(fe, noisy[perm][:1000] * 1.6, base[:1000] * 1.7, rng)   # every angle is acos(1): no match survives
    # squared norms just under 2^19 (float keys at their upper limit), many exact duplicates -> equal dot products
    u = np.full((48, 128), 64, np.int32)
    u[:, 0] = 63
    for r in range(48):
        k = rng.integers(1, 128, 6)
        u[r, k] -= rng.integers(1, 20, 6)
    assert ((u * u).sum(1) < (1 << 19)).all() and ((u * u).sum(1) > (1 << 19) - 40000).all()
    f = (u / 512.0).astype(np.float32)
    d2 = f[rng.integers(0, 48, 1024)]
    d1 = f[rng.integers(0, 48, 1000)]
    _check_nodes(fe, d1, d2, rng)
    d2n = np.abs(d2 + rng.normal(0, 2e-3, d2.shape).astype(np.float32))
    _check_nodes(fe, d1, np.minimum(d2n, 0.1249), rng)
    # one row over the limit switches the whole node to integer keys
    d1b = d1.copy()
    d1b[500, :] = 0.126
    _check_nodes(fe, d1b, d2, rng)
    # saturated / wrapping u8 values (SiftMatchCU.cpp:96-99 stores an unsigned char): 0.6 * 512 = 307 -> 51
    d1c = _rand_sift(rng, 300)
    d1c[::7, 3] = 0.6
    _check_nodes(fe, d1c, base[:400], rng)


def test_sift_block_shapes(fe):
    """Float keys run in two block shapes (sift_match.hip, launch_sift_dot): 128-row blocks (32 rows per wave) when the
    larger side of the batch fits one block, 256-row blocks (64 rows per wave, tiles loaded straight into LDS)
    otherwise.  Sizes around every boundary of both -- 32 / 64 rows per wave, 128 / 256 rows per block, 128-column tiles
    (the ragged last tile runs the same pipeline under a column mask), waves without rows -- with duplicated rows so
    that the tie rules are exercised in every shape."""
    rng = np.random.default_rng(4242)
    base = _rand_sift(rng, 1024)
    base[1::5] = base[0::5][: len(base[1::5])]          # exact duplicates
    noisy = np.abs(base + rng.normal(0, 0.01, base.shape).astype(np.float32))
    noisy /= np.linalg.norm(noisy, axis=1, keepdims=True)
    perm = rng.permutation(1024)
    for n1, n2 in ((1, 1), (31, 128), (128, 33), (129, 64), (64, 129), (127, 255), (256, 256), (257, 129), (193, 385),
                   (320, 511), (513, 640), (700, 1023), (1024, 767), (1024, 1024), (65, 1000)):
        _check_nodes(fe, noisy[perm][:n1], base[:n2], rng)


def test_sift_tie_rules(fe):
    """Exact duplicates force equal dot products: the row side must follow RowMatch_Kernel's
    32-thread butterfly (ProgramCU.cu:1715-1736), the column side "lowest row wins" (:1464-1467,
    :1773)."""
    rng = np.random.default_rng(5)
    base = _rand_sift(rng, 40)
    d2 = base[rng.integers(0, 40, 500)]          # many duplicate train rows
    d1 = base[rng.integers(0, 40, 300)]          # many duplicate query rows
    fe.upload_sift_node(1, d1, _xyz(rng, 300))
    fe.upload_sift_node(2, d2, _xyz(rng, 500))
    mq, mt, md = fe.sift_match_nodes(1, 2)
    oq, ot, od = po.sift_match(d1, d2)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)
    # distinct near-duplicates: ratio test passes, ties still decide the winners
    d2b = d2 + rng.normal(0, 1e-4, d2.shape).astype(np.float32)
    fe.upload_sift_node(2, d2b, _xyz(rng, 500))
    mq, mt, md = fe.sift_match_nodes(1, 2)
    oq, ot, od = po.sift_match(d1, d2b)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)
    fe.release_node(1)
    fe.release_node(2)


def test_sift_pair_op_vs_oracle(fe):
    from rgbdslam_v2_amd.frontend import inlier_indices
    F = 8
    seq = synth.make_sequence(n_frames=F, n_kp=1000, seed=4, depth_noise=synth.DEPTH_NOISE_R1)  # (0.01 z^2: the test below)
    sd = synth.sift_descriptors_like(seq["desc"], seed=4)
    for f in range(F):
        fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
    pq, pt = synth.candidate_pairs(F, per_frame=3, seed=4)
    out, dist = fe.match_sift_pair_list(pq, pt)
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
    n_edges = 0
    for rec, dd, q, t in zip(out, dist, pq, pt):
        ref = po.match_sift_node_pair(sd[q], seq["xyz1"][q], int(q), sd[t], seq["xyz1"][t], int(t), prm)
        n = ref["n_all"]
        assert rec["n_all"] == n
        assert np.array_equal(rec["all_q"][:n], ref["all_q"]) and np.array_equal(rec["all_t"][:n], ref["all_t"])
        assert np.array_equal(dd[:n], ref["all_dist"])
        assert np.all(np.diff(dd[:n]) >= 0)
        assert (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
        assert rec["n_inl"] == ref["n_inl"] and rec["real_iterations"] == ref["real_iterations"]
        assert np.array_equal(inlier_indices(rec), ref["inl_idx"])
        T = np.array(rec["trafo"], np.float32).reshape(4, 4).T
        assert np.abs(T - ref["T"]).max() <= 1e-4
        assert np.array_equal(T, ref["T"])
        if ref["id1"] >= 0:
            n_edges += 1
            assert np.abs(T - synth.relative_pose(seq["poses"], q, t)).max() < 0.03
    assert n_edges >= len(pq) * 3 // 4
    # ORB and SIFT nodes cannot be mixed in one pair
    fe.upload_node(100, seq["desc"][0], seq["xyz1"][0])
    from rgbdslam_v2_amd._lib import RgbdfeError
    with pytest.raises(RgbdfeError):
        fe.match_sift_pair_list([0], [100])
    with pytest.raises(RgbdfeError):
        fe.match_pair_list([100], [0])
    fe.release_node(100)
    for f in range(F):
        fe.release_node(f)


def test_sift_pair_op_ragged_and_empty_nodes(fe):
    """Regression (found by tools/fuzz_sift.py): an empty train node made the tile preload read before the slab.
    Ragged sizes around the 32-row / 128-column tiles, empty and single-row nodes, small max_matches."""
    from rgbdslam_v2_amd.frontend import inlier_indices
    rng = np.random.default_rng(13)
    sizes = [1, 127, 1000, 129, 300, 0, 33, 2]
    F = len(sizes)
    seq = synth.make_sequence(n_frames=F, n_kp=1000, n_world=3000, seed=313)
    sd = synth.sift_descriptors_like(seq["desc"], seed=13)
    nodes = [(sd[f][: sizes[f]].copy(), seq["xyz1"][f][: sizes[f]].copy()) for f in range(F)]
    nodes[3][0][1::2] = nodes[3][0][0::2][: len(nodes[3][0][1::2])]  # exact duplicates: tie rules
    for f in range(F):
        fe.upload_sift_node(f, *nodes[f])
    kw = dict(max_matches=5, min_matches=0, ransac_iterations=200)
    fe.set_params(**kw)
    try:
        pq = np.array([q for q in range(F) for t in range(F)], np.int32)
        pt = np.array([t for q in range(F) for t in range(F)], np.int32)
        out, dist = fe.match_sift_pair_list(pq, pt)
        prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov, **kw)
        for rec, dd, q, t in zip(out, dist, pq, pt):
            ref = po.match_sift_node_pair(nodes[q][0], nodes[q][1], int(q), nodes[t][0], nodes[t][1], int(t), prm)
            n = ref["n_all"]
            assert rec["n_all"] == n and np.array_equal(rec["all_q"][:n], ref["all_q"])
            assert np.array_equal(rec["all_t"][:n], ref["all_t"]) and np.array_equal(dd[:n], ref["all_dist"])
            assert (rec["id1"], rec["id2"], rec["n_inl"]) == (ref["id1"], ref["id2"], ref["n_inl"])
            assert np.array_equal(inlier_indices(rec), ref["inl_idx"])
            assert np.array_equal(np.array(rec["trafo"], np.float32).reshape(4, 4).T, ref["T"])
    finally:
        fe.set_params(max_matches=300, min_matches=20, ransac_iterations=200)
        for f in range(F):
            fe.release_node(f)


def test_sift_golden_vectors_from_the_reference_matcher(fe):
    """The HIP matcher against tests/golden/sift_golden.npz: real SIFT descriptors (box.siftgpu of the reference tree)
    matched by the reference's own matcher code (see tests/test_oracle_sift.py and tests/golden/make_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sift_golden.npz"))
    rng = np.random.default_rng(5)
    for name in ("view", "self", "dups"):
        d1 = g[name + "_d1"].astype(np.float32) / 512.0
        d2 = g[name + "_d2"].astype(np.float32) / 512.0
        fe.upload_sift_node(1, d1, _xyz(rng, len(d1)))
        fe.upload_sift_node(2, d2, _xyz(rng, len(d2)))
        mq, mt, md = fe.sift_match_nodes(1, 2)
        assert np.array_equal(mq, g[name + "_q"]) and np.array_equal(mt, g[name + "_t"]), name
        assert np.array_equal(md, g[name + "_dist"]), name
        fe.release_node(1)
        fe.release_node(2)


def test_sift_bench_step_properties():
    """configs[3] at the bench's size (2000 pairs of 1000 x 1000 descriptors, bench.py's sift sub-record) through
    properties that do not need the oracle on every pair: the records of a batch do not depend on its composition or order
    (a permuted list gives the permuted records, two half batches give the same records as one), every match list is
    mutual-best consistent and sorted by distance, and a sample of the pairs equals the oracle bit for bit."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    F = 100
    seq = synth.make_sequence(n_frames=F, n_kp=1000, seed=20260923, depth_noise=0.01)
    sd = synth.sift_descriptors_like(seq["desc"], seed=20260923)
    fe = FrontEnd(device_id=0, max_nodes=F, max_keypoints=1024, max_pairs_per_batch=2048)
    for f in range(F):
        fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
    pq, pt = synth.candidate_pairs(F, per_frame=20, seed=20260923)
    pq, pt = pq[:2000], pt[:2000]
    out, dist = fe.match_sift_pair_list(pq, pt)
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(pq))
    out_p, dist_p = fe.match_sift_pair_list(pq[perm], pt[perm])
    assert out_p.tobytes() == out[perm].tobytes() and np.array_equal(np.asarray(dist_p), np.asarray(dist)[perm])
    h = len(pq) // 2
    o1, d1 = fe.match_sift_pair_list(pq[:h], pt[:h])
    o2, d2 = fe.match_sift_pair_list(pq[h:], pt[h:])
    assert o1.tobytes() + o2.tobytes() == out.tobytes()
    assert np.array_equal(np.concatenate([np.asarray(d1), np.asarray(d2)]), np.asarray(dist))
    edges = 0
    for rec, dd in zip(out, dist):
        n = int(rec["n_all"])
        assert np.all(np.diff(np.asarray(dd)[:n]) >= 0)                    # keepStrongestMatches order
        assert len(set(rec["all_q"][:n].tolist())) == n and len(set(rec["all_t"][:n].tolist())) == n   # mutual best
        edges += rec["id1"] >= 0
    assert edges > 0.9 * len(pq)
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
    for k in rng.choice(len(pq), 12, replace=False):
        q, t = int(pq[k]), int(pt[k])
        ref = po.match_sift_node_pair(sd[q], seq["xyz1"][q], q, sd[t], seq["xyz1"][t], t, prm)
        rec = out[k]
        n = ref["n_all"]
        assert rec["n_all"] == n and np.array_equal(rec["all_q"][:n], ref["all_q"]) and np.array_equal(rec["all_t"][:n], ref["all_t"])
        assert np.array_equal(np.asarray(dist[k])[:n], ref["all_dist"])
        assert (rec["id1"], rec["id2"], rec["n_inl"]) == (ref["id1"], ref["id2"], ref["n_inl"])
        assert np.array_equal(np.array(rec["trafo"], np.float32).reshape(4, 4).T, ref["T"])
    fe.close()
