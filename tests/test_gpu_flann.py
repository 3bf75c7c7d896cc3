"""-m gpu: Node::featureMatching's FLANN branch for float descriptors (node.cpp:610-667) with exact neighbours --
l2_knn2_kernel / l2_ratio_kernel behind rgbdfe_match_flann_pair_list -- against oracle/rgbd_oracle.c::orc_flann_match /
orc_match_float_node_pair.  The squared distances are float sums in flann::L2's order: neighbours, ratios, match
lists, inlier sets and pose bits are identical."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import inlier_indices

pytestmark = pytest.mark.gpu


def _root_sift(rng, n, dim=128):
    v = rng.gamma(0.6, 1.0, (n, dim)).astype(np.float32)
    v /= np.abs(v).sum(1, keepdims=True)
    return np.sqrt(v).astype(np.float32)          # squareroot_descriptor_space (node.cpp:1557-1571)


@pytest.fixture(scope="module")
def seq():
    s = synth.make_sequence(n_frames=8, n_kp=700, n_world=2400, seed=77, depth_noise=synth.DEPTH_NOISE_R1)
    rng = np.random.default_rng(7)
    # float descriptors: one 128-d RootSIFT-like vector per world point, observed with a little noise; outliers random
    base = _root_sift(rng, 2400)
    desc = np.zeros((8, 700, 128), np.float32)
    for f in range(8):
        wid = s["world_id"][f]
        d = np.where(wid[:, None] >= 0, base[np.maximum(wid, 0)], _root_sift(rng, 700))
        desc[f] = d + rng.normal(0, 0.004, (700, 128)).astype(np.float32)
    return s, np.abs(desc).astype(np.float32)


@pytest.mark.parametrize("mode", [0, 1 << 20], ids=["one_wave_per_pair", "record_replay"])
def test_flann_branch_pairs_match_oracle(seq, mode):
    from rgbdslam_v2_amd.frontend import FrontEnd
    s, desc = seq
    fe = FrontEnd(device_id=0, max_nodes=12, max_keypoints=704, max_pairs_per_batch=64)
    fe.set_latency_mode(mode, 0)
    for f in range(8):
        fe.upload_float_node(f, desc[f], s["xyz1"][f])
    pq = np.array([1, 2, 3, 4, 5, 6, 7, 7, 5], np.int32)
    pt = np.array([0, 1, 2, 0, 4, 3, 6, 0, 1], np.int32)
    for ratio in (0.95, 0.7):
        out, dist = fe.match_flann_pair_list(pq, pt, ratio)
        prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
        for rec, dd, q, t in zip(out, dist, pq, pt):
            ref = po.match_float_node_pair(desc[q], s["xyz1"][q], int(q), desc[t], s["xyz1"][t], int(t), ratio, prm)
            n = ref["n_all"]
            assert rec["n_all"] == n and n > 50
            assert np.array_equal(rec["all_q"][:n], ref["all_q"]) and np.array_equal(rec["all_t"][:n], ref["all_t"])
            assert np.array_equal(dd[:n], ref["all_dist"])
            assert rec["n_inl"] == ref["n_inl"] and (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
            assert np.array_equal(inlier_indices(rec), ref["inl_idx"])
            T = np.array(rec["trafo"], np.float32).reshape(4, 4).T
            assert np.abs(T - ref["T"]).max() <= 1e-4 and np.array_equal(T, ref["T"])
        assert (out["id1"] >= 0).sum() >= 6
    fe.close()


@pytest.mark.parametrize("nq,nt,dim", [(300, 500, 128), (257, 64, 64), (5, 2, 128), (9, 1, 128), (1, 3, 32)])
def test_flann_match_lists(nq, nt, dim):
    """knn-2, ratio test, train-unique first come first served; ragged sizes; 64-d (SURF) rows zero padded."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(nq * 7 + nt)
    t = _root_sift(rng, nt, dim)
    q = _root_sift(rng, nq, dim)
    k = min(nq, nt) // 2
    q[:k] = t[rng.integers(0, nt, k)] + rng.normal(0, 0.003, (k, dim)).astype(np.float32)   # duplicates of train rows too
    fe = FrontEnd(device_id=0, max_nodes=4, max_keypoints=512, max_pairs_per_batch=4, min_matches=0)
    xq = np.tile(np.array([[0, 0, 2, 1]], np.float32), (nq, 1))
    xt = np.tile(np.array([[0, 0, 2, 1]], np.float32), (nt, 1))
    fe.upload_float_node(1, q, xq)
    fe.upload_float_node(2, t, xt)
    for ratio in (0.95, 0.5, 2.0):
        out, dist = fe.match_flann_pair_list([1], [2], ratio)
        mq, mt, md = po.flann_match(q, t, ratio)
        # the result POD lists the matches sorted by (ratio, queryIdx), at most max_matches of them
        o = np.lexsort((mq, md))[:300]
        n = int(out[0]["n_all"])
        assert n == len(o)
        assert np.array_equal(out[0]["all_q"][:n], mq[o]) and np.array_equal(out[0]["all_t"][:n], mt[o])
        assert np.array_equal(dist[0][:n], md[o])
        assert len(set(mt.tolist())) == len(mt)
    fe.close()


def test_wrong_node_kind_is_refused():
    from rgbdslam_v2_amd._lib import RgbdfeError
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(0)
    fe = FrontEnd(device_id=0, max_nodes=4, max_keypoints=64, max_pairs_per_batch=4)
    x = np.tile(np.array([[0, 0, 2, 1]], np.float32), (10, 1))
    fe.upload_float_node(1, _root_sift(rng, 10), x)
    fe.upload_node(2, rng.integers(0, 256, (10, 32), dtype=np.uint8), x)
    with pytest.raises(RgbdfeError):
        fe.match_flann_pair_list([1], [2])
    with pytest.raises(RgbdfeError):
        fe.match_pair_list([1], [2])
    with pytest.raises(RgbdfeError):
        fe.upload_float_node(3, np.zeros((4, 30), np.float32), x[:4])   # dim not a multiple of 4
    fe.close()
