"""-m gpu: the g2o refinement after RANSAC (params.g2o_iterations = "g2o_transformation_refinement"; node.cpp:1222-1268,
transformation_estimation.cpp:37-170) -- g2o_refine_kernel vs the oracle's restatement.  The kernel follows the oracle
operation by operation (same partial-sum order), so poses, inlier sets and errors are compared bit for bit, and within the
1e-4 pose tolerance of north_star."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import inlier_indices

pytestmark = pytest.mark.gpu
K = (521.0, 521.0, 319.5, 239.5)  # transformation_estimation.cpp:56


def keypoints_of(xyz1, rng, noise=0.2):
    u = K[0] * xyz1[:, 0] / xyz1[:, 2] + K[2]
    v = K[1] * xyz1[:, 1] / xyz1[:, 2] + K[3]
    return (np.stack([u, v], 1) + rng.normal(0, noise, (len(xyz1), 2))).astype(np.float32)


@pytest.mark.parametrize("iters,mode", [(1, 0), (3, 1 << 20), (8, 1 << 20)])
def test_pairs_with_g2o_refinement_match_oracle(iters, mode):
    from rgbdslam_v2_amd.frontend import FrontEnd
    seq = synth.make_sequence(n_frames=10, n_kp=600, n_world=2000, seed=55, depth_noise=0.004)
    rng = np.random.default_rng(9)
    kps = [keypoints_of(seq["xyz1"][f], rng) for f in range(10)]
    seq["desc"][9] = rng.integers(0, 256, seq["desc"][9].shape, dtype=np.uint8)   # an unrelated node: no edge, no refinement
    fe = FrontEnd(device_id=0, max_nodes=12, max_keypoints=640, max_pairs_per_batch=64, g2o_iterations=iters)
    fe.set_latency_mode(mode, 0)
    for f in range(10):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        fe.upload_node_keypoints(f, kps[f])
    pq = np.array([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 5], np.int32)
    pt = np.array([0, 1, 2, 0, 4, 3, 6, 7, 0, 8, 1], np.int32)
    out = fe.match_pair_list(pq, pt)
    plain = FrontEnd(device_id=0, max_nodes=12, max_keypoints=640, max_pairs_per_batch=64)
    plain.set_latency_mode(mode, 0)
    for f in range(10):
        plain.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    base = plain.match_pair_list(pq, pt)
    plain.close()
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
    changed = 0
    for rec, b, q, t in zip(out, base, pq, pt):
        ref = po.match_node_pair_g2o(seq["desc"][q], seq["xyz1"][q], kps[q], int(q), seq["desc"][t], seq["xyz1"][t], kps[t],
                                     int(t), iters, prm)
        assert rec["n_all"] == ref["n_all"] and rec["n_inl"] == ref["n_inl"], (q, t)
        assert (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
        assert np.array_equal(inlier_indices(rec), ref["inl_idx"])
        T = np.array(rec["trafo"], np.float32).reshape(4, 4).T
        assert np.abs(T - ref["T"]).max() <= 1e-4, (q, t, np.abs(T - ref["T"]).max())
        assert np.array_equal(T, ref["T"]) and rec["rmse"] == np.float32(ref["rmse"])
        assert rec["valid_iterations"] == ref["valid_iterations"]
        assert rec["info_scale"] == ref["info_scale"]
        changed += int(rec["valid_iterations"] != b["valid_iterations"])
        if rec["valid_iterations"] == b["valid_iterations"]:      # refinement not adopted: the RANSAC result stands
            assert rec.tobytes() == b.tobytes()
    assert changed >= 1, changed       # the refinement is adopted for some pairs (it must not lose inliers, :1252)
    assert out[9]["id1"] == -1
    fe.close()


def test_g2o_needs_keypoints():
    from rgbdslam_v2_amd._lib import RgbdfeError
    from rgbdslam_v2_amd.frontend import FrontEnd
    seq = synth.make_sequence(n_frames=3, n_kp=200, n_world=600, seed=1)
    fe = FrontEnd(device_id=0, max_nodes=4, max_keypoints=256, max_pairs_per_batch=8, g2o_iterations=2)
    for f in range(3):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    with pytest.raises(RgbdfeError, match="keypoints"):
        fe.match_pair_list([1], [0])
    with pytest.raises(RgbdfeError):
        fe.upload_node_keypoints(0, np.zeros((5, 2), np.float32))     # wrong count
    # keypoints belong to one upload of one node -- a re-upload, or a new node in a recycled slot, must not be
    # refined against the slot's previous KeyPoint.pt
    rng = np.random.default_rng(3)
    kps = [keypoints_of(seq["xyz1"][f], rng) for f in range(3)]
    for f in range(3):
        fe.upload_node_keypoints(f, kps[f])
    ok = fe.match_pair_list([1, 2], [0, 1])
    fe.upload_node(1, seq["desc"][1], seq["xyz1"][1])                  # re-upload in place: keypoints gone
    with pytest.raises(RgbdfeError, match="keypoints"):
        fe.match_pair_list([1], [0])
    fe.upload_node_keypoints(1, kps[1])
    assert fe.match_pair_list([1, 2], [0, 1]).tobytes() == ok.tobytes()
    fe.release_node(2)
    fe.upload_node(7, seq["desc"][2], seq["xyz1"][2])                  # a new node takes the recycled slot
    with pytest.raises(RgbdfeError, match="keypoints"):
        fe.match_pair_list([7], [1])
    fe.upload_node_keypoints(7, kps[2])
    fe.match_pair_list([7], [1])                                       # (another node id = another RANSAC stream)
    fe.release_node(7)
    fe.upload_node(2, seq["desc"][2], seq["xyz1"][2])                  # node 2 again, into the slot node 7 just left
    with pytest.raises(RgbdfeError, match="keypoints"):
        fe.match_pair_list([2], [1])
    fe.upload_node_keypoints(2, kps[2])
    assert fe.match_pair_list([1, 2], [0, 1]).tobytes() == ok.tobytes()
    fe.set_params(g2o_iterations=0)
    assert fe.match_pair_list([1], [0])["n_all"][0] > 0
    fe.close()
