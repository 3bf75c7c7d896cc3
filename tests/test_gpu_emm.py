"""-m gpu: depthToCV8UC1, createXYZRGBPointCloud and observationLikelihood kernels through the C ABI vs the
oracle -- bytes, float bits and integer counts must be identical."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from rgbdslam_v2_amd.frontend import FrontEnd
    f = FrontEnd(device_id=0, max_nodes=16, max_keypoints=64, max_pairs_per_batch=16)
    yield f
    f.close()


def test_depth_to_mono8_matches_oracle(fe):
    rng = np.random.default_rng(7)
    for rows, cols in ((480, 640), (47, 61)):
        d = rng.uniform(0, 4, (rows, cols)).astype(np.float32)
        d[rng.random((rows, cols)) < 0.1] = np.nan
        d[0, :8] = [np.nan, 0.005, 0.015, 0.025, 2.555, 2.56, 3e9, -np.inf]
        assert np.array_equal(fe.depth_to_mono8(d), po.depth_to_mono8(d))
        mm = rng.integers(0, 9000, (rows, cols)).astype(np.uint16)
        mm[0, :4] = [0, 500, 510, 65535]
        a, b = fe.depth_to_mono8(mm), po.depth_to_mono8(mm)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_point_cloud_matches_oracle(fe):
    rng = np.random.default_rng(8)
    for rows, cols, s, ch in ((480, 640, 2, 3), (480, 640, 1, 1), (960, 1280, 4, 3), (48, 64, 8, 0)):
        depth = rng.uniform(0.05, 5, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.1] = np.nan
        rgb = None if ch == 0 else rng.integers(0, 256, (rows, cols, 3) if ch == 3 else (rows, cols), dtype=np.uint8)
        f = 525.0 * cols / 640
        K = (f, f, (cols - 1) / 2, (rows - 1) / 2)
        for bgr in (False, True):
            got = fe.upload_node_cloud(1, depth, *K, rgb=rgb, encoding_bgr=bgr, depth_scaling=1.0, min_depth=0.4,
                                       cloud_skip=s, return_cloud=True)
            ref = po.create_point_cloud(depth, *K, rgb=rgb, encoding_bgr=bgr, depth_scaling=1.0, min_depth=0.4,
                                        cloud_skip=s)
            assert got.shape == ref.shape
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))  # float bits incl. NaN payload-free z
    fe.release_node_cloud(1)
    with pytest.raises(Exception):
        fe.upload_node_cloud(1, np.zeros((48, 64), np.float32), 50, 50, 32, 24, cloud_skip=5)


def _nan_equal_bits(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_observation_likelihood_matches_oracle(fe):
    F = 6
    seq = synth.make_depth_sequence(n_frames=F, nan_fraction=0.05)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    clouds = []
    for f in range(F):
        fe.upload_node_cloud(f, seq["depth"][f], *K, cloud_skip=2)
        clouds.append(po.create_point_cloud(seq["depth"][f], *K, cloud_skip=2))
    rng = np.random.default_rng(9)
    jobs = []
    for n in range(F):
        for o in range(F):
            T = synth.relative_pose(seq["poses"], n, o).astype(np.float32)
            jobs.append((n, o, T))
            Tp = T.copy()  # perturbed: mixes inliers, outliers and occluded points, and points leaving the raster
            Tp[:3, 3] += rng.normal(0, 0.08, 3).astype(np.float32)
            a = rng.normal(0, 0.03)
            Tp[:3, :3] = (Tp[:3, :3] @ np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])).astype(np.float32)
            jobs.append((n, o, Tp))
    jobs.append((0, 1, np.diag([1, 1, -1, 1]).astype(np.float32)))       # everything behind the camera
    jobs.append((0, 1, np.full((4, 4), np.nan, np.float32)))             # NaN transform
    for skip in (8, 3, 1):
        got = fe.observation_likelihood([j[0] for j in jobs], [j[1] for j in jobs], np.stack([j[2] for j in jobs]), skip)
        classes = np.zeros(3, np.int64)
        for (n, o, T), g in zip(jobs, got):
            ref = po.observation_likelihood(clouds[n], clouds[o], T, *K, cloud_skip=2, skip_step=skip,
                                            depth_cov=fe.params.depth_cov)
            assert list(g) == list(ref), (n, o, skip)
            classes += ref[:3].astype(np.int64)
        assert np.all(classes > 0)  # the job list exercises all three classes
    # emm__skip_step <= 0 -> (1, 0, 0, 1)
    assert fe.observation_likelihood([0], [1], np.eye(4)[None], -1).tolist() == [[1, 0, 0, 1]]
    # other depth covariance (the denominator of the cdf argument)
    fe.set_params(depth_cov=2.5e-5)
    got = fe.observation_likelihood([2], [0], synth.relative_pose(seq["poses"], 2, 0)[None], 8)
    ref = po.observation_likelihood(clouds[2], clouds[0], synth.relative_pose(seq["poses"], 2, 0), *K, cloud_skip=2,
                                    skip_step=8, depth_cov=2.5e-5)
    assert list(got[0]) == list(ref)
    fe.set_params(depth_cov=1e-4)
    with pytest.raises(Exception):
        fe.observation_likelihood([0], [99], np.eye(4)[None], 8)  # no cloud for node 99
    for f in range(F):
        fe.release_node_cloud(f)


def test_pairwise_likelihood_gates_ransac_edges(fe):
    """matchNodePair with observability_threshold > 0 (node.cpp:1340-1343): a correct RANSAC edge passes the
    criterion, an edge whose transform contradicts the depth images does not."""
    from rgbdslam_v2_amd._lib import RESULT_DTYPE
    seq = synth.make_depth_sequence(n_frames=2)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    for f in range(2):
        fe.upload_node_cloud(f, seq["depth"][f], *K, cloud_skip=2)
    res = np.zeros(2, RESULT_DTYPE)
    T = synth.relative_pose(seq["poses"], 1, 0).astype(np.float32)
    bad = T.copy(); bad[2, 3] -= 0.6
    for r, t in zip(res, (T, bad)):
        r["id1"], r["id2"] = 0, 1          # older, newer (node.cpp:1337-1338)
        r["trafo"] = t.T.reshape(-1)       # column-major
    counts, met = fe.pairwise_observation_likelihood(res, observability_threshold=0.6, emm_skip_step=8)
    assert met.tolist() == [True, False]
    assert counts[0, 0] > counts[0, 1] and counts[1, 1] > counts[1, 0]
    # both directions, as the reference sums them
    clouds = [po.create_point_cloud(d, *K, cloud_skip=2) for d in seq["depth"]]
    a = po.observation_likelihood(clouds[1], clouds[0], T, *K, cloud_skip=2, skip_step=8)
    b = po.observation_likelihood(clouds[0], clouds[1], np.linalg.inv(T.astype(np.float64)).astype(np.float32), *K,
                                  cloud_skip=2, skip_step=8)
    assert list(counts[0]) == list(a + b)
    for f in range(2):
        fe.release_node_cloud(f)
