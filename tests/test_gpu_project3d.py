"""-m gpu: removeDepthless + projectTo3D kernel vs the oracle (float ops in the same order: exact)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def test_project_to_3d_matches_oracle():
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=64, max_pairs_per_batch=2)
    rng = np.random.default_rng(4)
    for (rows, cols, n, maxk, scale) in [(480, 640, 1500, 1000, 1.0), (480, 640, 300, 1000, 1.0),
                                         (960, 1280, 6000, 4000, 1.0), (48, 64, 700, 50, 0.5),
                                         (48, 64, 0, 10, 1.0)]:
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.15] = np.nan
        kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
        if n > 10:
            kp[3] = [np.nan, 5.0]
            kp[4] = [cols - 0.25, rows - 0.25]  # round() reaches rows/cols: clamped (reference reads OOB)
            kp[5] = [10.5, 20.5]                # half-way cases: round half away from zero
        fx = 525.0 * cols / 640
        kept, xyz = fe.project_to_3d(kp, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk)
        kept2, xyz2 = po.project_to_3d(kp, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk)
        assert np.array_equal(kept, kept2)
        assert np.array_equal(xyz, xyz2)
    fe.close()


def test_sift_node_features_match_oracle():
    """a20: projectTo3DSiftGPU + squareroot_descriptor_space through the C ABI vs the oracle: exact."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=64, max_pairs_per_batch=2)
    rng = np.random.default_rng(14)
    for (rows, cols, n, maxk, scale, root) in [(480, 640, 1500, 1000, 1.0, True), (480, 640, 300, 1000, 1.0, False),
                                               (960, 1280, 5000, 4000, 1.0, True), (48, 64, 700, 50, 0.5, True),
                                               (48, 64, 0, 10, 1.0, True)]:
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.15] = np.nan
        kp = np.stack([rng.uniform(0, cols - 0.01, n), rng.uniform(0, rows - 0.01, n)], 1).astype(np.float32)
        desc = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
        if n > 10:
            kp[3] = [np.nan, 5.0]               # undefined in the reference; both sides clamp to column 0
            kp[4] = [cols + 7.0, -3.0]          # outside: clamped
            kp[5] = [10.9999959, 20.5]          # node.cpp:733's own example
            desc[6] = 0.0
            desc[8] *= -1.0
        fx = 525.0 * cols / 640
        got = fe.sift_node_features(kp, desc, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk, root)
        ref = po.sift_node_features(kp, desc, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk, root)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
        if n and root:
            assert not np.array_equal(got[2], got[3])
    fe.close()
