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


def test_sift_node_features_min_depth_match_oracle():
    """node.cpp:727-731: projectTo3DSiftGPU with use_feature_min_depth (rgbdfe_sift_node_features_min_depth): exact."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=64, max_pairs_per_batch=2)
    rng = np.random.default_rng(15)
    for (rows, cols, n, maxk, scale, nanf) in [(480, 640, 1500, 1000, 1.0, 0.3), (960, 1280, 3000, 4000, 1.0, 0.5),
                                               (48, 64, 700, 50, 0.5, 0.6), (48, 64, 200, 1000, 1.0, 0.97),
                                               (48, 64, 0, 10, 1.0, 0.1)]:
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < nanf] = np.nan
        depth[rng.random((rows, cols)) < 0.02] = 0.0
        kp = np.stack([rng.uniform(0, cols - 0.01, n), rng.uniform(0, rows - 0.01, n)], 1).astype(np.float32)
        size = (12.0 * rng.uniform(0.8, 12.0, n)).astype(np.float32)
        desc = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
        if n > 10:
            kp[:4] = [[0.2, 0.3], [cols - 0.5, rows - 0.5], [cols / 2, 0.1], [0.4, rows / 2]]
            size[7] = 1.0
        fx = 525.0 * cols / 640
        got = fe.sift_node_features(kp, desc, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk, True, kp_size=size)
        ref = po.sift_node_features(kp, desc, depth, fx, fx, (cols - 1) / 2, (rows - 1) / 2, scale, maxk, True, kp_size=size)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    fe.close()


def _cloud(rng, rows, cols):
    cloud = np.zeros((rows, cols, 4), np.float32)
    cloud[..., 0] = rng.uniform(-2, 2, (rows, cols))
    cloud[..., 1] = rng.uniform(-2, 2, (rows, cols))
    cloud[..., 2] = rng.uniform(0.4, 5.0, (rows, cols))
    for ch in range(3):
        cloud[..., ch][rng.random((rows, cols)) < 0.05] = np.nan
    return cloud


@pytest.mark.parametrize("rows,cols,n,maxk,maxd", [(480, 640, 1500, 1000, 3.5), (48, 64, 700, 50, 2.0),
                                                   (48, 64, 300, 1000, 1e9), (48, 64, 300, 1000, -1.0), (48, 64, 0, 10, 3.0)])
def test_project_to_3d_cloud_matches_oracle(rows, cols, n, maxk, maxd):
    """row a22 (i): Node::projectTo3D's point-cloud overload (node.cpp:855-898) on the device."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(rows * 31 + n)
    cloud = _cloud(rng, rows, cols)
    kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
    if n > 5:
        kp[3] = [np.nan, 5.0]
        kp[5] = [10.9999959, 20.5]
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=64, max_pairs_per_batch=2)
    kept, xyz = fe.project_to_3d_cloud(kp, cloud, maxd, maxk)
    okept, oxyz = po.project_to_3d_cloud(kp.reshape(-1, 2), cloud, maxd, maxk)
    assert np.array_equal(kept, okept) and np.array_equal(xyz, oxyz)
    fe.close()


def test_point_cloud_constructor_feature_path():
    """node.cpp:252-369: detect -> projectTo3D(cloud) -> compute (no removeDepthless, no retainBest); the 3-D points
    follow their keypoints through compute()'s border filter and regrouping (deviation D6)."""
    from oracle import pyorb
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    fr = synth.make_image_sequence(n_frames=2, seed=9)
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=1024, max_pairs_per_batch=2)
    fe.detector_configure(max_keypoints=800)
    st = pyorb.grid_state(800)
    rng = np.random.default_rng(2)
    for f in range(2):
        g, d = fr["gray"][f], fr["depth"][f]
        m = np.where(fr["mask"][f] > 0, 255, 0).astype(np.uint8)
        rows, cols = g.shape
        u, v = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
        cloud = np.zeros((rows, cols, 4), np.float32)
        cloud[..., 2] = d
        cloud[..., 0] = (u - fr["cx"]) * d / fr["fx"]
        cloud[..., 1] = (v - fr["cy"]) * d / fr["fy"]
        cloud[..., 2][rng.random((rows, cols)) < 0.03] = np.nan
        maxd = float(np.nanmedian(d)) + (1.0 if f == 0 else 0.0)   # frame 1: about half of the plane is too far away
        kp, desc, xyz = fe.detect_describe_cloud(g, m, cloud, maxd)
        det = pyorb.grid_detect(st, g, m)
        kept, pxyz = po.project_to_3d_cloud(np.stack([det["x"], det["y"]], 1), cloud, maxd, 800)
        k3 = det[kept]
        # compute(): border filter + stable regroup by octave; carry the positions along
        inside = (k3["x"] >= 31) & (k3["x"] < cols - 31) & (k3["y"] >= 31) & (k3["y"] < rows - 31)
        order = np.concatenate([np.flatnonzero(inside & (k3["octave"] == lv)) for lv in range(8)])
        rk, rdesc = pyorb.compute(g, k3)
        assert len(rk) == len(order) == len(kp) and 0 < len(kp) <= 800
        for fld in ("x", "y", "octave", "size", "response", "angle"):
            assert np.array_equal(kp[fld], rk[fld]), fld
        assert np.array_equal(desc, rdesc)
        assert np.array_equal(xyz, pxyz[order])
        assert np.array_equal(fe.detector_thresholds(), np.array(st.thresh[:9]))
    fe.close()


def test_use_feature_min_depth_mode():
    """The default-off variant of rows a4 / a7 (parameter "use_feature_min_depth"): the neighbourhood minimum of
    getMinDepthInNeighborhood (misc.cpp:774-793) on the device, in the A/B entry point and in the whole Node::Node feature
    path (removeDepthless, retainBest, compute, projectTo3D with the keypoints' own sizes)."""
    from oracle import pyorb
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(41)
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=1024, max_pairs_per_batch=8)
    for rows, cols, n, maxk, scale in ((480, 640, 1500, 1000, 1.0), (120, 160, 600, 1000, 0.5), (48, 64, 300, 40, 1.0)):
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.3] = np.nan
        depth[10:40, 20:50] = np.nan
        depth[5, 7] = 0.0
        kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
        size = (31.0 * 1.2 ** rng.integers(0, 8, n)).astype(np.float32)
        size[:10] = [1.0, 2.0, 2.9, 3.0, 0.5] * 2
        f = 525.0 * cols / 640
        K = (f, f * 1.01, (cols - 1) / 2, (rows - 1) / 2)
        kept, xyz = fe.project_to_3d_min_depth(kp, size, depth, *K, scale, maxk)
        okept, oxyz = po.project_to_3d_min_depth(kp, size, depth, *K, scale, maxk)
        assert np.array_equal(kept, okept) and np.array_equal(xyz, oxyz) and 0 < len(kept) <= maxk
    # the frame path: depth with holes, so that the two modes keep different keypoints
    fr = synth.make_image_sequence(n_frames=3, seed=5)
    K = (fr["fx"], fr["fy"], fr["cx"], fr["cy"])
    fe.detector_configure(max_keypoints=1000)
    fe.set_feature_min_depth(True)
    pyorb.set_use_feature_min_depth(True)
    try:
        st = pyorb.grid_state(1000)
        for f in range(3):
            g = fr["gray"][f]
            d = fr["depth"][f].copy()
            d[rng.random(d.shape) < 0.2] = np.nan
            m = np.full(g.shape, 255, np.uint8)
            kp, desc, xyz = fe.detect_describe(g, m, d, *K)
            rk, rdesc = pyorb.node_features(st, g, m, d, 1000)
            assert len(kp) == len(rk) and np.array_equal(kp["x"], rk["x"]) and np.array_equal(kp["y"], rk["y"])
            assert np.array_equal(desc, rdesc)
            okept, oxyz = po.project_to_3d_min_depth(np.stack([rk["x"], rk["y"]], 1), rk["size"], d, *K, 1.0, 1000)
            assert len(okept) == len(rk) and np.array_equal(xyz, oxyz)
            plain, _ = po.project_to_3d(np.stack([rk["x"], rk["y"]], 1), d, *K, 1.0, 1000)
            assert len(plain) < len(rk)          # the neighbourhood depth rescues keypoints on NaN pixels
    finally:
        pyorb.set_use_feature_min_depth(False)
        fe.set_feature_min_depth(False)
    kp2, _, _ = fe.detect_describe(fr["gray"][0], np.full(fr["gray"][0].shape, 255, np.uint8), fr["depth"][0], *K)
    assert len(kp2) > 100
    fe.close()


def test_frame_golden_from_the_reference_functions():
    """The HIP frame-level kernels against tests/golden/frame_golden.npz: the outputs of the reference's own
    Node::projectTo3D / removeDepthless / projectTo3DSiftGPU / squareroot_descriptor_space on the same inputs
    (tests/golden/make_golden.py; see tests/test_oracle_frame_golden.py)."""
    import os
    from rgbdslam_v2_amd.frontend import FrontEnd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_golden.npz"))
    fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=1024, max_pairs_per_batch=4)
    for tag in "ab":
        kp, depth, K, maxk = g[f"p3d_{tag}_kp"], g[f"p3d_{tag}_depth"], g[f"p3d_{tag}_K"], int(g[f"p3d_{tag}_maxk"])
        kept, xyz = fe.project_to_3d(kp, depth, *[float(v) for v in K[:4]], float(K[4]), maxk)
        assert np.array_equal(kept, g[f"p3d_{tag}_kept"]) and np.array_equal(xyz, g[f"p3d_{tag}_xyz"])
    K = g["sift_K"]
    got = fe.sift_node_features(g["sift_kp"], g["sift_desc"], g["sift_depth"], *[float(v) for v in K[:4]], float(K[4]),
                                int(g["sift_maxk"]), True)
    assert np.array_equal(got[0], g["sift_kept"]) and np.array_equal(got[1], g["sift_xyz"])
    assert np.array_equal(got[2], g["sift_raw"]) and np.array_equal(got[3], g["sift_root"])
    # point-cloud constructor's projection and the use_feature_min_depth variant
    kept, xyz = fe.project_to_3d_cloud(g["cloudp_kp"], g["cloudp_cloud"], float(g["cloudp_maxd"]), int(g["cloudp_maxk"]))
    assert np.array_equal(kept, g["cloudp_kept"]) and np.array_equal(xyz, g["cloudp_xyz"])
    K = [float(v) for v in g["mind_K"]]
    kept, xyz = fe.project_to_3d_min_depth(g["mind_kp"], g["mind_size"], g["mind_depth"], *K[:4], K[4], 1000)
    assert np.array_equal(kept, g["mind_kept"]) and np.array_equal(xyz, g["mind_xyz"])
    # createXYZRGBPointCloud + observationLikelihood
    K = [float(v) for v in g["emm_K"]]
    fe2 = FrontEnd(device_id=0, max_nodes=4, max_keypoints=64, max_pairs_per_batch=4)
    for f in range(3):
        c = fe2.upload_node_cloud(f, g["emm_depth"][f], *K, rgb=g["emm_gray"][f], encoding_bgr=False, depth_scaling=1.0,
                                  min_depth=0.1, cloud_skip=2, return_cloud=True)
        assert np.array_equal(c.view(np.uint32), g["emm_clouds"][f].view(np.uint32))
    got = fe2.observation_likelihood(g["emm_jobs"][:, 0], g["emm_jobs"][:, 1], g["emm_T"], 8)
    assert np.array_equal(got, g["emm_counts"])
    fe2.close()
    fe.close()
