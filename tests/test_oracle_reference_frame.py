"""CPU: the oracle's frame-level functions against the reference's OWN code, compiled from /root/reference where it
lies into oracle/_ref/libref_frame.so (oracle/Makefile; OpenCV / PCL / ROS types are the stand-ins of
oracle/ref_stubs/frame_prelude.h).  Skipped when the pin is not built (no reference tree, no prebuilt .so)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

R = po.ref_frame_lib()
pytestmark = pytest.mark.skipif(R is None, reason="reference pin (oracle/_ref/libref_frame.so) not built")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_project_to_3d_and_remove_depthless_on_reference_code():
    """rows a4, a7: removeDepthless (node.cpp:66-97) and Node::projectTo3D (:900-965)."""
    rng = np.random.default_rng(31)
    for rows, cols, n, maxk, scale in ((480, 640, 1500, 1000, 1.0), (48, 64, 700, 50, 0.5), (48, 64, 200, 1000, 1.0)):
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.15] = np.nan
        # keep clear of round(y) == rows / round(x) == cols: the reference reads out of bounds there (oracle clamps)
        kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
        kp[(kp[:, 0] >= cols - 0.5) & (kp[:, 0] < cols), 0] = 5.25
        kp[(kp[:, 1] >= rows - 0.5) & (kp[:, 1] < rows), 1] = 7.75
        kp[3] = [np.nan, 5.0]
        kp[5] = [10.5, 20.5]  # ties: round half away from zero
        f = 525.0 * cols / 640
        K = (f, f * 1.01, (cols - 1) / 2, (rows - 1) / 2)
        kept = np.zeros(n, np.int32)
        xyz = np.zeros((n, 4), np.float32)
        k = R.ref_project_to_3d(_p(kp), n, _p(depth), rows, cols, *K, scale, maxk, _p(kept), _p(xyz))
        okept, oxyz = po.project_to_3d(kp, depth, *K, scale, maxk)
        assert k == len(okept) and np.array_equal(kept[:k], okept) and np.array_equal(xyz[:k], oxyz)
        # removeDepthless is the same filter without the cut and without the scaling
        k2 = R.ref_remove_depthless(_p(kp), n, _p(depth), rows, cols, _p(kept))
        okept2, _ = po.project_to_3d(kp, depth, *K, 1.0, 10 ** 9)
        assert np.array_equal(kept[:k2], okept2)


def test_sift_node_features_on_reference_code():
    """row a20: Node::projectTo3DSiftGPU (node.cpp:695-769) and squareroot_descriptor_space (:1557-1571)."""
    rng = np.random.default_rng(32)
    for rows, cols, n, maxk in ((480, 640, 1200, 1000), (48, 64, 300, 40)):
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.15] = np.nan
        kp = np.stack([rng.uniform(0, cols - 0.01, n), rng.uniform(0, rows - 0.01, n)], 1).astype(np.float32)
        kp[5] = [10.9999959, 20.5]
        desc = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
        desc[6] = 0.0
        desc[8] *= -1.0
        K = (525.0 * cols / 640, 520.0 * cols / 640, (cols - 1) / 2, (rows - 1) / 2)
        kept = np.zeros(n, np.int32)
        xyz = np.zeros((n, 4), np.float32)
        dout = np.zeros((n, 128), np.float32)
        sgpu = np.zeros((n, 128), np.float32)
        k = R.ref_project_to_3d_sift(_p(kp), n, _p(desc), _p(depth), rows, cols, *K, 1.0, maxk, _p(kept), _p(xyz),
                                     _p(dout), _p(sgpu))
        okept, oxyz, oraw, ofeat = po.sift_node_features(kp, desc, depth, *K, 1.0, maxk, use_root_sift=True)
        assert k == len(okept) and np.array_equal(kept[:k], okept) and np.array_equal(xyz[:k], oxyz)
        assert np.array_equal(dout[:k], oraw) and np.array_equal(sgpu[:k], oraw)
        feat = dout[:k].copy()
        R.ref_root_sift(_p(feat), k, 128)
        assert np.array_equal(feat, ofeat)


def test_sift_node_features_min_depth_on_reference_code():
    """node.cpp:727-731: projectTo3DSiftGPU with use_feature_min_depth -- the reference function with the parameter switched
    on (sizes = 12 * scale as SiftGPUWrapper::detect sets them) against the oracle's restatement."""
    if not hasattr(R, "ref_project_to_3d_sift_min_depth"):
        pytest.skip("oracle/_ref/libref_frame.so predates this entry point")
    rng = np.random.default_rng(35)
    for rows, cols, n, maxk, nanf in ((480, 640, 900, 1000, 0.3), (48, 64, 300, 40, 0.6), (48, 64, 200, 1000, 0.97)):
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < nanf] = np.nan
        depth[rng.random((rows, cols)) < 0.02] = 0.0           # a minimum of 0 counts as "no depth" (misc.cpp:790)
        kp = np.stack([rng.uniform(0, cols - 0.01, n), rng.uniform(0, rows - 0.01, n)], 1).astype(np.float32)
        kp[:4] = [[0.2, 0.3], [cols - 0.5, rows - 0.5], [cols / 2, 0.1], [0.4, rows / 2]]   # windows clipped by the image
        size = (12.0 * rng.uniform(0.8, 12.0, n)).astype(np.float32)
        size[7] = 1.0                                             # radius 0: an empty window
        desc = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
        K = (525.0 * cols / 640, 520.0 * cols / 640, (cols - 1) / 2, (rows - 1) / 2)
        kept = np.zeros(n, np.int32)
        xyz = np.zeros((n, 4), np.float32)
        k = R.ref_project_to_3d_sift_min_depth(_p(kp), _p(size), n, _p(desc), _p(depth), rows, cols, *K, 1.0, maxk, _p(kept),
                                               _p(xyz))
        okept, oxyz, oraw, _ = po.sift_node_features(kp, desc, depth, *K, 1.0, maxk, use_root_sift=False, kp_size=size)
        assert k == len(okept) and np.array_equal(kept[:k], okept) and np.array_equal(xyz[:k], oxyz)
        assert np.array_equal(oraw, desc[okept])
        plain = po.sift_node_features(kp, desc, depth, *K, 1.0, maxk, use_root_sift=False)
        assert len(plain[0]) != k or not np.array_equal(plain[1], oxyz)   # the variant matters on this input


def test_point_cloud_on_reference_code():
    """createXYZRGBPointCloud (misc.cpp:467-556)."""
    rng = np.random.default_rng(33)
    for rows, cols, s, ch in ((480, 640, 2, 3), (480, 640, 1, 1), (48, 64, 8, 3), (96, 128, 4, 3)):
        depth = rng.uniform(0.05, 5, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.1] = np.nan
        rgb = rng.integers(0, 256, (rows, cols, 3) if ch == 3 else (rows, cols), dtype=np.uint8)
        K = (525.0 * cols / 640, 522.0 * cols / 640, (cols - 1) / 2, (rows - 1) / 2)
        for bgr in (False, True):
            ref = np.zeros((rows // s, cols // s, 4), np.float32)
            R.ref_create_point_cloud(_p(depth), rows, cols, _p(rgb), ch, int(bgr), *K, 1.0, 0.4, s, _p(ref))
            got = po.create_point_cloud(depth, *K, rgb=rgb, encoding_bgr=bgr, depth_scaling=1.0, min_depth=0.4, cloud_skip=s)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_observation_likelihood_on_reference_code():
    """observationLikelihood (misc.cpp:814-969) and observation_criterion_met (:1136-1148)."""
    F = 5
    seq = synth.make_depth_sequence(n_frames=F, width=320, height=240, nan_fraction=0.05)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    clouds = [po.create_point_cloud(d, *K, cloud_skip=2) for d in seq["depth"]]
    rng = np.random.default_rng(34)
    classes = np.zeros(3, np.int64)
    for n in range(F):
        for o in range(F):
            T = synth.relative_pose(seq["poses"], n, o).astype(np.float32)
            Tp = T.copy()
            Tp[:3, 3] += rng.normal(0, 0.08, 3).astype(np.float32)
            a = rng.normal(0, 0.03)
            Tp[:3, :3] = (Tp[:3, :3] @ np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])).astype(np.float32)
            for TT in (T, Tp):
                for skip, dc in ((8, 1e-4), (3, 2.5e-5)):
                    ref = np.zeros(4, np.uint32)
                    TT = np.ascontiguousarray(TT, np.float32)
                    R.ref_observation_likelihood(_p(clouds[n]), _p(clouds[o]), clouds[o].shape[0], clouds[o].shape[1], _p(TT),
                                                 *K, 2, skip, dc, _p(ref))
                    got = po.observation_likelihood(clouds[n], clouds[o], TT, *K, cloud_skip=2, skip_step=skip, depth_cov=dc)
                    assert list(got) == list(ref), (n, o, skip)
                    classes += ref[:3].astype(np.int64)
    assert np.all(classes > 0)
    q1, q2 = C.c_double(0), 0.0
    for inl, outl, occ, th in ((100, 10, 5, 0.6), (10, 100, 5, 0.6), (30, 1, 200, 0.6), (0, 5, 5, -0.6), (50, 50, 0, 0.5)):
        a = R.ref_observation_criterion_met(inl, outl, occ + inl + outl, th, C.byref(q1))
        b, q2 = po.observation_criterion_met(inl, outl, occ + inl + outl, th)
        assert bool(a) == b and (th < 0 or q1.value == q2)


def test_project_to_3d_cloud_on_reference_code():
    """row a22 (i): Node::projectTo3D, point-cloud overload (node.cpp:855-898): truncating lookup, maximum_depth,
    NaN coordinates, the max_keypoints cut."""
    rng = np.random.default_rng(41)
    for rows, cols, n, maxk, maxd in ((480, 640, 1500, 1000, 3.5), (48, 64, 700, 50, 2.0), (48, 64, 300, 1000, 1e9),
                                      (48, 64, 300, 1000, -1.0)):
        cloud = np.zeros((rows, cols, 4), np.float32)
        cloud[..., 0] = rng.uniform(-2, 2, (rows, cols))
        cloud[..., 1] = rng.uniform(-2, 2, (rows, cols))
        cloud[..., 2] = rng.uniform(0.4, 5.0, (rows, cols))
        cloud[..., 3] = rng.uniform(0, 1, (rows, cols))
        for ch in range(3):
            cloud[..., ch][rng.random((rows, cols)) < 0.05] = np.nan
        kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
        kp[3] = [np.nan, 5.0]
        kp[5] = [10.9999959, 20.5]  # the coordinates the reference's comment is about: truncation, not rounding
        kept = np.zeros(n, np.int32)
        xyz = np.zeros((n, 4), np.float32)
        k = R.ref_project_to_3d_cloud(_p(kp), n, _p(cloud), rows, cols, maxd, maxk, _p(kept), _p(xyz))
        okept, oxyz = po.project_to_3d_cloud(kp, cloud, maxd, maxk)
        assert k == len(okept) and np.array_equal(kept[:k], okept) and np.array_equal(xyz[:k], oxyz)
        assert maxd < 0 or k > 0


def test_use_feature_min_depth_on_reference_code():
    """The default-off variant of rows a4 / a7 ("use_feature_min_depth", parameter_server.cpp:90): getMinDepthInNeighborhood
    (misc.cpp:774-793) compiled from the source, inside removeDepthless (node.cpp:82) and Node::projectTo3D (:940) with
    the parameter switched on.  (cv::minMaxLoc is a stand-in: its NaN / empty-window behaviour is restated.)"""
    rng = np.random.default_rng(33)
    R.ref_min_depth_in_neighborhood.restype = C.c_float
    R.ref_min_depth_in_neighborhood.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    for rows, cols, n, maxk, scale in ((120, 160, 600, 1000, 1.0), (48, 64, 300, 40, 0.5)):
        depth = rng.uniform(0.4, 5.0, (rows, cols)).astype(np.float32)
        depth[rng.random((rows, cols)) < 0.3] = np.nan
        depth[10:40, 20:50] = np.nan            # windows without a single valid depth
        depth[5, 7] = 0.0                       # a zero minimum becomes NaN (:786-789)
        kp = np.stack([rng.uniform(-3, cols + 3, n), rng.uniform(-3, rows + 3, n)], 1).astype(np.float32)
        size = (31.0 * 1.2 ** rng.integers(0, 8, n)).astype(np.float32)
        size[:20] = [1.0, 2.0, 2.9, 3.0, 0.5] * 4   # radius 0: an empty window
        kp[7] = [7.2, 5.4]
        for i in range(0, n, 7):
            a = R.ref_min_depth_in_neighborhood(_p(depth), rows, cols, float(kp[i, 0]), float(kp[i, 1]), float(size[i]))
            b = po.min_depth_in_neighborhood(depth, float(kp[i, 0]), float(kp[i, 1]), float(size[i]))
            assert (np.isnan(a) and np.isnan(b)) or np.float32(a) == np.float32(b)
        f = 525.0 * cols / 640
        K = (f, f * 1.01, (cols - 1) / 2, (rows - 1) / 2)
        kept = np.zeros(n, np.int32)
        xyz = np.zeros((n, 4), np.float32)
        k = R.ref_project_to_3d_min_depth(_p(kp), _p(size), n, _p(depth), rows, cols, C.c_double(K[0]), C.c_double(K[1]),
                                          C.c_double(K[2]), C.c_double(K[3]), C.c_double(scale), maxk, _p(kept), _p(xyz))
        okept, oxyz = po.project_to_3d_min_depth(kp, size, depth, *K, scale, maxk)
        assert k == len(okept) and np.array_equal(kept[:k], okept) and np.array_equal(xyz[:k], oxyz)
        assert 0 < k <= maxk
        k2 = R.ref_remove_depthless_min_depth(_p(kp), _p(size), n, _p(depth), rows, cols, _p(kept))
        okept2 = po.remove_depthless_min_depth(kp, size, depth)
        assert np.array_equal(kept[:k2], okept2)
        # and it differs from the plain lookup (more keypoints survive: a NaN pixel has valid neighbours)
        plain, _ = po.project_to_3d(kp, depth, *K, 1.0, 10 ** 9)
        assert len(okept2) != len(plain)
