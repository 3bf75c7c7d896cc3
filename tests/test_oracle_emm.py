"""CPU: the oracle's restatement of the frame-level functions either side of the pair path
(SURVEY.md 8(f) rows 3 and 2): depthToCV8UC1, createXYZRGBPointCloud, observationLikelihood."""
import math

import numpy as np

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth


def test_depth_to_mono8():
    rng = np.random.default_rng(3)
    d = rng.uniform(0, 4, (48, 64)).astype(np.float32)
    d[0, :8] = [np.nan, 0.005, 0.015, 0.025, 2.555, 2.56, 10.0, -1.0]  # ties round to even, saturation
    m = po.depth_to_mono8(d)
    t = (d * np.float32(100)).astype(np.float32)
    exp = np.where(np.isnan(t), 0, np.clip(np.rint(np.nan_to_num(t)), 0, 255)).astype(np.uint8)
    assert np.array_equal(m, exp)
    assert list(m[0, :8]) == [0, 0, 2, 2, 255, 255, 255, 0]
    mm = rng.integers(0, 9000, (48, 64)).astype(np.uint16)
    mm[0, :4] = [0, 500, 510, 65535]
    m8, dm = po.depth_to_mono8(mm)
    t = (mm.astype(np.float32) * np.float32(0.05) + np.float32(-25)).astype(np.float32)
    assert np.array_equal(m8, np.clip(np.rint(t), 0, 255).astype(np.uint8))
    assert np.array_equal(dm, (mm.astype(np.float32) * np.float32(0.001)).astype(np.float32))


def test_create_point_cloud_closed_form():
    rng = np.random.default_rng(4)
    rows, cols, s = 48, 64, 2
    depth = rng.uniform(0.05, 4, (rows, cols)).astype(np.float32)
    depth[rng.random((rows, cols)) < 0.1] = np.nan
    rgb = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    fx, fy, cx, cy = 52.5, 52.5, 31.5, 23.5
    for bgr in (False, True):
        c = po.create_point_cloud(depth, fx, fy, cx, cy, rgb=rgb, encoding_bgr=bgr, min_depth=0.1, cloud_skip=s)
        assert c.shape == (rows // s, cols // s, 4)
        v, u = np.mgrid[0:rows:s, 0:cols:s]
        Z = depth[::s, ::s]
        valid = Z >= np.float32(0.1)  # NaN -> False
        fxinv, fyinv = np.float32(1.0 / np.float32(fx)), np.float32(1.0 / np.float32(fy))
        x = ((u.astype(np.float32) - np.float32(cx)) * Z * fxinv).astype(np.float32)
        y = ((v.astype(np.float32) - np.float32(cy)) * Z * fyinv).astype(np.float32)
        assert np.array_equal(c[..., 0][valid], x[valid]) and np.array_equal(c[..., 1][valid], y[valid])
        assert np.array_equal(c[..., 2][valid], Z[valid]) and np.all(np.isnan(c[..., 2][~valid]))
        x1 = ((u.astype(np.float32) - np.float32(cx)).astype(np.float64) * fxinv.astype(np.float64)).astype(np.float32)
        assert np.array_equal(c[..., 0][~valid], x1[~valid])  # "as at 1 meter" (misc.cpp:527)
        bits = c[..., 3].copy().view(np.uint32)
        px = rgb[::s, ::s].astype(np.uint32)
        r, b = (px[..., 2], px[..., 0]) if bgr else (px[..., 0], px[..., 2])
        exp = b | (px[..., 1] << 8) | (r << 16)
        exp[0, 0] = 0  # quirk: color_idx > 0 (misc.cpp:536)
        assert np.array_equal(bits, exp)
    g = po.create_point_cloud(depth, fx, fy, cx, cy, rgb=rgb[..., 0].copy(), cloud_skip=4)
    assert g.shape == (12, 16, 4)
    assert g[1, 1, 3:4].view(np.uint32)[0] == int(rgb[4, 4, 0]) * 0x010101


def test_erf_boundaries_reproduce_the_cdf_tests():
    lo, hi = po.emm_erf_boundaries()
    assert abs(lo + hi) < 1e-12 and 2.18 < hi < 2.19
    def direct(q):
        p = 0.5 * (1 + math.erf(q))
        return 0 if p < 0.001 else (1 if p < 0.999 else 2)
    def by_boundary(q):
        return 0 if q < lo else (1 if q < hi else 2)
    rng = np.random.default_rng(5)
    qs = list(rng.normal(0, 2.5, 20000))
    for b in (lo, hi):
        q = b
        for _ in range(200):
            q = math.nextafter(q, -math.inf)
        for _ in range(400):
            qs.append(q)
            q = math.nextafter(q, math.inf)
    qs += [math.inf, -math.inf, 0.0]
    assert all(direct(q) == by_boundary(q) for q in qs)
    assert direct(math.nan) == 2 and by_boundary(math.nan) == 2  # NaN falls through to "bad" both ways


def test_observation_likelihood_semantics():
    seq = synth.make_depth_sequence(n_frames=3, width=160, height=120)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    clouds = [po.create_point_cloud(d, *K, cloud_skip=2) for d in seq["depth"]]
    n_valid = lambda c: int(np.isfinite(c[::4, ::4, 2]).sum())
    # the same frame against itself: every finite sampled point is an inlier
    c = po.observation_likelihood(clouds[0], clouds[0], np.eye(4), *K, cloud_skip=2, skip_step=4)
    assert c[3] == 15 * 20 and c[1] == 0 and c[2] == 0 and c[0] >= 0.9 * n_valid(clouds[0])
    # consecutive frames with the true relative pose: overwhelmingly inliers
    T = synth.relative_pose(seq["poses"], 1, 0)
    c = po.observation_likelihood(clouds[1], clouds[0], T, *K, cloud_skip=2, skip_step=4)
    assert c[0] > 0.8 * (c[0] + c[1] + c[2]) and c[0] > 100
    ok, q = po.observation_criterion_met(c[0], c[1], c[0] + c[1] + c[2], 0.6)
    assert ok and q > 0.8
    # a transform that pushes the new cloud 0.5 m towards the old camera: it would have blocked the view -> outliers
    Tb = T.copy(); Tb[2, 3] -= 0.5
    c = po.observation_likelihood(clouds[1], clouds[0], Tb, *K, cloud_skip=2, skip_step=4)
    assert c[1] > 0.8 * (c[0] + c[1] + c[2])
    ok, _ = po.observation_criterion_met(c[0], c[1], c[0] + c[1] + c[2], 0.6)
    assert not ok
    # 0.5 m behind the old surface: occluded
    Tf = T.copy(); Tf[2, 3] += 0.5
    c = po.observation_likelihood(clouds[1], clouds[0], Tf, *K, cloud_skip=2, skip_step=4)
    assert c[2] > 0.8 * (c[0] + c[1] + c[2])
    # emm__skip_step < 0 / unstructured clouds: (1, 0, 0, 1) (misc.cpp:829-843)
    assert list(po.observation_likelihood(clouds[1], clouds[0], T, *K, skip_step=-1)) == [1, 0, 0, 1]
    assert po.observation_criterion_met(0, 5, 5, -0.6)[0]  # negative threshold: always met (:1139)
