"""CPU checks of oracle/orb_oracle.c (the OpenCV-3.3 ORB restatement; "parity unpinned") against
independent numpy formulations of the same published algorithms."""
import numpy as np
import pytest

from oracle import pyorb
from rgbdslam_v2_amd import synth

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
          (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


@pytest.fixture(scope="module")
def img():
    return synth.make_image_sequence(n_frames=1, seed=8)["gray"][0]


def test_level_geometry_and_tables():
    sc, lw, lh = pyorb.level_geometry(640, 480)
    assert list(lw) == [640, 533, 444, 370, 309, 257, 214, 179]
    assert list(lh) == [480, 400, 333, 278, 231, 193, 161, 134]
    assert abs(sc[7] - 1.2 ** 7) < 1e-5
    # u_max of the radius-15 disc (orb.cpp) as every ORB implementation tabulates it
    assert list(pyorb.umax()[:16]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    k = pyorb.gauss_kernel()
    assert list(k) == [18, 34, 49, 55, 49, 34, 18] and list(k) == list(k[::-1])
    pat = pyorb.pattern().reshape(256, 4)
    assert pat.shape == (256, 4) and np.abs(pat).max() <= 13
    assert list(pat[0]) == [8, -3, 9, 5] and list(pat[255]) == [-1, -6, 0, -11]
    assert len({tuple(r) for r in pat}) == 256  # no duplicated test


def test_resize_against_float_bilinear(img):
    out = pyorb.resize(img, 533, 400)
    # float bilinear with OpenCV's pixel-centre convention; the fixed-point result stays within 1 LSB
    sx, sy = 640 / 533, 480 / 400
    xs = (np.arange(533) + 0.5) * sx - 0.5
    ys = (np.arange(400) + 0.5) * sy - 0.5
    x0 = np.clip(np.floor(xs).astype(int), 0, 638); y0 = np.clip(np.floor(ys).astype(int), 0, 478)
    fx = np.clip(xs - x0, 0, 1)[None, :]; fy = np.clip(ys - y0, 0, 1)[:, None]
    I = img.astype(np.float64)
    ref = (I[y0][:, x0] * (1 - fx) * (1 - fy) + I[y0][:, x0 + 1] * fx * (1 - fy) +
           I[y0 + 1][:, x0] * (1 - fx) * fy + I[y0 + 1][:, x0 + 1] * fx * fy)
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0
    assert np.array_equal(pyorb.resize(img, 640, 480), img)  # identity size: exact copy


def fast_numpy(img, t):
    h, w = img.shape
    I = img.astype(np.int32)
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])  # [16, h-6, w-6]
    c = I[3:h - 3, 3:w - 3]
    darker = ring < c - t
    brighter = ring > c + t

    def has_run(b):
        bb = np.concatenate([b, b[:9]], 0)
        run = np.zeros(b.shape[1:], bool)
        for s in range(16):
            run |= bb[s:s + 9].all(0)
        return run
    out = np.zeros((h, w), bool)
    out[3:h - 3, 3:w - 3] = has_run(darker) | has_run(brighter)
    return out


@pytest.mark.parametrize("t", [10, 30])
def test_fast_corner_test_and_score(img, t):
    sub = np.ascontiguousarray(img[100:260, 200:420])
    score = pyorb.fast_score_map(sub, t)
    corners = fast_numpy(sub, t)
    assert np.array_equal(score > 0, corners)
    assert corners.sum() > 50
    # the score is the largest threshold for which the pixel is still a corner
    ys, xs = np.nonzero(corners)
    for y, x in list(zip(ys, xs))[:40]:
        s = int(score[y, x])
        assert s >= t
        assert fast_numpy(sub, s)[y, x]
        assert not fast_numpy(sub, s + 1)[y, x]


def _closed_form_scores(a, t):
    """orb_fast_nms_kernel's formulation (csrc/orb_kernels.hip): with d = centre - ring, P = max over the 16 arcs of 9 of
    min d, N = the same for -d:  corner <=> max(P, N) > t,  score = max(P, N) - 1."""
    h, w = a.shape
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
              (-3, 1), (-2, 2), (-1, 3)]
    c = a[3:h - 3, 3:w - 3].astype(np.int32)
    d = np.stack([c - a[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in circle])
    P = np.max([np.min([d[(i + j) & 15] for j in range(9)], 0) for i in range(16)], 0)
    N = np.max([np.min([-d[(i + j) & 15] for j in range(9)], 0) for i in range(16)], 0)
    m = np.maximum(P, N)
    out = np.zeros((h, w), np.uint8)
    out[3:h - 3, 3:w - 3] = np.where(m > t, np.clip(m - 1, 0, 255), 0)
    return out


def test_fast_closed_form_equals_the_reference_loops(img):
    """The kernel does not run cv::FAST's counting loop and cornerScore<16>'s two pruned loops (fast.cpp, fast_score.cpp; restated
    in orb_fast_score_at) but their closed form: same integers for every pixel and threshold -- textured, flat, saturated
    and random images, thresholds from 0 to 255."""
    rng = np.random.default_rng(5)
    images = [np.ascontiguousarray(img[60:180, 100:260]),
              rng.integers(0, 256, (64, 96), dtype=np.uint8),                               # noise: many arcs, both polarities
              rng.integers(100, 104, (48, 64), dtype=np.uint8),                             # nearly flat
              (rng.integers(0, 2, (48, 64), dtype=np.uint8) * 255),                         # saturated differences (+-255)
              np.tile(np.arange(64, dtype=np.uint8) * 4, (48, 1))]                          # a ramp: long monotone arcs
    for a in images:
        for t in (0, 1, 2, 7, 20, 60, 127, 254, 255):
            assert np.array_equal(pyorb.fast_score_map(a, t), _closed_form_scores(a, t)), t


def test_nms_mask_border(img):
    sub = np.ascontiguousarray(img[:200, :300])
    score = pyorb.fast_score_map(sub, 15)
    mask = np.full(sub.shape, 255, np.uint8)
    mask[:, 150:] = 0
    kp = pyorb.fast_keypoints(score, mask, 15)
    S = score.astype(np.int32)
    exp = []
    for y in range(3, 197):
        for x in range(3, 297):
            s = S[y, x]
            if s == 0:
                continue
            nb = S[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s > nb.max() and mask[y, x] and 15 <= x < 285 and 15 <= y < 185:
                exp.append((x, y, s))
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kp] == exp
    assert len(exp) > 20


def test_harris_and_angle(img):
    I = img.astype(np.int64)
    for (x, y) in [(100, 80), (320, 240), (500, 400), (33, 47)]:
        a = b = c = 0
        for i in range(-3, 4):
            for j in range(-3, 4):
                p = lambda dx, dy: I[y + i + dy, x + j + dx]
                Ix = (p(1, 0) - p(-1, 0)) * 2 + (p(1, -1) - p(-1, -1)) + (p(1, 1) - p(-1, 1))
                Iy = (p(0, 1) - p(0, -1)) * 2 + (p(-1, 1) - p(-1, -1)) + (p(1, 1) - p(1, -1))
                a += Ix * Ix; b += Iy * Iy; c += Ix * Iy
        scale = 1.0 / (4 * 7 * 255.0)
        ref = (a * b - c * c - 0.04 * (a + b) ** 2) * scale ** 4
        got = pyorb.harris_at(img, x, y)
        assert abs(got - ref) <= 2e-6 * max(abs(ref), 1e-12) + 1e-12
        u = pyorb.umax()
        m10 = m01 = 0
        for v in range(-15, 16):
            for uu in range(-u[abs(v)], u[abs(v)] + 1):
                m10 += uu * I[y + v, x + uu]
                m01 += v * I[y + v, x + uu]
        ang = np.degrees(np.arctan2(m01, m10)) % 360
        got = pyorb.ic_angle_at(img, x, y)
        assert min(abs(got - ang), 360 - abs(got - ang)) < 0.02  # fastAtan2 is a 0.01-degree polynomial


def test_blur_and_descriptor_invariants(img):
    b = pyorb.gaussian_blur7(img)
    k = pyorb.gauss_kernel().astype(np.float64)
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")
    ref = sum(k[i] * pad[:, i:i + 640] for i in range(7))
    ref = sum(k[j] * ref[j:j + 480] for j in range(7))
    assert np.array_equal(b, np.clip(np.floor((ref + 32768) / 65536), 0, 255).astype(np.uint8))
    kp = pyorb.detect(img, None, 20)
    k2, d = pyorb.compute(img, kp)
    assert len(k2) == len(d) and np.all(np.diff(k2["octave"]) >= 0)
    assert np.all((k2["x"] >= 31) & (k2["x"] < 609) & (k2["y"] >= 31) & (k2["y"] < 449))
    # deterministic, and distinct keypoints have distinct descriptors (practically always)
    k3, d3 = pyorb.compute(img, kp)
    assert np.array_equal(d, d3)
    assert len({bytes(r) for r in d}) > 0.98 * len(d)


def test_grid_detector_state_and_repeatability():
    seq = synth.make_image_sequence(n_frames=2, seed=12)
    st = pyorb.grid_state(1000)
    assert (st.cell_min, st.cell_max, st.max_total, st.edge) == (111, 167, 1500, 31)  # features.cpp:47-53
    m = np.where(seq["mask"][0] > 0, 255, 0).astype(np.uint8)
    kp0, d0 = pyorb.node_features(st, seq["gray"][0], m, seq["depth"][0], 1000)
    thr_after_first = list(st.thresh[:9])
    assert all(t != 20.0 for t in thr_after_first)  # every cell adapted (too many -> x1.3)
    kp1, d1 = pyorb.node_features(st, seq["gray"][1], np.where(seq["mask"][1] > 0, 255, 0).astype(np.uint8),
                                  seq["depth"][1], 1000)
    assert 500 < len(kp0) <= 1000 and 500 < len(kp1) <= 1000
    # the same plane is seen with a small similarity motion: most descriptors find a close partner
    D = np.unpackbits(d0[:, None, :] ^ d1[None, :, :], axis=2).sum(2)
    assert (D.min(1) < 50).mean() > 0.5


@pytest.mark.skipif(pyorb.ref_adjuster_lib() is None, reason="reference pin (oracle/_ref/libref_adjuster.so) not built")
def test_grid_and_threshold_adaptation_match_live_reference_code():
    """Rows a1-a3: the reference's own createDetector("ORB") wiring (features.cpp:35-60) and
    feature_adjuster.cpp -- DetectorAdjuster, VideoDynamicAdaptedFeatureDetector (re-detect with threshold x0.7,
    x1.3 for the next frame, <= 5 iterations), VideoGridAdaptedFeatureDetector (3x3 cells with +-31 px overlap,
    keepStrongest per cell, aggregation) -- compiled from /root/reference and run around the oracle's cv::ORB::detect
    restatement, against the oracle's orb_grid_detect: same keypoints on every frame of a sequence (the per-cell
    thresholds persist across frames on both sides)."""
    R = pyorb.ref_adjuster_lib()
    key = lambda k: sorted(zip(k["octave"].tolist(), k["y"].tolist(), k["x"].tolist(), k["response"].tolist(),
                               k["angle"].tolist(), k["size"].tolist()))
    rng = np.random.default_rng(13)
    for max_kp, grid, iters, frames in ((1000, 3, 5, 4), (300, 2, 3, 3), (4000, 3, 5, 2)):
        seq = synth.make_image_sequence(n_frames=frames, seed=21 + grid)
        st = pyorb.grid_state(max_kp, grid, iters)
        h = R.ref_grid_detector_create(max_kp, grid, iters)
        try:
            for f in range(frames):
                img = seq["gray"][f].copy()
                mask = np.where(seq["mask"][f] > 0, 255, 0).astype(np.uint8)
                if f == 1:   # a texture-poor frame: the x0.7 re-detection loop runs
                    img = (img.astype(np.float32) * 0.15 + 100).astype(np.uint8)
                if f == 2:   # a frame whose right third has no depth: cells with an all-zero mask break out
                    mask[:, 430:] = 0
                a = pyorb.grid_detect(st, img, mask)
                b = pyorb.ref_grid_detect(h, img, mask)
                assert len(a) == len(b) and len(a) > 0
                assert key(a) == key(b), (max_kp, grid, f)
        finally:
            R.ref_grid_detector_destroy(h)
    # no mask at all
    seq = synth.make_image_sequence(n_frames=1, seed=5)
    st = pyorb.grid_state(600, 3, 5)
    h = R.ref_grid_detector_create(600, 3, 5)
    assert key(pyorb.grid_detect(st, seq["gray"][0], None)) == key(pyorb.ref_grid_detect(h, seq["gray"][0], None))
    R.ref_grid_detector_destroy(h)
