"""-m gpu: SIFT extraction (rgbdfe_sift_detect = SiftGPUWrapper::detect, sift_gpu_wrapper.cpp:113-167) against the
reference's own pipeline -- SiftGPU's CUDA kernels + PyramidCU / SiftPyramid host code compiled from the reference tree on
the CPU emulation (oracle/_ref/libref_siftgpu.so, prebuilt, travels to the GPU box) and its frozen outputs
(tests/golden/sift_extract_golden.npz).

What is compared how:
  * Gaussian planes, keypoint candidates (position, extremum sign, sub-pixel offsets), per-level counts, the feature-count
    limit, feature order and positions: BIT FOR BIT -- this arithmetic has no transcendental function in it;
  * scale (powf), orientation (atan2f / expf histogram) and descriptors (expf, sincosf, atan2f): the device's libm differs
    from glibc's by ulps.  Scale: 2e-6 relative.  Orientation: 1e-4 rad for >= 99.5 % of the features; a gradient angle
    that sits on the border of two of the 36 histogram bins can fall to either side, which moves an interpolated peak by a
    fraction of the 0.17 rad bin width -- every feature within 2e-2 rad (measured: one feature in 1554 at 1.9e-3).
    Descriptors: relative L2 difference 1e-3 for >= 99.5 %, 5e-2 for all (the descriptor frame turns with the
    orientation)."""
import os
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "sift_extract_golden.npz")
SCALE_RTOL, ORI_TOL, ORI_TOL_ALL, DESC_RTOL, DESC_RTOL_ALL, TIGHT_FRACTION = 1e-6, 1e-4, 2e-2, 1e-3, 5e-2, 0.995


@pytest.fixture(scope="module")
def fe():
    from rgbdslam_v2_amd.frontend import FrontEnd
    f = FrontEnd(device_id=0, max_nodes=8, max_keypoints=2048, max_pairs_per_batch=8)
    yield f
    f.close()


def image(w, h, seed):
    return synth.make_image_sequence(n_frames=1, seed=seed, width=w, height=h)["gray"][0]


def check_features(kp, desc, rkeys, rdesc):
    assert len(kp) == len(rkeys)
    if len(kp) == 0:
        return
    assert np.array_equal(kp["x"], rkeys[:, 0]) and np.array_equal(kp["y"], rkeys[:, 1])      # positions: exact
    s = kp["size"].astype(np.float64) / 12.0                                                      # the wrapper's 12 * scale
    assert np.all(np.abs(s - rkeys[:, 2]) <= 2 * SCALE_RTOL * rkeys[:, 2])
    do = np.abs(kp["angle"].astype(np.float64) * 3.1415927 / 180.0 - rkeys[:, 3])
    do = np.minimum(do, 2 * np.pi - do)
    assert do.max() <= ORI_TOL_ALL and (do <= ORI_TOL).mean() >= TIGHT_FRACTION, (do.max(), (do <= ORI_TOL).mean())
    rel = np.linalg.norm(desc - rdesc, axis=1) / np.maximum(np.linalg.norm(rdesc, axis=1), 1e-12)
    assert rel.max() <= DESC_RTOL_ALL and (rel <= DESC_RTOL).mean() >= TIGHT_FRACTION, (rel.max(), (rel <= DESC_RTOL).mean())
    assert np.all(kp["response"] == 0) and np.all(kp["octave"] == 0)


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_frozen_reference_outputs(fe, name):
    """a, b: seeded synthetic images; c: external/SiftGPU/data/640-1.jpg whole (640 x 480, 1263 features under "-tc2 1000");
    d: a 320 x 240 window of 800-2.jpg -- SiftGPU's own test photographs, luminance stored in the fixture."""
    g = np.load(GOLD)
    w, h, maxf, seed, omin, onum = [int(v) for v in g[name + "_meta"]]
    img = g[name + "_img"] if name + "_img" in g else image(w, h, seed)
    assert img.shape == (h, w)
    kp, desc = fe.sift_detect(img, None, maxf)
    geo = fe.sift_geometry()
    assert (geo["octave_min"], geo["octave_num"]) == (omin, onum)
    crcs = [zlib.crc32(fe.sift_debug_plane(o, l).tobytes()) for o in range(onum) for l in range(geo["levels"])]
    assert np.array_equal(np.array(crcs, np.uint32), g[name + "_plane_crc"])                     # every Gaussian plane
    k = 0
    for o in range(onum):
        for j in range(geo["dog_levels"]):
            got = fe.sift_debug_candidates(o, j)
            assert len(got) == int(g[name + "_cand_n"][k]), (o, j)
            assert zlib.crc32(np.ascontiguousarray(got).tobytes()) == int(g[name + "_cand_crc"][k]), (o, j)
            key = "%s_cand_%d_%d" % (name, o, j)
            if key in g:
                assert np.array_equal(got.view(np.uint32), g[key].view(np.uint32)), (o, j)
            k += 1
    check_features(kp, desc, g[name + "_keys"], g[name + "_desc"])


@pytest.mark.parametrize("w,h,maxf,seed", [(320, 240, 1000, 1), (644, 481, 300, 2), (640, 480, 1000, 7), (100, 36, 50, 4)])
def test_against_the_compiled_reference(fe, w, h, maxf, seed):
    """Sizes with every width / height parity (the width is cut to a multiple of 4, odd heights halve with a remainder),
    with and without the "-tc2" feature-count limit skipping the fine octaves."""
    if po.ref_siftgpu_lib() is None:
        pytest.skip("oracle/_ref/libref_siftgpu.so was not built (no reference tree when the snapshot was made)")
    img = image(w, h, seed)
    kp, desc = fe.sift_detect(img, None, maxf)
    rkeys, rdesc, rcnt = po.ref_sift_detect(img, maxf)
    geo, rgeo = fe.sift_geometry(), po.ref_sift_geometry()
    assert geo == rgeo
    for o in range(geo["octave_num"]):
        for l in range(geo["levels"]):
            a, b = fe.sift_debug_plane(o, l), po.ref_sift_level(o, l, 0)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("plane", o, l)
        for j in range(geo["dog_levels"]):
            a, b = fe.sift_debug_candidates(o, j), po.ref_sift_candidates(o, j)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("candidates", o, j)
    check_features(kp, desc, rkeys, rdesc)


@pytest.mark.parametrize("kind", ["noise", "flat", "steps", "binary"])
def test_adversarial_images_against_the_compiled_reference(fe, kind):
    """Images a camera never delivers: pure noise (candidates everywhere: tens of thousands per level, the truncation by the
    feature limit cuts whole octaves), a flat image (nothing anywhere), intensity steps (extrema on straight edges: the edge
    ratio test), a 0 / 255 pattern (the largest gradients) -- planes and candidates bit for bit, features within the stated
    tolerances."""
    if po.ref_siftgpu_lib() is None:
        pytest.skip("oracle/_ref/libref_siftgpu.so was not built (no reference tree when the snapshot was made)")
    rng = np.random.default_rng(17)
    h, w = 120, 164
    if kind == "noise":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "flat":
        img = np.full((h, w), 131, np.uint8)
    elif kind == "steps":
        img = (((np.arange(w)[None, :] // 23) * 60 + (np.arange(h)[:, None] // 31) * 35) % 256).astype(np.uint8)
    else:
        img = (rng.integers(0, 2, (h // 4, w // 4), dtype=np.uint8).repeat(4, 0).repeat(4, 1) * 255)
    img = np.ascontiguousarray(img)
    maxf = 600
    kp, desc = fe.sift_detect(img, None, maxf)
    rkeys, rdesc, rcnt = po.ref_sift_detect(img, maxf)
    geo = fe.sift_geometry()
    assert geo == po.ref_sift_geometry()
    for o in range(geo["octave_num"]):
        for l in range(geo["levels"]):
            a, b = fe.sift_debug_plane(o, l), po.ref_sift_level(o, l, 0)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("plane", o, l)
        for j in range(geo["dog_levels"]):
            a, b = fe.sift_debug_candidates(o, j), po.ref_sift_candidates(o, j)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("candidates", o, j)
    check_features(kp, desc, rkeys, rdesc)
    if kind == "flat":
        assert len(kp) == 0
    if kind == "noise":
        assert len(kp) > 50          # the limit keeps the coarse octaves only (whole fine levels are dropped)


def test_full_size_properties(fe):
    """BASELINE configs[3]'s frame size through size-independent properties: determinism, the feature-count limit's
    semantics (coarse octaves first, whole levels), keypoints inside the image, histogram descriptors non-negative."""
    img = image(640, 480, 11)
    a = fe.sift_detect(img, None, 1000)
    b = fe.sift_detect(img, None, 1000)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    kp, desc = a
    assert 300 < len(kp) and np.all(desc >= 0) and np.isfinite(desc).all()
    assert np.all((kp["x"] >= 0) & (kp["x"] <= 640) & (kp["y"] >= 0) & (kp["y"] <= 480))
    assert np.all(kp["size"] > 0) and np.all((kp["angle"] >= 0) & (kp["angle"] <= 360.001))
    # without a limit every level contributes; the limited list is a suffix of it (fine levels dropped whole, the others
    # untouched) -- SiftPyramid::LimitFeatureCount erases from the front
    kall, dall = fe.sift_detect(img, None, 1 << 30)
    assert len(kall) >= len(kp)
    assert kall[len(kall) - len(kp):].tobytes() == kp.tobytes() and dall[len(kall) - len(kp):].tobytes() == desc.tobytes()
    small = fe.sift_detect(img, None, 100)[0]
    assert 0 < len(small) < len(kp) and kp[len(kp) - len(small):].tobytes() == small.tobytes()


def test_batch_equals_single_calls(fe):
    """rgbdfe_sift_detect_batch: 8 frames per launch chain (11 frames = one full chain + a ragged one), every frame's
    output identical to its single call -- textured frames, a featureless one in the middle, with and without the limit."""
    frames = [image(320, 240, 20 + k) for k in range(11)]
    frames[4] = np.full((240, 320), 90, np.uint8)
    for maxf in (300, 1 << 20):
        single = [fe.sift_detect(g, None, maxf) for g in frames]
        batch = fe.sift_detect_batch(frames, maxf)
        assert len(batch) == len(frames) and len(batch[4][0]) == 0
        for (ka, da), (kb, db) in zip(single, batch):
            assert ka.tobytes() == kb.tobytes() and da.tobytes() == db.tobytes()
    # a single call after a batch sees the first frame's stage data again (debug accessors) and the same result
    again = fe.sift_detect(frames[0], None, 300)
    assert again[0].tobytes() == fe.sift_detect_batch(frames[:1], 300)[0][0].tobytes()
    # a stride that is too small: the counts come back, the error is raised
    from rgbdslam_v2_amd.frontend import RgbdfeError
    with pytest.raises(RgbdfeError):
        fe.sift_detect_batch(frames[:3], 300, out_stride=8)


def _check_descriptors(desc, rdesc):
    rel = np.linalg.norm(desc - rdesc, axis=1) / np.maximum(np.linalg.norm(rdesc, axis=1), 1e-12)
    nz = np.linalg.norm(rdesc, axis=1) > 0
    assert np.array_equal(np.linalg.norm(desc, axis=1) > 0, nz)
    assert rel[nz].max() <= DESC_RTOL_ALL and (rel[nz] <= DESC_RTOL).mean() >= TIGHT_FRACTION, (rel[nz].max(), (rel[nz] <= DESC_RTOL).mean())


def test_describe_given_keypoints(fe):
    """rgbdfe_sift_describe = SiftGPUWrapper::detect with a keypoint list (sift_gpu_wrapper.cpp:132-142; extractor SIFTGPU behind
    another detector, node.cpp:166-171): descriptors at given positions / sizes / angles, in the callers' order, against the
    frozen outputs of the reference pipeline and -- where the pin library travelled -- against the pin itself on ORB
    keypoints."""
    g = np.load(GOLD)
    img = image(322, 241, 5)
    keys = g["k_keys"]
    kp = np.zeros(len(keys), fe.sift_detect(img, None, 10)[0].dtype)
    kp["x"], kp["y"] = keys[:, 0], keys[:, 1]
    # the ABI takes cv::KeyPoint fields: size = 12 * scale, angle in degrees; the fixture's floats survive the wrapper's
    # two conversions only approximately, so the comparison keys are those the library reports back
    kp["size"] = (12.0 * keys[:, 2].astype(np.float64)).astype(np.float32)
    kp["angle"] = (keys[:, 3].astype(np.float64) * 180.0 / 3.1415927).astype(np.float32)
    out_kp, desc = fe.sift_describe(img, kp)
    assert np.array_equal(out_kp["x"], kp["x"]) and np.all(out_kp["response"] == 0) and np.all(out_kp["octave"] == 0)
    back = np.stack([out_kp["x"], out_kp["y"], out_kp["size"] / 12.0, out_kp["angle"] * 3.1415927 / 180.0], 1)
    assert np.abs(back[:, 2] - keys[:, 2]).max() <= 2e-6 * keys[:, 2].max() and np.abs(back[:, 3] - keys[:, 3]).max() < 2e-6
    if po.ref_siftgpu_lib() is not None:
        s = (out_kp["size"].astype(np.float64) / 12.0).astype(np.float32)
        o = (out_kp["angle"].astype(np.float64) / 180.0 * 3.1415927).astype(np.float32)
        _check_descriptors(desc, po.ref_sift_describe(img, np.stack([out_kp["x"], out_kp["y"], s, o], 1)))
    else:
        _check_descriptors(desc, g["k_desc"])
    # ORB keypoints of a frame (the mixed configuration): the order is the detector's, nothing is dropped
    seq = synth.make_image_sequence(n_frames=1, seed=2)
    okp = fe.orb_detect(seq["gray"][0], None, 20)[:400]
    k2, d2 = fe.sift_describe(seq["gray"][0], okp)
    assert len(k2) == len(okp) == len(d2) and np.array_equal(k2["x"], okp["x"]) and np.all(np.linalg.norm(d2, axis=1) > 0)
    if po.ref_siftgpu_lib() is not None:
        s = (okp["size"].astype(np.float64) / 12.0).astype(np.float32)
        o = (okp["angle"].astype(np.float64) / 180.0 * 3.1415927).astype(np.float32)
        _check_descriptors(d2, po.ref_sift_describe(seq["gray"][0], np.stack([okp["x"], okp["y"], s, o], 1)))
    assert fe.sift_describe(img, kp[:0])[1].shape == (0, 128)


def test_extraction_feeds_the_sift_pair_path(fe):
    """detect -> projectTo3DSiftGPU (rgbdfe_sift_node_features) -> upload -> SiftGPU matcher + RANSAC: two views of the same
    textured plane give an edge."""
    seq = synth.make_image_sequence(n_frames=2, seed=5)
    nodes = []
    for f in range(2):
        kp, desc = fe.sift_detect(seq["gray"][f], None, 1000)
        xy = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
        kept, xyz1, raw, feat = fe.sift_node_features(xy, desc, seq["depth"][f], seq["fx"], seq["fy"], seq["cx"], seq["cy"],
                                                      max_keypoints=2048)
        assert len(kept) > 200
        # SiftGPU's matcher quantises 512 * d to bytes (SiftMatchCU.cpp:96-99): it expects normalised descriptors, so the
        # node keeps the L2-normalised copy the reference would get without "-unn"
        d = raw / np.maximum(np.linalg.norm(raw, axis=1, keepdims=True), 1e-12)
        d = np.minimum(d, 0.2)
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)).astype(np.float32)
        fe.upload_sift_node(f, d, xyz1)
        nodes.append((d, xyz1))
    out, dist = fe.match_sift_pair_list([1], [0])
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
    ref = po.match_sift_node_pair(nodes[1][0], nodes[1][1], 1, nodes[0][0], nodes[0][1], 0, prm)
    assert out["n_all"][0] == ref["n_all"] and out["n_inl"][0] == ref["n_inl"] and out["id1"][0] == ref["id1"]
    assert out["id1"][0] == 0 and out["n_inl"][0] > 50
    for f in range(2):
        fe.release_node(f)


def test_error_paths(fe):
    from rgbdslam_v2_amd._lib import RgbdfeError
    with pytest.raises(RgbdfeError):
        fe.sift_detect(np.zeros((8, 8), np.uint8), None, 100)            # too small for one octave
    img = image(200, 152, 3)
    kp, desc = fe.sift_detect(img, None, 400)
    import ctypes as C
    n = C.c_int32(0)
    k = np.zeros(4, kp.dtype)
    d = np.zeros((4, 128), np.float32)
    st = fe._L.rgbdfe_sift_detect(fe._ctx, img.ctypes.data, None, 152, 200, 400, k.ctypes.data, d.ctypes.data, 4, C.byref(n))
    assert st == -5 and n.value == len(kp)                                # RGBDFE_ERR_CAPACITY reports the needed rows
    flat = fe.sift_detect(np.full((120, 160), 77, np.uint8), None, 100)   # no structure: no features, no error
    assert len(flat[0]) == 0 and flat[1].shape == (0, 128)
