"""CPU: the SIFT extraction pin -- the reference's SiftGPU CUDA pipeline compiled from /root/reference on the fiber-based
CUDA emulation (oracle/_ref/libref_siftgpu.so, oracle/Makefile) -- is repeatable, equals its frozen outputs
(tests/golden/sift_extract_golden.npz, what the GPU tests compare the HIP path with) and behaves like SIFT."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sift_extract_golden.npz")
needs_pin = pytest.mark.skipif(po.ref_siftgpu_lib() is None, reason="oracle/_ref/libref_siftgpu.so not built")


@needs_pin
def test_pin_reproduces_its_golden_outputs():
    g = np.load(GOLD)
    w, h, maxf, seed, omin, onum = [int(v) for v in g["a_meta"]]
    img = synth.make_image_sequence(n_frames=1, seed=seed, width=w, height=h)["gray"][0]
    for _ in range(2):          # twice: every call starts from a fresh SiftGPU state
        keys, desc, cnt = po.ref_sift_detect(img, maxf)
        assert np.array_equal(keys, g["a_keys"]) and np.array_equal(desc, g["a_desc"]) and np.array_equal(cnt, g["a_counts"])
    geo = po.ref_sift_geometry()
    assert (geo["octave_min"], geo["octave_num"], geo["levels"], geo["dog_levels"]) == (omin, onum, 8, 5)
    for o in range(onum):
        for j in range(5):
            assert np.array_equal(po.ref_sift_candidates(o, j), g["a_cand_%d_%d" % (o, j)])


@needs_pin
def test_pin_reproduces_the_photo_fixture():
    """Case d: a window of external/SiftGPU/data/800-2.jpg, luminance stored in the fixture (no JPEG decoder, no
    reference tree needed)."""
    g = np.load(GOLD)
    w, h, maxf, seed, omin, onum = [int(v) for v in g["d_meta"]]
    keys, desc, cnt = po.ref_sift_detect(g["d_img"], maxf)
    assert np.array_equal(keys, g["d_keys"]) and np.array_equal(desc, g["d_desc"]) and np.array_equal(cnt, g["d_counts"])
    assert len(keys) > 300 and np.all(desc >= 0)


@needs_pin
def test_pin_finds_a_blob_at_its_place_and_scale():
    """A Gaussian blob of standard deviation s is a DoG extremum at scale ~ s * sqrt(2) ... within the sampling of 5 levels
    per octave: the strongest feature sits on the blob, its scale within a factor 1.5 of s."""
    h, w, s, cx, cy = 128, 160, 6.0, 83.0, 61.0
    y, x = np.mgrid[0:h, 0:w]
    img = (40 + 180 * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s))).astype(np.uint8)
    keys, desc, cnt = po.ref_sift_detect(img, 1000)
    assert len(keys) >= 1
    d = np.hypot(keys[:, 0] - (cx + 0.5), keys[:, 1] - (cy + 0.5))
    near = keys[d < 1.5]
    assert len(near) >= 1
    assert np.any((near[:, 2] > s / 1.5) & (near[:, 2] < s * 1.5)), near
    assert np.all(desc >= 0)


@needs_pin
def test_feature_count_limit_keeps_the_coarse_octaves():
    """'-tc2 N' (_TruncateMethod = 1): levels are taken from the coarsest octave down while fewer than N features have been
    found, then whole fine levels are dropped while more than N remain without them (SiftPyramid.cpp:170-210)."""
    img = synth.make_image_sequence(n_frames=1, seed=9, width=320, height=240)["gray"][0]
    kall, dall, call = po.ref_sift_detect(img, 1 << 30)
    k100, d100, c100 = po.ref_sift_detect(img, 100)
    assert len(k100) < len(kall) and np.array_equal(kall[len(kall) - len(k100):], k100)
    nz = np.flatnonzero(c100)
    assert np.all(c100[nz[0]:] == call[nz[0]:]) and np.all(c100[: nz[0]] == 0)   # a suffix of whole levels
    assert c100.sum() - c100[nz[0]] <= 100 < c100.sum() + (call[nz[0] - 1] if nz[0] > 0 else 1 << 30)
