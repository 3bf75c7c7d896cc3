"""-m gpu: the batch entry points' read-backs written by the kernels themselves (round 6: orb_measure_kernel, sift_row_scan /
sift_orientation / sift_descriptor kernels store into the page-locked buffers the host reads) against the
copies behind the kernels they replace (RGBDFE_DETECT_HOSTWRITE=0 / RGBDFE_SIFT_HOSTWRITE=0), and the Python mirror's reused
output arrays (copy=False) against its copies."""
import os

import numpy as np
import pytest

from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _frames(n, w=640, h=480, seed=5):
    seq = synth.make_image_sequence(n_frames=min(n, 10), seed=seed, width=w, height=h)
    idx = synth.forth_and_back(n, len(seq["gray"]))
    grays = [seq["gray"][i] for i in idx]
    for f in range(n // 2, n // 2 + 3):            # a dark stretch: adjuster iterations, re-passes
        grays[f] = (grays[f].astype(np.float32) * 0.25 + 70).astype(np.uint8)
    return seq, idx, grays


def test_orb_pass_readback_by_the_measure_kernel_equals_the_copy():
    from rgbdslam_v2_amd.frontend import FrontEnd
    n = 33                                          # two full super-frames of 14 and a ragged one
    seq, idx, grays = _frames(n)
    depths = [seq["depth"][i] for i in idx]
    masks = [np.where(seq["mask"][i] > 0, 255, 0).astype(np.uint8) for i in idx]
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    outs = {}
    for hw in ("1", "0"):
        with _Env(RGBDFE_DETECT_HOSTWRITE=hw):
            fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=1024, max_pairs_per_batch=8)
            fe.detector_configure(max_keypoints=1000)
            res = fe.detect_describe_batch(grays, masks, depths, *K)
            again = fe.detect_describe_batch(grays, masks, depths, *K, copy=False)     # views of the reused output arrays
            again = [tuple(a.copy() for a in r) for r in again]
            thr = fe.detector_thresholds().copy()
            fe.close()
            outs[hw] = (res, again, thr)
    (r1, a1, t1), (r0, a0, t0) = outs["1"], outs["0"]
    assert np.array_equal(t1, t0) and sum(len(r[0]) for r in r1) > 300 * n
    for got, want in ((r1, r0), (a1, a0)):
        for (k1, d1, x1), (k0, d0, x0) in zip(got, want):
            assert k1.tobytes() == k0.tobytes() and np.array_equal(d1, d0) and x1.tobytes() == x0.tobytes()


def test_sift_chunk_readbacks_by_their_kernels_equal_the_copies():
    from rgbdslam_v2_amd.frontend import FrontEnd
    n = 19                                          # two chunks of 8 and a ragged one
    _, _, grays = _frames(n, seed=9)
    outs = {}
    for hw in ("1", "0"):
        with _Env(RGBDFE_SIFT_HOSTWRITE=hw):
            fe = FrontEnd(device_id=0, max_nodes=2, max_keypoints=64, max_pairs_per_batch=8)
            res = fe.sift_detect_batch(grays)
            one = fe.sift_detect(grays[3], None)
            fe.close()
            outs[hw] = (res, one)
    (r1, o1), (r0, o0) = outs["1"], outs["0"]
    assert sum(len(r[0]) for r in r1) > 200 * n
    for (k1, d1), (k0, d0) in zip(r1, r0):
        assert k1.tobytes() == k0.tobytes() and d1.tobytes() == d0.tobytes()
    assert o1[0].tobytes() == o0[0].tobytes() and o1[1].tobytes() == o0[1].tobytes()
