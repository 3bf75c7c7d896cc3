"""Writes inputs.pin: everything pin_third_party.cpp needs, from this repository alone (seeded synthetic frames, the committed
pair fixture tests/golden/pair_golden.npz, seeded matrices).      python tools/pin_third_party/export_inputs.py inputs.pin

  images      two 640x480 frames of the bench's synthetic sequence + one 320x240 crop with a partly masked field of view and a
              dark half (re-detection territory), three FAST thresholds each (20 = the adjuster's start value,
              feature_adjuster.h:17; 14 and 9 = one and two x0.7 steps), max_keypoints 600;
  given       the oracle's own detections of image 0 (threshold 20, best 600): pins cv::ORB::compute independently of detect;
  retain      keypoint lists with ties at the cut (KeyPointsFilter::retainBest keeps the ties: node.cpp:187-191 resizes);
  fits        the four pairs of tests/golden/pair_golden.npz: all matches in match order, the inlier sets, 4-point samples,
              one list with a NaN depth in it (transformation_estimation_euclidean.cpp:22-23 skips it);
  svd_in      3x3 float matrices: covariances of those fits, random ones, rank-deficient and zero ones;
  llt_S/d     3x3 double SPD matrices as errorFunction2 builds them (misc.cpp:751-760) with the deltas, and a few
              badly conditioned ones."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import pinfile  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyorb  # noqa: E402
from rgbdslam_v2_amd import synth  # noqa: E402
import oracle_side  # noqa: E402


def build():
    a = {}
    seq = synth.make_image_sequence(n_frames=2, seed=11)
    imgs = [(seq["gray"][0], np.where(seq["mask"][0] > 0, 255, 0).astype(np.uint8)),
            (seq["gray"][1], np.where(seq["mask"][1] > 0, 255, 0).astype(np.uint8))]
    small = np.ascontiguousarray(seq["gray"][0][100:340, 200:520]).copy()
    small[:, :160] = (small[:, :160].astype(np.int32) // 6).astype(np.uint8)          # a dark half
    m = np.full(small.shape, 255, np.uint8)
    m[:40, :] = 0
    m[:, -50:] = 0                                                                      # part of the field of view masked
    imgs.append((small, m))
    a["n_images"] = np.array([len(imgs)], np.int32)
    for i, (g, mk) in enumerate(imgs):
        a["img%d_gray" % i] = np.ascontiguousarray(g, np.uint8)
        a["img%d_mask" % i] = np.ascontiguousarray(mk, np.uint8)
        a["img%d_fast_thresholds" % i] = np.array([20, 14, 9], np.int32)
        a["img%d_max_keypoints" % i] = np.array([600], np.int32)
    kp = pyorb.detect(imgs[0][0], imgs[0][1], 20)
    best = oracle_side.retain_best(kp, 600)[:600]
    a["img0_given_f"], a["img0_given_octave"] = oracle_side.kp_arrays(best)
    rng = np.random.default_rng(20260924)
    lists = []
    for n, cut, levels in ((500, 100, 7), (64, 64, 3), (300, 299, 2), (40, 10, 1)):
        k = np.zeros(n, pyorb.KP_DTYPE)
        k["x"] = rng.uniform(20, 600, n).astype(np.float32)
        k["y"] = rng.uniform(20, 440, n).astype(np.float32)
        k["size"] = 31.0
        k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        k["response"] = (rng.integers(0, levels, n).astype(np.float32) + 1.0) * 1e-4    # few distinct responses: ties everywhere
        k["octave"] = rng.integers(0, 8, n)
        lists.append((k, cut))
    a["n_retain"] = np.array([len(lists)], np.int32)
    for i, (k, cut) in enumerate(lists):
        a["retain%d_f" % i], a["retain%d_octave" % i] = oracle_side.kp_arrays(k)
        a["retain%d_n" % i] = np.array([cut], np.int32)
    g = np.load(os.path.join(ROOT, "tests", "golden", "pair_golden.npz"))
    fits, covs = [], []
    for p in range(4):
        qi, ti = g["pairs"][p]
        n_all = int(g["p%d_n_all" % p])
        mq, mt = g["p%d_all_q" % p][:n_all], g["p%d_all_t" % p][:n_all]
        f, t = g["xyz1"][qi][mq][:, :3], g["xyz1"][ti][mt][:, :3]
        inl = g["p%d_inl_idx" % p]
        fits += [(f, t), (f[inl], t[inl]), (f[:4], t[:4]), (f[5:9], t[5:9])]
    f_nan = fits[0][0][:12].copy()
    f_nan[3, 2] = np.nan
    fits.append((f_nan, fits[0][1][:12]))
    a["n_fits"] = np.array([len(fits)], np.int32)
    for i, (f, t) in enumerate(fits):
        a["fit%d_from" % i] = np.ascontiguousarray(f, np.float32)
        a["fit%d_to" % i] = np.ascontiguousarray(t, np.float32)
        fc, tc = f[~np.isnan(f[:, 2])], t[~np.isnan(f[:, 2])]
        covs.append(((tc - tc.mean(0)).T @ (fc - fc.mean(0)) / max(len(fc), 1)).astype(np.float32))
    mats = covs + [rng.normal(size=(3, 3)).astype(np.float32) * s for s in (1.0, 1e-3, 1e3) for _ in range(6)]
    v = rng.normal(size=3).astype(np.float32)
    mats += [np.outer(v, v).astype(np.float32), np.zeros((3, 3), np.float32), np.eye(3, dtype=np.float32),
             np.diag([3.0, 3.0, 1.0]).astype(np.float32), -np.eye(3, dtype=np.float32)]
    a["svd_in"] = np.stack(mats).astype(np.float32)
    rcx, rcy = po.raster_cov()
    S, d = [], []
    for k in range(48):
        th = rng.uniform(-0.3, 0.3, 3)
        cx, cy, cz = np.cos(th)
        sx, sy, sz = np.sin(th)
        R = np.array([[cy * cz, -cy * sz, sy], [sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy],
                      [-cx * sy * cz + sx * sz, cx * sy * sz + sx * cz, cx * cy]])
        z1, z2, dc = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), 10.0 ** rng.uniform(-6, -2)
        S.append(R.T @ np.diag([rcx * z1, rcy * z1, dc]) @ R + np.diag([rcx * z2, rcy * z2, dc]))
        d.append(rng.normal(size=3) * 10.0 ** rng.uniform(-3, -1))
    for eps in (1e-8, 1e-12, 0.0):                      # badly conditioned / singular: D5 territory
        B = np.array([[1.0, 1.0, 0.0], [1.0, 1.0 + eps, 0.0], [0.0, 0.0, 1.0]])
        S.append(B)
        d.append(np.array([1.0, -1.0, 0.5]))
    a["llt_S"] = np.stack(S).astype(np.float64)
    a["llt_d"] = np.stack(d).astype(np.float64)
    return a


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "inputs.pin"
    arrays = build()
    pinfile.write(out, arrays)
    print("wrote %s: %d arrays, %.1f MB" % (out, len(arrays), os.path.getsize(out) / 1e6))
