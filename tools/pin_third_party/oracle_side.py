"""What the oracle (oracle/orb_oracle.c, oracle/rgbd_oracle.c) says the third-party calls of pin_third_party.cpp return,
array for array, under the names the harness writes: the reference side of the comparison (compare_pins.py) and -- written to a
pin file -- the stand-in for a harness run that tests/test_pin_third_party.py feeds to the comparator."""
import ctypes as C
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from oracle import pyorb  # noqa: E402


def kp_arrays(kp):
    f = np.stack([kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"]], axis=1).astype(np.float32) if len(kp) else \
        np.zeros((0, 5), np.float32)
    return f, kp["octave"].astype(np.int32)


def kp_struct(f, octave):
    kp = np.zeros(len(octave), pyorb.KP_DTYPE)
    for i, name in enumerate(("x", "y", "size", "angle", "response")):
        kp[name] = f[:, i]
    kp["octave"] = octave
    return kp


def retain_best(kp, n):
    kp = np.ascontiguousarray(kp.copy())
    lib = pyorb.lib()
    lib.orb_retain_best.restype = C.c_int
    m = lib.orb_retain_best(kp.ctypes.data_as(C.c_void_p), len(kp), int(n))
    return kp[:m].copy()


def llt_quadform(S, d):
    """d^T * S.llt().solve(d) in the operation order of rgbd_oracle.c (orc_error_function2, the default variant): unblocked
    Cholesky on the lower triangle, forward and backward substitution.  Python floats are IEEE doubles: the same bits."""
    x = S[0][0]
    if not x > 0.0:
        return float(np.finfo(np.float64).max)
    l00 = math.sqrt(x)
    l10 = S[1][0] / l00
    l20 = S[2][0] / l00
    x = S[1][1] - l10 * l10
    if not x > 0.0:
        return float(np.finfo(np.float64).max)
    l11 = math.sqrt(x)
    l21 = (S[2][1] - l20 * l10) / l11
    x = S[2][2] - (l20 * l20 + l21 * l21)
    if not x > 0.0:
        return float(np.finfo(np.float64).max)
    l22 = math.sqrt(x)
    y0 = d[0] / l00
    y1 = (d[1] - l10 * y0) / l11
    y2 = (d[2] - (l20 * y0 + l21 * y1)) / l22
    z2 = y2 / l22
    z1 = (y1 - l21 * z2) / l11
    z0 = (y0 - (l10 * z1 + l20 * z2)) / l00
    return (d[0] * z0 + d[1] * z1) + d[2] * z2


def oracle_pins(inp):
    """inputs (dict from pinfile.read) -> dict of every array pin_third_party.cpp writes, computed by the oracle."""
    out = {}
    for i in range(int(inp["n_images"][0])):
        tag = "img%d" % i
        gray, mask = inp[tag + "_gray"], inp[tag + "_mask"]
        for t, thr in enumerate(inp[tag + "_fast_thresholds"]):
            kp = pyorb.detect(gray, mask, int(thr))
            out["%s_detect%d_f" % (tag, t)], out["%s_detect%d_octave" % (tag, t)] = kp_arrays(kp)
            if t == 0:
                max_keyp = int(inp[tag + "_max_keypoints"][0])
                best = kp
                if len(best) > max_keyp:
                    best = retain_best(best, max_keyp)
                    out[tag + "_retain_best_size"] = np.array([len(best)], np.int32)
                    best = best[:max_keyp]
                out[tag + "_retained_f"], out[tag + "_retained_octave"] = kp_arrays(best)
                kept, desc = pyorb.compute(gray, best)
                out[tag + "_described_f"], out[tag + "_described_octave"] = kp_arrays(kept)
                out[tag + "_desc"] = desc.reshape(-1, 32)
        if tag + "_given_f" in inp:
            kept, desc = pyorb.compute(gray, kp_struct(inp[tag + "_given_f"], inp[tag + "_given_octave"]))
            out[tag + "_given_described_f"], out[tag + "_given_described_octave"] = kp_arrays(kept)
            out[tag + "_given_desc"] = desc.reshape(-1, 32)
    for i in range(int(inp["n_retain"][0])):
        tag = "retain%d" % i
        kept = retain_best(kp_struct(inp[tag + "_f"], inp[tag + "_octave"]), int(inp[tag + "_n"][0]))
        out[tag + "_kept_f"], out[tag + "_kept_octave"] = kp_arrays(kept)
    for i in range(int(inp["n_fits"][0])):
        tag = "fit%d" % i
        f, t = inp[tag + "_from"], inp[tag + "_to"]
        n = len(f)
        q = np.concatenate([f, np.ones((n, 1), np.float32)], axis=1)
        tt = np.concatenate([t, np.ones((n, 1), np.float32)], axis=1)
        idx = np.arange(n, dtype=np.int32)
        out[tag + "_T"] = po.fit_transform(q, tt, idx, idx, idx).astype(np.float32)
    A = inp["svd_in"]
    U, S, V = np.zeros_like(A), np.zeros((len(A), 3), np.float32), np.zeros_like(A)
    for k in range(len(A)):
        U[k], S[k], V[k] = po.svd3(A[k])
    out["svd_U"], out["svd_S"], out["svd_V"] = U, S, V
    out["llt_q"] = np.array([llt_quadform(inp["llt_S"][k], inp["llt_d"][k]) for k in range(len(inp["llt_S"]))], np.float64)
    return out
