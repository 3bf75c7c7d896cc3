"""Sensitivity of the pair op to the UNPINNED third-party arithmetic ("what's weak" 1).

Eigen 3.2 (JacobiSVD<Matrix3f>, LLT<Matrix3d>, fixed-size products) and PCL 1.7 (TransformationFromCorrespondences)
are not in the reference tree and not installable here, so the oracle -- and the kernels, which follow it bit for
bit -- restate them from the published algorithms.  This harness asks what that is worth: it re-runs whole bench steps
(BASELINE configs[1]) under alternative roundings of every restated piece (oracle/rgbd_oracle.h ORC_VAR_*), and under a
build of the same code with fused multiply-adds (what the reference's own build produces on an FMA host), and reports
per variant: how many pairs keep their edge decision, their inlier set, and how far the pose moves.

    python tools/sensitivity.py [n_pairs] [depth_noise]      -> JSON + a markdown table on stdout
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from rgbdslam_v2_amd import synth  # noqa: E402

VARIANTS = [
    ("LLT column scaling by the reciprocal (A21 *= 1/x)", 0x001),
    ("triangular solves term by term, reciprocal pivots", 0x002),
    ("Jacobi sweep order (2,1) (2,0) (1,0)", 0x004),
    ("Jacobi threshold relative to the pair's own diagonal", 0x008),
    ("JacobiSVD without pre-scaling", 0x010),
    ("PCL covariance update re-associated", 0x020),
    ("R = U (S V^T), sums from the last term", 0x040),
    ("errorFunction2: R^T (cov1 R)", 0x080),
    ("all of the above", 0x0FF),
]


def has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def _load(name):
    path = os.path.join(ROOT, "oracle", name)
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ROOT, "oracle", "rgbd_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), name], stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    L.orc_match_pairs_mt.restype = None
    L.orc_match_pairs_mt.argtypes = [C.c_void_p] * 6 + [C.c_int, C.POINTER(po.OrcParams), C.c_void_p, C.c_int]
    L.orc_set_variant.restype = None
    L.orc_set_variant.argtypes = [C.c_uint]
    L.orc_set_trace.restype = None
    L.orc_set_trace.argtypes = [C.c_void_p]
    return L


def run(L, variant, descs, xyzs, pq, pt, prm, threads):
    L.orc_set_variant(variant)
    n_nodes = len(descs)
    dptr = (C.c_void_p * n_nodes)(*[d.ctypes.data for d in descs])
    xptr = (C.c_void_p * n_nodes)(*[x.ctypes.data for x in xyzs])
    counts = np.array([d.shape[0] for d in descs], np.uint32)
    ids = np.arange(n_nodes, dtype=np.int32)
    out = (po.OrcResult * len(pq))()
    trace = np.zeros(len(pq), np.uint64)
    L.orc_set_trace(trace.ctypes.data)
    L.orc_match_pairs_mt(dptr, xptr, counts.ctypes.data, ids.ctypes.data, pq.ctypes.data, pt.ctypes.data, len(pq),
                         C.byref(prm), out, threads)
    L.orc_set_variant(0)
    L.orc_set_trace(None)
    res = []
    for o, src in zip(out, trace):
        n_inl = o.n_inl
        res.append((o.id1, n_inl, tuple(o.inl_idx[:n_inl]), np.array(o.T, np.float32), o.rmse, o.real_iterations, int(src)))
    return res


def compare(base, var):
    n = len(base)
    same_edge = same_set = same_path = 0
    dev_same_set = dev_same_edge = dev_same_path = 0.0
    rmse_rel = 0.0
    for b, v in zip(base, var):
        eb, ev = b[0] >= 0, v[0] >= 0
        if eb == ev:
            same_edge += 1
        if eb == ev and b[2] == v[2]:
            same_set += 1
            if b[6] == v[6]:   # ... and the transform was fitted from the same inlier set: nothing discrete flipped
                same_path += 1
                if eb:
                    dev_same_path = max(dev_same_path, float(np.abs(b[3] - v[3]).max()))
            if eb:
                dev_same_set = max(dev_same_set, float(np.abs(b[3] - v[3]).max()))
                if b[4] > 0:
                    rmse_rel = max(rmse_rel, abs(b[4] - v[4]) / b[4])
        if eb and ev:
            dev_same_edge = max(dev_same_edge, float(np.abs(b[3] - v[3]).max()))
    return {"pairs": n, "edge_decision_kept_pct": round(100.0 * same_edge / n, 3),
            "inlier_set_kept_pct": round(100.0 * same_set / n, 3),
            "nothing_flipped_pct": round(100.0 * same_path / n, 3), "max_pose_dev_nothing_flipped": dev_same_path,
            "max_pose_dev_same_inlier_set": dev_same_set, "max_rel_rmse_dev_same_inlier_set": rmse_rel,
            "max_pose_dev_any_edge": dev_same_edge}


def study(n_pairs=4000, depth_noise=0.01, threads=0, seed=20260923):
    seq = synth.make_sequence(n_frames=200, n_kp=1000, seed=seed, depth_noise=depth_noise)
    pq, pt = synth.candidate_pairs(200, per_frame=20, seed=seed)
    sel = np.linspace(0, len(pq) - 1, min(n_pairs, len(pq))).astype(np.int64)
    pq, pt = np.ascontiguousarray(pq[sel], np.int32), np.ascontiguousarray(pt[sel], np.int32)
    descs = [np.ascontiguousarray(d, np.uint8) for d in seq["desc"]]
    xyzs = [np.ascontiguousarray(x, np.float32) for x in seq["xyz1"]]
    prm = po.default_params(seed=seed, depth_cov=1e-4)
    threads = threads or po.usable_cpus()
    L = _load("liboracle.so")
    base = run(L, 0, descs, xyzs, pq, pt, prm, threads)
    rows = []
    for name, flag in VARIANTS:
        rows.append({"variant": name, "flags": flag, **compare(base, run(L, flag, descs, xyzs, pq, pt, prm, threads))})
    if has_fma():
        F = _load("liboracle_fma.so")
        rows.append({"variant": "fused multiply-adds (gcc -mfma -ffp-contract=fast)", "flags": "fma",
                     **compare(base, run(F, 0, descs, xyzs, pq, pt, prm, threads))})
        rows.append({"variant": "fused multiply-adds + all of the above", "flags": "fma|0xFF",
                     **compare(base, run(F, 0x0FF, descs, xyzs, pq, pt, prm, threads))})
    edges = sum(1 for b in base if b[0] >= 0)
    return {"workload": "configs[1], %d pairs of one bench step, depth noise %.3f z^2" % (len(pq), depth_noise),
            "edges_in_baseline": edges, "rows": rows}


def markdown(st):
    out = ["| variant | edge decision kept | final inlier set kept | nothing flipped (final set and the set the pose was fitted from) | max pose dev, nothing flipped | max pose dev, any pair with an edge on both sides |",
           "|---|---|---|---|---|---|"]
    for r in st["rows"]:
        out.append("| %s | %.2f %% | %.2f %% | %.2f %% | %.2e | %.2e |" % (
            r["variant"], r["edge_decision_kept_pct"], r["inlier_set_kept_pct"], r["nothing_flipped_pct"],
            r["max_pose_dev_nothing_flipped"], r["max_pose_dev_any_edge"]))
    return "\n".join(out)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    dn = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    st = study(n, dn)
    print(json.dumps(st, indent=1))
    print(markdown(st))
