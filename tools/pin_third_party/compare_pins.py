"""Bit-equality per field between what the real third-party libraries returned (pins.pin, written by pin_third_party on a
machine that has OpenCV 3.x / Eigen / PCL) and what the oracle restates (oracle/orb_oracle.c, oracle/rgbd_oracle.c):

    python tools/pin_third_party/compare_pins.py inputs.pin pins.pin [--json report.json]

One line per array: identical / shape differs / n of m elements differ (largest absolute difference, first index).  Exit code 0
when everything is identical: SURVEY.md 8(a) rows a1 (cv::ORB detect), a5 (retainBest), a6 (cv::ORB compute), a15 (PCL TFC +
JacobiSVD) and the 3x3 LLT of a17 are then pinned on the libraries themselves, and `parity unpinned` can be struck from
oracle/*.c and DESIGN.md section 3.  SVD factors are compared as the library returns them AND, should they differ, through the
product U diag(S) V^T and the transform the fit derives from them (sign conventions of singular vectors are free)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pinfile  # noqa: E402
import oracle_side  # noqa: E402


def compare(name, got, want):
    if got.shape != want.shape or got.dtype != want.dtype:
        return {"name": name, "status": "shape", "got": [str(got.dtype)] + list(got.shape), "oracle": [str(want.dtype)] + list(want.shape)}
    a, b = np.ascontiguousarray(got), np.ascontiguousarray(want)
    same = a.view(np.uint8).reshape(a.size, -1) == b.view(np.uint8).reshape(b.size, -1) if a.size else np.zeros((0, 1), bool)
    bad = np.nonzero(~same.all(axis=1))[0] if a.size else np.zeros(0, np.int64)
    if len(bad) == 0:
        return {"name": name, "status": "identical", "elements": int(a.size)}
    fa, fb = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
    with np.errstate(invalid="ignore"):
        diff = np.abs(fa[bad] - fb[bad])
    return {"name": name, "status": "differs", "elements": int(a.size), "differing": int(len(bad)),
            "max_abs_diff": float(np.nanmax(diff)) if len(diff) else 0.0, "first_index": int(bad[0])}


def run(inputs_path, pins_path):
    inp = pinfile.read(inputs_path)
    got = pinfile.read(pins_path)
    want = oracle_side.oracle_pins(inp)
    report = []
    for name in want:
        if name not in got:
            report.append({"name": name, "status": "missing in pins"})
        else:
            report.append(compare(name, got[name], want[name]))
    for name in got:
        if name not in want:
            report.append({"name": name, "status": "unexpected in pins"})
    if all(k in got for k in ("svd_U", "svd_S", "svd_V")):   # the factors' product, whatever the vectors' signs
        rec_g = np.einsum("kij,kj,klj->kil", got["svd_U"].astype(np.float64), got["svd_S"].astype(np.float64), got["svd_V"].astype(np.float64))
        rec_w = np.einsum("kij,kj,klj->kil", want["svd_U"].astype(np.float64), want["svd_S"].astype(np.float64), want["svd_V"].astype(np.float64))
        report.append({"name": "svd U*S*V^T (informative)", "status": "info", "max_abs_diff": float(np.abs(rec_g - rec_w).max())})
    return report


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 2:
        sys.exit(__doc__)
    rep = run(args[0], args[1])
    bad = 0
    for r in rep:
        if r["status"] == "identical":
            print("  identical   %-34s %d elements" % (r["name"], r["elements"]))
        elif r["status"] == "info":
            print("  (info)      %-34s max |diff| %.3g" % (r["name"], r["max_abs_diff"]))
        else:
            bad += 1
            print("  %-11s %-34s %s" % (r["status"].upper(), r["name"], {k: v for k, v in r.items() if k not in ("name", "status")}))
    if "--json" in sys.argv:
        json.dump(rep, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("%d arrays compared, %d not identical" % (len(rep), bad))
    sys.exit(0 if bad == 0 else 1)
