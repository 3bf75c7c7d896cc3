"""CPU: the oracle's frame-level functions against tests/golden/frame_golden.npz -- inputs and the outputs of the
reference's OWN Node::projectTo3D / removeDepthless (src/node.cpp:66-97, 900-965), projectTo3DSiftGPU (:695-769) and
squareroot_descriptor_space (:1557-1571), compiled from the reference tree when the fixture was made
(tests/golden/make_golden.py, oracle/_ref/libref_frame.so).  Needs neither the reference tree nor the pin."""
import os

import numpy as np

from oracle import pyoracle as po

GOLD = os.path.join(os.path.dirname(__file__), "golden", "frame_golden.npz")


def test_project_to_3d_golden():
    g = np.load(GOLD)
    for tag in "ab":
        kp, depth, K, maxk = g[f"p3d_{tag}_kp"], g[f"p3d_{tag}_depth"], g[f"p3d_{tag}_K"], int(g[f"p3d_{tag}_maxk"])
        kept, xyz = po.project_to_3d(kp, depth, *[float(v) for v in K[:4]], float(K[4]), maxk)
        assert np.array_equal(kept, g[f"p3d_{tag}_kept"]) and np.array_equal(xyz, g[f"p3d_{tag}_xyz"])
        kept2, _ = po.project_to_3d(kp, depth, *[float(v) for v in K[:4]], 1.0, 10 ** 9)  # removeDepthless: no cut
        assert np.array_equal(kept2, g[f"p3d_{tag}_depthless_kept"])


def test_sift_node_features_golden():
    g = np.load(GOLD)
    K = g["sift_K"]
    kept, xyz, raw, feat = po.sift_node_features(g["sift_kp"], g["sift_desc"], g["sift_depth"], *[float(v) for v in K[:4]],
                                                 float(K[4]), int(g["sift_maxk"]), use_root_sift=True)
    assert np.array_equal(kept, g["sift_kept"]) and np.array_equal(xyz, g["sift_xyz"])
    assert np.array_equal(raw, g["sift_raw"]) and np.array_equal(feat, g["sift_root"])
