"""CPU: the oracle's frame-level functions against tests/golden/frame_golden.npz -- inputs and the outputs of the
reference's OWN Node::projectTo3D / removeDepthless (src/node.cpp:66-97, 900-965), projectTo3DSiftGPU (:695-769),
squareroot_descriptor_space (:1557-1571), the point-cloud and min-depth projections, createXYZRGBPointCloud and
observationLikelihood (src/misc.cpp), compiled from the reference tree when the fixture was made
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


def test_point_cloud_and_observation_likelihood_golden():
    """createXYZRGBPointCloud (misc.cpp:467-556) and observationLikelihood (misc.cpp:814-969) as the reference's own code
    computed them for three small depth frames."""
    g = np.load(GOLD)
    K = [float(v) for v in g["emm_K"]]
    clouds = []
    for f in range(3):
        c = po.create_point_cloud(g["emm_depth"][f], *K, rgb=g["emm_gray"][f], encoding_bgr=False, depth_scaling=1.0,
                                  min_depth=0.1, cloud_skip=2)
        assert np.array_equal(c.view(np.uint32), g["emm_clouds"][f].view(np.uint32))
        clouds.append(c)
    for (n, o), T, ref in zip(g["emm_jobs"], g["emm_T"], g["emm_counts"]):
        got = po.observation_likelihood(clouds[n], clouds[o], T, *K, cloud_skip=2, skip_step=8, depth_cov=1e-4)
        assert list(got) == list(ref), (n, o)
    assert np.all(g["emm_counts"].sum(0)[:3] > 0)


def test_cloud_projection_and_min_depth_golden():
    """Node::projectTo3D, point-cloud overload (node.cpp:855-898), and the use_feature_min_depth variant
    (getMinDepthInNeighborhood, misc.cpp:774-793, inside projectTo3D :940)."""
    g = np.load(GOLD)
    kept, xyz = po.project_to_3d_cloud(g["cloudp_kp"], g["cloudp_cloud"], float(g["cloudp_maxd"]), int(g["cloudp_maxk"]))
    assert np.array_equal(kept, g["cloudp_kept"]) and np.array_equal(xyz, g["cloudp_xyz"])
    K = [float(v) for v in g["mind_K"]]
    kept, xyz = po.project_to_3d_min_depth(g["mind_kp"], g["mind_size"], g["mind_depth"], *K[:4], K[4], 1000)
    assert np.array_equal(kept, g["mind_kept"]) and np.array_equal(xyz, g["mind_xyz"])
