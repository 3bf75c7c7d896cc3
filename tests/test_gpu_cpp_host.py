"""-m gpu: the C++ host side (include/rgbdfe.hpp: Node::matchNodePair, GraphManager::nodeComparisons)
built with plain g++ against the C ABI gives the same results as the oracle."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "examples", "cpp", "match_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples", "cpp")])
    F = 5
    seq = synth.make_sequence(n_frames=F, n_kp=700, n_world=2800, seed=6)
    path = tmp_path / "nodes.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<i", F))
        for k in range(F):
            f.write(struct.pack("<i", 700))
            f.write(seq["desc"][k].tobytes())
            f.write(seq["xyz1"][k].tobytes())
    out = subprocess.check_output([exe, str(path)], text=True, timeout=120)
    lines = [json.loads(l) for l in out.strip().splitlines()]
    assert len(lines) == F + 2
    # getPotentialEdgeTargetsWithDijkstra(2 sequential, 1 geodesic, 1 sampled) over the 4 earlier nodes: no more nodes
    # than targets, so all of them, sequentially from the predecessor (graph_manager.cpp:212-227)
    assert lines[-3]["candidates"] == [2, 1, 0]
    prm = po.default_params()
    for t, rec in enumerate(lines[:-3]):
        ref = po.match_node_pair(seq["desc"][F - 1], seq["xyz1"][F - 1], F - 1, seq["desc"][t], seq["xyz1"][t], t, prm)
        assert (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
        assert rec["n_all"] == ref["n_all"] and rec["n_inl"] == ref["n_inl"]
        T = np.array(rec["T"], np.float32).reshape(4, 4).T
        assert np.array_equal(T, ref["T"])
        assert np.float32(rec["rmse"]) == ref["rmse"] and rec["info"] == ref["info_scale"]
    assert lines[-2]["single_id1"] == lines[0]["id1"] and lines[-2]["single_n_inl"] == lines[0]["n_inl"]
    # GraphManager::getNeighbours through the C++ layer == the Python binding's ranking == the numpy oracle
    ref_pos, _ = po.place_recognition(seq["desc"][F - 1], [seq["desc"][t] for t in range(F - 1)], 2, 256)
    assert lines[-1]["devices"] == 1 and lines[-1]["neighbours"] == [int(i) for i in ref_pos]   # node id == position
    # the same program over two device contexts behind one handle (rgbdfe_create_multi): same lines
    out2 = subprocess.check_output([exe, str(path), "multi"], text=True, timeout=120)
    lines2 = [json.loads(l) for l in out2.strip().splitlines()]
    assert lines2[-1]["devices"] == 2
    assert lines2[:-1] == lines[:-1] and lines2[-1]["neighbours"] == lines[-1]["neighbours"]


def test_cpp_depth_image_node_constructor(tmp_path):
    """include/rgbdfe.hpp's Node(gray, mask, depth, ...) -- the reference's depth-image constructor (node.cpp:139-210) --
    from a plain g++ program: the same feature counts, descriptor bytes and edge as the Python binding's detect_describe
    + upload + match on the same two frames."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    exe = os.path.join(ROOT, "examples", "cpp", "match_demo")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples", "cpp")], stdout=subprocess.DEVNULL)
    seq = synth.make_sequence(n_frames=2, n_kp=300, n_world=1200, seed=6)
    nodes = tmp_path / "nodes.bin"
    with open(nodes, "wb") as f:
        f.write(struct.pack("<i", 2))
        for k in range(2):
            f.write(struct.pack("<i", 300))
            f.write(seq["desc"][k].tobytes())
            f.write(seq["xyz1"][k].tobytes())
    img = synth.make_image_sequence(n_frames=2, seed=1)
    K = (float(img["fx"]), float(img["fy"]), float(img["cx"]), float(img["cy"]))
    masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in img["mask"]]
    frames = tmp_path / "frames.bin"
    with open(frames, "wb") as f:
        rows, cols = img["gray"][0].shape
        f.write(struct.pack("<ii", rows, cols))
        f.write(struct.pack("<dddd", *K))
        for k in range(2):
            f.write(np.ascontiguousarray(img["gray"][k], np.uint8).tobytes())
            f.write(masks[k].tobytes())
            f.write(np.ascontiguousarray(img["depth"][k], np.float32).tobytes())
        # an organised cloud of the second frame for the point-cloud constructor
        cloud = po.create_point_cloud(np.ascontiguousarray(img["depth"][1], np.float32), *K, cloud_skip=1)
        assert cloud.shape == (rows, cols, 4)
        f.write(np.ascontiguousarray(cloud, np.float32).tobytes())
    out = subprocess.check_output([exe, str(nodes), "single", str(frames)], text=True, timeout=120)
    rec_sift = json.loads(out.strip().splitlines()[-1])
    rec = json.loads(out.strip().splitlines()[-2])
    rec_cloud = json.loads(out.strip().splitlines()[-3])
    fe = FrontEnd(device_id=0, max_nodes=4, max_keypoints=2048, max_pairs_per_batch=8)
    fe.detector_configure(max_keypoints=1000)
    feats = [fe.detect_describe(img["gray"][k], masks[k], img["depth"][k], *K) for k in range(2)]
    assert rec["frame_features"] == [len(feats[0][0]), len(feats[1][0])]
    h = 0
    for b in feats[1][1].reshape(-1).tolist():
        h = (h * 131 + b) % (1 << 64)
    assert rec["desc_hash"] == h
    # the point-cloud constructor ran third on the demo's detector (its thresholds carry over): same call order here
    ck, cd, cx3 = fe.detect_describe_cloud(img["gray"][1], masks[1], cloud, 3.5)
    hc = 0
    for b in cd.reshape(-1).tolist():
        hc = (hc * 131 + b) % (1 << 64)
    assert rec_cloud["cloud_features"] == len(ck) and rec_cloud["cloud_desc_hash"] == hc and len(ck) > 100
    zs = np.float32(0)
    for z in cx3[:, 2]:
        zs = np.float32(zs + z)
    assert np.float32(rec_cloud["cloud_zsum"]) == zs
    for k in range(2):   # the demo numbers its frame nodes behind the two file nodes: ids 2 and 3 (the draws depend on them)
        fe.upload_node(2 + k, feats[k][1], feats[k][2])
    r = fe.match_pair_list(np.array([3], np.int32), np.array([2], np.int32))[0]
    assert rec["frame_edge"] == [int(r["id1"]), int(r["id2"])] and rec["frame_edge"] == [2, 3]
    assert rec["frame_inliers"] == int(r["n_inl"]) and rec["frame_inliers"] > 20
    # the SiftGPU node constructor (SiftGPUWrapper::detect -> projectTo3DSiftGPU -> upload) and the SIFTGPU matcher branch
    sn = []
    for k in range(2):
        kp, d = fe.sift_detect(img["gray"][k], None, 1000)
        xy = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
        kept, xyz1, raw, _ = fe.sift_node_features(xy, d, img["depth"][k], *K, 1.0, 1000, False)
        fe.upload_sift_node(200 + k, raw, xyz1)
        sn.append(raw)
    rs = fe.match_sift_pair_list(np.array([201], np.int32), np.array([200], np.int32))[0][0]
    assert rec_sift["siftgpu_features"] == [len(sn[0]), len(sn[1])] and len(sn[1]) > 200
    # (whether this pair yields an edge is not the point: the reference extracts with "-unn" and its matcher quantises
    # 512 * d to bytes, so the unnormalised descriptors saturate -- reproduced as found; the C++ layer and
    # the Python mirror must agree on whatever comes out)
    assert rec_sift["siftgpu_edge"] == [int(rs["id1"]), int(rs["id2"])]
    assert rec_sift["siftgpu_inliers"] == int(rs["n_inl"])
    assert abs(rec_sift["siftgpu_desc_sum"] - float(sn[1].astype(np.float64).sum())) < 1e-6 * rec_sift["siftgpu_desc_sum"]
    fe.close()


def test_cpp_sift_nodes(tmp_path):
    """include/rgbdfe.hpp with 128-d float descriptors (Node::SiftDescriptors: matcher_type == "SIFTGPU"): nodeComparisons and
    matchNodePair from a plain g++ program give the oracle's match counts, inliers, edge ids and DMatch distances."""
    exe = os.path.join(ROOT, "examples", "cpp", "match_demo")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples", "cpp")], stdout=subprocess.DEVNULL)
    F = 4
    seq = synth.make_sequence(n_frames=F, n_kp=500, n_world=2000, seed=8)
    sd = synth.sift_descriptors_like(seq["desc"], seed=8)
    nodes = tmp_path / "nodes.bin"
    with open(nodes, "wb") as f:   # the ORB section of the demo needs its file too
        f.write(struct.pack("<i", 2))
        for k in range(2):
            f.write(struct.pack("<i", 500))
            f.write(seq["desc"][k].tobytes())
            f.write(seq["xyz1"][k].tobytes())
    sift = tmp_path / "sift.bin"
    with open(sift, "wb") as f:
        f.write(struct.pack("<i", F))
        for k in range(F):
            f.write(struct.pack("<i", 500))
            f.write(np.ascontiguousarray(sd[k], np.float32).tobytes())
            f.write(seq["xyz1"][k].tobytes())
    out = subprocess.check_output([exe, str(nodes), "single", "-", str(sift)], text=True, timeout=120)
    lines = [json.loads(l) for l in out.strip().splitlines() if "sift_" in l]
    assert len(lines) == F   # F - 1 comparisons + the single call
    prm = po.default_params()
    for t, rec in enumerate(lines[:-1]):
        ref = po.match_sift_node_pair(sd[F - 1], seq["xyz1"][F - 1], 100 + F - 1, sd[t], seq["xyz1"][t], 100 + t, prm)
        assert (rec["sift_id1"], rec["sift_id2"]) == (ref["id1"], ref["id2"])
        assert rec["sift_n_all"] == ref["n_all"] and rec["sift_n_inl"] == ref["n_inl"]
        assert abs(rec["sift_dist_sum"] - float(np.sum(ref["all_dist"].astype(np.float64)))) < 1e-6 * max(1.0, rec["sift_dist_sum"])
    assert lines[-1]["sift_single_n_inl"] == lines[0]["sift_n_inl"]
