"""SIFT extraction -> matcher -> RANSAC, end to end against the reference (VERDICT r3 "missing" 2).

Two views of a scene go through the whole SIFTGPU configuration of the reference twice:
  reference side   SiftGPU's own CUDA kernels + host code compiled on a CPU emulation (oracle/_ref/libref_siftgpu.so:
                   SiftGPUWrapper::detect, sift_gpu_wrapper.cpp:113-167)
  product side     rgbdfe_sift_detect (csrc/sift_extract.hip)
then, for both alike: projectTo3DSiftGPU (node.cpp:695-769) -> the node's siftgpu_descriptors exactly as the wrapper
returns them ("-unn": unnormalised) -> SiftMatchGPU (quantises 512 * d to bytes, SiftMatchCU.cpp:87-100) -> keepStrongest
-> RANSAC, the last three on the GPU pair path (bit-equal to the oracle: tests/test_gpu_sift.py), so that every
difference reported here comes from the extraction alone.

Views: the picture pairs SiftGPU ships (640-k.jpg / 800-k.jpg = one photograph at two sizes; as a camera motion that is a
pure translation towards a fronto-parallel plane: depth 2.0 m for the small picture, 1.6 m for the large one, same
intrinsics) and seeded synthetic views of a textured plane (rgbdslam_v2_amd.synth.make_image_sequence).
Reports, per pair: features, position differences, quantised descriptor bytes that differ, match-list differences, the edge
decision, inlier counts and the pose difference.

    python tools/sift_e2e.py            (GPU box; prints one JSON report)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FX = FY = 525.0
CX, CY = 319.5, 239.5


def plane_depth(shape, z, seed, noise=0.001):
    """A fronto-parallel plane at depth z with seeded sensor noise sigma = noise * z^2 (f32, metres)."""
    rng = np.random.default_rng(seed)
    return (z + rng.normal(0.0, noise * z * z, shape)).astype(np.float32)


def quantise(desc):
    """SiftMatchCU.cpp:96-99: pub[i] = int(512 * descriptors[i] + 0.5) stored as unsigned char."""
    return (512.0 * np.asarray(desc, np.float64) + 0.5).astype(np.int64).astype(np.uint8)


def views():
    """[(name, image A (newer node), depth A, image B (older node), depth B)]"""
    from rgbdslam_v2_amd import synth
    out = []
    g = np.load(os.path.join(ROOT, "tests", "golden", "sift_photo_pairs.npz"))
    for k in (1, 2):
        big, small = g["img_800_%d" % k], g["img_640_%d" % k]
        out.append(("photo_%d" % k, big, plane_depth(big.shape, 1.6, 10 + k), small, plane_depth(small.shape, 2.0, 20 + k)))
    for seed in (5, 9):
        seq = synth.make_image_sequence(n_frames=2, seed=seed)
        out.append(("synthetic_%d" % seed, seq["gray"][1], seq["depth"][1], seq["gray"][0], seq["depth"][0]))
    return out


def siftgpu_normalise(raw):
    """What SiftGPU delivers WITHOUT "-unn" (NormalizeDescriptor_Kernel, ProgramCU.cu:1117-1160): L2-normalise, clamp at 0.2,
    normalise again."""
    d = raw / np.maximum(np.linalg.norm(raw, axis=1, keepdims=True), 1e-12)
    d = np.minimum(d, 0.2)
    return (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)).astype(np.float32)


def node_from(fe, po, gray, depth, side, max_features=1000):
    """keys [n, 4] (x, y, scale, orientation), kept rows, xyz1 [k, 4], siftgpu_descriptors [k, 128] (raw, "-unn"),
    feature_descriptors_ [k, 128] (root-SIFT, node.cpp:1557-1571)"""
    if side == "reference":
        keys, desc, _ = po.ref_sift_detect(gray, max_features)
        xy = keys[:, :2].astype(np.float32)
        kept, xyz1, raw, feat = po.sift_node_features(xy, desc, depth, FX, FY, CX, CY, max_keypoints=4096)
    else:
        kp, desc = fe.sift_detect(gray, None, max_features)
        keys = np.stack([kp["x"], kp["y"], kp["size"] / 12.0, kp["angle"] * 3.1415927 / 180.0], 1).astype(np.float32)
        xy = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
        kept, xyz1, raw, feat = fe.sift_node_features(xy, desc, depth, FX, FY, CX, CY, max_keypoints=4096)
    return dict(keys=keys, kept=kept, xyz1=xyz1, raw=raw, feat=feat)


# the three ways the reference can be configured to match SIFTGPU features
VARIANTS = (
    # matcher_type SIFTGPU as the wrapper is written ("-unn": descriptors of norm ~2 go to a matcher that stores
    # int(512 d + 0.5) in a byte): the bytes wrap and next to nothing matches -- on both sides alike
    ("siftgpu_matcher_unn", "sift", lambda nd: nd["raw"]),
    # the same matcher on what SiftGPU delivers without "-unn"
    ("siftgpu_matcher_normalised", "sift", lambda nd: siftgpu_normalise(nd["raw"])),
    # matcher_type FLANN (the reference's default with SIFTGPU features): root-SIFT descriptors, ratio test 0.95
    ("flann_rootsift", "float", lambda nd: nd["feat"]),
)


def evaluate(fe, po, only=None):
    from rgbdslam_v2_amd.frontend import inlier_indices
    report = {}
    for name, ga, da, gb, db in views():
        if only and name not in only:
            continue
        rec = {}
        nodes = {}
        for side in ("reference", "product"):
            for tag, g, d in (("a", ga, da), ("b", gb, db)):
                nodes[side, tag] = node_from(fe, po, g, d, side)
        flips = total = pos_diff = 0
        max_step = 0
        worst_rel = 0.0
        same_lists = True
        for tag in ("a", "b"):
            r, p = nodes["reference", tag], nodes["product", tag]
            rec["features_" + tag] = [int(len(r["keys"])), int(len(p["keys"]))]
            if len(r["keys"]) != len(p["keys"]) or not np.array_equal(r["kept"], p["kept"]):
                same_lists = False
                continue
            pos_diff += int((r["keys"][:, :2] != p["keys"][:, :2]).any(1).sum()) + int((r["xyz1"] != p["xyz1"]).any(1).sum())
            qa, qb = quantise(siftgpu_normalise(r["raw"])), quantise(siftgpu_normalise(p["raw"]))
            dq = np.abs(qa.astype(np.int32) - qb.astype(np.int32))
            flips += int((dq != 0).sum())
            total += int(dq.size)
            max_step = max(max_step, int(dq.max()) if dq.size else 0)
            rel = np.linalg.norm(r["raw"] - p["raw"], axis=1) / np.maximum(np.linalg.norm(r["raw"], axis=1), 1e-12)
            worst_rel = max(worst_rel, float(rel.max()) if len(rel) else 0.0)
        rec.update(feature_lists_identical=same_lists, position_or_point_differences=pos_diff,
                   quantised_bytes=total, quantised_bytes_that_differ=flips, largest_byte_step=max_step,
                   largest_descriptor_relative_l2_difference=worst_rel)
        for vname, kind, pick in VARIANTS:
            out = {}
            for side in ("reference", "product"):
                ia, ib = 1, 0   # the same node ids on both sides: the pair's RANSAC draws are a function of them
                up = fe.upload_sift_node if kind == "sift" else fe.upload_float_node
                up(ia, pick(nodes[side, "a"]), nodes[side, "a"]["xyz1"])
                up(ib, pick(nodes[side, "b"]), nodes[side, "b"]["xyz1"])
                r, _ = (fe.match_sift_pair_list if kind == "sift" else fe.match_flann_pair_list)([ia], [ib])
                fe.release_node(ia)
                fe.release_node(ib)
                r = r[0]
                n = int(r["n_all"])
                out[side] = dict(rec=r, matches=list(zip(r["all_q"][:n].tolist(), r["all_t"][:n].tolist())),
                                 inliers=inlier_indices(r), T=np.array(r["trafo"], np.float32).reshape(4, 4).T)
            ra, rb = out["reference"], out["product"]
            ma, mb = set(ra["matches"]), set(rb["matches"])
            ia_ = set(ra["matches"][i] for i in ra["inliers"])
            ib_ = set(rb["matches"][i] for i in rb["inliers"])
            rec[vname] = dict(
                matches=[len(ma), len(mb)], matches_only_on_one_side=len(ma ^ mb), match_lists_identical=ra["matches"] == rb["matches"],
                edge=[bool(ra["rec"]["id1"] >= 0), bool(rb["rec"]["id1"] >= 0)],
                inliers=[int(ra["rec"]["n_inl"]), int(rb["rec"]["n_inl"])], inliers_only_on_one_side=len(ia_ ^ ib_),
                pose_max_abs_diff=float(np.abs(ra["T"] - rb["T"]).max()),
                translation=[[round(float(v), 5) for v in ra["T"][:3, 3]], [round(float(v), 5) for v in rb["T"][:3, 3]]],
                rmse=[float(ra["rec"]["rmse"]), float(rb["rec"]["rmse"])])
        report[name] = rec
    return report


def main():
    from oracle import pyoracle as po
    from rgbdslam_v2_amd.frontend import FrontEnd
    if po.ref_siftgpu_lib() is None:
        raise SystemExit("oracle/_ref/libref_siftgpu.so is not built (python __graft_entry__.py where /root/reference exists)")
    fe = FrontEnd(device_id=0, max_nodes=8, max_keypoints=4096, max_pairs_per_batch=8)
    try:
        print(json.dumps(evaluate(fe, po), indent=1))
    finally:
        fe.close()


if __name__ == "__main__":
    main()
