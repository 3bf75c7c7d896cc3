"""One-off fuzz of the frame-level entry points (GPU) against the oracle: image sizes, masks, empty inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po, pyorb
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
bad = 0
def chk(name, ok, info=""):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", name, info, flush=True)
def kps_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("x", "y", "size", "angle", "response", "octave"))
fe = FrontEnd(device_id=0, max_nodes=8, max_keypoints=2048, max_pairs_per_batch=16)
rng = np.random.default_rng(77)
# ---- Hamming NN, host entry point: sizes around the tile / split boundaries
for nq, nt in ((0, 5), (5, 0), (1, 1), (1, 2), (2, 1), (63, 64), (65, 513), (512, 2), (513, 1025), (2048, 2047), (3, 2048)):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nq and nt: q[: min(nq, nt)] = t[: min(nq, nt)][::-1]
    print("hamming", nq, nt, flush=True)
    hd, idx = fe.bruteForceSearchORB_batch(q, t)
    rhd, ridx = po.hamming_nn_batch(q, t)
    chk("hamming", np.array_equal(hd, rhd) and np.array_equal(idx, ridx), (nq, nt))
# ---- images of several sizes through detect / compute / detect_describe
for (h, w, maxkp, grid) in ((480, 640, 1000, 3), (240, 320, 300, 3), (120, 160, 100, 2), (96, 128, 50, 1), (479, 641, 600, 3), (960, 1280, 2000, 3)):
    seq = synth.make_image_sequence(n_frames=3, width=w, height=h, seed=h + w)
    fe.detector_configure(max_keypoints=maxkp, grid_resolution=grid, adjuster_max_iterations=5)
    st = pyorb.grid_state(maxkp, grid, 5)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    for f in range(3):
        g, d = seq["gray"][f], seq["depth"][f]
        m = np.where(seq["mask"][f] > 0, 255, 0).astype(np.uint8)
        if f == 1: m[:, : w // 2] = 0                      # half of the frame without depth
        if f == 2: g = np.full_like(g, 90)                 # no texture at all: zero keypoints
        print("detect_describe", h, w, f, flush=True)
        kp, desc, xyz = fe.detect_describe(g, m, d, *K)
        rk, rdesc = pyorb.node_features(st, g, m, d, maxkp)
        chk("detect_describe kps", kps_equal(kp, rk), (h, w, f, len(kp), len(rk)))
        chk("detect_describe desc", np.array_equal(desc, rdesc), (h, w, f))
        chk("thresholds", np.array_equal(fe.detector_thresholds()[: grid * grid], np.array(st.thresh[: grid * grid])), (h, w, f))
    g = seq["gray"][0]
    for thr in (5, 20, 80):
        print("orb_detect", h, w, thr, flush=True)
        a, b = fe.orb_detect(g, None, thr), pyorb.detect(g, None, thr)
        chk("orb_detect", kps_equal(a, b), (h, w, thr, len(a), len(b)))
    for nk in (0, 1, 7):
        sel = b[:nk]
        k1, d1 = fe.orb_compute(g, sel); k2, d2 = pyorb.compute(g, sel)
        chk("orb_compute", kps_equal(k1, k2) and np.array_equal(d1, d2), (h, w, nk))
# ---- clouds + EMM with odd sizes
for (h, w, s, skip) in ((480, 640, 2, 8), (480, 640, 1, 16), (96, 128, 4, 3), (48, 64, 8, 1), (30, 40, 2, 7)):
    base = synth.make_depth_sequence(n_frames=3, width=w, height=h, nan_fraction=0.1)
    K = (base["fx"], base["fy"], base["cx"], base["cy"])
    clouds = []
    for f in range(3):
        print("cloud", h, w, s, f, flush=True)
        c = fe.upload_node_cloud(f, base["depth"][f], *K, cloud_skip=s, return_cloud=True)
        r = po.create_point_cloud(base["depth"][f], *K, cloud_skip=s)
        chk("cloud", np.array_equal(c.view(np.uint32), r.view(np.uint32)), (h, w, s))
        clouds.append(r)
    ids_n = [0, 1, 2, 0, 2]; ids_o = [1, 2, 0, 0, 1]
    Ts = np.stack([synth.relative_pose(base["poses"], a, b) for a, b in zip(ids_n, ids_o)]).astype(np.float32)
    Ts[4, :3, 3] += 0.3
    print("emm", h, w, s, skip, flush=True)
    got = fe.observation_likelihood(ids_n, ids_o, Ts, skip)
    for k in range(5):
        ref = po.observation_likelihood(clouds[ids_n[k]], clouds[ids_o[k]], Ts[k], *K, cloud_skip=s, skip_step=skip, depth_cov=fe.params.depth_cov)
        chk("emm", list(got[k]) == list(ref), (h, w, s, skip, k, list(got[k]), list(ref)))
    for f in range(3): fe.release_node_cloud(f)
# ---- projection helpers with empty / tiny inputs
d = rng.uniform(0.5, 3, (48, 64)).astype(np.float32)
for n in (0, 1, 2, 300):
    kp = np.stack([rng.uniform(0, 63, n), rng.uniform(0, 47, n)], 1).astype(np.float32).reshape(-1, 2)
    a = fe.project_to_3d(kp, d, 50, 50, 32, 24, 1.0, 10); b = po.project_to_3d(kp, d, 50, 50, 32, 24, 1.0, 10)
    chk("project", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), n)
    desc = rng.random((n, 128)).astype(np.float32)
    a = fe.sift_node_features(kp, desc, d, 50, 50, 32, 24, 1.0, 10); b = po.sift_node_features(kp, desc, d, 50, 50, 32, 24, 1.0, 10)
    chk("sift_node_features", all(np.array_equal(x, y) for x, y in zip(a, b)), n)
print("misc fuzz done, mismatches:", bad)
