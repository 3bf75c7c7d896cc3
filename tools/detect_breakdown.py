import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
seq = synth.make_image_sequence(n_frames=8, seed=1)
masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
fe = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
fe.detector_configure(max_keypoints=1000)
K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
for f in range(8):
    fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], *K)
def t(fn, n=40):
    t0 = time.perf_counter()
    for i in range(n): fn(i % 8)
    return (time.perf_counter() - t0) / n * 1e6
print("detect_describe   %.0f us" % t(lambda f: fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], *K)))
kps = [fe.orb_detect(seq["gray"][f], masks[f], 20) for f in range(8)]
print("orb_detect (one threshold, whole frame, %d kp) %.0f us" % (len(kps[0]), t(lambda f: fe.orb_detect(seq["gray"][f], masks[f], 20))))
print("orb_compute       %.0f us" % t(lambda f: fe.orb_compute(seq["gray"][f], kps[f][:1000])))
xy = [np.stack([k["x"], k["y"]], 1) for k in kps]
print("project_to_3d     %.0f us" % t(lambda f: fe.project_to_3d(xy[f], seq["depth"][f], *K)))
print("depth_to_mono8    %.0f us" % t(lambda f: fe.depth_to_mono8(seq["depth"][f])))
