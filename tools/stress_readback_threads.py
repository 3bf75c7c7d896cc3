"""Three threads, a context each, running ORB and SIFT batch calls (read-backs written by the kernels) at once: every output
against a single-threaded reference of the same call.   python tools/stress_readback_threads.py"""
import os, sys, zlib, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
seq = synth.make_image_sequence(n_frames=20, seed=3, width=640, height=480)
masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
def run(fe, kind, first, n):
    idx = [(first + k) % 20 for k in range(n)]
    if kind == "sift":
        out = fe.sift_detect_batch([seq["gray"][i] for i in idx])
        return zlib.crc32(b"".join(k.tobytes() + d.tobytes() for k, d in out))
    fe.detector_configure(max_keypoints=1000)
    out = fe.detect_describe_batch([seq["gray"][i] for i in idx], [masks[i] for i in idx], [seq["depth"][i] for i in idx], *K)
    return zlib.crc32(b"".join(k.tobytes() + d.tobytes() + x.tobytes() for k, d, x in out))
jobs = [("orb", 0, 30), ("sift", 3, 17), ("orb", 5, 14), ("orb", 11, 41), ("sift", 0, 9), ("orb", 2, 1)]
fe0 = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
ref = {j: run(fe0, *j) for j in jobs}
fe0.close()
bad = []
def worker(t):
    fe = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
    for r in range(25):
        j = jobs[(r + t) % len(jobs)]
        if run(fe, *j) != ref[j]:
            bad.append((t, r, j))
    fe.close()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads soak done, mismatches:", len(bad), bad[:3])
