"""Soak of the batch entry points whose read-backs are written by the kernels themselves (round 6): runs of random length through
rgbdfe_detect_describe_batch / rgbdfe_sift_detect_batch, again and again on ONE context, every output compared with the first
result for the same frames from the same detector state (fresh detector per run).   python tools/stress_readback.py [runs]"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(7)
seq = synth.make_image_sequence(n_frames=20, seed=3, width=640, height=480)
masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
fe = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
seen, bad = {}, 0
for r in range(runs):
    n = int(rng.integers(1, 45))
    first = int(rng.integers(0, 20))
    idx = [(first + k) % 20 for k in range(n)]
    if r % 3 == 2:
        out = fe.sift_detect_batch([seq["gray"][i] for i in idx], copy=bool(r & 1))
        key = ("sift", first, n)
        crc = zlib.crc32(b"".join(k.tobytes() + d.tobytes() for k, d in out))
    else:
        fe.detector_configure(max_keypoints=1000)          # createDetector's state: the run's outputs depend on the frames only
        out = fe.detect_describe_batch([seq["gray"][i] for i in idx], [masks[i] for i in idx], [seq["depth"][i] for i in idx], *K,
                                       copy=bool(r & 1))
        key = ("orb", first, n)
        crc = zlib.crc32(b"".join(k.tobytes() + d.tobytes() + x.tobytes() for k, d, x in out))
    # the same (kind, first, n) must give the same bytes whenever it comes up again; shorter runs are prefixes of longer ones' inputs
    # only for SIFT (stateless), so only exact repeats are compared
    if key in seen and seen[key] != crc:
        bad += 1
        print("MISMATCH", key)
    seen.setdefault(key, crc)
# exact repeats are rare with random lengths: repeat every recorded key once more, in another order
for key in list(seen)[::-1]:
    kind, first, n = key
    idx = [(first + k) % 20 for k in range(n)]
    if kind == "sift":
        out = fe.sift_detect_batch([seq["gray"][i] for i in idx])
        crc = zlib.crc32(b"".join(k.tobytes() + d.tobytes() for k, d in out))
    else:
        fe.detector_configure(max_keypoints=1000)
        out = fe.detect_describe_batch([seq["gray"][i] for i in idx], [masks[i] for i in idx], [seq["depth"][i] for i in idx], *K)
        crc = zlib.crc32(b"".join(k.tobytes() + d.tobytes() + x.tobytes() for k, d, x in out))
    if crc != seen[key]:
        bad += 1
        print("MISMATCH on repeat", key)
fe.close()
print("readback soak done: %d runs + %d repeats, mismatches: %d" % (runs, len(seen), bad))
