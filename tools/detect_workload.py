"""Profiling workload for the frame-level paths (tools/profile_round.sh): runs N frames through one entry point and prints
{"frames": total frames processed in this process} so that per-frame counter sums can be formed.
    python tools/detect_workload.py orb 640 480 1000 [frames] [reps]      rgbdfe_detect_describe_batch
    python tools/detect_workload.py sift 640 480 0 [frames] [reps]        rgbdfe_sift_detect
    python tools/detect_workload.py sift_batch 640 480 0 [frames] [reps]  rgbdfe_sift_detect_batch"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

kind, w, h, n_kp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 8
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
n_base = min(n_frames, 28 if w <= 640 else 14)      # bench.py's detect sub-record: the generated frames forth and back
seq = synth.make_image_sequence(n_frames=n_base, seed=1, width=w, height=h)
idx = synth.forth_and_back(n_frames, n_base)
seq["gray"], seq["depth"] = [seq["gray"][i] for i in idx], [seq["depth"][i] for i in idx]
masks = [np.where(seq["mask"][i] > 0, 255, 0).astype(np.uint8) for i in idx]
fe = FrontEnd(max_nodes=4, max_keypoints=max(64, ((n_kp + 63) // 64) * 64), max_pairs_per_batch=8)
total = 0
if kind == "orb":
    fe.detector_configure(max_keypoints=n_kp)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    for _ in range(reps):
        fe.detect_describe_batch(list(seq["gray"]), masks, list(seq["depth"]), *K)
        total += n_frames
elif kind == "sift_batch":
    for _ in range(reps):
        fe.sift_detect_batch(list(seq["gray"]))
        total += n_frames
else:
    for _ in range(reps):
        for f in range(n_frames):
            fe.sift_detect(seq["gray"][f], None)
            total += 1
fe.close()
print(json.dumps({"kind": kind, "width": w, "height": h, "n_kp": n_kp, "frames": total}))
