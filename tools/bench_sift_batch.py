"""rgbdfe_sift_detect_batch alone: ms per frame for a run of frames, per repetition (bench.py's sift_extract batch figure).
    python tools/bench_sift_batch.py [width height frames reps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

w, h, n_frames, reps = [int(v) for v in (sys.argv[1:5] + ["640", "480", "32", "9"][len(sys.argv) - 1:])]
n_base = min(n_frames, 28 if w <= 640 else 14)
seq = synth.make_image_sequence(n_frames=n_base, seed=1, width=w, height=h)
gray = [seq["gray"][i] for i in synth.forth_and_back(n_frames, n_base)]
fe = FrontEnd(max_nodes=4, max_keypoints=64, max_pairs_per_batch=8)
fe.sift_detect_batch(gray, copy=False)
fe.sift_detect_batch(gray, copy=False)
ms, feats = [], 0
for _ in range(reps):
    t0 = time.perf_counter()
    out = fe.sift_detect_batch(gray, copy=False)
    ms.append((time.perf_counter() - t0) * 1e3 / n_frames)
    feats = sum(len(o[0]) for o in out)
fe.close()
ms.sort()
print(json.dumps({"width": w, "height": h, "frames": n_frames, "features_per_frame": round(feats / n_frames, 1),
                  "ms_per_frame_median": round(ms[len(ms) // 2], 4), "ms_per_frame_min": round(ms[0], 4),
                  "frames_per_s_median": round(1e3 / ms[len(ms) // 2], 1),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("RGBDFE_")}}))
