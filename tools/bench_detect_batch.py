"""rgbdfe_detect_describe_batch alone: ms per frame for a run of frames (a recorded bag file), per repetition.
    python tools/bench_detect_batch.py [width height n_kp frames reps]
Environment switches of the library are read once per process (RGBDFE_SUPER_DEPTH, RGBDFE_SUPER_PARALLEL_REPLAY,
RGBDFE_DETECT_SUPER, RGBDFE_DETECT_TIMING): run one process per setting."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

w, h, n_kp, n_frames, reps = [int(v) for v in (sys.argv[1:6] + ["640", "480", "1000", "56", "5"][len(sys.argv) - 1:])]
base = synth.make_image_sequence(n_frames=min(n_frames, 28), seed=1, width=w, height=h)
idx = synth.forth_and_back(n_frames, len(base["gray"]))   # a continuous camera path of any length
gray = [base["gray"][i] for i in idx]
depth = [base["depth"][i] for i in idx]
masks = [np.where(base["mask"][i] > 0, 255, 0).astype(np.uint8) for i in idx]
K = (base["fx"], base["fy"], base["cx"], base["cy"])
fe = FrontEnd(max_nodes=4, max_keypoints=max(64, ((n_kp + 63) // 64) * 64), max_pairs_per_batch=8)
fe.detector_configure(max_keypoints=n_kp)
fe.detect_describe_batch(gray[:14], masks[:14], depth[:14], *K)     # warm-up: allocations, thresholds settle
ms, tot = [], 0
for _ in range(reps):
    t0 = time.perf_counter()
    out = fe.detect_describe_batch(gray, masks, depth, *K, copy=os.environ.get("BENCH_COPY", "0") == "1")   # (bench.py: reused output arrays)
    ms.append((time.perf_counter() - t0) * 1e3 / n_frames)
    tot = sum(len(o[0]) for o in out)
fe.close()
ms.sort()
print(json.dumps({"width": w, "height": h, "n_kp": n_kp, "frames": n_frames, "keypoints_per_frame": round(tot / n_frames, 1),
                  "ms_per_frame_min": round(ms[0], 4), "ms_per_frame_median": round(ms[len(ms) // 2], 4),
                  "ms_per_frame_all": [round(v, 4) for v in ms], "frames_per_s_median": round(1e3 / ms[len(ms) // 2], 1),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("RGBDFE_")}}))
