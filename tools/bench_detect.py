"""Level-B measurement: per-frame detect + describe (rgbdfe_detect_describe) on
synthetic 640x480 frames, ORB 1000 keypoints; reports frames/s (host buffers in, host buffers out:
PCIe and the adjuster's host round trips included) next to the CPU oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seq = synth.make_image_sequence(n_frames=n_frames, seed=1)
masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
fe = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
fe.detector_configure(max_keypoints=1000)
for f in range(3):
    fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], seq["fx"], seq["fy"], seq["cx"], seq["cy"])
t0 = time.perf_counter()
tot = 0
for rep in range(3):
    for f in range(n_frames):
        kp, d, x = fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], seq["fx"], seq["fy"], seq["cx"], seq["cy"])
        tot += len(kp)
dt = time.perf_counter() - t0
out = {"metric": "frames detected+described/sec, 640x480 ORB-1000 (Level B)", "value": round(3 * n_frames / dt, 2),
       "unit": "frames/s", "ms_per_frame": round(dt / (3 * n_frames) * 1e3, 3), "mean_keypoints": tot / (3 * n_frames)}
K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
grays, depths = list(seq["gray"]), list(seq["depth"])
fe.detect_describe_batch(grays, masks, depths, *K)
t0 = time.perf_counter()
for rep in range(3):
    fe.detect_describe_batch(grays, masks, depths, *K)
dtb = time.perf_counter() - t0
out["batch_api_ms_per_frame"] = round(dtb / (3 * n_frames) * 1e3, 3)
# the same single calls with page-locked image buffers (rgbdfe_host_register): no staging copy
pg = [np.ascontiguousarray(g).copy() for g in grays]
pm = [m.copy() for m in masks]
for a in pg + pm:
    fe.host_register(a)
for f in range(3):
    fe.detect_describe(pg[f], pm[f], seq["depth"][f], *K)
t0 = time.perf_counter()
for rep in range(3):
    for f in range(n_frames):
        fe.detect_describe(pg[f], pm[f], seq["depth"][f], *K)
out["page_locked_images_ms_per_frame"] = round((time.perf_counter() - t0) / (3 * n_frames) * 1e3, 3)
for a in pg + pm:
    fe.host_unregister(a)
out["batch_api_frames_per_s"] = round(3 * n_frames / dtb, 2)
try:
    from oracle import pyorb
    st = pyorb.grid_state(1000)
    t0 = time.perf_counter()
    for f in range(min(n_frames, 10)):
        pyorb.node_features(st, seq["gray"][f], masks[f], seq["depth"][f], 1000)
    out["cpu_oracle_frames_per_s_1thread"] = round(min(n_frames, 10) / (time.perf_counter() - t0), 2)
except Exception as e:  # oracle is optional here
    out["cpu_oracle_error"] = str(e)
print(json.dumps(out))
