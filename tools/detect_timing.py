"""Per-phase host wall clock of the detect / describe path (RGBDFE_DETECT_TIMING=1 prints the table at context teardown):
    RGBDFE_DETECT_TIMING=1 python tools/detect_timing.py single|batch"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
mode = sys.argv[1]
seq = synth.make_image_sequence(n_frames=20, seed=1)
masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
fe = FrontEnd(max_nodes=4, max_keypoints=1024, max_pairs_per_batch=8)
fe.detector_configure(max_keypoints=1000)
K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
if mode == "single":
    for rep in range(4):
        for f in range(20):
            fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], *K)
else:
    for rep in range(4):
        fe.detect_describe_batch(list(seq["gray"]), masks, list(seq["depth"]), *K)
fe.close()
