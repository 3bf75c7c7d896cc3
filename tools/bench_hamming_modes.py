"""The Hamming kernels of the library side by side (rgbdfe_set_hamming_mode): serial stage time per batch
(HIP events, one batch in flight) at 1000 and 4000 keypoints, and a byte comparison of the batch results."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import KERNEL_HAMMING, KERNEL_RANSAC, RESULT_DTYPE
from rgbdslam_v2_amd.frontend import FrontEnd

import torch

MODES = [int(m) for m in os.environ.get("RGBDFE_BENCH_HAMMING_MODES", "0,1,2,3").split(",")]   # the first one is the byte reference
out = {}
for n_kp, frames, per in ((1000, 200, 20), (4000, 100, 10)):
    seq = synth.make_sequence(n_frames=frames, n_kp=n_kp, n_world=4 * n_kp, seed=20260923, depth_noise=0.01)
    pq, pt = synth.candidate_pairs(frames, per_frame=per, seed=20260923)
    ref = None
    for mode in MODES:
        fe = FrontEnd(device_id=0, max_nodes=frames, max_keypoints=((n_kp + 63) // 64) * 64, max_pairs_per_batch=len(pq))
        fe.set_hamming_mode(mode)
        for f in range(frames):
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        buf = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
        fe.set_profiling(True)
        fe.reset_kernel_time()
        for _ in range(5):
            fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
        fe.synchronize()
        fe.set_profiling(False)
        ham, nl, _ = fe.kernel_time(KERNEL_HAMMING)
        rsc, _, _ = fe.kernel_time(KERNEL_RANSAC)
        got = buf.cpu().numpy().tobytes()
        if ref is None:
            ref = got
        out["kp%d_mode%d" % (n_kp, mode)] = {"pairs": len(pq), "hamming_ms": round(ham / nl, 4), "ransac_ms": round(rsc / nl, 4),
                                             "pairs_per_s_hamming_only": round(len(pq) / (ham / nl) * 1e3),
                                             "bytes_equal_mode0": got == ref}
        # live-SLAM shape: one node against 20 candidates (split train rows)
        small = fe.match_node_pairs(frames - 1, np.arange(frames - 21, frames - 1))
        out["kp%d_mode%d" % (n_kp, mode)]["live20_first_n_all"] = int(small["n_all"][0])
        fe.close()
print(json.dumps(out, indent=1))
