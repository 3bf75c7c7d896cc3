"""GPU box, diagnostics (librgbdfe_stats.so = make -C rgbdslam_v2_amd/csrc stats): what the refinement kernel's workgroups do per
batch of the bench workload -- half-rounds, scorings, SVD requests, units -- next to the kernel's duration (HIP events).
    RGBDFE_LIB=.../librgbdfe_stats.so python tools/refine_stats.py [depth_noise=0.01]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbdslam_v2_amd import _lib, synth  # noqa: E402
from rgbdslam_v2_amd._lib import KERNEL_RANSAC, RESULT_DTYPE  # noqa: E402
from rgbdslam_v2_amd.frontend import FrontEnd  # noqa: E402

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
L = C.CDLL(_lib.LIB_PATH)
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=noise)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
buf = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
for i in range(2):
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
fe.synchronize()
st = (C.c_ulonglong * 40)()
L.rgbdfe_debug_split_stats(st, 1)
fe.reset_kernel_time()
fe.set_profiling(True)
B = 5
for i in range(B):
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
fe.synchronize()
fe.set_profiling(False)
ms, nl, _ = fe.kernel_time(KERNEL_RANSAC)
L.rgbdfe_debug_split_stats(st, 0)
v = [int(x) for x in st]
wgs = max(v[1], 1)
print(json.dumps({"depth_noise": noise, "batches": B, "stage_ms_per_batch": round(ms / max(nl, 1), 4),
                  "workgroup_launches_with_work_per_batch": v[1] / B, "half_rounds_per_workgroup": round(v[0] / wgs, 1),
                  "longest_workgroup_half_rounds": v[7], "ticket_half_round_fraction": round(v[2] / max(v[0], 1), 3),
                  "scorings_per_batch": v[3] / B, "scorings_per_half_round": round(v[3] / max(v[0], 1), 2),
                  "svd_requests_per_batch": v[4] / B, "iterations_ended_per_batch": v[20] / B,
                  "ended_after_first_scoring_per_batch": v[21] / B, "ended_without_refined_set_per_batch": v[22] / B,
                  "recurrence_passes_per_batch": v[23] / B, "recurrence_steps_per_pass": round(v[24] / max(v[23], 1), 1),
                  "refits_per_recurrence_pass": round(v[25] / max(v[23], 1), 2), "scorings_with_pass2_per_batch": v[26] / B,
                  "pass2_rounds_per_batch": v[27] / B, "error_sum_additions_per_batch": v[28] / B,
                  "pass1_candidates_per_scoring": round(v[29] / max(v[3], 1), 1), "units_loaded_per_batch": v[5] / B, "items_per_batch": v[6] / B,
                  "server_us_per_half_round": dict(zip(("svd", "recycle", "complete_loads", "hand_out", "list_active", "issue_loads",
                                                        "ticket_scoring", "barrier_wait"), [round(x / 100.0 / max(v[0], 1), 2) for x in v[8:16]])),
                  "worker0_us_per_half_round": dict(zip(("scoring", "bookkeeping_refit", "barrier_wait"),
                                                        [round(x / 100.0 / max(v[0], 1), 2) for x in v[16:19]])),
                  "workers_us_per_half_round": dict(zip(("scoring_longest", "scoring_mean", "bookkeeping_refit_longest", "bookkeeping_refit_mean",
                                                         "busy_longest", "busy_mean"), [round(x / 100.0 / max(v[0], 1), 2) for x in v[30:36]]))}))
