"""Measurement of the environment measurement model (observationLikelihood, misc.cpp:814-969) and of the
cloud builder (createXYZRGBPointCloud, misc.cpp:467-556) -- SURVEY.md 8(f) rows 2 and 3.

Workload: the BASELINE configs[1] shape -- 200 frames of 640x480 depth, cloud_creation_skip_step 2
(240x320 clouds resident in HBM), 20 candidate edges per frame, both directions per edge
(pairwiseObservationLikelihood) = 8000 jobs per step, emm__skip_step 8 (1200 sampled points per job).
Prints one JSON line: edges/s with resident clouds, the kernel's HIP-event time and its algorithmic bytes,
the cloud builder's rate, and the CPU oracle timed beside it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import KERNEL_EMM
from rgbdslam_v2_amd.frontend import FrontEnd

F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SKIP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
STEPS = 10
base = synth.make_depth_sequence(n_frames=8)
K = (base["fx"], base["fy"], base["cx"], base["cy"])
fe = FrontEnd(max_nodes=4, max_keypoints=64, max_pairs_per_batch=8)
rng = np.random.default_rng(1)
# 200 frames: the 8 rendered frames, re-used with fresh sensor noise (the kernel does not care)
t0 = time.perf_counter()
for f in range(F):
    d = base["depth"][f % 8]
    fe.upload_node_cloud(f, d, *K, cloud_skip=2)
t_cloud = time.perf_counter() - t0
pq, pt = synth.candidate_pairs(F, 20)
T = np.stack([synth.relative_pose(base["poses"], q % 8, t % 8) for q, t in zip(pq, pt)]).astype(np.float32)
Tinv = np.stack([np.linalg.inv(t.astype(np.float64)).astype(np.float32) for t in T])
new_ids = np.concatenate([pq, pt]).astype(np.int32)
old_ids = np.concatenate([pt, pq]).astype(np.int32)
TT = np.concatenate([T, Tinv])
fe.observation_likelihood(new_ids, old_ids, TT, SKIP)
fe.set_profiling(True)
fe.reset_kernel_time()
t0 = time.perf_counter()
for _ in range(STEPS):
    c = fe.observation_likelihood(new_ids, old_ids, TT, SKIP)
dt = time.perf_counter() - t0
ms, launches, jobs = fe.kernel_time(KERNEL_EMM)
fe.set_profiling(False)
ch, cw = 240, 320
nsx, nsy = -(-cw // SKIP), -(-ch // SKIP)
pts = nsx * nsy
# algorithmic bytes per job: every sampled point of the new cloud (16 B) + up to 9 depth values of the old cloud
# (4 B each) for the points that project into the raster + 16 B of counts
proj = int(c[:, :3].sum())  # classified points had a neighbourhood walk (upper bound: 9 reads each)
alg_bytes = len(new_ids) * (pts * 16 + 16) + proj * 9 * 4
k_ms = ms / max(launches, 1)
out = {
    "metric": "edges checked by the environment measurement model / s (both directions), 240x320 clouds",
    "value": round(len(pq) * STEPS / dt, 1), "unit": "edges/s", "jobs_per_step": int(len(new_ids)),
    "emm_skip_step": SKIP, "sampled_points_per_job": pts, "ms_per_step_host": round(dt / STEPS * 1e3, 3),
    "kernel_ms_per_launch": round(k_ms, 4),
    "roofline": {"bound": "hbm", "kernel": "emm_kernel", "achieved": round(alg_bytes / (k_ms * 1e-3) / 1e9, 2),
                 "peak": 8000.0, "unit": "GB/s", "frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / 8000.0, 5),
                 "algorithmic_bytes_per_launch": alg_bytes},
    "mean_counts": [round(float(x), 1) for x in c.mean(0)],
    "cloud_builder_frames_per_s_host_in": round(F / t_cloud, 1),
}
try:
    from oracle import pyoracle as po
    clouds = [po.create_point_cloud(base["depth"][f], *K, cloud_skip=2) for f in range(8)]
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        po.observation_likelihood(clouds[new_ids[i] % 8], clouds[old_ids[i] % 8], TT[i], *K, cloud_skip=2,
                                  skip_step=SKIP, depth_cov=fe.params.depth_cov)
    out["cpu_baseline"] = {"value": round(n / 2 / (time.perf_counter() - t0), 1), "unit": "edges/s", "cores": 1,
                           "kind": "port", "sample": "%d jobs, oracle/liboracle.so, one thread" % n}
except Exception as e:
    out["cpu_baseline_error"] = str(e)
print(json.dumps(out))
