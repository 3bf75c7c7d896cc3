"""Diagnostics: one-wave vs record/replay schedules at a large ransac_iterations value, repeated runs, against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
from oracle import pyoracle as po
F, N = 60, 600
seq = synth.make_sequence(n_frames=F, n_kp=N)
rng = np.random.default_rng(1)
n, I = 3000, 1500
pq = rng.integers(1, F, n).astype(np.int32)
pt = (pq - rng.integers(1, 12, n)).clip(0).astype(np.int32)
runs = {}
for name, mode in (("one_wave", (0, 0)), ("default", None)):
    for rep in range(3):
        fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=n, ransac_iterations=I)
        if mode is not None:
            fe.set_latency_mode(*mode)
        for f in range(F):
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        runs[(name, rep)] = fe.match_pair_list(pq, pt)
        fe.close()
prm = po.default_params(seed=20260923, depth_cov=1e-4)
prm.ransac_iterations = I
bad_all = set()
for name in ("one_wave", "default"):
    for rep in (1, 2):
        bad = [k for k in range(n) if runs[(name, 0)][k].tobytes() != runs[(name, rep)][k].tobytes()]
        print(name, "run 0 vs run", rep, "differing pairs:", len(bad), bad[:6])
        bad_all |= set(bad)
bad = [k for k in range(n) if runs[("one_wave", 0)][k].tobytes() != runs[("default", 0)][k].tobytes()]
print("one_wave vs default:", len(bad), bad[:8])
bad_all |= set(bad)
for k in sorted(bad_all)[:8]:
    q, t = int(pq[k]), int(pt[k])
    r = po.match_node_pair(seq["desc"][q], seq["xyz1"][q], q, seq["desc"][t], seq["xyz1"][t], t, prm)
    print("pair", k, "oracle valid", r["valid_iterations"], "real", r["real_iterations"],
          "| one_wave", [int(runs[("one_wave", i)][k]["valid_iterations"]) for i in range(3)],
          "| default", [int(runs[("default", i)][k]["valid_iterations"]) for i in range(3)])
