"""Diagnostics: wall-cycle split of the select+RANSAC kernel per phase (needs librgbdfe_prof.so,
`make -C rgbdslam_v2_amd/csrc prof`).  RGBDFE_LIB=rgbdslam_v2_amd/librgbdfe_prof.so python tools_phase_profile.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
fe.set_latency_mode(0)  # the phase counters live in the one-wave-per-pair schedule
out = fe.match_pair_list(pq, pt)
dbg = out["all_q"][:, :64].copy().view(np.int64)  # signed: a reordered timer read shows up as a small negative delta
names = ["select", "load_pts", "hyp_gen", "score", "refit", "other", "n_score", "n_refit"]
tot = (dbg[:, :6].sum(axis=1) + dbg[:, 8] + dbg[:, 11] + dbg[:, 12]).astype(np.float64)
print("pairs", len(out), "mean wall cycles/pair %.3g" % tot.mean(), "max %.3g" % tot.max())
for i, n in enumerate(names):
    if i < 6:
        print("%-9s %6.2f%%  mean cycles %.3g" % (n, 100 * dbg[:, i].sum() / tot.sum(), dbg[:, i].mean()))
    else:
        print("%-9s mean %.1f" % (n, dbg[:, i].mean()))
print("cycles per score %.0f, per refit %.0f" % (dbg[:, 3].sum() / dbg[:, 6].sum(), dbg[:, 4].sum() / dbg[:, 7].sum()))

for i, n in ((11, "score:sequential error sum"), (8, "fit:compact"), (4, "fit:recurrence (all slots)"), (12, "fit:gather + batched SVD")):
    print("%-28s %6.2f%%  mean cycles %.3g" % (n, 100 * dbg[:, i].sum() / tot.sum(), dbg[:, i].mean()))
print("batched refit rounds/pair %.1f, mean longest list %.1f" % (dbg[:, 9].mean(), dbg[:, 10].sum() / max(1, dbg[:, 9].sum())))
print("per scoring: candidates %.1f, inliers (when pass 2 ran) %.1f, hopeless early-outs %.1f%%" % (
    dbg[:, 13].sum() / dbg[:, 6].sum(), dbg[:, 14].sum() / max(1, (dbg[:, 6].sum() - dbg[:, 15].sum())), 100.0 * dbg[:, 15].sum() / dbg[:, 6].sum()))
