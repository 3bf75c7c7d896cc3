"""Diagnostics: where the waves of ransac_refine_kernel spend their wall cycles (prof build: `make -C rgbdslam_v2_amd/csrc prof`).
RGBDFE_LIB=$PWD/rgbdslam_v2_amd/librgbdfe_prof.so python tools/phase_profile_split.py [noise]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth, _lib
from rgbdslam_v2_amd.frontend import FrontEnd

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=noise)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
L = C.CDLL(_lib.LIB_PATH)
tot = (C.c_ulonglong * 32)()
fe.match_pair_list(pq, pt)
L.rgbdfe_debug_split_totals(tot, 1)
reps = 3
for _ in range(reps):
    fe.match_pair_list(pq, pt)
L.rgbdfe_debug_split_totals(tot, 0)
t = np.array(list(tot), np.float64)
names = ["prologue: barrier", "close (record writes)", "item fetch (vmask scan)", "open (hypothesis load)", "scoring + error sums",
         "bookkeeping", "compaction", "recurrence + mailbox", "SVD: waiting / polling", "SVD: serving", "exit"]
wall = t[:11].sum() + t[17:21].sum()
pairs = len(pq) * reps
waves = t[24]
print("noise %.3f: %.0f working waves per batch, %.3g wall cycles per wave (100 MHz counter => %.1f us)" % (noise, waves / reps, wall / waves, wall / waves / 100.0))
for i, nm in enumerate(names):
    print("  %-32s %5.1f %%   %9.0f cycles per wave" % (nm, 100 * t[i] / wall, t[i] / waves))
for i, nm in ((17, "prologue: unit parameters (loads)"), (18, "prologue: M loads issued, slot init"), (19, "prologue: first refill"), (20, "prologue: vmcnt(0)")):
    print("  %-32s %5.1f %%   %9.0f cycles per wave" % (nm, 100 * t[i] / wall, t[i] / waves))
print("  waves with work: %.0f of %.0f per batch" % (t[25] / reps, waves / reps))
print("  per wave: %.2f rounds, %.2f items, %.2f scorings, %.2f services of %.2f requests each" % (
    t[13] / waves, t[12] / waves, t[14] / waves, t[15] / waves, t[11] / max(t[15], 1)))
