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
import torch
from rgbdslam_v2_amd._lib import RESULT_DTYPE
buf = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
def one_batch():
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
    fe.synchronize()
one_batch()
L.rgbdfe_debug_split_totals(tot, 1)
reps = 1
one_batch()
rows = np.zeros((1 << 18, 26), np.uint64)
nrows = L.rgbdfe_debug_split_rows(rows.ctypes.data_as(C.c_void_p), C.c_uint(1 << 18))
rows = rows[:nrows]
t = rows.sum(axis=0).astype(np.float64)
t[24] = nrows
t[25] = rows[:, 25].sum()
work = rows[rows[:, 25] != 0]
life = (work[:, 23] - work[:, 22]).astype(np.float64) / 100.0  # us (100 MHz)
# the launches with work: cluster by start time
st = np.sort(work[:, 22].astype(np.float64)) / 100.0
print("working waves: %d; lifetime us: mean %.1f  p50 %.1f  p90 %.1f  max %.1f" % (len(work), life.mean(), np.percentile(life, 50), np.percentile(life, 90), life.max()))
gaps = np.where(np.diff(st) > 3.0)[0]
starts = np.concatenate([[0], gaps + 1]); ends = np.concatenate([gaps + 1, [len(st)]])
order = np.argsort(work[:, 22])
for a, b in list(zip(starts, ends))[:8]:
    w = work[order[a:b]]
    if len(w) < 64: continue
    s0 = w[:, 22].min() / 100.0
    e = w[:, 23].astype(np.float64) / 100.0 - s0
    wg = (w[:, 24] // 8)
    print("  launch: %d working waves in %d workgroups, start spread %.1f us, ends: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (
        len(w), len(np.unique(wg)), (w[:, 22].max() / 100.0 - s0), np.percentile(e, 10), np.percentile(e, 50), np.percentile(e, 90), e.max()))

names = ["init + barrier", "close (record writes)", "refill: claim items / load units / wait", "refill: hypothesis load", "scoring + error sums",
         "bookkeeping", "compaction", "recurrence + mailbox", "SVD: waiting / polling", "SVD: serving", "exit"]
wall = t[:11].sum()
pairs = len(pq) * reps
waves = t[24]
print("noise %.3f: %.0f working waves per batch, %.3g wall cycles per wave (100 MHz counter => %.1f us)" % (noise, waves / reps, wall / waves, wall / waves / 100.0))
for i, nm in enumerate(names):
    print("  %-32s %5.1f %%   %9.0f cycles per wave" % (nm, 100 * t[i] / wall, t[i] / waves))
print("  units loaded: %.0f per batch" % (t[21] / reps))
print("  waves with work: %.0f of %.0f per batch" % (t[25] / reps, waves / reps))
print("  per wave: %.2f rounds, %.2f items, %.2f scorings, %.2f services of %.2f requests each" % (
    t[13] / waves, t[12] / waves, t[14] / waves, t[15] / waves, t[11] / max(t[15], 1)))
