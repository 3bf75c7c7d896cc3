"""Diagnostics: where the recording waves of the record / replay schedule spend their wall cycles (needs the prof build:
`make -C rgbdslam_v2_amd/csrc prof`).  RGBDFE_LIB=$PWD/rgbdslam_v2_amd/librgbdfe_prof.so python tools/phase_profile_record.py [noise]
Wall cycles are per wave (3 waves share a SIMD), summed over all recording waves of 3 bench-shaped batches."""
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
L = _lib.load()
tot = (C.c_ulonglong * 24)()
fe.match_pair_list(pq, pt)
L.rgbdfe_debug_phase_totals(tot, 1)
reps = 3
for _ in range(reps):
    out = fe.match_pair_list(pq, pt)
L.rgbdfe_debug_phase_totals(tot, 0)
t = np.array(list(tot), np.float64)
names = {1: "prologue (PairPrep -> LDS)", 2: "hypotheses + pre-screen", 3: "scoring passes", 11: "error sums: staging + chain", 8: "refit: compaction",
         4: "refit: recurrences", 12: "refit: gather + batched SVD", 5: "slot bookkeeping / other",
         0: "error sums: load latency", 20: "error sums: entry", 21: "error sums: store drain (vmcnt 0)"}
wall = sum(t[i] for i in names)
pairs = len(pq) * reps
print("noise %.3f: %d recording waves per batch, %.1f iterations per wave, %.3g wall cycles per pair" % (
    noise, t[16] / reps, t[17] / max(t[16], 1), wall / pairs))
for i in (1, 2, 3, 20, 21, 0, 11, 8, 4, 12, 5):
    print("  %-30s %5.1f %%   %9.0f cycles per pair" % (names[i], 100 * t[i] / wall, t[i] / pairs))
print("  per pair: %.1f refinement rounds, mean longest error list %.1f" % (t[18] / pairs, t[19] / max(t[18], 1)))
print("  per pair: %.1f scorings (%.0f %% hopeless early-outs), %.1f refits in %.1f batched rounds; mean real iterations %.1f, valid %.1f" % (
    t[6] / pairs, 100 * t[15] / max(t[6], 1), t[7] / pairs, t[9] / pairs, out["real_iterations"].mean(), out["valid_iterations"].mean()))
