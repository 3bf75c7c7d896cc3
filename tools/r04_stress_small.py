"""Stress aid (GPU box): many small batches through the split RANSAC path vs a reference run (RGBDFE_RANSAC_SPLIT=0).
python tools/r04_stress_small.py ref | run"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F = 12
seq = synth.make_sequence(n_frames=F, n_kp=600, n_world=3000, seed=5)
pq, pt = synth.candidate_pairs(F, 6)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=32)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
if sys.argv[1] == "ref":
    np.save("gpurun_out/stress_ref.npy", fe.match_pair_list(pq, pt))
    print("ref: pairs", len(pq))
else:
    ref = np.load("gpurun_out/stress_ref.npy")
    bad_runs = 0
    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        out = fe.match_pair_list(pq, pt)
        bad = [i for i in range(len(ref)) if out[i].tobytes() != ref[i].tobytes()]
        if bad:
            bad_runs += 1
            for i in bad[:3]:
                print("rep %d pair %d (chunk pos %d): n_inl %d/%d valid %d/%d real %d/%d rmse %.6g/%.6g id %d/%d" % (
                    rep, i, i % 32, out[i]["n_inl"], ref[i]["n_inl"], out[i]["valid_iterations"], ref[i]["valid_iterations"],
                    out[i]["real_iterations"], ref[i]["real_iterations"], out[i]["rmse"], ref[i]["rmse"], out[i]["id1"], ref[i]["id1"]))
    print("runs with mismatches:", bad_runs)
