import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
for dn in (0.01, 0.005, 0.002):
    seq = synth.make_sequence(n_frames=200, n_kp=1000, seed=20260923, depth_noise=dn)
    pq, pt = synth.candidate_pairs(200, per_frame=20, seed=20260923)
    fe = FrontEnd(max_nodes=200, max_keypoints=1024, max_pairs_per_batch=4000, seed=20260923)
    for f in range(200):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    r = fe.match_pair_list(pq, pt)
    print(dn, "valid mean %.1f median %.0f max %d | real %.1f | n_inl mean %.1f | n_all %.1f" % (
        r["valid_iterations"].mean(), np.median(r["valid_iterations"]), r["valid_iterations"].max(), r["real_iterations"].mean(),
        r["n_inl"].mean(), r["n_all"].mean()))
    fe.close()
