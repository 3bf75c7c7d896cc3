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
    print(dn, "valid mean %.1f | real %.1f | class histogram" % (r["valid_iterations"].mean(), r["real_iterations"].mean()), np.bincount(r["pad0"], minlength=3))
    fe.close()
