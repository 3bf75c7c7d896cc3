"""Bisecting aid (GPU box): one small workload through the recording stage's variants, compared pair by pair.
python tools/r04_debug_split.py run <tag>   -> writes gpurun_out/dbg_<tag>.npy ;  python tools/r04_debug_split.py cmp"""
import os, sys, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    F, N = 40, 1000
    noise = float(os.environ.get("DBG_NOISE", "0.01"))
    seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=noise)
    pq, pt = synth.candidate_pairs(F, 20)
    npairs = int(os.environ.get("DBG_PAIRS", "0")) or len(pq)
    pq, pt = pq[:npairs], pt[:npairs]
    fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
    for f in range(F):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    outs = [fe.match_pair_list(pq, pt).copy() for _ in range(3)]
    same = [outs[0].tobytes() == o.tobytes() for o in outs[1:]]
    print(sys.argv[2], "pairs", len(pq), "repeat-identical", same)
    np.save("gpurun_out/dbg_%s.npy" % sys.argv[2], outs[0])
else:
    ref = np.load("gpurun_out/dbg_ref.npy")
    for f in sorted(glob.glob("gpurun_out/dbg_*.npy")):
        o = np.load(f)
        if o.shape != ref.shape:
            print(f, "shape", o.shape, ref.shape); continue
        bad = [i for i in range(len(ref)) if o[i].tobytes() != ref[i].tobytes()]
        print(os.path.basename(f), "mismatching pairs: %d of %d" % (len(bad), len(ref)))
        for i in bad[:6]:
            print("   pair %d: n_inl %d/%d  valid_it %d/%d real_it %d/%d rmse %.6g/%.6g n_all %d" % (
                i, o[i]["n_inl"], ref[i]["n_inl"], o[i]["valid_iterations"], ref[i]["valid_iterations"],
                o[i]["real_iterations"], ref[i]["real_iterations"], o[i]["rmse"], ref[i]["rmse"], ref[i]["n_all"]))
