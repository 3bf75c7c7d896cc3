"""The one-wave-per-pair RANSAC schedule (rgbdfe_set_latency_mode(ctx, 0, 0); diagnostics only) at many iterations per pair:
run-to-run differences of the result records, by field, against the record / replay schedule (the product path)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
F, N = 200, 600
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=float(sys.argv[3]) if len(sys.argv) > 3 else synth.DEPTH_NOISE_R1)
rng = np.random.default_rng(1)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24000
pq = rng.integers(1, F, n).astype(np.int32)
pt = (pq - rng.integers(1, 12, n)).clip(0).astype(np.int32)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=n)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
fe.set_params(ransac_iterations=iters)
ref = fe.match_pair_list(pq, pt)                    # record / replay
ref2 = fe.match_pair_list(pq, pt)
print("record/replay run-to-run identical:", ref.tobytes() == ref2.tobytes())
fe.set_latency_mode(0, 0)
outs = [fe.match_pair_list(pq, pt) for _ in range(runs)]
for k, o in enumerate(outs):
    diff = [name for name in o.dtype.names if not np.array_equal(o[name], ref[name])]
    bad = np.flatnonzero(np.array([o[i].tobytes() != ref[i].tobytes() for i in range(n)]))
    print("one-wave run", k, "pairs differing from record/replay:", len(bad), "fields:", diff)
    for i in bad[:5]:
        print("   pair", i, "valid", int(o["valid_iterations"][i]), int(ref["valid_iterations"][i]), "real", int(o["real_iterations"][i]),
              int(ref["real_iterations"][i]), "n_inl", int(o["n_inl"][i]), int(ref["n_inl"][i]))
fe.close()

# unrelated places: most pairs never exit early, every iteration runs (the reject path)
desc, xyz, lq, lt = synth.loop_closure_places(n_frames=60, n_kp=500, frames_per_place=6)
fe = FrontEnd(max_nodes=64, max_keypoints=512, max_pairs_per_batch=len(lq))
for f in range(60):
    fe.upload_node(f, desc[f], xyz[f])
fe.set_params(ransac_iterations=iters)
ref = fe.match_pair_list(lq, lt)
fe.set_latency_mode(0, 0)
for k in range(runs):
    o = fe.match_pair_list(lq, lt)
    print("places: one-wave run", k, "pairs", len(lq), "differing from record/replay:",
          int(sum(o[i].tobytes() != ref[i].tobytes() for i in range(len(lq)))), "edges", int((o["id1"] >= 0).sum()))
fe.close()
