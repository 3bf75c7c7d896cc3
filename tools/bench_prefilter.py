"""Loop-closure search WITH and WITHOUT the place-recognition prefilter (BASELINE configs[4]
shape): every frame of a trajectory looks for edges to ALL earlier frames.
  all-pairs : match + RANSAC every (frame, earlier frame) pair, one batch;
  prefilter : per frame rgbdfe_place_recognition over all earlier frames, then match + RANSAC only its top-K.
Reports frames/s for the whole sweep, pairs handed to RANSAC, and the recall of the all-pairs edges."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd

F = int(sys.argv[1]) if len(sys.argv) > 1 else 90
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
W, H = (1280, 960) if N > 2000 else (640, 480)
places = [synth.make_sequence(n_frames=10, n_kp=N, n_world=4 * N, seed=1000 + p, width=W, height=H, depth_noise=0.01)
          for p in range((F + 9) // 10)]
desc = [pl["desc"][i] for pl in places for i in range(10)][:F]
xyz = [pl["xyz1"][i] for pl in places for i in range(10)][:F]
pq = np.array([q for q in range(F) for t in range(q)], np.int32)
pt = np.array([t for q in range(F) for t in range(q)], np.int32)
fe = FrontEnd(max_nodes=F, max_keypoints=((N + 63) // 64) * 64, max_pairs_per_batch=max(len(pq), 64))
for f in range(F):
    fe.upload_node(f, desc[f], xyz[f])
fe.match_pair_list(pq[:64], pt[:64])
t0 = time.perf_counter()
full = fe.match_pair_list(pq, pt)
t_full = time.perf_counter() - t0
edges_full = {(int(a), int(b)) for a, b, e in zip(pq, pt, full["id1"]) if e >= 0}

fe.place_recognition_batch([F - 1], [np.arange(F - 1)], k_neighbours=2, max_out=K)
t0 = time.perf_counter()
sel_q, sel_t = [], []
ta = time.perf_counter()
ranked = fe.place_recognition_batch(np.arange(1, F), [np.arange(q) for q in range(1, F)], k_neighbours=2, max_hd=128, max_out=K)
t_place = time.perf_counter() - ta
for q, (ids, sc) in zip(range(1, F), ranked):
    sel_q += [q] * len(ids)
    sel_t += list(ids)
sel_q, sel_t = np.array(sel_q, np.int32), np.array(sel_t, np.int32)
pre = fe.match_pair_list(sel_q, sel_t)
t_pre = time.perf_counter() - t0
edges_pre = {(int(a), int(b)) for a, b, e in zip(sel_q, sel_t, pre["id1"]) if e >= 0}
print(json.dumps({
    "frames": F, "keypoints": N, "top_k": K,
    "all_pairs": {"pairs": int(len(pq)), "seconds": round(t_full, 4), "frames_per_s": round(F / t_full, 1),
                  "edges": len(edges_full)},
    "prefilter": {"pairs": int(len(sel_q)), "seconds": round(t_pre, 4), "frames_per_s": round(F / t_pre, 1),
                  "place_recognition_seconds": round(t_place, 4), "edges": len(edges_pre),
                  "recall_of_all_pairs_edges": round(len(edges_pre & edges_full) / max(len(edges_full), 1), 4),
                  "edges_not_in_all_pairs": len(edges_pre - edges_full)},
    "speedup": round(t_full / t_pre, 2)}))
fe.close()
