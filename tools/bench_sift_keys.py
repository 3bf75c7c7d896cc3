"""A/B of sift_match.hip's two key formats on configs[3]-shaped pairs (1000 x 1000 SIFT descriptors):
RGBDFE_SIFT_FAST_KEYS=0/1 and RGBDFE_SIFT_ROWS64=0/1 in separate processes (equal crc = equal bytes).  Usage: python tools/bench_sift_keys.py [n_pairs]"""
import os, subprocess, sys, json

CHILD = r'''
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
n_pairs = int(sys.argv[1])
F = 40
seq = synth.make_sequence(n_frames=F, n_kp=1000, seed=9)
sd = synth.sift_descriptors_like(seq["desc"], seed=9)
fe = FrontEnd(device_id=0, max_nodes=F, max_keypoints=1024, max_pairs_per_batch=n_pairs)
for f in range(F):
    fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
pq, pt = synth.candidate_pairs(F, per_frame=n_pairs // F + 1, seed=9)
pq, pt = pq[:n_pairs], pt[:n_pairs]
out, dist = fe.match_sift_pair_list(pq, pt)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); fe.match_sift_pair_list(pq, pt); ts.append(time.perf_counter() - t0)
import zlib
print(json.dumps({"fast": os.environ.get("RGBDFE_SIFT_FAST_KEYS", "1"), "rows64": os.environ.get("RGBDFE_SIFT_ROWS64", "1"), "ms": min(ts) * 1e3,
                  "pairs_per_s": len(pq) / min(ts), "crc": zlib.crc32(out.tobytes()) ^ zlib.crc32(np.asarray(dist).tobytes())}))
'''
n = sys.argv[1] if len(sys.argv) > 1 else "2000"
for fast, rows64 in (("0", "0"), ("1", "0"), ("1", "1")):
    env = dict(os.environ, RGBDFE_SIFT_FAST_KEYS=fast, RGBDFE_SIFT_ROWS64=rows64)
    r = subprocess.run([sys.executable, "-c", CHILD, n], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-2000:])
