"""Stress: many pipelined 4000-pair batches over the two internal lanes (the bench's submission pattern) must give
byte-identical results every time -- record / replay scratch (records, walk states, error pool, PairPrep) is reused
by consecutive batches of a lane while the other lane runs."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd, RESULT_DTYPE

F, N = 200, 1000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq = synth.make_sequence(n_frames=F, n_kp=N)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
n = len(pq)
rec_bytes = RESULT_DTYPE.itemsize
bufs = [torch.zeros(n * rec_bytes, dtype=torch.uint8, device="cuda") for _ in range(4)]
sizes = [n, n, 1500, n, 300, n, 64, n]
ref = {}
bad = 0
tickets = []
for rep in range(reps):
    m = sizes[rep % len(sizes)]
    b = bufs[rep % 4]
    if len(tickets) >= 4:
        tk, bb, mm = tickets.pop(0)
        fe.wait_ticket(tk, None)
        got = np.frombuffer(bb.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:mm].tobytes()
        if mm not in ref:
            ref[mm] = got
        elif got != ref[mm]:
            bad += 1
            print("MISMATCH at batch of", mm)
    tickets.append((fe.submit_pair_list(pq[:m], pt[:m], b.data_ptr()), b, m))
for tk, bb, mm in tickets:
    fe.wait_ticket(tk, None)
    got = np.frombuffer(bb.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:mm].tobytes()
    if mm in ref and got != ref[mm]:
        bad += 1
        print("MISMATCH at batch of", mm)
    ref.setdefault(mm, got)
# and against the synchronous call
sync = fe.match_pair_list(pq, pt).tobytes()
if sync != ref[n]:
    bad += 1
    print("pipelined result differs from the synchronous call")
print("pipeline stress done, batches", reps, "mismatches:", bad)
