"""Stress aid (GPU box): the steps of tests/test_gpu_async.py::test_submit_wait_tickets, repeated, with details of what differs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import RESULT_DTYPE
from rgbdslam_v2_amd.frontend import FrontEnd
seq = synth.make_sequence(n_frames=12, n_kp=400, n_world=1600, seed=14)
pq, pt = synth.candidate_pairs(12, per_frame=6, seed=14)
def show(tag, out, ref):
    bad = [i for i in range(len(ref)) if out[i].tobytes() != ref[i].tobytes()]
    for i in bad[:2]:
        print("   fields:", [n for n in ref.dtype.names if np.asarray(out[i][n]).tobytes() != np.asarray(ref[i][n]).tobytes()],
              "trafo", np.asarray(out[i]["trafo"])[:4], np.asarray(ref[i]["trafo"])[:4])
        print("%s pair %d (q %d t %d): n_all %d/%d n_inl %d/%d valid %d/%d real %d/%d rmse %.6g/%.6g id %d/%d" % (
            tag, i, pq[i], pt[i], out[i]["n_all"], ref[i]["n_all"], out[i]["n_inl"], ref[i]["n_inl"], out[i]["valid_iterations"], ref[i]["valid_iterations"],
            out[i]["real_iterations"], ref[i]["real_iterations"], out[i]["rmse"], ref[i]["rmse"], out[i]["id1"], ref[i]["id1"]))
    return len(bad)
gold = None
if os.path.exists("gpurun_out/async_gold.npy"): gold = np.load("gpurun_out/async_gold.npy")
nbad = 0
def fe_of(cap):
    fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=cap)
    for f in range(12):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    return fe
if len(sys.argv) > 2:  # the module's first test
    big = fe_of(256); r1 = big.match_pair_list(pq, pt); big.close()
    small = fe_of(7); r2 = small.match_pair_list(pq, pt)
    if gold is not None: nbad += show("first test big", r1, gold) + show("first test small", r2, gold)
    r3 = small.match_pair_list(pq[::-1].copy(), pt[::-1].copy()); small.close()
for rep in range(int(sys.argv[1])):
    fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=32)
    for f in range(12):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    ref = fe.match_pair_list(pq, pt)
    if gold is None:
        gold = ref.copy(); np.save("gpurun_out/async_gold.npy", gold); print("gold saved"); 
    nbad += show("rep %d ref-vs-gold" % rep, ref, gold)
    rec = RESULT_DTYPE.itemsize
    chunks = [slice(i, min(i + 20, len(pq))) for i in range(0, len(pq), 20)]
    bufs = [torch.zeros(20 * rec, dtype=torch.uint8, device="cuda") for _ in chunks]
    tickets = [fe.submit_pair_list(pq[c], pt[c], b.data_ptr()) for c, b in zip(chunks, bufs)]
    st = torch.cuda.Stream()
    fe.wait_ticket(tickets[-1], st.cuda_stream)
    for t in tickets[:-1][::-1]:
        fe.wait_ticket(t, None)
    st.synchronize()
    for c, b in zip(chunks, bufs):
        n = c.stop - c.start
        got = np.frombuffer(b.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:n]
        nbad += show("rep %d async chunk %d" % (rep, c.start), got, gold[c])
    b0 = torch.zeros(16 * rec, dtype=torch.uint8, device="cuda")
    fe.match_pair_list_device(pq[:16], pt[:16], b0.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = np.frombuffer(b0.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
    nbad += show("rep %d device variant" % rep, got, gold[:16])
    if os.environ.get("STEP"): print("rep", rep, "before again: reopened", __import__("ctypes").CDLL(__import__("rgbdslam_v2_amd._lib", fromlist=["x"]).LIB_PATH).rgbdfe_debug_reopened())
    fe.release_node(3)
    fe.upload_node(3, seq["desc"][3], seq["xyz1"][3])
    again = fe.match_pair_list(pq, pt)
    nbad += show("rep %d again" % rep, again, gold)
    fe.close()
print("mismatching records in total:", nbad)
try:
    import ctypes as C
    from rgbdslam_v2_amd import _lib
    print("reopened iterations:", C.CDLL(_lib.LIB_PATH).rgbdfe_debug_reopened())
except Exception as e:
    print("no debug counter", e)
