"""-m gpu: every number bench.py prints comes with oracle constants (tests/golden/bench_expected.json, made offline by
tools/make_bench_expected.py).  Here: configs[3] at the bench's size record by record against the oracle (12
sampled pairs of 2000 before), and the detect / sift_extract / front_end sub-records against their constants -- the
sub-record functions themselves are run, so what the driver's bench line checks is what is checked here."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def test_sift_subrecord_every_pair_matches_oracle():
    """bench.py's `sift` sub-record (configs[3]: 2000 pairs of 1000 x 1000 128-d float descriptors): match lists,
    distances, inlier sets and poses of ALL pairs equal the oracle's, and their sums are the constants the bench asserts."""
    import bench
    from rgbdslam_v2_amd.frontend import FrontEnd, inlier_indices
    seq, _, _ = bench.orb_workload(1)
    sd, pq, pt = bench.sift_workload(seq)
    assert len(pq) == 2000
    fe = FrontEnd(device_id=0, max_nodes=bench.SIFT_FRAMES, max_keypoints=1024, max_pairs_per_batch=len(pq), seed=bench.SEED)
    try:
        for f in range(bench.SIFT_FRAMES):
            fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
        out, dist = fe.match_sift_pair_list(pq, pt)
        prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
    finally:
        fe.close()

    def one(k):
        q, t = int(pq[k]), int(pt[k])
        return po.match_sift_node_pair(sd[q], seq["xyz1"][q], q, sd[t], seq["xyz1"][t], t, prm)
    with ThreadPoolExecutor(po.usable_cpus()) as ex:
        refs = list(ex.map(one, range(len(pq))))
    for k, (rec, ref) in enumerate(zip(out, refs)):
        n = ref["n_all"]
        assert rec["n_all"] == n, k
        assert np.array_equal(rec["all_q"][:n], ref["all_q"]) and np.array_equal(rec["all_t"][:n], ref["all_t"]), k
        assert np.array_equal(np.asarray(dist[k])[:n], ref["all_dist"]), k
        assert (rec["id1"], rec["id2"], rec["n_inl"]) == (ref["id1"], ref["id2"], ref["n_inl"]), k
        assert (rec["real_iterations"], rec["valid_iterations"]) == (ref["real_iterations"], ref["valid_iterations"]), k
        assert np.array_equal(inlier_indices(rec), ref["inl_idx"]), k
        assert np.array_equal(np.array(rec["trafo"], np.float32).reshape(4, 4).T, ref["T"]), k
        assert np.float32(rec["rmse"]) == ref["rmse"], k
    exp = bench.expected("sift", "0.01")
    assert exp is not None and bench.pair_aggregates(out) == exp
    assert exp["edges"] > 0.9 * len(pq)


def test_bench_subrecords_match_their_oracle_constants():
    """detect (640x480 ORB-1000 and 1280x960 ORB-4000 through the batch entry point), sift_extract and front_end: the
    sub-record functions of bench.py end with a comparison against constants made by oracle/orb_oracle.c, the compiled
    reference SiftGPU pipeline and oracle/liboracle.so -- they raise when an output differs."""
    import bench
    for key in ("detect", "sift_extract", "front_end"):
        assert bench.expected(key) is not None, key
    det = bench.detect_subrecord(0)
    for key, rec in det.items():
        assert rec["parity_check"]["checked"] and rec["parity_check"]["ok"], key
        assert rec["parity_check"]["oracle_aggregates"]["keypoints"] > 0
    sx = bench.sift_extract_subrecord(0)
    assert sx["parity_check"]["checked"] and sx["parity_check"]["ok"]
    fr = bench.front_end_subrecord(0)
    assert fr["parity_check"]["checked"] and fr["parity_check"]["ok"]
    assert fr["edges_found"] == fr["parity_check"]["oracle_aggregates"]["edges"] > 0


def test_a_wrong_result_stops_the_bench():
    """check_against is what stands between a wrong kernel and a printed number: it must raise."""
    import bench
    exp = bench.expected("orb", 0.01, 1)
    assert exp is not None
    with pytest.raises(SystemExit):
        bench.check_against(exp, dict(exp, inliers=exp["inliers"] + 1), "a doctored record set", "test")
    assert bench.check_against(exp, dict(exp), "the same sums", "test")["ok"]


@pytest.mark.parametrize("n_kp,per_frame", [(500, 8), (1000, 20)])
def test_pack_inliers_equals_its_host_twin(n_kp, per_frame):
    """rgbdfe_pack_inliers (the default payload of bench.py --gpus N): headers, list positions and the (query row, train
    row) lists of a shard's records equal the numpy restatement of the format, byte for byte; padding headers say "no
    edge"; an empty shard gives an empty list block."""
    import torch
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import INLIER_HEADER_DTYPE, RESULT_DTYPE, inlier_pairs, inlier_stream_of, parse_inlier_stream
    from rgbdslam_v2_amd.frontend import FrontEnd, inlier_indices
    F = 24
    seq = synth.make_sequence(n_frames=F, n_kp=n_kp, n_world=4 * n_kp, seed=31)
    pq, pt = synth.candidate_pairs(F, per_frame=per_frame, seed=31)
    n, n_hdr = len(pq), len(pq) + 5
    fe = FrontEnd(device_id=0, max_nodes=F, max_keypoints=1024, max_pairs_per_batch=n)
    try:
        for f in range(F):
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        d_rec = torch.zeros(n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        fe.wait_ticket(fe.submit_pair_list(pq, pt, d_rec.data_ptr()), None)
        rec = np.frombuffer(d_rec.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
        cap = n_hdr * INLIER_HEADER_DTYPE.itemsize + 4 * 320 * n
        d_stream = torch.full((cap,), 0xAB, dtype=torch.uint8, device="cuda")
        d_tot = torch.zeros(1, dtype=torch.int32, device="cuda")
        fe.pack_inliers(d_rec.data_ptr(), n, n_hdr, d_stream.data_ptr(), d_tot.data_ptr())
        fe.synchronize()
        total = int(d_tot.item())
        hdr, lst = inlier_stream_of(rec, n_hdr)
        assert total == len(lst) == int(rec["n_inl"].sum()) > 10 * n // 2
        got = d_stream.cpu().numpy()
        want = np.concatenate([hdr.view(np.uint8).reshape(-1), lst.view(np.uint8).reshape(-1)])
        assert np.array_equal(got[: len(want)], want)
        assert np.all(got[len(want):] == 0xAB)                                   # nothing written behind the stream
        h2, l2 = parse_inlier_stream(got, n_hdr, total)
        for k in (0, 1, n // 2, n - 1):
            q_rows, t_rows = inlier_pairs(h2, l2, k)
            ii = inlier_indices(rec[k])
            assert np.array_equal(q_rows, rec[k]["all_q"][ii]) and np.array_equal(t_rows, rec[k]["all_t"][ii])
        # an empty shard
        fe.pack_inliers(d_rec.data_ptr(), 0, 3, d_stream.data_ptr(), d_tot.data_ptr())
        fe.synchronize()
        e = d_stream.cpu().numpy()[: 3 * 104].view(INLIER_HEADER_DTYPE)
        assert int(d_tot.item()) == 0 and np.all(e["id1"] == -1) and np.all(e["n_inl"] == 0)
    finally:
        fe.close()
