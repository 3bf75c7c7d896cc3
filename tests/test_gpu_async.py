"""-m gpu: batching machinery of the C ABI -- chunking of long pair lists over the two internal
streams, the staging ring, and the asynchronous submit / wait-ticket entry points -- must not change
any result."""
import numpy as np
import pytest

from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import RESULT_DTYPE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    seq = synth.make_sequence(n_frames=12, n_kp=400, n_world=1600, seed=14)
    pq, pt = synth.candidate_pairs(12, per_frame=6, seed=14)
    return seq, pq, pt


def _fe(cap, seq):
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=cap)
    for f in range(seq["desc"].shape[0]):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    return fe


def test_chunked_list_equals_single_batch(data):
    seq, pq, pt = data
    big = _fe(256, seq)
    ref = big.match_pair_list(pq, pt)
    big.close()
    small = _fe(7, seq)  # 72 pairs -> 11 chunks alternating over both lanes, ring reused
    out = small.match_pair_list(pq, pt)
    assert out.tobytes() == ref.tobytes()
    out2 = small.match_pair_list(pq[::-1].copy(), pt[::-1].copy())
    assert out2[::-1].tobytes() == ref.tobytes()
    small.close()


def test_submit_wait_tickets(data):
    import torch
    seq, pq, pt = data
    fe = _fe(32, seq)
    ref = fe.match_pair_list(pq, pt)
    rec = RESULT_DTYPE.itemsize
    chunks = [slice(i, min(i + 20, len(pq))) for i in range(0, len(pq), 20)]
    bufs = [torch.zeros(20 * rec, dtype=torch.uint8, device="cuda") for _ in chunks]
    tickets = [fe.submit_pair_list(pq[c], pt[c], b.data_ptr()) for c, b in zip(chunks, bufs)]
    assert tickets == sorted(tickets) and len(set(tickets)) == len(tickets)
    # wait out of order: once on a torch stream, once on the host
    s = torch.cuda.Stream()
    fe.wait_ticket(tickets[-1], s.cuda_stream)
    for t in tickets[:-1][::-1]:
        fe.wait_ticket(t, None)
    s.synchronize()
    for c, b in zip(chunks, bufs):
        n = c.stop - c.start
        got = np.frombuffer(b.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:n]
        assert got.tobytes() == ref[c].tobytes()
    # in-order variant on a caller stream
    b0 = torch.zeros(len(pq[:16]) * rec, dtype=torch.uint8, device="cuda")
    fe.match_pair_list_device(pq[:16], pt[:16], b0.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = np.frombuffer(b0.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
    assert got.tobytes() == ref[:16].tobytes()
    # releasing / re-uploading a node while nothing is in flight, then matching again
    fe.release_node(3)
    fe.upload_node(3, seq["desc"][3], seq["xyz1"][3])
    again = fe.match_pair_list(pq, pt)
    assert again.tobytes() == ref.tobytes()
    fe.close()


def test_errors_are_reported_not_crashed(data):
    from rgbdslam_v2_amd._lib import RgbdfeError
    seq, pq, pt = data
    fe = _fe(8, seq)
    with pytest.raises(RgbdfeError):
        fe.match_pair_list([0], [99])            # unknown node
    with pytest.raises(ValueError):
        fe.upload_node(50, np.zeros((5, 32), np.uint8), np.zeros((6, 4), np.float32))  # shape mismatch
    with pytest.raises(RgbdfeError):
        fe.set_params(max_matches=1000)          # beyond RGBDFE_MAX_MATCHES
    fe.set_params(max_matches=300)
    big = np.zeros((600, 32), np.uint8)
    with pytest.raises(RgbdfeError):
        fe.upload_node(51, big, np.zeros((600, 4), np.float32))  # more rows than max_keypoints
    out = fe.match_pair_list(pq[:3], pt[:3])    # the context still works
    assert out["n_all"].min() >= 0
    fe.close()


def test_small_batches_repeat_bytes(data):
    """Small batches take the split-train (atomicMin) Hamming path; repeated runs through fresh
    contexts must be byte-identical to the big-batch (plain store) path."""
    seq, pq, pt = data
    big = _fe(256, seq)
    ref = big.match_pair_list(pq, pt)
    big.close()
    for rep in range(6):
        fe = _fe(64, seq)
        for j in range(0, 24, 4):
            b = fe.match_pair_list(pq[j:j + 4], pt[j:j + 4])
            assert b.tobytes() == ref[j:j + 4].tobytes()
        fe.close()


def test_graph_replay_equals_plain_launches(data, monkeypatch):
    """The launch chain of an ORB batch is captured into a hipGraph once per batch shape and replayed afterwards
    (api_batches.hip enqueue_pairs): first use (capture), replays, another shape, new pair lists through the same graph, a
    parameter change (new key), the device-output entry points -- every result equals the plain stream launches of a
    context created with RGBDFE_GRAPHS=0."""
    import torch
    seq, pq, pt = data
    monkeypatch.setenv("RGBDFE_GRAPHS", "0")
    plain = _fe(128, seq)
    monkeypatch.setenv("RGBDFE_GRAPHS", "1")
    graph = _fe(128, seq)
    try:
        for lists in ((pq, pt), (pq, pt), (pq[::-1].copy(), pt[::-1].copy()), (pq[:31], pt[:31]), (pq, pt), (pq[:31], pt[:31])):
            assert graph.match_pair_list(*lists).tobytes() == plain.match_pair_list(*lists).tobytes()
        for kw in (dict(ransac_iterations=50), dict(max_matches=128, min_matches=10), dict(ransac_iterations=200)):
            plain.set_params(**kw)
            graph.set_params(**kw)
            for _ in range(2):
                assert graph.match_pair_list(pq, pt).tobytes() == plain.match_pair_list(pq, pt).tobytes()
        plain.set_params(max_matches=300, min_matches=20)
        graph.set_params(max_matches=300, min_matches=20)
        ref = plain.match_pair_list(pq, pt)
        bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(3)]
        for rep in range(3):                      # pipelined submissions: ring slots x output buffers = several keys
            tks = [graph.submit_pair_list(pq, pt, b.data_ptr()) for b in bufs]
            for tk, b in zip(tks, bufs):
                graph.wait_ticket(tk, None)
                assert b.cpu().numpy().tobytes() == ref.tobytes()
        graph.set_latency_mode(0, 0)              # the one-wave schedule: its own shape
        plain.set_latency_mode(0, 0)
        assert graph.match_pair_list(pq, pt).tobytes() == plain.match_pair_list(pq, pt).tobytes()
    finally:
        plain.close()
        graph.close()


def test_host_output_jobs_pipeline_and_equal_the_synchronous_call(data):
    """rgbdfe_submit_pair_list_host / rgbdfe_wait_host (round 5): results in HOST memory with the download of
    batch k behind batch k while batch k+1 computes.  Same bytes as rgbdfe_match_pair_list -- into pageable and into pinned
    caller memory, two jobs in flight, waited in either order; a third submit before a wait is refused; the inlier payload is
    the stream the host twin of rgbdfe_pack_inliers makes of the same records."""
    import torch
    from rgbdslam_v2_amd._lib import INLIER_HEADER_DTYPE, RGBDFE_MAX_MATCHES, inlier_stream_of
    from rgbdslam_v2_amd.frontend import RgbdfeError
    seq, pq, pt = data
    fe = _fe(64, seq)
    try:
        ref_a = fe.match_pair_list(pq[:40], pt[:40])
        ref_b = fe.match_pair_list(pq[40:], pt[40:])
        out_a = np.zeros(40, RESULT_DTYPE)                                           # pageable
        pinned = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
        out_b = pinned.numpy()[: len(ref_b) * RESULT_DTYPE.itemsize].view(RESULT_DTYPE)  # pinned: the download writes it
        for order in ((0, 1), (1, 0)):
            out_a[:] = 0
            out_b.view(np.uint8)[:] = 0
            ta = fe.submit_pair_list_host(pq[:40], pt[:40], out_a)
            tb = fe.submit_pair_list_host(pq[40:], pt[40:], out_b)
            with pytest.raises(RgbdfeError, match="wait for an earlier ticket"):
                fe.submit_pair_list_host(pq[:5], pt[:5], np.zeros(5, RESULT_DTYPE))
            sizes = [fe.wait_host((ta, tb)[k]) for k in order]
            assert sorted(sizes) == sorted([ref_a.nbytes, ref_b.nbytes])
            assert out_a.tobytes() == ref_a.tobytes() and out_b.tobytes() == ref_b.tobytes()
        with pytest.raises(RgbdfeError, match="no host job"):
            fe.wait_host(ta)
        # the inlier payload (~104 + 4 * inliers bytes per pair instead of 1744)
        buf = np.zeros(40 * (INLIER_HEADER_DTYPE.itemsize + 4 * RGBDFE_MAX_MATCHES), np.uint8)
        nb = fe.wait_host(fe.submit_pair_list_host(pq[:40], pt[:40], buf, inliers=True))
        hdr, lst = inlier_stream_of(ref_a, 40)
        assert nb == hdr.nbytes + lst.nbytes and nb < ref_a.nbytes // 3
        assert buf[: hdr.nbytes].tobytes() == hdr.tobytes() and buf[hdr.nbytes: nb].tobytes() == lst.tobytes()
        small = np.zeros(40 * INLIER_HEADER_DTYPE.itemsize + 8, np.uint8)            # headers fit, the list does not
        t = fe.submit_pair_list_host(pq[:40], pt[:40], small, inliers=True)
        with pytest.raises(RgbdfeError, match="does not fit"):
            fe.wait_host(t)
        # ... and the job stays: the same ticket into a buffer that is large enough, then the ticket is gone
        buf[:] = 0
        assert fe.wait_host_into(t, buf) == nb
        assert buf[: hdr.nbytes].tobytes() == hdr.tobytes() and buf[hdr.nbytes: nb].tobytes() == lst.tobytes()
        with pytest.raises(RgbdfeError, match="no host job"):
            fe.wait_host(t)
        t = fe.submit_pair_list_host(pq[:40], pt[:40], small, inliers=True)          # ... or the job is dropped
        with pytest.raises(RgbdfeError, match="does not fit"):
            fe.wait_host(t)
        assert fe.wait_host_into(t, None) == 0
        assert fe.match_pair_list(pq[:40], pt[:40]).tobytes() == ref_a.tobytes()     # the context is still usable
    finally:
        fe.close()
