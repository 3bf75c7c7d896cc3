"""-m gpu: several devices behind one handle (rgbdfe_create_multi), the all-gather of the result
PODs, and the ABI hardening (many threads on one context, batches beyond 65536 pairs, upload ordering).

The GPU box has ONE device, so the group lists it twice: two device contexts, two host threads, two shards --
real kernels on both shards, gathered bytes compared with the single-context result."""
import threading

import numpy as np
import pytest

from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd._lib import COMPACT_DTYPE, RESULT_DTYPE, RgbdfeError, compact_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    seq = synth.make_sequence(n_frames=14, n_kp=500, n_world=2000, seed=21)
    pq, pt = synth.candidate_pairs(14, per_frame=7, seed=21)
    return seq, pq, pt


def _fe(seq, device_ids=None, cap=256, **kw):
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=24, max_keypoints=512, max_pairs_per_batch=cap, device_ids=device_ids, **kw)
    for f in range(seq["desc"].shape[0]):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    return fe


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0]])
def test_sharded_match_equals_single_device(data, ids):
    seq, pq, pt = data
    one = _fe(seq)
    ref = one.match_pair_list(pq, pt)
    ref_np = one.match_node_pairs(13, np.arange(13))
    one.close()
    grp = _fe(seq, device_ids=ids)
    assert grp.device_count == len(ids)
    out = grp.match_pair_list(pq, pt)                 # pair k -> device k mod G, written to out[k]
    assert out.tobytes() == ref.tobytes()
    assert grp.match_node_pairs(13, np.arange(13)).tobytes() == ref_np.tobytes()   # the blockingMapped call shape
    assert grp.match_pair_list(pq[:1], pt[:1]).tobytes() == ref[:1].tobytes()       # fewer pairs than devices
    assert len(grp.match_pair_list(pq[:0], pt[:0])) == 0
    # node lifetime is replicated too
    grp.release_node(3)
    with pytest.raises(RgbdfeError):
        grp.match_pair_list([3], [1])
    grp.upload_node(3, seq["desc"][3], seq["xyz1"][3])
    assert grp.match_pair_list(pq, pt).tobytes() == ref.tobytes()
    grp.close()


@pytest.mark.parametrize("ids,transport", [([0, 0], "p2p"), ([0], None)])
def test_allgather_leaves_all_results_on_every_device(data, ids, transport):
    import torch
    seq, pq, pt = data
    one = _fe(seq)
    ref = one.match_pair_list(pq, pt)
    one.close()
    grp = _fe(seq, device_ids=ids)
    G, n = len(ids), len(pq)
    per = (n + G - 1) // G
    rec = RESULT_DTYPE.itemsize
    bufs = [torch.zeros(G * per * rec, dtype=torch.uint8, device="cuda:0") for _ in ids]
    torch.cuda.synchronize()
    got_per = grp.match_pair_list_allgather(pq, pt, [b.data_ptr() for b in bufs])
    assert got_per == per
    if transport is not None:
        assert grp.gather_transport() == transport
    else:
        assert grp.gather_transport() in ("rccl", "none (one device)")   # one rank: RCCL when librccl loads
    for b in bufs:
        allrec = np.frombuffer(b.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
        for k in range(n):
            assert allrec[(k % G) * per + k // G].tobytes() == ref[k].tobytes()
        for d in range(G):                             # unused tail records: ids -1
            used = len(range(d, n, G))
            assert np.all(allrec[d * per + used:(d + 1) * per]["id1"] == -1)
    grp.close()


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0]])
def test_allgather_of_accepted_edges_only(ids):
    """in an all-pairs sweep most pairs are rejected and need not travel.  Two unrelated places: only pairs
    inside a place become edges; every device ends up with exactly the accepted records, in shard order, plus their
    positions in the pair list."""
    import torch
    a = synth.make_sequence(n_frames=6, n_kp=400, n_world=1500, seed=31)
    b = synth.make_sequence(n_frames=6, n_kp=400, n_world=1500, seed=32)
    desc = list(a["desc"]) + list(b["desc"])
    xyz = list(a["xyz1"]) + list(b["xyz1"])
    F = 12
    pq = np.array([q for q in range(F) for t in range(q)], np.int32)
    pt = np.array([t for q in range(F) for t in range(q)], np.int32)
    from rgbdslam_v2_amd.frontend import FrontEnd

    def make(device_ids):
        fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=128, device_ids=device_ids)
        for f in range(F):
            fe.upload_node(f, desc[f], xyz[f])
        return fe

    one = make(None)
    ref = one.match_pair_list(pq, pt)
    one.close()
    is_edge = ref["id1"] >= 0
    assert 10 <= is_edge.sum() <= 40 and not is_edge[(pq // 6) != (pt // 6)].any()
    grp = make(ids)
    G, n = len(ids), len(pq)
    per = (n + G - 1) // G
    rec = RESULT_DTYPE.itemsize
    bufs = [torch.zeros(G * per * rec, dtype=torch.uint8, device="cuda:0") for _ in ids]
    idxs = [torch.full((G * per,), -7, dtype=torch.int32, device="cuda:0") for _ in ids]
    torch.cuda.synchronize()
    counts, stride = grp.match_pair_list_allgather_edges(pq, pt, [b.data_ptr() for b in bufs], [i.data_ptr() for i in idxs])
    assert counts.sum() == is_edge.sum() and stride == counts.max()
    for d in range(G):
        shard = np.arange(d, n, G)
        want = shard[is_edge[shard]]
        assert counts[d] == len(want)
        for b, ix in zip(bufs, idxs):
            got = np.frombuffer(b.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[d * stride: d * stride + counts[d]]
            assert got.tobytes() == ref[want].tobytes()
            assert np.array_equal(ix.cpu().numpy()[d * stride: d * stride + counts[d]], want)
    # no index buffers, and a sweep without any edge
    counts2, stride2 = grp.match_pair_list_allgather_edges(pq, pt, [b.data_ptr() for b in bufs])
    assert np.array_equal(counts2, counts) and stride2 == stride
    cross = (pq // 6) != (pt // 6)
    counts3, stride3 = grp.match_pair_list_allgather_edges(pq[cross], pt[cross], [b.data_ptr() for b in bufs])
    assert counts3.sum() == 0 and stride3 == 0
    grp.close()


def test_group_refuses_device_pointer_entry_points(data):
    import torch
    seq, pq, pt = data
    grp = _fe(seq, device_ids=[0, 0])
    buf = torch.zeros(4 * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    with pytest.raises(RgbdfeError, match="device"):
        grp.submit_pair_list(pq[:4], pt[:4], buf.data_ptr())
    grp.close()


def test_sift_pairs_shard_too(data):
    seq, pq, pt = data
    from rgbdslam_v2_amd.frontend import FrontEnd
    sd = synth.sift_descriptors_like(seq["desc"], seed=5)
    outs = []
    for ids in (None, [0, 0]):
        fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=128, device_ids=ids)
        for f in range(sd.shape[0]):
            fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
        outs.append(fe.match_sift_pair_list(pq, pt))
        fe.close()
    a, b = outs
    if isinstance(a, tuple):
        for x, y in zip(a, b):
            assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
    else:
        assert a.tobytes() == b.tobytes()


def test_one_context_from_many_threads(data):
    """rgbdfe.h: a context may be called from any thread; calls serialise, errors stay per thread."""
    seq, pq, pt = data
    fe = _fe(seq, cap=64)
    ref = fe.match_pair_list(pq, pt)
    errors, wrong = [], []

    def worker(tid):
        try:
            for it in range(6):
                lo = (tid * 7 + it * 3) % (len(pq) - 20)
                out = fe.match_pair_list(pq[lo:lo + 20], pt[lo:lo + 20])
                if out.tobytes() != ref[lo:lo + 20].tobytes():
                    wrong.append((tid, it))
                try:
                    fe.match_pair_list([0], [1000 + tid])     # unknown node: this thread's own error text
                    wrong.append((tid, it, "no error"))
                except RgbdfeError as e:
                    if "not resident" not in str(e):
                        wrong.append((tid, it, str(e)))
                if tid % 2 == 0:                              # uploads interleave with other threads' batches
                    fe.upload_node(100 + tid, seq["desc"][tid % 14], seq["xyz1"][tid % 14])
                    fe.release_node(100 + tid)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors and not wrong, (errors, wrong)
    assert fe.match_pair_list(pq, pt).tobytes() == ref.tobytes()
    fe.close()


def test_one_group_handle_from_many_threads(data):
    """a rgbdfe_create_multi handle is one context -- calls from several threads (match / upload / release /
    set_params / gathers) serialise on the group's own lock; no caller may lose its shard or see another caller's
    status."""
    import torch
    seq, pq, pt = data
    one = _fe(seq, cap=64)
    ref = one.match_pair_list(pq, pt)
    one.close()
    grp = _fe(seq, device_ids=[0, 0], cap=64)
    G, rec = 2, RESULT_DTYPE.itemsize
    errors, wrong = [], []

    def worker(tid):
        try:
            bufs = [torch.zeros(G * 10 * rec, dtype=torch.uint8, device="cuda:0") for _ in range(G)]
            torch.cuda.synchronize()
            for it in range(6):
                lo = (tid * 5 + it * 3) % (len(pq) - 20)
                out = grp.match_pair_list(pq[lo:lo + 19], pt[lo:lo + 19])      # odd count: uneven shards
                if out.tobytes() != ref[lo:lo + 19].tobytes():
                    wrong.append((tid, it, "match"))
                try:
                    grp.match_pair_list([0], [1000 + tid])
                    wrong.append((tid, it, "no error"))
                except RgbdfeError as e:
                    if "not resident" not in str(e):
                        wrong.append((tid, it, str(e)))
                if tid % 3 == 0:
                    grp.upload_node(100 + tid, seq["desc"][tid % 14], seq["xyz1"][tid % 14])
                    grp.release_node(100 + tid)
                if tid % 3 == 1:
                    per = grp.match_pair_list_allgather(pq[lo:lo + 20], pt[lo:lo + 20], [b.data_ptr() for b in bufs])
                    got = np.frombuffer(bufs[it % G].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
                    for k in range(20):
                        if got[(k % G) * per + k // G].tobytes() != ref[lo + k].tobytes():
                            wrong.append((tid, it, "gather", k))
                            break
                if tid % 3 == 2:
                    grp.set_params(seed=grp.params.seed)            # same values: results must not move
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors and not wrong, (errors, wrong)
    assert grp.match_pair_list(pq, pt).tobytes() == ref.tobytes()
    grp.close()


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0]])
def test_compact_allgather_carries_everything_but_the_match_lists(data, ids):
    """The default gather payload: rgbdfe_compact_result (144 B instead of 1744 B).  Every device ends up with every
    pair's header + inlier mask; the full record of any pair is recomputed locally, byte-identical (no fetch)."""
    import torch
    seq, pq, pt = data
    one = _fe(seq)
    ref = one.match_pair_list(pq, pt)
    # the single-context packer (what bench.py --gpus N runs before its own all-gather)
    d_rec = torch.from_numpy(ref.view(np.uint8).reshape(-1).copy()).cuda()
    d_cmp = torch.zeros(len(ref) * COMPACT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    one.pack_compact(d_rec.data_ptr(), len(ref), d_cmp.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_cmp.cpu().numpy().tobytes() == compact_of(ref).tobytes()
    one.close()
    assert COMPACT_DTYPE.itemsize == 144
    grp = _fe(seq, device_ids=ids)
    G, n = len(ids), len(pq)
    per = (n + G - 1) // G
    bufs = [torch.zeros(G * per * COMPACT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0") for _ in ids]
    torch.cuda.synchronize()
    assert grp.match_pair_list_allgather_compact(pq, pt, [b.data_ptr() for b in bufs]) == per
    want = compact_of(ref)
    for b in bufs:
        got = np.frombuffer(b.cpu().numpy().tobytes(), dtype=COMPACT_DTYPE)
        for k in range(n):
            assert got[(k % G) * per + k // G].tobytes() == want[k].tobytes()
        for d in range(G):
            used = len(range(d, n, G))
            assert np.all(got[d * per + used:(d + 1) * per]["id1"] == -1)
    # "lazy fetch": any device recomputes a pair's full record on its own
    k = int(np.flatnonzero(ref["id1"] >= 0)[3])
    assert grp.match_pair_list(pq[k:k + 1], pt[k:k + 1]).tobytes() == ref[k:k + 1].tobytes()
    grp.close()


def test_sift_and_flann_batches_split_into_pieces_keep_their_distance_rows():
    """a SIFT / FLANN batch that runs as several record / replay pieces (pairs x ransac_iterations > 2^24)
    with host outputs: every piece's DMatch.distance rows must land at the piece's offset, not on top of piece 0."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    seq = synth.make_sequence(n_frames=5, n_kp=96, n_world=260, seed=13)
    sd = synth.sift_descriptors_like(seq["desc"], seed=2)
    iters = 100000                       # pieces of 2^24 / 100000 = 167 pairs
    n = 400
    rng = np.random.default_rng(4)
    pq = rng.integers(1, 5, n).astype(np.int32)
    pt = (pq - 1 - rng.integers(0, 4, n) % pq).astype(np.int32)

    def run(flann, pairs_q, pairs_t, cap):
        fe = FrontEnd(device_id=0, max_nodes=8, max_keypoints=128, max_pairs_per_batch=cap, min_matches=5,
                      ransac_iterations=iters)
        for f in range(5):
            (fe.upload_float_node if flann else fe.upload_sift_node)(f, sd[f], seq["xyz1"][f])
        out = fe.match_flann_pair_list(pairs_q, pairs_t) if flann else fe.match_sift_pair_list(pairs_q, pairs_t)
        fe.close()
        return out

    for flann in (False, True):
        rec, dist = run(flann, pq, pt, n)                       # one batch of 400 pairs = three pieces
        uniq = sorted({(int(a), int(b)) for a, b in zip(pq, pt)})
        r1, d1 = run(flann, [a for a, _ in uniq], [b for _, b in uniq], 16)   # the same pairs, single-piece batches
        lut = {kq: i for i, kq in enumerate(uniq)}
        assert any(r1["n_all"] > 0)
        for k in range(n):
            i = lut[(int(pq[k]), int(pt[k]))]
            assert rec[k].tobytes() == r1[i].tobytes(), (flann, k)
            assert np.array_equal(dist[k][: rec[k]["n_all"]], d1[i][: r1[i]["n_all"]]), (flann, k)


def test_batch_beyond_65536_pairs():
    """a batch with more pairs than error-pool regions (65536) must not spin: it runs as record / replay
    pieces, with the results of the same pairs in bench-sized batches."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    seq = synth.make_sequence(n_frames=6, n_kp=64, n_world=160, seed=3)
    n = 66000
    rng = np.random.default_rng(0)
    pq = rng.integers(1, 6, n).astype(np.int32)
    pt = (pq - 1 - rng.integers(0, 5, n) % pq).astype(np.int32)
    fe = FrontEnd(device_id=0, max_nodes=8, max_keypoints=64, max_pairs_per_batch=n, min_matches=5, ransac_iterations=20)
    for f in range(6):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    # one submit = one batch (match_pair_list would cut the list in two halves for its two streams)
    import torch
    buf = torch.zeros(n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
    out = np.frombuffer(buf.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
    fe.close()
    small = FrontEnd(device_id=0, max_nodes=8, max_keypoints=64, max_pairs_per_batch=512, min_matches=5,
                     ransac_iterations=20)
    for f in range(6):
        small.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    uniq = {}
    for k in range(n):
        uniq.setdefault((int(pq[k]), int(pt[k])), k)
    keys = sorted(uniq)
    ref = small.match_pair_list([a for a, _ in keys], [b for _, b in keys])
    small.close()
    lut = {kq: ref[i] for i, kq in enumerate(keys)}
    for k in range(0, n, 97):
        assert out[k].tobytes() == lut[(int(pq[k]), int(pt[k]))].tobytes()


def test_upload_node_device_orders_before_later_batches(data):
    """rgbdfe_upload_node_device on a caller stream returns at once; the next batch still sees the node."""
    import torch
    seq, pq, pt = data
    fe = _fe(seq)
    ref = fe.match_pair_list(pq, pt)
    s = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        f = 5 + rep
        d_desc = torch.from_numpy(seq["desc"][f].copy()).cuda()
        d_xyz = torch.from_numpy(seq["xyz1"][f].copy()).cuda()
        torch.cuda.synchronize()
        fe.upload_node(f, np.zeros_like(seq["desc"][f]), np.zeros_like(seq["xyz1"][f]))   # wipe the slot
        with torch.cuda.stream(s):
            big.fill_(rep)                      # keeps the caller's stream busy ahead of the copies
            fe.upload_node_device(f, d_desc.data_ptr(), d_xyz.data_ptr(), d_desc.shape[0], s.cuda_stream)
        out = fe.match_pair_list(pq, pt)        # must wait for the copies on the device
        assert out.tobytes() == ref.tobytes()
        s.synchronize()
    fe.close()


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0], [0]])
def test_allgather_of_the_inlier_form(data, ids):
    """rgbdfe_match_pair_list_allgather_inliers: every device ends up with every device's inlier stream -- the 104-byte
    headers of its shard in shard order and the (query row, train row) of every inlier match -- equal to the host twin of
    the format applied to the single-context records; pair k of the list = header k // G of device k % G."""
    import torch
    from rgbdslam_v2_amd._lib import INLIER_HEADER_DTYPE, inlier_pairs, inlier_stream_of, parse_inlier_stream
    from rgbdslam_v2_amd.frontend import inlier_indices
    seq, pq, pt = data
    one = _fe(seq)
    ref = one.match_pair_list(pq, pt)
    one.close()
    grp = _fe(seq, device_ids=ids)
    G, n = len(ids), len(pq)
    per = (n + G - 1) // G
    cap = G * per * (INLIER_HEADER_DTYPE.itemsize + 4 * 320)
    bufs = [torch.full((cap,), 0xCD, dtype=torch.uint8, device="cuda:0") for _ in ids]
    torch.cuda.synchronize()
    # The group's first call sizes the exchange after reading the counts (stride = the longest list exactly); from then on
    # the exchange is sized BEFORE the counts are known -- the longest list seen + a quarter -- and is the call's only one.
    first = grp.match_pair_list_allgather_inliers(pq, pt, [b.data_ptr() for b in bufs])
    assert first[0] == per and first[2] == per * 104 + 4 * int(first[1].max()) and grp.gather_exchanges() == 1
    for b in bufs:
        b.fill_(0xCD)
    torch.cuda.synchronize()
    got_per, totals, stride = grp.match_pair_list_allgather_inliers(pq, pt, [b.data_ptr() for b in bufs])
    longest = int(totals.max())
    assert got_per == per and np.array_equal(totals, first[1]) and grp.gather_exchanges() == 1
    assert stride == per * 104 + 4 * min(longest + longest // 4 + 64, per * 320)
    assert int(totals.sum()) == int(ref["n_inl"].sum()) > 20 * n // 2
    for b in bufs:
        raw = b.cpu().numpy()
        for d in range(G):
            shard = ref[d::G]
            hdr_want, lst_want = inlier_stream_of(shard, per)
            hdr, lst = parse_inlier_stream(raw[d * stride:(d + 1) * stride], per, int(totals[d]))
            assert hdr.tobytes() == hdr_want.tobytes() and np.array_equal(lst, lst_want), d
        for k in (0, 1, n // 2, n - 1):                 # what GraphManager::updateInlierFeatures reads, pair k of the list
            d, j = k % G, k // G
            hdr, lst = parse_inlier_stream(raw[d * stride:(d + 1) * stride], per, int(totals[d]))
            q_rows, t_rows = inlier_pairs(hdr, lst, j)
            ii = inlier_indices(ref[k])
            assert np.array_equal(q_rows, ref[k]["all_q"][ii]) and np.array_equal(t_rows, ref[k]["all_t"][ii])
            assert hdr["id1"][j] == ref[k]["id1"] and np.array_equal(hdr["trafo"][j], ref[k]["trafo"])
    # a list shorter than the device count, and an empty one
    p1, t1, s1 = grp.match_pair_list_allgather_inliers(pq[:1], pt[:1], [b.data_ptr() for b in bufs])
    assert p1 == 1 and int(t1.sum()) == int(ref["n_inl"][0]) and s1 == 104 + 4 * 320   # (the known capacity, cut to the worst case)
    assert grp.gather_exchanges() == 1
    p0, t0, s0 = grp.match_pair_list_allgather_inliers(pq[:0], pt[:0], [b.data_ptr() for b in bufs])
    assert p0 == 0 and s0 == 0
    one = _fe(seq)
    with pytest.raises(RgbdfeError):
        one.match_pair_list_allgather_inliers(pq, pt, [bufs[0].data_ptr()])       # needs a multi-device handle
    one.close()
    grp.close()
    # lists that outgrow what the group has seen: the exchange sized from the one-pair call is too small for the whole list --
    # a second exchange at the exact size, the same streams
    grp = _fe(seq, device_ids=ids)
    grp.match_pair_list_allgather_inliers(pq[:1], pt[:1], [b.data_ptr() for b in bufs])
    _, totals2, stride2 = grp.match_pair_list_allgather_inliers(pq, pt, [b.data_ptr() for b in bufs])
    assert grp.gather_exchanges() == 2 and np.array_equal(totals2, totals) and stride2 == per * 104 + 4 * int(totals.max())
    raw = bufs[-1].cpu().numpy()
    for d in range(G):
        hdr_want, lst_want = inlier_stream_of(ref[d::G], per)
        hdr, lst = parse_inlier_stream(raw[d * stride2:(d + 1) * stride2], per, int(totals2[d]))
        assert hdr.tobytes() == hdr_want.tobytes() and np.array_equal(lst, lst_want), d
    grp.close()
