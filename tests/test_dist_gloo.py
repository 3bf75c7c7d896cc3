"""world_size-2 gloo test of the N>1 path: pair sharding + all-gather of the MatchingResult PODs.
Without a GPU the per-rank records are synthetic (a pure function of the pair); the sharding /
padding / all-gather / reassembly logic is the same code bench.py and a multi-GPU host use."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from rgbdslam_v2_amd import dist as rdist
from rgbdslam_v2_amd._lib import (COMPACT_DTYPE, INLIER_HEADER_DTYPE, RESULT_DTYPE, compact_of, inlier_pairs, inlier_stream_of,
                                  parse_inlier_stream)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(pq, pt):
    """Deterministic per-pair records standing in for GPU output."""
    out = np.zeros(len(pq), RESULT_DTYPE)
    out["id1"], out["id2"] = pt, pq
    out["n_all"] = (pq * 7 + pt) % 300
    out["rmse"] = (pq + 0.5 * pt).astype(np.float32)
    out["trafo"][:, 0] = pq
    out["inlier_mask"][:, 0] = pq.astype(np.uint64) << np.uint64(7)
    out["inlier_mask"][:, 3] = (pt.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
    out["n_inl"] = np.unpackbits(out["inlier_mask"].view(np.uint8).reshape(len(pq), -1), axis=1).sum(1)
    m = np.arange(320, dtype=np.uint16)
    out["all_q"] = (m[None, :] * 3 + pq[:, None].astype(np.uint16)) % 1000
    out["all_t"] = (m[None, :] * 5 + pt[:, None].astype(np.uint16)) % 1000
    return out


def _inliers_of(rec):
    """(query rows, train rows) of a record's inlier matches, straight from its mask and lists."""
    bits = np.unpackbits(np.ascontiguousarray(rec["inlier_mask"]).view(np.uint8), bitorder="little")
    m = np.nonzero(bits)[0]
    return rec["all_q"][m].astype(np.int32), rec["all_t"][m].astype(np.int32)


def _worker(rank, world, port, n_pairs, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    pq = rng.integers(0, 50, n_pairs).astype(np.int32)
    pt = rng.integers(0, 50, n_pairs).astype(np.int32)
    sq, st = rdist.shard_pairs(pq, pt, rank, world)
    local = _fake_records(sq, st)
    allrec = rdist.all_gather_results(local, n_pairs)
    ok = allrec.tobytes() == _fake_records(pq, pt).tobytes()
    # the default payload: compact records (header + inlier mask, 144 B)
    allcmp = rdist.all_gather_compact(compact_of(local), n_pairs)
    ok = ok and allcmp.dtype == COMPACT_DTYPE and allcmp.tobytes() == compact_of(_fake_records(pq, pt)).tobytes()
    ok = ok and rdist.collective_device().type == "cpu"
    # accepted edges only (all-pairs sweeps): rejected pairs (id1 == -1) do not travel
    full = _fake_records(pq, pt)
    rej = (pq + pt) % 3 != 0
    full["id1"][rej] = -1
    full["id2"][rej] = -1
    idx, edges = rdist.all_gather_edges(full[rank::world], n_pairs)
    want = np.flatnonzero(~rej)
    ok = ok and np.array_equal(idx, want) and edges.tobytes() == full[want].tobytes()
    # the inlier form (bench.py's default payload): headers + (query row, train row) of every inlier
    hdr, pairs = rdist.all_gather_inliers(local, n_pairs)
    want_all = _fake_records(pq, pt)
    for f in INLIER_HEADER_DTYPE.names:
        if f != "first_inlier":
            ok = ok and np.array_equal(hdr[f], want_all[f])
    for k in range(n_pairs):
        a, b = _inliers_of(want_all[k])
        ok = ok and np.array_equal(pairs[k][0], a) and np.array_equal(pairs[k][1], b)
    # the same gather as ONE collective sized in advance: the lengths are read from the gathered streams' own headers
    import torch
    n_pad = max(rdist.shard_sizes(n_pairs, world))
    h_loc, l_loc = inlier_stream_of(local, n_pad)
    buf = torch.from_numpy(np.concatenate([h_loc.view(np.uint8).reshape(-1), l_loc.view(np.uint8).reshape(-1)]).copy())
    g2, tot2 = rdist.gather_inlier_streams(buf, len(l_loc), n_pad)                    # (the two-collective form: exact size)
    g1, tot1 = rdist.gather_inlier_streams_sized(buf, n_pad, max(tot2) + max(tot2) // 4 + 64)
    ok = ok and tot1.tolist() == tot2 and g1.shape[1] > g2.shape[1]
    h1, p1 = rdist.unshard_inlier_streams(g1.numpy(), tot1.tolist(), n_pairs, world)
    ok = ok and h1.tobytes() == hdr.tobytes() and all(np.array_equal(p1[k][0], pairs[k][0]) and
                                                     np.array_equal(p1[k][1], pairs[k][1]) for k in range(n_pairs))
    # a capacity that is too small: the headers still arrive whole, the totals say which lists were cut
    g0, tot0 = rdist.gather_inlier_streams_sized(buf, n_pad, 3)
    ok = ok and tot0.tolist() == tot2 and min(tot2) > 3 and g0.shape[1] == n_pad * INLIER_HEADER_DTYPE.itemsize + 12
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_shard_and_unshard_roundtrip():
    for n, w in [(0, 2), (1, 2), (7, 2), (8, 8), (4000, 8), (13, 4)]:
        pq = np.arange(n, dtype=np.int32)
        shards = [rdist.shard_pairs(pq, pq, r, w)[0] for r in range(w)]
        assert [len(s) for s in shards] == rdist.shard_sizes(n, w)
        n_pad = max(len(s) for s in shards) if n else 0
        g = np.zeros((w, n_pad), RESULT_DTYPE)
        for r, s in enumerate(shards):
            g["id2"][r, : len(s)] = s
        assert np.array_equal(rdist.unshard(g, n, w)["id2"], pq)


def test_compact_record_layout():
    """rgbdfe_compact_result = the record's first 104 bytes + its inlier mask (include/rgbdfe.h)."""
    assert COMPACT_DTYPE.itemsize == 144 and RESULT_DTYPE.itemsize == 1744
    rec = _fake_records(np.arange(5, dtype=np.int32), np.arange(5, dtype=np.int32) + 1)
    c = compact_of(rec)
    raw, craw = rec.view(np.uint8).reshape(5, 1744), c.view(np.uint8).reshape(5, 144)
    assert np.array_equal(craw[:, :104], raw[:, :104]) and np.array_equal(craw[:, 104:], raw[:, 1704:])


def test_inlier_stream_layout():
    """rgbdfe_inlier_header = the record's first 104 bytes with pad0 = first_inlier; the list block holds
    query | train << 16 of every set bit of inlier_mask in ascending order; padding headers carry ids -1."""
    assert INLIER_HEADER_DTYPE.itemsize == 104
    rec = _fake_records(np.arange(6, dtype=np.int32) * 9 + 1, np.arange(6, dtype=np.int32) + 2)
    hdr, lst = inlier_stream_of(rec, 8)
    raw, hraw = rec.view(np.uint8).reshape(6, 1744), hdr.view(np.uint8).reshape(8, 104)
    keep = np.r_[0:84, 88:104]
    assert np.array_equal(hraw[:6][:, keep], raw[:, :104][:, keep])
    assert np.array_equal(hdr["first_inlier"][:6], np.concatenate([[0], np.cumsum(rec["n_inl"])[:-1]]))
    assert np.all(hdr["id1"][6:] == -1) and np.all(hdr["n_inl"][6:] == 0) and len(lst) == rec["n_inl"].sum() > 20
    buf = np.concatenate([hdr.view(np.uint8).reshape(-1), lst.view(np.uint8).reshape(-1), np.zeros(40, np.uint8)])
    h2, l2 = parse_inlier_stream(buf, 8, len(lst))
    for k in range(6):
        a, b = inlier_pairs(h2, l2, k)
        wa, wb = _inliers_of(rec[k])
        assert np.array_equal(a, wa) and np.array_equal(b, wb)
    e, _ = inlier_stream_of(rec[:0], 3)
    assert np.all(e["id1"] == -1) and np.all(e["first_inlier"] == 0)


def test_all_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
