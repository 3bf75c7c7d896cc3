"""world_size-2 gloo test of the N>1 path: pair sharding + all-gather of the MatchingResult PODs.
Without a GPU the per-rank records are synthetic (a pure function of the pair); the sharding /
padding / all-gather / reassembly logic is the same code bench.py and a multi-GPU host use."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from rgbdslam_v2_amd import dist as rdist
from rgbdslam_v2_amd._lib import COMPACT_DTYPE, RESULT_DTYPE, compact_of


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
    return out


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
