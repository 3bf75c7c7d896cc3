"""Multi-GPU pair sharding: one process per GPU, every rank holds all node
features (48 KB per 1000-keypoint node -- trivial against 288 GB of HBM), the candidate pair
list is sharded round-robin and the fixed-size MatchingResult PODs are exchanged with ONE
all-gather (RCCL over xGMI when the tensors live on the GPU, gloo on CPU in the tests).
No other collective exists on this path: pairs are independent
(QtConcurrent::blockingMapped has no cross-task dependency, graph_manager.cpp:548)."""
import numpy as np

from ._lib import COMPACT_DTYPE, INLIER_HEADER_DTYPE, RESULT_DTYPE, inlier_stream_of, parse_inlier_stream


def collective_device(group=None):
    """Where the tensors of a collective must live: the current GPU under the nccl (= RCCL) backend, the host otherwise."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def shard_pairs(pair_q, pair_t, rank: int, world: int):
    """pair k -> rank k mod world."""
    pair_q = np.asarray(pair_q, np.int32)
    pair_t = np.asarray(pair_t, np.int32)
    return np.ascontiguousarray(pair_q[rank::world]), np.ascontiguousarray(pair_t[rank::world])


def shard_sizes(n_pairs: int, world: int):
    return [len(range(r, n_pairs, world)) for r in range(world)]


def unshard(gathered: np.ndarray, n_pairs: int, world: int) -> np.ndarray:
    """gathered: [world, n_pad] records as all-gathered (rank-major, shards padded to n_pad).
    Returns the records in global pair order."""
    out = np.zeros(n_pairs, gathered.dtype)
    for r in range(world):
        k = len(range(r, n_pairs, world))
        out[r::world] = gathered[r, :k]
    return out


def all_gather_results(local_records, n_pairs: int, group=None, dtype=RESULT_DTYPE):
    """All-gather of the per-rank result records.  local_records: numpy array of `dtype` (RESULT_DTYPE, or
    COMPACT_DTYPE for the 144-byte payload) or a uint8 torch tensor already in HBM (RCCL).  Returns all
    records in global order (numpy)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_pairs, world)
    n_pad = max(sizes) if sizes else 0
    rec = dtype.itemsize
    if isinstance(local_records, np.ndarray):
        buf = np.zeros(n_pad, dtype)
        buf[: len(local_records)] = local_records
        local = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()).to(collective_device(group))
    else:
        local = local_records.reshape(-1)
        if local.numel() != n_pad * rec:
            pad = torch.zeros(n_pad * rec, dtype=torch.uint8, device=local.device)
            pad[: local.numel()] = local
            local = pad
    out = torch.empty(world * n_pad * rec, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    g = np.frombuffer(out.cpu().numpy().tobytes(), dtype=dtype).reshape(world, n_pad)
    return unshard(g, n_pairs, world)


def all_gather_compact(local_records, n_pairs: int, group=None):
    """The default payload of the multi-GPU gather: COMPACT_DTYPE records (header + inlier mask, 144 B instead of
    1744 B per pair).  local_records: numpy COMPACT_DTYPE array, or a uint8 tensor in HBM as rgbdfe_pack_compact
    wrote it."""
    return all_gather_results(local_records, n_pairs, group, dtype=COMPACT_DTYPE)


def gather_inlier_streams(local_stream, local_total: int, n_pad: int, group=None):
    """The two collectives of the inlier-stream gather (include/rgbdfe.h: rgbdfe_inlier_header): the ranks exchange the
    lengths of their list blocks (one tiny all-gather), then their streams padded to the longest.  local_stream: uint8 tensor
    holding at least n_pad * 104 + 4 * local_total bytes (HBM under RCCL, host under gloo).  Returns (the gathered uint8
    tensor [world, n_pad * 104 + 4 * max_total], the list of totals)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_stream.device
    t = torch.tensor([int(local_total)], dtype=torch.int64, device=dev)
    tots = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(tots, t, group=group)
    totals = [int(v) for v in tots.cpu().tolist()]
    nbytes = n_pad * INLIER_HEADER_DTYPE.itemsize + 4 * max(totals)
    if local_stream.numel() < nbytes:
        pad = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        pad[: local_stream.numel()] = local_stream.reshape(-1)
        local_stream = pad
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, local_stream.reshape(-1)[:nbytes], group=group)
    return out.reshape(world, nbytes), totals


def inlier_stream_totals(gathered, n_pad: int):
    """Lengths of the list blocks of gathered inlier streams ([world, bytes] uint8 tensor), read from the streams themselves: a
    stream's last header holds first_inlier + n_inl = the length of its list block (padding headers carry first_inlier = the
    length and n_inl = 0).  An int32 tensor [world] on the streams' device -- no host read."""
    import torch
    hb = INLIER_HEADER_DTYPE.itemsize
    w32 = gathered.reshape(gathered.shape[0], -1).view(torch.int32)
    last = (n_pad - 1) * hb // 4
    return w32[:, last + INLIER_HEADER_DTYPE.fields["first_inlier"][1] // 4] + w32[:, last + INLIER_HEADER_DTYPE.fields["n_inl"][1] // 4]


def gather_inlier_streams_sized(local_stream, n_pad: int, cap_entries: int, group=None):
    """The inlier-stream gather as ONE collective sized before the ranks have counted their lists: every rank sends n_pad
    headers + cap_entries list entries (cap_entries: agreed beforehand, e.g. the longest list of an earlier gather plus a
    quarter -- bench.py).  The lengths travel inside the streams (inlier_stream_totals), so the receiver needs no second
    collective and the sender no host read.  Returns (gathered [world, n_pad * 104 + 4 * cap_entries], the totals as a
    device tensor [world]); a total above cap_entries means that rank's list was cut: gather again with a larger capacity
    (every rank sees the same totals and decides alike)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_stream.device
    nbytes = n_pad * INLIER_HEADER_DTYPE.itemsize + 4 * int(cap_entries)
    flat = local_stream.reshape(-1)
    if flat.numel() < nbytes:
        pad = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        pad[: flat.numel()] = flat
        flat = pad
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, flat[:nbytes], group=group)
    out = out.reshape(world, nbytes)
    return out, inlier_stream_totals(out, n_pad)


def unshard_inlier_streams(gathered: np.ndarray, totals, n_pairs: int, world: int):
    """gathered [world, bytes] (numpy uint8) -> (headers in global pair order [n_pairs], list of (query rows, train rows) per
    pair): what updateInlierFeatures (graph_manager.cpp:409-419) reads, for every pair of the global list."""
    sizes = shard_sizes(n_pairs, world)
    n_pad = max(sizes) if sizes else 0
    hdr_all = np.zeros(n_pairs, INLIER_HEADER_DTYPE)
    pairs = [None] * n_pairs
    for r in range(world):
        hdr, lst = parse_inlier_stream(gathered[r], n_pad, totals[r])
        for j in range(sizes[r]):
            k = r + j * world
            hdr_all[k] = hdr[j]
            e = lst[int(hdr["first_inlier"][j]): int(hdr["first_inlier"][j]) + int(hdr["n_inl"][j])]
            pairs[k] = ((e & 0xFFFF).astype(np.int32), (e >> 16).astype(np.int32))
    return hdr_all, pairs


def all_gather_inliers(local_records: np.ndarray, n_pairs: int, group=None):
    """All-gather of the inlier form of the per-rank results (host records in, global order out): see gather_inlier_streams /
    unshard_inlier_streams.  local_records: this rank's RESULT_DTYPE records in shard order (numpy)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_pad = max(shard_sizes(n_pairs, world))
    hdr, lst = inlier_stream_of(local_records, n_pad)
    buf = np.concatenate([hdr.view(np.uint8).reshape(-1), lst.view(np.uint8).reshape(-1)])
    g, totals = gather_inlier_streams(torch.from_numpy(buf.copy()).to(collective_device(group)), len(lst), n_pad, group)
    return unshard_inlier_streams(g.cpu().numpy(), totals, n_pairs, world)


def all_gather_edges(local_records, n_pairs: int, group=None):
    """All-gather of the ACCEPTED edges only (for an all-pairs loop-closure sweep most pairs are
    rejected, id1 == -1, and need not travel).  Every rank compacts its accepted records, the ranks exchange their
    counts (one small all-gather), pad to the largest count and exchange the records plus their global pair indices.
    local_records: this rank's RESULT_DTYPE records in shard order (numpy).  Returns (pair_index int64 [E], records [E])
    in global pair order, identical on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local_records = np.asarray(local_records)
    keep = np.flatnonzero(local_records["id1"] >= 0)
    gidx = rank + world * keep.astype(np.int64)
    assert len(local_records) == len(range(rank, n_pairs, world))
    dev = collective_device(group)  # RCCL moves device tensors only
    cnt = torch.tensor([len(keep)], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    n_pad = int(cnts.max().item())
    rec = RESULT_DTYPE.itemsize
    if n_pad == 0:
        return np.zeros(0, np.int64), np.zeros(0, RESULT_DTYPE)
    buf = np.zeros(n_pad, RESULT_DTYPE)
    buf[: len(keep)] = local_records[keep]
    ibuf = np.full(n_pad, -1, np.int64)
    ibuf[: len(keep)] = gidx
    t_rec = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()).to(dev)
    t_idx = torch.from_numpy(ibuf).to(dev)
    o_rec = torch.empty(world * n_pad * rec, dtype=torch.uint8, device=dev)
    o_idx = torch.empty(world * n_pad, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(o_rec, t_rec, group=group)
    dist.all_gather_into_tensor(o_idx, t_idx, group=group)
    allrec = np.frombuffer(o_rec.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)
    allidx = o_idx.cpu().numpy()
    sel = np.flatnonzero(allidx >= 0)
    order = sel[np.argsort(allidx[sel], kind="stable")]
    return allidx[order].copy(), allrec[order].copy()
