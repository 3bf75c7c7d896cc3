"""-m gpu: two RANKS of bench.py's step on the one GPU of the test box -- two processes, each with its own context on
cuda:0, the pair list sharded pair k -> rank k mod 2, REAL kernels on both shards, the result PODs exchanged with the
all-gather bench.py uses (gloo here: RCCL refuses two ranks on one device); the gathered bytes must equal the
single-rank result (results do not depend on the sharding)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from rgbdslam_v2_amd import dist as rdist
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq = synth.make_sequence(n_frames=16, n_kp=600, n_world=2400, seed=33)
        pq, pt = synth.candidate_pairs(16, per_frame=9, seed=33)
        # an all-pairs flavour: a few unrelated pairs that RANSAC rejects
        rng = np.random.default_rng(1)
        seq["desc"][15] = rng.integers(0, 256, seq["desc"][15].shape, dtype=np.uint8)
        fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=640, max_pairs_per_batch=256)
        for f in range(16):
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        sq, st = rdist.shard_pairs(pq, pt, rank, world)
        local = fe.match_pair_list(sq, st)
        allrec = rdist.all_gather_results(local, len(pq))
        idx, edges = rdist.all_gather_edges(local, len(pq))
        ok = True
        if rank == 0:
            ref = fe.match_pair_list(pq, pt)
            ok = allrec.tobytes() == ref.tobytes()
            want = np.flatnonzero(ref["id1"] >= 0)
            ok = ok and np.array_equal(idx, want) and edges.tobytes() == ref[want].tobytes()
            ok = ok and 0 < len(want) < len(pq)
        fe.close()
        q.put((rank, bool(ok)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_real_kernels():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(0, True), (1, True)], res
