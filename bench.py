#!/usr/bin/env python3
"""bench.py -- frame-pairs matched+RANSAC'd per second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (Hamming match -> select -> RANSAC, the whole
Node::matchNodePair pair op) over one batch of candidate pairs of the synthetic
BASELINE configs[1] workload: 640x480 RGB-D, ORB 1000 keypoints/frame, 200 frames,
20 candidate pairs per frame = 4000 pairs per GPU per step.  Node features are resident
in HBM before the timed region starts.  With N > 1 (one rank per GPU, launched by
torch.distributed.run) every rank holds all nodes, the global pair list (N x 4000 pairs)
is sharded pair k -> rank k mod N, and each step ends with the RCCL all-gather of the
MatchingResult PODs: weak scaling.

Rank 0 prints ONE JSON line.  The timed region (K steps between barrier + synchronize) is run REPEATS = 7 times in
the invocation: `value` / `ms_per_step` are the median repetition, `repeats` lists all of them.  The head of the line is
self-sufficient: `roofline` carries, as flat numbers, the fraction and the average launch time of every
kernel family the line reports (`match_*`, `sift_*`, `detect_*`, `sift_extract_*`, `heavy_*`) and the headline measured a
second time with the library's DEFAULT launch path (`default_path_*`: plain launches, no hipGraph); every `*_avg_launch_ms`
there is a serial (one batch in flight) time that the serial traces under profiles/r06/ reproduce.  Its `roofline` block is
computed from times measured in THIS run: per-stage HIP-event times of steps that run one batch at a time (`serial`),
next to the pipelined step time the metric uses (`frac_serial` / `frac_pipelined`); counter-derived figures (`traffic`,
`issue_roofline`) are static and name the `profiles/` file they come from.  At N = 1 the line also carries the
`ransac_heavy` (depth noise 0.002 z^2), `sift` (configs[3]), `sift_extract`, `detect` (Level B frames/s), `front_end`,
`host_io` and `loop_closure` sub-records and the CPU baselines, and the results of the last step are checked against aggregate figures
the oracle produced for the same seeded workload (EXPECTED; tests/test_gpu_pairs.py re-derives them bit for bit).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_KP = 1000
N_FRAMES = 200
PAIRS_PER_FRAME = 20
MAX_MATCHES = 300
SEED = 20260923
REPEATS = 7                                # repetitions of the timed region (median reported)
PMC_SUMMARY = "profiles/r06_pmc_summary.json"   # static counter figures (tools/profile_round.sh + tools/make_pmc_summary.py)
PMC_FALLBACK = "profiles/r05_pmc_summary.json"
# Aggregates of every workload this file prints a number for, as the ORACLE computes them (tools/make_bench_expected.py:
# oracle/liboracle.so, oracle/orb_oracle.c and the compiled reference SiftGPU pipeline over the same seeded workloads,
# offline on the CPU; committed as tests/golden/bench_expected.json).  The -m gpu tests compare the same workloads record by
# record (tests/test_gpu_pairs.py::test_whole_bench_step_matches_oracle, test_loop_closure_subrecord_matches_oracle,
# tests/test_gpu_bench_parity.py).  bench.py refuses to print a number whose results differ.
EXPECTED_FILE = "tests/golden/bench_expected.json"
try:
    EXPECTED = json.load(open(os.path.join(ROOT, EXPECTED_FILE)))
except Exception:  # noqa: BLE001
    EXPECTED = {}
SIFT_DESC_RTOL = 2e-3   # sift_extract: sum of |descriptor elements| per frame vs the reference's (libm differences)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz (int32 VALU lanes/s)


def algorithmic_bytes(n_kp, m):
    """SURVEY.md 8(d).  Whole pair path: 104*N + 24*M + 64 (= 111264 B at N=1000, M=300).
    Per kernel (DESIGN.md 'Kernels'): Hamming = both descriptor sets read once + packed
    (hd,idx) key written per query; select+RANSAC = keys read + matched xyz1 gathered +
    result POD written."""
    pair = 104 * n_kp + 24 * m + 64
    hamming = 2 * n_kp * 32 + 4 * n_kp
    ransac = 4 * n_kp + 2 * m * 16 + 1744
    return pair, hamming, ransac


def expected(*path):
    """EXPECTED[path[0]][path[1]]... (keys as strings), or None."""
    d = EXPECTED
    for k in path:
        if not isinstance(d, dict) or str(k) not in d:
            return None
        d = d[str(k)]
    return d


def pair_aggregates(res):
    """Sums over result records (RESULT_DTYPE or COMPACT_DTYPE): what the oracle constants of the pair workloads are."""
    return {"edges": int((res["id1"] >= 0).sum()), "real_iterations": int(res["real_iterations"].astype(np.int64).sum()),
            "inliers": int(res["n_inl"].astype(np.int64).sum())}


def features_checksum(frames):
    """(keypoints, descriptors, points) of a run of frames -> {"keypoints": total, "crc32": of every keypoint field,
    descriptor byte and point coordinate in frame order}: the constants of the detect / front_end sub-records."""
    import zlib
    crc, n = 0, 0
    for kp, desc, xyz in frames:
        n += len(kp)
        for f in ("x", "y", "size", "angle", "response", "octave"):
            crc = zlib.crc32(np.ascontiguousarray(kp[f]).tobytes(), crc)
        crc = zlib.crc32(np.ascontiguousarray(desc, np.uint8).tobytes(), crc)
        crc = zlib.crc32(np.ascontiguousarray(xyz, np.float32).tobytes(), crc)
    return {"keypoints": int(n), "crc32": int(crc)}


def sift_features_checksum(frames):
    """SIFT extraction outputs of a run of frames: feature counts and positions are bit-exact against the reference
    (crc32 over x, y), descriptors within a libm tolerance (sum of |elements| per run, compared with SIFT_DESC_RTOL)."""
    import zlib
    crc, counts, dsum = 0, [], 0.0
    for kp, desc in frames:
        counts.append(int(len(kp)))
        crc = zlib.crc32(np.ascontiguousarray(kp["x"], np.float32).tobytes(), crc)
        crc = zlib.crc32(np.ascontiguousarray(kp["y"], np.float32).tobytes(), crc)
        dsum += float(np.abs(np.asarray(desc, np.float64)).sum())
    return {"features_per_frame": counts, "xy_crc32": int(crc), "descriptor_abs_sum": dsum}


def check_against(exp, got, what, source, approx=()):
    """parity_check block of a (sub-)record: raises when `got` differs from the oracle constants `exp`."""
    if exp is None:
        return {"checked": False, "got": got, "note": "no oracle constants for this workload (non-default flags)"}
    bad = []
    for k, v in exp.items():
        if k in approx:
            if abs(got[k] - v) > SIFT_DESC_RTOL * abs(v):
                bad.append(k)
        elif got.get(k) != v:
            bad.append(k)
    if bad:
        raise SystemExit("bench.py: results of %s differ from the oracle's constants in %s: got %r, expected %r" % (what, bad, got, exp))
    # (`source` -- which oracle produced the constants and which test compares record by record -- documents the call site; the
    # line itself names the constants' file only: thirteen such blocks made the line longer than what the driver keeps of it)
    out = {"checked": True, "ok": True, "oracle_aggregates": exp, "source": EXPECTED_FILE}
    if approx:
        out["tolerance"] = {k: SIFT_DESC_RTOL for k in approx}
    return out


def parity_check(path, res):
    """Sums over one step's records vs the oracle's (EXPECTED[path...]): raises when they differ."""
    return check_against(expected(*path) if path else None, pair_aggregates(res), "the timed workload %r" % (path,),
                         "oracle/liboracle.so on the same seeded workload (%s); every pair bit for bit in tests/test_gpu_pairs.py, "
                         "tests/test_gpu_bench_parity.py" % EXPECTED_FILE)


# ---- the workloads of the sub-records (tools/make_bench_expected.py builds the same ones for the oracle) ------------------
def orb_workload(world, frames=N_FRAMES, n_kp=N_KP, pairs_per_frame=PAIRS_PER_FRAME, depth_noise=0.01):
    """configs[1] at `world` ranks (weak scaling): the nodes and the GLOBAL pair list (world x 4000 pairs; rank r owns pairs r::world)."""
    from rgbdslam_v2_amd import synth
    seq = synth.make_sequence(n_frames=frames, n_kp=n_kp, seed=SEED, depth_noise=depth_noise)
    per_frame = min(pairs_per_frame * world, frames - 1)
    pq, pt = synth.candidate_pairs(frames, per_frame=per_frame, seed=SEED)
    return seq, pq, pt


SIFT_FRAMES = 100


def sift_workload(seq):
    """configs[3]: the first 100 nodes of configs[1] with 128-d float descriptors, 20 candidates each = 2000 pairs."""
    from rgbdslam_v2_amd import synth
    sd = synth.sift_descriptors_like(seq["desc"][:SIFT_FRAMES], seed=SEED)
    pq, pt = synth.candidate_pairs(SIFT_FRAMES, per_frame=20, seed=SEED)
    return sd, pq, pt


DETECT_WORKLOADS = ((640, 480, 1000, 28, 112), (1280, 960, 4000, 14, 56))   # w, h, keypoints, generated frames, frames per run


def detect_workload(w, h, n_base, n_run):
    """A recorded sequence for the detect sub-records: n_base generated frames walked forth and back for n_run frames."""
    from rgbdslam_v2_amd import synth
    seq = synth.make_image_sequence(n_frames=n_base, seed=1, width=w, height=h)
    masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in seq["mask"]]
    idx = synth.forth_and_back(n_run, n_base)
    K = (seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    return seq, masks, [seq["gray"][i] for i in idx], [masks[i] for i in idx], [seq["depth"][i] for i in idx], K


FRONT_END = dict(n_base=28, n_run=112, cand=20, n_kp=1000)


def front_end_pairs(n_run, cand):
    pq = np.array([f for f in range(1, n_run) for c in range(1, min(cand, f) + 1)], np.int32)
    pt = np.array([f - c for f in range(1, n_run) for c in range(1, min(cand, f) + 1)], np.int32)
    return pq, pt


def sift_extract_workload():
    from rgbdslam_v2_amd import synth
    seq = synth.make_image_sequence(n_frames=8, seed=1)
    return seq, [seq["gray"][i] for i in synth.forth_and_back(32, len(seq["gray"]))]


def load_pmc():
    for rel in (PMC_SUMMARY, PMC_FALLBACK):
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
            # (every section of the summary carries the tree it was collected on: tools/make_pmc_summary.py)
            return d, rel + (" @ commit " + ", ".join(c[:12] for c in d["commits"]) if d.get("commits") else "")
        except Exception:
            continue
    return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=25)   # (~28 ms: the first ~20 ms after an idle period run 6 - 7 % slow)
    ap.add_argument("--frames", type=int, default=N_FRAMES)
    ap.add_argument("--kp", type=int, default=N_KP)
    ap.add_argument("--pairs-per-frame", type=int, default=PAIRS_PER_FRAME)
    ap.add_argument("--config", choices=["orb", "sift"], default="orb",
                    help="orb = BASELINE configs[1] (the headline metric); sift = configs[3]: SIFT 128-d "
                         "float descriptors, dot-product matrix on the bf16 MFMA")
    ap.add_argument("--ransac-path", choices=["default", "one_wave", "record_replay"], default="default",
                    help="which select+RANSAC schedule the library uses for the batches of this run: its default "
                         "(record / replay up to the library's batch limit, one wave per pair above it), or forced")
    ap.add_argument("--chunk-iterations", type=int, default=0, help="iterations per wave on the record / replay path")
    ap.add_argument("--depth-noise", type=float, default=0.01,
                    help="sigma of the synthetic depth noise as a multiple of z^2 (SURVEY.md 8(d): 0.01 = sigma_depth; "
                         "round 1 measured with 0.002)")
    ap.add_argument("--hamming-mode", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="-1 = the library's default (fp4 MFMA contraction), 0 = xor+popcount kernel, 2 = MFMA with VALU row term")
    ap.add_argument("--no-extras", action="store_true", help="skip the sub-records")
    ap.add_argument("--gather", choices=["inliers", "compact", "full"], default="inliers",
                    help="N > 1: payload of the per-step all-gather -- the inlier stream (default: 104-byte header + 4 bytes per "
                         "inlier match, what GraphManager reads of a MatchingResult), rgbdfe_compact_result (144 B per pair: "
                         "header + inlier mask, the lists stay behind) or the whole 1744-byte record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    # The launch chain of an ORB pair batch as a cached hipGraph (rgbdfe_set_graph_capture): off by default in the library --
    # an open capture makes device-wide synchronisations on OTHER threads of the process fail -- on here, where every HIP
    # call of the process comes from this thread (the line's config says so).
    os.environ.setdefault("RGBDFE_GRAPHS", "1")

    # `python bench.py --gpus N` creates its N ranks itself: without WORLD_SIZE in the environment and with
    # N > 1 the process re-executes under torch.distributed.run exactly as the driver would launch it (one rank per GPU,
    # rendezvous on 127.0.0.1, a free port).  A WORLD_SIZE that disagrees with --gpus -- in EITHER direction -- is an error:
    # eight "N = 1" lines must never pass for a scaling curve.  (GraphManager's pair fan-out this replaces:
    # graph_manager.cpp:541-560.)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
                 + sys.argv[1:])

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or leave WORLD_SIZE unset and "
                         f"let bench.py start them)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # RGBDFE_BENCH_BACKEND=gloo (tests only): several ranks on ONE GPU -- RCCL refuses two ranks on a device -- to exercise
    # the sharding / gather / timing logic of the N > 1 path; collectives then go through host tensors.
    backend = os.environ.get("RGBDFE_BENCH_BACKEND", "nccl")
    # RGBDFE_BENCH_FORCE_GATHER=1 (tests only): the N > 1 gather code with ONE rank -- the only way to run its RCCL branch
    # (process group, pack kernels, the inlier gather, parity check on the gathered records) on a one-GPU box
    force_gather = os.environ.get("RGBDFE_BENCH_FORCE_GATHER") == "1"
    gather_on = world > 1 or force_gather
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1 or force_gather:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    host_coll = (world > 1 or force_gather) and backend != "nccl"

    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import (KERNEL_HAMMING, KERNEL_RANSAC, KERNEL_SIFT_DOT,
                                      KERNEL_SIFT_FINISH, RESULT_DTYPE)
    from rgbdslam_v2_amd.frontend import FrontEnd
    from rgbdslam_v2_amd.dist import shard_pairs

    F, N = args.frames, args.kp
    # global pair list: per frame 20*world candidates (weak scaling), sharded round-robin
    seq, pq_all, pt_all = orb_workload(world, F, N, args.pairs_per_frame, args.depth_noise)
    pq, pt = shard_pairs(pq_all, pt_all, rank, world)
    n_local = len(pq)
    counts = [n_local]
    if world > 1 or force_gather:
        t_cnt = torch.tensor([n_local], device="cpu" if host_coll else "cuda")
        all_cnt = [torch.zeros_like(t_cnt) for _ in range(world)]
        dist.all_gather(all_cnt, t_cnt)
        counts = [int(c.item()) for c in all_cnt]
    n_pad = max(counts)

    fe = FrontEnd(device_id=local_rank, max_nodes=F, max_keypoints=((N + 63) // 64) * 64,
                  max_pairs_per_batch=max(n_pad, 1), seed=SEED)
    if args.hamming_mode >= 0:
        fe.set_hamming_mode(args.hamming_mode)
    if args.ransac_path == "one_wave":
        fe.set_latency_mode(0, 0)
    elif args.ransac_path == "record_replay":
        fe.set_latency_mode((1 << 31) - 1, args.chunk_iterations)
    # node features -> HBM (resident before the timed region)
    sift = args.config == "sift"
    sift_desc = synth.sift_descriptors_like(seq["desc"], seed=SEED) if sift else None
    for f in range(F):
        if sift:
            fe.upload_sift_node(f, sift_desc[f], seq["xyz1"][f])
        else:
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])

    rec_bytes = RESULT_DTYPE.itemsize
    from rgbdslam_v2_amd._lib import COMPACT_DTYPE, INLIER_HEADER_DTYPE, RGBDFE_MAX_MATCHES
    compact = gather_on and args.gather == "compact"
    inliers = gather_on and args.gather == "inliers"
    hdr_bytes = INLIER_HEADER_DTYPE.itemsize
    stream_cap = n_pad * (hdr_bytes + 4 * RGBDFE_MAX_MATCHES)      # a shard's inlier stream at its largest
    gat_bytes = COMPACT_DTYPE.itemsize if compact else rec_bytes    # (fixed-size payloads)
    rccl_ranks = None
    rank_devices = None
    cdev = "cpu" if host_coll else "cuda"
    if gather_on:
        # what the collective library itself saw: every rank contributes 1 through the backend the steps use
        ones = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        # ... and which device every rank really sits on (N ranks on N distinct devices is what a scaling line claims)
        prop = torch.cuda.get_device_properties(local_rank)
        mine = "rank %d: cuda:%d %s" % (rank, local_rank, getattr(prop, "uuid", None) or getattr(prop, "pci_bus_id", "?"))
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    # Steps are pipelined: step k is submitted to one of the context's internal streams while
    # step k-1 still runs (its RANSAC tail overlaps step k's Hamming kernel).  A ring of result
    # buffers keeps every step's output alive until its all-gather has consumed it.
    NBUF = 4
    d_local = [torch.zeros(n_pad * rec_bytes, dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
    if inliers:
        d_send = [torch.zeros(stream_cap, dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
        d_tot = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(NBUF)]
        d_all = torch.zeros(world * stream_cap, dtype=torch.uint8, device="cuda")
        d_tots = torch.zeros(world, dtype=torch.int64, device=cdev)
    else:
        d_send = [torch.zeros(n_pad * gat_bytes, dtype=torch.uint8, device="cuda") for _ in range(NBUF)] if compact else d_local
        d_all = torch.zeros(world * n_pad * gat_bytes, dtype=torch.uint8, device="cuda") if gather_on else None
    consumed = [None] * NBUF
    stream = torch.cuda.current_stream().cuda_stream
    state = {"k": 0, "bytes": 0, "gathers": 0, "gathers_all": 0, "last": None}

    # The inlier streams' gather: ONE collective per step and no host read in front of it.  The collective is
    # sized BEFORE the ranks have counted their lists: n_pad headers + `cap` list entries per rank, cap = the longest list of
    # the first (warm-up) gather plus a quarter -- the only gather that exchanges the lengths first.  Every stream carries
    # its own length (its last header: first_inlier + n_inl), so what every rank got is checked on the device from the
    # gathered headers; the check's result is read by the host two steps later (an event that has long fired by then), and
    # a list that has outgrown `cap` -- every rank sees the same gathered headers and decides alike, at the same step --
    # repeats that step's gather at a size that fits from the ring buffer that still holds its stream.
    hdr_dt = INLIER_HEADER_DTYPE
    off_first, off_ninl = hdr_dt.fields["first_inlier"][1], hdr_dt.fields["n_inl"][1]
    # (RGBDFE_BENCH_INLIER_CAP: tests only -- a capacity to start from instead of the learned one, e.g. one that is too small)
    inl = {"cap": int(os.environ.get("RGBDFE_BENCH_INLIER_CAP", "0")), "pending": [], "collectives": 0, "regathers": 0, "idx": None}
    if inliers:
        h_chk = [torch.zeros(world, dtype=torch.int32).pin_memory() for _ in range(NBUF)]
        ev_chk = [torch.cuda.Event() for _ in range(NBUF)]

    def exchange_streams(b, entries):
        nbytes = n_pad * hdr_bytes + 4 * entries
        out = d_all[: world * nbytes]
        if host_coll:
            h_all = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(h_all, d_send[b][:nbytes].cpu())
            out.copy_(h_all)
        else:
            dist.all_gather_into_tensor(out, d_send[b][:nbytes])
        inl["collectives"] += 1
        state["bytes"] += world * nbytes
        return nbytes, out

    def learn_capacity(b):
        """The first gather: the list lengths (one int64 per rank) through the host, then the streams at the longest."""
        torch.cuda.current_stream().synchronize()
        mine = torch.tensor([int(d_tot[b].item())], dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(d_tots, mine)
        inl["collectives"] += 1
        totals = [int(v) for v in d_tots.cpu().tolist()]
        nbytes, _ = exchange_streams(b, max(totals))
        inl["cap"] = min(max(totals) + max(totals) // 4 + 64, n_pad * RGBDFE_MAX_MATCHES)
        inl["idx"] = None
        state["last"] = (nbytes, totals)

    def gather_inliers(b):
        if inl["cap"] == 0:
            learn_capacity(b)
        else:
            nbytes, out = exchange_streams(b, inl["cap"])
            if inl["idx"] is None:   # word positions of every rank's last header's (first_inlier, n_inl) in the gathered buffer
                base = [(r * nbytes + (n_pad - 1) * hdr_bytes) // 4 for r in range(world)]
                inl["idx"] = (torch.tensor([x + off_first // 4 for x in base], device="cuda"),
                              torch.tensor([x + off_ninl // 4 for x in base], device="cuda"))
            w32 = out.view(torch.int32)
            h_chk[b].copy_(w32[inl["idx"][0]] + w32[inl["idx"][1]], non_blocking=True)
            ev_chk[b].record()
            inl["pending"].append((b, nbytes, inl["cap"], state["k"]))
        consumed[b] = torch.cuda.Event()
        consumed[b].record()
        state["gathers"] += 1
        state["gathers_all"] += 1

    def check_gathers(upto):
        """The length checks of the gathers of steps <= upto (None: all of them).  Which gathers are checked at which step is
        a function of the step count alone -- never of whether an event happens to have fired -- so that every rank issues a
        repeated gather at the same position of its sequence of collectives; the events of steps two behind have long fired."""
        redo = False   # behind a repeated gather every later one is repeated too: d_all ends up holding the LAST step's streams
        while inl["pending"] and (upto is None or inl["pending"][0][3] <= upto or redo):
            b, nbytes, cap_used, _ = inl["pending"].pop(0)
            ev_chk[b].synchronize()
            totals = [int(v) for v in h_chk[b].tolist()]
            if max(totals) > cap_used or redo:   # (the same on every rank: all of them read the same gathered headers)
                redo = True
                inl["regathers"] += 1
                if max(totals) > inl["cap"]:
                    inl["cap"] = min(max(totals) + max(totals) // 4 + 64, n_pad * RGBDFE_MAX_MATCHES)
                    inl["idx"] = None
                nbytes, _ = exchange_streams(b, inl["cap"])
                consumed[b] = torch.cuda.Event()
                consumed[b].record()
            state["last"] = (nbytes, totals)

    def flush():
        check_gathers(None)

    def step():
        b = state["k"] % NBUF
        state["k"] += 1
        if consumed[b] is not None:
            consumed[b].synchronize()  # the all-gather that read this buffer (4 steps ago) is done
        if sift:
            ticket = fe.submit_sift_pair_list(pq, pt, d_local[b].data_ptr())
        else:
            ticket = fe.submit_pair_list(pq, pt, d_local[b].data_ptr())
        if gather_on:
            fe.wait_ticket(ticket, stream)  # torch's stream waits for this batch only
            if inliers:
                # headers + (query row, train row) of every inlier match: inlier_scan_kernel + inlier_list_kernel on torch's
                # stream, and behind them -- stream ordered, no host read -- the step's one collective (gather_inliers)
                fe.pack_inliers(d_local[b].data_ptr(), n_local, n_pad, d_send[b].data_ptr(), d_tot[b].data_ptr(), stream)
                check_gathers(state["k"] - 2)   # the length checks of the steps two and more behind this one (state["k"]: this step, from 1)
                gather_inliers(b)
                return
            if compact:  # header + inlier mask of every record (144 of 1744 B): compact_pack_kernel on torch's stream
                fe.pack_compact(d_local[b].data_ptr(), n_local, d_send[b].data_ptr(), stream)
            if host_coll:
                h_all = torch.empty(d_all.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(h_all, d_send[b].cpu())
                d_all.copy_(h_all)
            else:
                dist.all_gather_into_tensor(d_all, d_send[b])
            consumed[b] = torch.cuda.Event()
            consumed[b].record()
            state["bytes"] += d_all.numel()
            state["gathers"] += 1

    fe.set_profiling(True)          # before the warm-up: its steps then take the very path the timed steps take
    for _ in range(args.warmup):
        step()
    if inliers:
        flush()
    fe.synchronize()
    torch.cuda.synchronize()
    state["bytes"] = state["gathers"] = 0
    fe.reset_kernel_time()
    # The timed region -- exactly K steps between barrier + synchronize on both sides, max over ranks -- REPEATS times;
    # the median repetition is the reported one (one region lasts a few tens of ms: single runs spread by several %).
    rep_elapsed = []
    for _rep in range(REPEATS):
        if world > 1:
            dist.barrier()
        fe.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if inliers:
            flush()                 # the last step's gather (K steps = K gathers inside the timed region)
        fe.synchronize()            # every internal stream of the context
        torch.cuda.synchronize()    # device-wide
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        el = t1 - t0
        if world > 1:
            te = torch.tensor([el], device="cpu" if host_coll else "cuda", dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        rep_elapsed.append(el)
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]
    gathers_timed, gather_bytes_timed = state["gathers"], state["bytes"]   # (the default-path leg below gathers too)
    fe.set_profiling(False)
    # Official per-kernel HIP-event times: measured inside the timed region (batches overlap there).
    k_match = KERNEL_SIFT_DOT if sift else KERNEL_HAMMING
    ham_ms, ham_launches, ham_pairs = fe.kernel_time(k_match)
    rsc_ms, rsc_launches, rsc_pairs = fe.kernel_time(KERNEL_RANSAC)
    fin_ms, fin_launches, _ = fe.kernel_time(KERNEL_SIFT_FINISH)
    # The same timed region once more on the library's DEFAULT launch path (graph capture off: plain launches -- what a
    # multi-threaded integration runs): reported beside the headline, never instead of it.
    plain_elapsed = None
    if not sift and os.environ.get("RGBDFE_GRAPHS") != "0":
        fe.set_graph_capture(False)
        for _ in range(max(args.warmup, 4)):
            step()
        if inliers:
            flush()
        pe = []
        for _rep in range(3):
            if world > 1:
                dist.barrier()
            fe.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if inliers:
                flush()
            fe.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t0
            if world > 1:
                te = torch.tensor([el], device="cpu" if host_coll else "cuda", dtype=torch.float64)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te.item())
            pe.append(el)
        plain_elapsed = sorted(pe)[1]
        fe.set_graph_capture(True)
        for _ in range(4):
            step()
        if inliers:
            flush()
        fe.synchronize()
        torch.cuda.synchronize()
    # Second, clearly separate measurement: SERIAL steps (one batch in flight at a time), with their own wall clock.
    # In the timed region two batches share the chip, so a stage's HIP-event span there is inflated by its neighbour
    # and can exceed ms_per_step; the serial steps give stage times that add up to a step time measured the same way.
    iso = {}
    fe.reset_kernel_time()
    fe.set_profiling(True)
    n_serial = 5
    torch.cuda.synchronize()
    ts0 = time.perf_counter()
    for _ in range(n_serial):
        if sift:
            tk = fe.submit_sift_pair_list(pq, pt, d_local[0].data_ptr())
        else:
            tk = fe.submit_pair_list(pq, pt, d_local[0].data_ptr())
        fe.wait_ticket(tk, None)
    fe.synchronize()
    serial_ms = (time.perf_counter() - ts0) / n_serial * 1e3
    fe.set_profiling(False)
    for nme, kk in (("match", k_match), ("ransac", KERNEL_RANSAC), ("sift_finish", KERNEL_SIFT_FINISH)):
        ms, nl, _ = fe.kernel_time(kk)
        if nl:
            iso["serial_%s_ms" % nme] = round(ms / nl, 4)
    iso["serial_ms_per_step"] = round(serial_ms, 4)

    total_pairs = sum(counts) * args.steps
    value = total_pairs / elapsed

    # sanity: results of the last step are real (edges found)
    last = d_local[(state["k"] - 1) % NBUF]
    res = np.frombuffer(last.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:n_local]
    edge_frac = float((res["id1"] >= 0).mean()) if n_local else 0.0
    mean_iters = float(res["real_iterations"].mean()) if n_local else 0.0
    default_workload = (not sift and F == N_FRAMES and N == N_KP and args.pairs_per_frame == PAIRS_PER_FRAME)
    # N = 1: this rank's records; N > 1: what the all-gather of the last step left on rank 0 -- every rank's records, as the
    # consumer of the gather sees them -- against the oracle's sums over the world x 4000 pairs of the global list
    list_entries = None
    if gather_on:
        torch.cuda.synchronize()
        if inliers:
            from rgbdslam_v2_amd._lib import parse_inlier_stream
            nbytes, totals = state["last"]
            g = d_all[: world * nbytes].cpu().numpy().reshape(world, nbytes)
            parsed = [parse_inlier_stream(g[r], n_pad, totals[r]) for r in range(world)]
            res_all = np.concatenate([parsed[r][0][:counts[r]] for r in range(world)])
            list_entries = int(sum(totals))
            # every header's list lies inside its rank's list block, lists follow each other without gaps
            for r in range(world):
                h = parsed[r][0][:counts[r]]
                assert np.array_equal(h["first_inlier"], np.concatenate([[0], np.cumsum(h["n_inl"])[:-1]])) and \
                    int(h["n_inl"].sum()) == totals[r], "inlier stream of rank %d is inconsistent" % r
        else:
            g = np.frombuffer(d_all.cpu().numpy().tobytes(), dtype=COMPACT_DTYPE if compact else RESULT_DTYPE).reshape(world, n_pad)
            res_all = np.concatenate([g[r, :counts[r]] for r in range(world)])
    else:
        res_all = res
    parity = parity_check(("orb", args.depth_noise, world) if default_workload else None, res_all) if len(res_all) else None
    if parity is not None and gather_on:
        parity["records"] = "gathered on rank 0: %d records of %d ranks" % (len(res_all), world)
        if list_entries is not None:
            parity["inlier_list_entries"] = list_entries
            if parity.get("checked") and list_entries != parity["oracle_aggregates"]["inliers"]:
                raise SystemExit("bench.py: the gathered inlier lists hold %d matches, the oracle counts %d"
                                 % (list_entries, parity["oracle_aggregates"]["inliers"]))
    rep_values = [sum(counts) * args.steps / e for e in rep_elapsed]

    if rank == 0:
        b_pair, b_ham, b_rsc = algorithmic_bytes(N, MAX_MATCHES)
        ms_per_step = elapsed / args.steps * 1e3
        # spans inside the timed region (two batches in flight: a span can exceed ms_per_step)
        ham_ovl = ham_ms / max(ham_launches, 1)
        rsc_ovl = rsc_ms / max(rsc_launches, 1)
        fin_ovl = fin_ms / max(fin_launches, 1)
        # serial stage times (one batch in flight): what `roofline` is computed from
        ham_ser = iso.get("serial_match_ms", 0.0)
        rsc_ser = iso.get("serial_ransac_ms", 0.0)
        repeats = {"values": [round(v, 2) for v in rep_values], "ms_per_step": [round(e / args.steps * 1e3, 4) for e in rep_elapsed],
                   "reported": "median", "spread_pct": round((max(rep_values) - min(rep_values)) / value * 100.0, 2)}
        timing = {
            "ms_per_step": round(ms_per_step, 4),
            "serial_ms_per_step": iso.get("serial_ms_per_step"),
            "serial_stage_ms": {"match": ham_ser, "sift_finish": iso.get("serial_sift_finish_ms"), "select_ransac": rsc_ser},
            "overlapped_stage_span_ms": {"match": round(ham_ovl, 4), "sift_finish": round(fin_ovl, 4) if fin_launches else None,
                                         "select_ransac": round(rsc_ovl, 4)},
            "note": "value / ms_per_step: pipelined timed region; serial_*: 5 extra steps with one batch in flight (roofline uses these)",
        }
        if sift:
            # configs[3]: the dense contraction.  FLOP per pair = 2 * Nq * Nt * 128 on the bf16 MFMA
            # (dense peak 2.5 PFLOP/s, MI355X_MICROARCH.md); reported for the MFMA kernel itself.
            flop = 2.0 * N * N * 128 * n_local
            tf = flop / (ham_ser * 1e-3) / 1e12 if ham_ser else 0.0
            out = {
                "metric": "frame-pairs matched+RANSAC/sec, 640x480 SIFT-1000 (128-d float, bf16 MFMA)",
                "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16 MFMA (exact u8 dot products) + f32/f64 (RANSAC)", "data": "synthetic",
                "config": {"workload": "configs[3]: synthetic SIFT 128-d float descriptors, %d kp, %d candidate "
                                       "pairs/frame, %d frames" % (N, args.pairs_per_frame, F),
                           "pairs_per_gpu_per_step": n_local, "edge_fraction": round(edge_frac, 4),
                           "mean_ransac_iterations": round(mean_iters, 2)},
                "roofline": {"bound": "mfma", "kernel": "sift_dot_top2", "achieved": round(tf, 3),
                             "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5),
                             "traffic": None, "flop_per_pair": 2.0 * N * N * 128,
                             "pairs_per_launch": n_local, "avg_launch_ms": ham_ser, "time_basis": "serial"},
                "repeats": repeats, "timing": timing,
            }
            print(json.dumps(out), flush=True)
            fe.close()
            if gather_on:
                dist.destroy_process_group()
            return
        dominant = "hamming_nn" if ham_ser >= rsc_ser else "select_ransac"
        dom_ms, dom_bytes = (ham_ser, b_ham) if dominant == "hamming_nn" else (rsc_ser, b_rsc)
        achieved = (dom_bytes * n_local) / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        achieved_pipe = (dom_bytes * n_local) / (ms_per_step * 1e-3) / 1e9
        pmc, pmc_src = load_pmc()
        mrf = match_roofline(N, n_local, ham_ser, fe.hamming_mode)
        roofline = {
            # the contract's block: ALGORITHMIC bytes of the dominant kernel / its launch time / HBM peak
            "bound": "hbm", "limiter": "valu_issue", "kernel": dominant, "achieved": round(achieved, 3),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "frac_pipelined": round(achieved_pipe / HBM_PEAK_GBS, 6),    # / ms_per_step of the timed region (batches overlap)
            "traffic": static_traffic(pmc, "orb" if args.depth_noise >= 0.005 else "ransac_heavy", dominant, n_local),
            "traffic_source": pmc_src,
            "algorithmic_bytes_per_pair": dom_bytes, "pairs_per_launch": n_local,
            "avg_launch_ms": round(dom_ms, 4), "time_basis": "serial, HIP events, this run",
            "step_ms_serial": iso.get("serial_ms_per_step"), "step_ms_pipelined": round(ms_per_step, 4),
            "pair_path_GBs": round(value / world * b_pair / 1e9, 3),
            # ---- every kernel family of the line, flat (filled in below as the sub-records run): fraction of ITS roofline and
            #      the serial launch time the fraction was formed with
            "match_frac": mrf["frac"] if mrf else None, "match_avg_launch_ms": mrf["avg_launch_ms"] if mrf else None,
            "match_peak_tflops": mrf["peak"] if mrf and mrf.get("bound") == "mfma" else None,
        }
        if plain_elapsed is not None:   # the headline on the library's default launch path (no hipGraph)
            roofline["default_path_value"] = round(total_pairs / plain_elapsed, 2)
            roofline["default_path_ms_per_step"] = round(plain_elapsed / args.steps * 1e3, 4)
        iss = issue_roofline(pmc, pmc_src, "orb" if args.depth_noise >= 0.005 else "ransac_heavy", N, n_local,
                             ham_ser, rsc_ser, fe.hamming_mode)
        if (iss.get("stages") or {}).get("select_ransac"):
            roofline["issue_frac"] = iss["stages"]["select_ransac"]["frac"]
        out = {
            "metric": "frame-pairs matched+RANSAC/sec, 640x480 ORB-1000",
            "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp4 MFMA, exact (match) + f32/f64 (RANSAC)" if fe.hamming_mode != 0 else
                     "u32 popcount (match) + f32/f64 (RANSAC)", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 640x480 RGB-D, ORB %d kp, %d candidate pairs/frame, %d frames"
                                   % (N, args.pairs_per_frame, F),
                       "pairs_per_gpu_per_step": n_local, "max_matches": MAX_MATCHES,
                       "ransac_iterations": 200, "parallelism": "pair-sharded x%d" % world,
                       "graph_capture": os.environ.get("RGBDFE_GRAPHS") != "0",
                       "depth_noise_sigma_over_z2": args.depth_noise,
                       "edge_fraction": round(edge_frac, 4), "mean_ransac_iterations": round(mean_iters, 2)},
            "roofline": roofline,
            "cpu_baseline": None,
            "match_roofline": mrf,
            "issue_roofline": iss,
            "repeats": repeats, "parity_check": parity,
            "timing": timing,
        }
        if world != 1 or args.no_cpu_baseline:
            del out["cpu_baseline"]
        if gather_on:
            per_step = gather_bytes_timed / max(gathers_timed, 1)
            out["gather"] = {"payload_option": args.gather,
                             "payload": "inlier stream: rgbdfe_inlier_header (104 B) + 4 B per inlier match" if inliers else
                                        ("rgbdfe_compact_result" if compact else "rgbdfe_match_result"),
                             "bytes_per_record": round(per_step / world / max(n_pad, 1), 1) if inliers else gat_bytes,
                             "bytes_per_step_per_rank": round(per_step),
                             "bytes_per_step_per_rank_by_payload": {
                                 "inliers": round(per_step) if inliers else None,
                                 "compact": world * n_pad * COMPACT_DTYPE.itemsize, "full": world * n_pad * rec_bytes},
                             "gathers_in_timed_regions": gathers_timed,
                             "collectives_per_step": 1,
                             "inlier_gather": {"list_capacity_entries": inl["cap"], "collectives_issued": inl["collectives"],
                                               "gathers_issued": state["gathers_all"], "regathers": inl["regathers"],
                                               "host_reads_before_a_collective": "the first (warm-up) gather only"}
                                              if inliers else None,
                             "backend": backend, "rccl_ranks": rccl_ranks, "rank_devices": rank_devices,
                             "distinct_devices": len({d.split(" ", 2)[2] for d in rank_devices}) if rank_devices else None,
                             "transport": "RCCL ncclAllGather via torch.distributed (nccl backend)" if backend == "nccl"
                                          else "host tensors (%s; test mode)" % backend}
        if world == 1 and not args.no_extras:
            fe.close()
            fe = None
            try:
                out["sift"] = sift_subrecord(seq, local_rank, default_workload and args.depth_noise == 0.01)
            except Exception as e:  # noqa: BLE001 -- a sub-record must not take the headline line down
                out["sift"] = {"error": repr(e)}
            try:
                out["detect"] = detect_subrecord(local_rank)
            except Exception as e:  # noqa: BLE001
                out["detect"] = {"error": repr(e)}
            try:
                out["sift_extract"] = sift_extract_subrecord(local_rank)
            except Exception as e:  # noqa: BLE001
                out["sift_extract"] = {"error": repr(e)}
            try:
                out["front_end"] = front_end_subrecord(local_rank)
            except Exception as e:  # noqa: BLE001
                out["front_end"] = {"error": repr(e)}
            try:
                out["host_io"] = host_io_subrecord(seq, pq, pt, local_rank, default_workload and args.depth_noise == 0.01)
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                out["host_io"] = {"error": repr(e)}
            try:
                out["loop_closure"] = loop_closure_subrecord(local_rank, args.depth_noise)
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                out["loop_closure"] = {"error": repr(e)}
            if args.depth_noise >= 0.005:
                try:
                    out["ransac_heavy"] = ransac_heavy_subrecord(local_rank, N, F, args.pairs_per_frame)
                except SystemExit:
                    raise
                except Exception as e:  # noqa: BLE001
                    out["ransac_heavy"] = {"error": repr(e)}
        # the sub-records' roofline figures, flat, into the head of the line (the tail may be cut by whoever stores it)
        def head(prefix, rec, fields):
            if isinstance(rec, dict) and "error" not in rec:
                for name, path in fields:
                    v = rec
                    for k in path:
                        v = v.get(k) if isinstance(v, dict) else None
                    roofline["%s_%s" % (prefix, name)] = v
        head("sift", out.get("sift"), (("frac", ("roofline", "frac")), ("avg_launch_ms", ("roofline", "avg_launch_ms")),
                                      ("pairs_per_s", ("value",))))
        for key, rec in (out.get("detect") or {}).items():
            head("detect_" + key.split("_")[0], rec, (("kernel_frac", ("roofline", "kernel_time_frac")),
                                                      ("kernel_us_per_frame", ("roofline", "kernel_us_per_frame")),
                                                      ("traffic", ("roofline", "traffic")), ("batch_fps", ("batch_api", "value"))))
        head("sift_extract", out.get("sift_extract"), (("frac", ("roofline", "frac")), ("kernel_frac", ("roofline", "kernel_time_frac")),
                                                       ("kernel_us_per_frame", ("roofline", "kernel_us_per_frame")),
                                                       ("batch_fps", ("batch_api", "value"))))
        head("heavy", out.get("ransac_heavy"), (("ms_per_step", ("ms_per_step",)), ("serial_stage_ms", ("serial_stage_ms", "select_ransac")),
                                                ("issue_frac", ("issue_roofline", "stages", "select_ransac", "frac")),
                                                ("traffic", ("roofline", "traffic"))))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq, pq, pt, SEED, 1e-4, args.cpu_seconds)
            ref = cpu_reference_code(seq, pq, pt, SEED, 1e-4, 5.0)
            if ref is not None:
                out["cpu_baseline_reference_code"] = ref
        print(json.dumps(out), flush=True)

    if fe is not None:
        fe.close()
    if world > 1 or force_gather:
        dist.destroy_process_group()


# Measured issue cost per wave-instruction per SIMD on this chip (tools/ubench/valu_rate.hip, profiles/r01_ubench): the
# classes the two ORB kernels are made of.  ns per wave-instruction per SIMD at 4 waves per SIMD, all 256 CUs busy.
ISSUE_NS = {"f32_or_simple_int": 1.10, "vop3_int": 1.90, "f64": 2.20, "xor_sgpr_plus_bcnt_pair": 3.34}
N_SIMD = 256 * 4


def issue_roofline(pmc, pmc_src, section, n_kp, n_pairs, ham_ms, rsc_ms, hamming_mode):
    """The instruction-issue roofline of the two ORB stages: wave-instructions per batch (counted by rocprofv3 PMC,
    SQ_INSTS_VALU and friends, in the committed profile named in `source` -- a static figure of the same workload, not
    collected in this run) x the measured issue cost of their instruction class / the stage time of THIS run.  frac =
    the share of the stage time the SIMDs need just to issue the stage's VALU instructions.  `section` picks the
    workload of the summary ("orb": depth noise 0.01 z^2, "ransac_heavy": 0.002 z^2)."""
    out = {"unit": "fraction of the stage time spent issuing its VALU instructions on %d SIMDs" % N_SIMD,
           "issue_ns_per_wave_instruction": ISSUE_NS, "source": None, "stages": {}}
    sec = pmc.get(section) if isinstance(pmc.get(section), dict) else (pmc if section == "orb" and "hamming" in pmc else None)
    if not sec:
        return out
    out["source"] = "%s [%s] (static: %s)" % (pmc_src, section, pmc.get("collected_with", "rocprofv3 --pmc"))
    for stage, ms in (("hamming", ham_ms), ("select_ransac", rsc_ms)):
        rec = sec.get(stage)
        if not rec or not ms:
            continue
        if stage == "hamming" and int(rec.get("hamming_mode", -1)) != int(hamming_mode):
            continue
        scale = n_pairs / float(rec.get("pairs_per_batch", n_pairs))
        insts = rec.get("valu_wave_instructions_per_batch", 0.0) * scale
        ns = rec.get("mean_issue_ns", ISSUE_NS["f32_or_simple_int"])
        t_issue_ms = insts * ns / N_SIMD * 1e-6
        out["stages"][stage] = {"valu_wave_instructions": round(insts), "mean_issue_ns": ns,
                                "issue_time_ms": round(t_issue_ms, 4), "stage_time_ms": round(ms, 4),
                                "frac": round(t_issue_ms / ms, 4),
                                "valu_busy_frac_pmc": rec.get("valu_busy_frac"),
                                "lds_bank_conflict_cycles_pmc": rec.get("lds_bank_conflict_cycles_per_batch"),
                                "hbm_bytes_per_batch_pmc": rec.get("hbm_bytes_per_launch")}
    return out


def static_traffic(pmc, section, dominant, n_pairs):
    """HBM bytes per launch of the dominant stage from the committed PMC profile (scaled to this run's pairs), or None."""
    try:
        sec = pmc[section] if section in pmc else pmc
        rec = sec["hamming" if dominant == "hamming_nn" else "select_ransac"]
        return round(rec["hbm_bytes_per_launch"] * n_pairs / float(rec["pairs_per_batch"]))
    except Exception:
        return None


def match_roofline(n_kp, n_pairs, ham_ms, hamming_mode):
    """The Hamming stage against the unit it actually runs on: the fp4 matrix cores (2 * N * N * 256 FLOP per pair,
    dense fp4 peak 10 PFLOP/s, MI355X_MICROARCH.md) or, for the popcount kernel, the integer VALU lanes
    (16 * N * (N - 1) lane-ops per pair: 8 xor + 8 bcnt per 256-bit compare)."""
    if not ham_ms:
        return None
    if hamming_mode != 0:
        tf = 2.0 * n_kp * n_kp * 256 * n_pairs / (ham_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "%s (v_mfma_f32_32x32x64_f8f6f4, fp4 x fp4)"
                                           % ("hamming_mfma_pipe_kernel" if hamming_mode == 3 else "hamming_mfma_kernel"),
                "achieved": round(tf, 2), "peak": 10000.0, "unit": "TFLOP/s", "frac": round(tf / 10000.0, 5),
                "flop_per_pair": 2.0 * n_kp * n_kp * 256, "avg_launch_ms": round(ham_ms, 4), "time_basis": "serial"}
    ops = 16.0 * n_kp * (n_kp - 1) * n_pairs / (ham_ms * 1e-3)
    return {"bound": "valu", "kernel": "hamming_nn_kernel (v_xor_b32 + v_bcnt_u32_b32)", "achieved": round(ops / 1e12, 3),
            "peak": round(VALU_PEAK_LANEOPS / 1e12, 3), "unit": "T lane-ops/s", "frac": round(ops / VALU_PEAK_LANEOPS, 4),
            "avg_launch_ms": round(ham_ms, 4), "time_basis": "serial"}


def sift_subrecord(seq, device, default=True):
    """configs[3] on a bounded workload (100 frames x 20 candidates = 2000 pairs per step): pairs/s and the MFMA
    fraction of the dot-product kernel, serial stage times."""
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import KERNEL_RANSAC, KERNEL_SIFT_DOT, KERNEL_SIFT_FINISH, RESULT_DTYPE
    from rgbdslam_v2_amd.frontend import FrontEnd
    import torch
    F = SIFT_FRAMES
    N = seq["desc"].shape[1]
    sd, pq, pt = sift_workload(seq)
    fe = FrontEnd(device_id=device, max_nodes=F, max_keypoints=((N + 63) // 64) * 64, max_pairs_per_batch=len(pq), seed=SEED)
    for f in range(F):
        fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
    bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for k in range(2):
        fe.submit_sift_pair_list(pq, pt, bufs[k % 4].data_ptr())
    fe.synchronize()
    steps = 8
    t0 = time.perf_counter()
    for k in range(steps):
        fe.submit_sift_pair_list(pq, pt, bufs[k % 4].data_ptr())
    fe.synchronize()
    dt = time.perf_counter() - t0
    fe.set_profiling(True)
    fe.reset_kernel_time()
    for k in range(3):
        fe.wait_ticket(fe.submit_sift_pair_list(pq, pt, bufs[0].data_ptr()), None)
    fe.synchronize()
    fe.set_profiling(False)
    dot_ms, nl, _ = fe.kernel_time(KERNEL_SIFT_DOT)
    fin_ms, _, _ = fe.kernel_time(KERNEL_SIFT_FINISH)
    rsc_ms, _, _ = fe.kernel_time(KERNEL_RANSAC)
    dot_ms, fin_ms, rsc_ms = dot_ms / max(nl, 1), fin_ms / max(nl, 1), rsc_ms / max(nl, 1)
    res = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[: len(pq)]
    fe.close()
    parity = parity_check(("sift", "0.01") if default else None, res)
    tf = 2.0 * N * N * 128 * len(pq) / (dot_ms * 1e-3) / 1e12 if dot_ms else 0.0
    pmc, pmc_src = load_pmc()
    traffic = None
    try:
        rec = pmc["sift"]["sift_dot"]
        traffic = round(rec["hbm_bytes_per_launch"] * len(pq) / float(pmc["sift"]["pairs_per_batch"]))
    except Exception:
        pass
    return {"metric": "frame-pairs matched+RANSAC/sec, SIFT 128-d float, %d kp (configs[3])" % N,
            "value": round(steps * len(pq) / dt, 2), "unit": "frame-pairs/s", "pairs_per_step": len(pq),
            "ms_per_step": round(dt / steps * 1e3, 4),
            "roofline": {"bound": "mfma", "kernel": "sift dot-product + top-2", "achieved": round(tf, 3), "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5), "flop_per_pair": 2.0 * N * N * 128,
                         "avg_launch_ms": round(dot_ms, 4), "time_basis": "serial", "traffic": traffic,
                         "traffic_source": "%s [sift]" % pmc_src,
                         "executed_over_algorithmic_flop": round(float(pmc["sift"]["sift_dot"]["mfma_instructions_per_batch"]) * 32768.0 /
                                                                 (2.0 * N * N * 128 * float(pmc["sift"]["pairs_per_batch"])), 3)
                         if isinstance((pmc.get("sift") or {}).get("sift_dot"), dict) and pmc["sift"]["sift_dot"].get("mfma_instructions_per_batch") else None},
            "serial_stage_ms": {"dot_top2": round(dot_ms, 4), "finish": round(fin_ms, 4), "select_ransac": round(rsc_ms, 4)},
            "parity_check": parity}


def host_io_subrecord(seq, pq, pt, device, default=True):
    """The headline workload through the synchronous host-buffer entry point (the timed region of the
    headline leaves its results in HBM): rgbdfe_match_pair_list -- pair ids in, the 1744-byte result records of all 4000
    pairs back in the caller's memory over PCIe -- and the cost of making a node resident from host arrays."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    F, N = seq["desc"].shape[0], seq["desc"].shape[1]
    fe = FrontEnd(device_id=device, max_nodes=F, max_keypoints=((N + 63) // 64) * 64, max_pairs_per_batch=len(pq), seed=SEED)
    try:
        t0 = time.perf_counter()
        for f in range(F):
            fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
        t_up = (time.perf_counter() - t0) / F
        fe.match_pair_list(pq, pt)
        per = []
        for _ in range(REPEATS + 2):
            t0 = time.perf_counter()
            out = fe.match_pair_list(pq, pt)
            per.append(time.perf_counter() - t0)
        # the same records through the asynchronous host-output jobs (rgbdfe_submit_pair_list_host / rgbdfe_wait_host):
        # batch k's download runs behind batch k while batch k+1 computes; two result buffers in caller memory take turns
        from rgbdslam_v2_amd._lib import INLIER_HEADER_DTYPE, RESULT_DTYPE, RGBDFE_MAX_MATCHES

        def pipelined(bufs, inliers, steps=12):
            tk = [fe.submit_pair_list_host(pq, pt, bufs[0], inliers=inliers)]
            fe.wait_host(tk.pop())                                   # warm: stages allocated
            nb = 0
            t0 = time.perf_counter()
            tk.append(fe.submit_pair_list_host(pq, pt, bufs[0], inliers=inliers))
            for k in range(1, steps):
                tk.append(fe.submit_pair_list_host(pq, pt, bufs[k % 2], inliers=inliers))
                nb = fe.wait_host(tk.pop(0))
            nb = fe.wait_host(tk.pop(0))
            return (time.perf_counter() - t0) / steps, nb
        recs = [np.zeros(len(pq), RESULT_DTYPE) for _ in range(2)]
        dt_rec, _ = pipelined(recs, False)
        async_parity = parity_check(("orb", "0.01", 1) if default else None, recs[1])
        streams = [np.zeros(len(pq) * (INLIER_HEADER_DTYPE.itemsize + 4 * RGBDFE_MAX_MATCHES), np.uint8) for _ in range(2)]
        dt_inl, nb_inl = pipelined(streams, True)
        # ... and into PINNED caller buffers (hipHostMalloc / rgbdfe_host_register): the download itself writes them
        import torch
        pins = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        dt_pin, _ = pipelined([p_.numpy().view(RESULT_DTYPE) for p_ in pins], False)
        pinned_parity = parity_check(("orb", "0.01", 1) if default else None, pins[1].numpy().view(RESULT_DTYPE))
    finally:
        fe.close()
    per.sort()
    dt = per[len(per) // 2]
    return {"metric": "frame-pairs matched+RANSAC/sec, results returned to host memory (rgbdfe_match_pair_list)",
            "value": round(len(pq) / dt, 1), "unit": "frame-pairs/s", "pairs_per_call": int(len(pq)),
            "ms_per_call": round(dt * 1e3, 4), "ms_per_call_repeats": [round(v * 1e3, 4) for v in per],
            "result_bytes_per_call": int(out.nbytes), "node_upload_us": round(t_up * 1e6, 1),
            "note": "host wall clock, PCIe both ways and the library's pinned staging included; never the headline `value`",
            "parity_check": parity_check(("orb", "0.01", 1) if default else None, out),
            "pipelined": {"entry_points": "rgbdfe_submit_pair_list_host / rgbdfe_wait_host, two jobs in flight, pageable caller buffers",
                          "records": {"value": round(len(pq) / dt_rec, 1), "ms_per_batch": round(dt_rec * 1e3, 4),
                                      "bytes_per_pair": RESULT_DTYPE.itemsize, "parity_check": async_parity},
                          "records_pinned_buffers": {"value": round(len(pq) / dt_pin, 1), "ms_per_batch": round(dt_pin * 1e3, 4),
                                                     "parity_check": pinned_parity},
                          "inlier_stream": {"value": round(len(pq) / dt_inl, 1), "ms_per_batch": round(dt_inl * 1e3, 4),
                                            "bytes_per_pair": round(nb_inl / len(pq), 1)}}}


def loop_closure_subrecord(device, depth_noise):
    """The reject path (every pair of the headline workload is a true edge): an all-pairs loop-closure search
    over frames of unrelated places -- 180 frames in 18 places of 10, every frame against every earlier one (16 110
    pairs per step, ~5 % true edges); the other pairs are chance matches that the min_matches gate or RANSAC rejects."""
    import torch
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import RESULT_DTYPE
    from rgbdslam_v2_amd.frontend import FrontEnd
    desc, xyz, pq, pt = synth.loop_closure_places(depth_noise=depth_noise)
    F = len(desc)
    fe = FrontEnd(device_id=device, max_nodes=F, max_keypoints=1024, max_pairs_per_batch=len(pq))
    for f in range(F):
        fe.upload_node(f, desc[f], xyz[f])
    bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:%d" % device) for _ in range(4)]

    def run(steps):
        tk = []
        for st in range(steps):
            if len(tk) >= 2:
                fe.wait_ticket(tk.pop(0), None)
            tk.append(fe.submit_pair_list(pq, pt, bufs[st % 4].data_ptr()))
        for t in tk:
            fe.wait_ticket(t, None)
        fe.synchronize()

    run(2)
    steps = 6
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    res = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[: len(pq)]
    same_place = (pq // 10) == (pt // 10)
    edges = res["id1"] >= 0
    fe.close()
    parity = parity_check(("loop_closure", depth_noise), res)
    return {"metric": "frame-pairs matched+RANSAC/sec, all-pairs loop-closure search, ORB-1000", "parity_check": parity,
            "value": round(len(pq) * steps / dt, 1), "unit": "frame-pairs/s", "pairs_per_step": int(len(pq)),
            "ms_per_step": round(dt / steps * 1e3, 3), "frames": F,
            "pairs_reaching_ransac": round(float((res["real_iterations"] > 0).mean()), 4),
            "edge_fraction": round(float(edges.mean()), 4),
            "true_pairs_found": round(float(edges[same_place].mean()), 4),
            "false_edges": int(edges[~same_place].sum())}


def ransac_heavy_subrecord(device, n_kp, n_frames, pairs_per_frame):
    """The slower regime as a driver-visible figure: configs[1] with round 1's depth noise 0.002 z^2 -- 56 %
    of the hypotheses are valid instead of 13 %, so the refinement rounds dominate the RANSAC stage.  Same 4000 pairs per
    step, pipelined timing + serial stage times + its own issue roofline section; results checked against the oracle."""
    import torch
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import KERNEL_HAMMING, KERNEL_RANSAC, RESULT_DTYPE
    from rgbdslam_v2_amd.frontend import FrontEnd
    noise = synth.DEPTH_NOISE_R1
    seq = synth.make_sequence(n_frames=n_frames, n_kp=n_kp, seed=SEED, depth_noise=noise)
    pq, pt = synth.candidate_pairs(n_frames, per_frame=pairs_per_frame, seed=SEED)
    fe = FrontEnd(device_id=device, max_nodes=n_frames, max_keypoints=((n_kp + 63) // 64) * 64, max_pairs_per_batch=len(pq), seed=SEED)
    for f in range(n_frames):
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
    bufs = [torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:%d" % device) for _ in range(4)]

    def run(steps):
        for st in range(steps):
            fe.submit_pair_list(pq, pt, bufs[st % 4].data_ptr())
        fe.synchronize()

    run(2)
    steps = 6
    vals = []
    for _ in range(REPEATS):
        t0 = time.perf_counter()
        run(steps)
        vals.append(len(pq) * steps / (time.perf_counter() - t0))
    value = sorted(vals)[len(vals) // 2]
    fe.set_profiling(True)
    fe.reset_kernel_time()
    for _ in range(3):
        fe.wait_ticket(fe.submit_pair_list(pq, pt, bufs[0].data_ptr()), None)
    fe.synchronize()
    fe.set_profiling(False)
    ham_ms, nl, _ = fe.kernel_time(KERNEL_HAMMING)
    rsc_ms, _, _ = fe.kernel_time(KERNEL_RANSAC)
    ham_ms, rsc_ms = ham_ms / max(nl, 1), rsc_ms / max(nl, 1)
    res = np.frombuffer(bufs[0].cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[: len(pq)]
    mode = fe.hamming_mode
    fe.close()
    default = n_kp == N_KP and n_frames == N_FRAMES and pairs_per_frame == PAIRS_PER_FRAME
    parity = parity_check(("orb", noise, 1) if default else None, res)
    pmc, pmc_src = load_pmc()
    _, _, b_rsc = algorithmic_bytes(n_kp, MAX_MATCHES)
    gbs = b_rsc * len(pq) / (rsc_ms * 1e-3) / 1e9 if rsc_ms else 0.0
    return {"metric": "frame-pairs matched+RANSAC/sec, 640x480 ORB-%d, depth noise 0.002 z^2 (round 1's regime)" % n_kp,
            "value": round(value, 1), "unit": "frame-pairs/s", "pairs_per_step": int(len(pq)),
            "ms_per_step": round(len(pq) / value * 1e3, 4),
            "repeats": [round(v, 1) for v in vals],
            "edge_fraction": round(float((res["id1"] >= 0).mean()), 4),
            "mean_ransac_iterations": round(float(res["real_iterations"].mean()), 2),
            "mean_valid_iterations": round(float(res["valid_iterations"].mean()), 2),
            "serial_stage_ms": {"match": round(ham_ms, 4), "select_ransac": round(rsc_ms, 4)},
            "roofline": {"bound": "hbm", "limiter": "valu_issue", "kernel": "select_ransac", "achieved": round(gbs, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
                         "traffic": static_traffic(pmc, "ransac_heavy", "select_ransac", len(pq)),
                         "traffic_source": "%s [ransac_heavy]" % pmc_src, "time_basis": "serial"},
            "issue_roofline": issue_roofline(pmc, pmc_src, "ransac_heavy", n_kp, len(pq), ham_ms, rsc_ms, mode),
            "parity_check": parity}


def sift_extract_subrecord(device):
    """SIFT extraction (SiftGPUWrapper::detect): frames/s of rgbdfe_sift_detect on synthetic
    640x480 frames, host buffers in and out; algorithmic bytes per frame stated in DESIGN.md 4.11."""
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    seq, run = sift_extract_workload()
    fe = FrontEnd(device_id=device, max_nodes=4, max_keypoints=64, max_pairs_per_batch=8)
    try:
        for f in range(2):
            fe.sift_detect(seq["gray"][f], None)
        reps, tot = 3, 0
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in range(len(seq["gray"])):
                kp, _ = fe.sift_detect(seq["gray"][f], None)
                tot += len(kp)
        dt = time.perf_counter() - t0
        # the same frames as a run through the batch entry point (8 frames per launch chain), 32 frames per call
        for _ in range(2):
            fe.sift_detect_batch(run, copy=False)
        per_frame = []
        for _ in range(REPEATS + 2):
            t0 = time.perf_counter()
            fe.sift_detect_batch(run, copy=False)          # output arrays reused, as an integration's buffers are
            per_frame.append((time.perf_counter() - t0) / len(run))
        per_frame.sort()
        dt_batch = per_frame[len(per_frame) // 2]
        # the outputs of the batch entry point over the generated frames against the compiled reference's (constants)
        got = sift_features_checksum(fe.sift_detect_batch(list(seq["gray"])))
    finally:
        fe.close()
    parity = check_against(expected("sift_extract", "640x480"), got, "the sift_extract sub-record",
                           "SiftGPU's own kernels + host code compiled on a CPU emulation (oracle/_ref/libref_siftgpu.so) over the "
                           "same frames (%s); planes / candidates / descriptors per feature in tests/test_gpu_sift_extract.py" % EXPECTED_FILE,
                           approx=("descriptor_abs_sum",))
    frames = reps * len(seq["gray"])
    w, h = seq["gray"][0].shape[1], seq["gray"][0].shape[0]
    b_frame = sift_extract_bytes(w, h, tot / frames)
    gbs = frames * b_frame / dt / 1e9
    gbs_batch = b_frame / dt_batch / 1e9
    pmc, pmc_src = load_pmc()
    prof = (pmc.get("sift_extract") or {}).get("%dx%d" % (w, h)) or {}
    k_ns = prof.get("kernel_ns_per_frame")
    return {"metric": "frames SIFT-detected+described/sec, %dx%d (SiftGPUWrapper::detect)" % (w, h),
            "value": round(frames / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt / frames * 1e3, 4),
            "mean_keypoints": round(tot / frames, 1),
            "batch_api": {"value": round(1.0 / dt_batch, 2), "unit": "frames/s", "ms_per_frame": round(dt_batch * 1e3, 4),
                          "frames_per_call": 32, "ms_per_frame_repeats": [round(v * 1e3, 4) for v in per_frame],
                          "note": "rgbdfe_sift_detect_batch: 8 frames per launch chain, the outputs of single calls"},
            "roofline": {"bound": "hbm", "achieved": round(gbs_batch, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs_batch / HBM_PEAK_GBS, 6), "frac_single_calls": round(gbs / HBM_PEAK_GBS, 6),
                         "algorithmic_bytes_per_frame": b_frame,
                         "time_basis": "host wall clock per frame, batch entry point",
                         "traffic": prof.get("hbm_bytes_per_frame"),
                         "kernel_time_frac": round(b_frame / (k_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 6) if k_ns else None,
                         "kernel_us_per_frame": round(k_ns / 1e3, 2) if k_ns else None,
                         "traffic_source": "%s [sift_extract]" % pmc_src},
            "parity_check": parity}


def sift_extract_bytes(w, h, n_kp):
    """Algorithmic bytes of one frame of SIFT extraction.  "-fo -1": the first octave is the 2x up-sampled
    image, 4 W H pixels; octave o has (4 W H) / 4^o pixels (sum <= 4/3) and "-d 5" gives 8 Gaussian levels per octave,
    f32.  Every level is written once, read once by the filter that produces the next level and once by the keypoint
    scan (the DoG planes are differences formed in registers, never stored): 24 plane passes x 4 B = 96 B per pixel; the
    extremum flags add one byte written + one read for each of the 5 scanned levels.  Per keypoint: 24 B candidate,
    16 B key, 512 B descriptor out (their gradient windows re-read planes already counted)."""
    px = 4.0 * w * h * (4.0 / 3.0)
    return w * h + px * (24 * 4 + 5 * 2) + n_kp * (24 + 16 + 512)


def front_end_subrecord(device):
    """Level C: a recorded sequence through the whole front end, the way an offline run (OpenNIListener reading a bag file)
    would drive it -- rgbdfe_detect_describe_batch_nodes over a stretch of frames (features to the host, nodes resident),
    rgbdfe_match_pair_list for every new node against its 20 predecessors -- host buffers in and out."""
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    n_base, n_run, cand, n_kp = FRONT_END["n_base"], FRONT_END["n_run"], FRONT_END["cand"], FRONT_END["n_kp"]
    _, _, grays, masks, depths, K = detect_workload(640, 480, n_base, n_run)
    pq, pt = front_end_pairs(n_run, cand)
    fe = FrontEnd(device_id=device, max_nodes=n_run, max_keypoints=1024, max_pairs_per_batch=len(pq), seed=SEED)
    per, parts, edges = [], None, 0
    ids = np.arange(n_run, dtype=np.int32)
    try:
        for rep in range(REPEATS + 1):
            for f in range(n_run):
                if rep:
                    fe.release_node(f)
            # every pass starts from createDetector's state (the per-cell thresholds adapt over a run): the passes are the same
            # work, and their outputs the ones the oracle constants were made from
            fe.detector_configure(max_keypoints=n_kp)
            t0 = time.perf_counter()
            feats = fe.detect_describe_batch(grays, masks, depths, *K, node_ids=ids, copy=False)   # features to the host (reused output arrays) AND resident nodes
            t1 = time.perf_counter()
            res = fe.match_pair_list(pq, pt)
            t3 = time.perf_counter()
            if rep:                                  # the first pass warms buffers and threads up
                per.append((t3 - t0) / n_run)
                parts = ((t1 - t0) / n_run, (t3 - t1) / n_run)
            edges = int((res["id1"] >= 0).sum())
    finally:
        fe.close()
    got = dict(features_checksum(feats), **pair_aggregates(res))
    parity = check_against(expected("front_end", "640x480_orb1000"), got, "the front_end sub-record",
                           "oracle/orb_oracle.c (detect + describe + projectTo3D, frame after frame) and oracle/liboracle.so (the "
                           "pairs over the oracle's features) on the same frames (%s)" % EXPECTED_FILE)
    per.sort()
    dt = per[len(per) // 2]
    return {"metric": "frames through detect + describe + node upload + 20 candidate pairs each, per second (640x480, ORB-1000)",
            "value": round(1.0 / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt * 1e3, 4), "frames_per_run": n_run,
            "pairs_per_run": int(len(pq)), "edges_found": edges,
            "ms_per_frame_parts_last_run": {"detect_describe_batch_nodes": round(parts[0] * 1e3, 4),
                                            "match_pair_list": round(parts[1] * 1e3, 4)},
            "note": "host wall clock; rgbdfe_detect_describe_batch_nodes returns the features to the host and leaves them resident "
                    "as nodes (device-to-device), rgbdfe_match_pair_list returns the result records; median of %d runs" % REPEATS,
            "parity_check": parity}


def detect_subrecord(device):
    """Level B: rgbdfe_detect_describe on synthetic frames, host buffers in and out (PCIe and the
    adjuster's round trips included).  B_frame = 13.4 * W * H + 64 * N algorithmic bytes per frame."""
    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd.frontend import FrontEnd
    out = {}
    # single calls over the generated frames; the batch entry point over a run of 16 / 8 super-frames of 7 (a recorded sequence:
    # the generated frames forth and back) -- its pipeline is three super-frames deep, a short run would time fill and drain
    for (w, h, n_kp, n_base, n_run) in DETECT_WORKLOADS:
        seq, masks, grays, mks, depths, K = detect_workload(w, h, n_base, n_run)
        fe = FrontEnd(device_id=device, max_nodes=4, max_keypoints=((n_kp + 63) // 64) * 64, max_pairs_per_batch=8)
        fe.detector_configure(max_keypoints=n_kp)
        for f in range(min(3, n_base)):
            fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], seq["fx"], seq["fy"], seq["cx"], seq["cy"])
        tot = 0
        t0 = time.perf_counter()
        for f in range(n_base):
            kp, _, _ = fe.detect_describe(seq["gray"][f], masks[f], seq["depth"][f], seq["fx"], seq["fy"],
                                          seq["cx"], seq["cy"])
            tot += len(kp)
        dt = (time.perf_counter() - t0) / n_base
        for _ in range(2):                      # the first calls of a run length pay page faults and thread wake-ups
            fe.detect_describe_batch(grays, mks, depths, *K, copy=False)
        per_frame = []
        for _ in range(REPEATS + 2):
            t0 = time.perf_counter()
            fe.detect_describe_batch(grays, mks, depths, *K, copy=False)   # output arrays reused, as an integration's buffers are
            per_frame.append((time.perf_counter() - t0) / n_run)
        per_frame.sort()
        dt_batch = per_frame[len(per_frame) // 2]
        # one more run from createDetector's state (the thresholds adapt from call to call): the outputs the oracle constants
        # were made from
        fe.detector_configure(max_keypoints=n_kp)
        got = features_checksum(fe.detect_describe_batch(grays, mks, depths, *K))
        fe.close()
        key = "%dx%d_orb%d" % (w, h, n_kp)
        parity = check_against(expected("detect", key), got, "the detect sub-record %s" % key,
                               "oracle/orb_oracle.c (detect + describe + projectTo3D, frame after frame from a fresh detector) on the "
                               "same frames (%s); every field per frame in tests/test_gpu_orb.py" % EXPECTED_FILE)
        b_frame = 13.4 * w * h + 64 * n_kp
        gbs = b_frame / dt / 1e9
        gbs_batch = b_frame / dt_batch / 1e9
        pmc, pmc_src = load_pmc()
        prof = (pmc.get("detect") or {}).get(key) or {}
        k_ns = prof.get("kernel_ns_per_frame")
        out[key] = {
            "value": round(1.0 / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt * 1e3, 4),
            "batch_api": {"value": round(1.0 / dt_batch, 2), "unit": "frames/s",
                          "ms_per_frame": round(dt_batch * 1e3, 4), "frames_per_call": n_run,
                          "ms_per_frame_repeats": [round(v * 1e3, 4) for v in per_frame],
                          "note": "rgbdfe_detect_describe_batch, run of %d frames into reused output arrays; median of %d"
                                  % (n_run, REPEATS + 2)},
            "mean_keypoints": round(tot / n_base, 1),
            "roofline": {"bound": "hbm", "achieved": round(gbs_batch, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs_batch / HBM_PEAK_GBS, 6), "frac_single_calls": round(gbs / HBM_PEAK_GBS, 6),
                         "algorithmic_bytes_per_frame": b_frame,
                         "time_basis": "host wall clock per frame, batch entry point",
                         "traffic": prof.get("hbm_bytes_per_frame"),
                         "kernel_time_frac": round(b_frame / (k_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 6) if k_ns else None,
                         "kernel_us_per_frame": round(k_ns / 1e3, 2) if k_ns else None,
                         "traffic_source": "%s [detect]" % pmc_src},
            "parity_check": parity}
    return out


def usable_cpus(hardware_threads):
    """Threads worth running: the hardware threads this process may use, capped by the container's CPU quota
    (cgroup v2 cpu.max / v1 cfs quota) -- oversubscribing a 16-CPU quota with 256 threads more than halves the rate."""
    n = hardware_threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(seq, pq, pt, seed, depth_cov, budget_s):
    """The oracle (CPU restatement of the reference pair path, kind='port') timed pair-parallel
    on this host's cores over a bounded sample of the same pair list."""
    from oracle import pyoracle as po
    prm = po.default_params(seed=seed, depth_cov=depth_cov)
    cores = usable_cpus(po.num_cores())
    descs, xyzs = list(seq["desc"]), list(seq["xyz1"])
    ids = np.arange(len(descs))
    probe = min(len(pq), max(2 * cores, 16))
    t0 = time.perf_counter()
    po.match_pairs_mt(descs, xyzs, ids, pq[:probe], pt[:probe], prm, cores)
    dt = time.perf_counter() - t0
    n = int(min(len(pq), max(probe, budget_s / max(dt / probe, 1e-6))))
    sel = np.linspace(0, len(pq) - 1, n).astype(np.int64)
    t0 = time.perf_counter()
    po.match_pairs_mt(descs, xyzs, ids, pq[sel], pt[sel], prm, cores)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d pairs of one step, oracle/liboracle.so, OpenMP pair-parallel, %d threads "
                      "(= the CPUs this container may use: %d hardware threads, cgroup quota applied)"
                      % (n, len(pq), cores, po.num_cores())}


def cpu_reference_code(seq, pq, pt, seed, depth_cov, budget_s):
    """The reference's OWN pair op (Node::matchNodePair and everything below it, compiled from the reference
    sources into oracle/_ref/libref_ransac.so with Eigen / PCL stand-ins) timed on one host
    thread over a bounded sample.  Reported next to cpu_baseline; None when the prebuilt pin is absent."""
    from oracle import pyoracle as po
    if po.ref_ransac_lib() is None:
        return None
    prm = po.default_params(seed=seed, depth_cov=depth_cov)
    descs, xyzs = seq["desc"], seq["xyz1"]
    sel = np.linspace(0, len(pq) - 1, min(len(pq), 400)).astype(np.int64)
    n = 0
    t0 = time.perf_counter()
    for k in sel:
        q, t = int(pq[k]), int(pt[k])
        po.ref_match_node_pair(descs[q], xyzs[q], q, descs[t], xyzs[t], t, prm)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frame-pairs/s", "cores": 1, "kind": "reference",
            "sample": "%d pairs of one step, the reference's matchNodePair compiled from its sources "
                      "(oracle/_ref/libref_ransac.so, third-party arithmetic from stand-ins), one thread" % n}


if __name__ == "__main__":
    main()
