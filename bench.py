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

Rank 0 prints ONE JSON line.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from rgbdslam_v2_amd import synth
    from rgbdslam_v2_amd._lib import (KERNEL_HAMMING, KERNEL_RANSAC, KERNEL_SIFT_DOT,
                                      KERNEL_SIFT_FINISH, RESULT_DTYPE)
    from rgbdslam_v2_amd.frontend import FrontEnd
    from rgbdslam_v2_amd.dist import shard_pairs

    F, N = args.frames, args.kp
    seq = synth.make_sequence(n_frames=F, n_kp=N, seed=SEED)
    # global pair list: per frame 20*world candidates (weak scaling), sharded round-robin
    per_frame = min(args.pairs_per_frame * world, F - 1)
    pq_all, pt_all = synth.candidate_pairs(F, per_frame=per_frame, seed=SEED)
    pq, pt = shard_pairs(pq_all, pt_all, rank, world)
    n_local = len(pq)
    counts = [n_local]
    if world > 1:
        t_cnt = torch.tensor([n_local], device="cuda")
        all_cnt = [torch.zeros_like(t_cnt) for _ in range(world)]
        dist.all_gather(all_cnt, t_cnt)
        counts = [int(c.item()) for c in all_cnt]
    n_pad = max(counts)

    fe = FrontEnd(device_id=local_rank, max_nodes=F, max_keypoints=((N + 63) // 64) * 64,
                  max_pairs_per_batch=max(n_pad, 1), seed=SEED)
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
    # Steps are pipelined: step k is submitted to one of the context's internal streams while
    # step k-1 still runs (its RANSAC tail overlaps step k's Hamming kernel).  A ring of result
    # buffers keeps every step's output alive until its all-gather has consumed it.
    NBUF = 4
    d_local = [torch.zeros(n_pad * rec_bytes, dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
    d_all = torch.zeros(world * n_pad * rec_bytes, dtype=torch.uint8, device="cuda") if world > 1 else None
    consumed = [None] * NBUF
    stream = torch.cuda.current_stream().cuda_stream
    state = {"k": 0}

    def step():
        b = state["k"] % NBUF
        state["k"] += 1
        if consumed[b] is not None:
            consumed[b].synchronize()  # the all-gather that read this buffer (4 steps ago) is done
        if sift:
            ticket = fe.submit_sift_pair_list(pq, pt, d_local[b].data_ptr())
        else:
            ticket = fe.submit_pair_list(pq, pt, d_local[b].data_ptr())
        if world > 1:
            fe.wait_ticket(ticket, stream)  # torch's stream waits for this batch only
            dist.all_gather_into_tensor(d_all, d_local[b])
            consumed[b] = torch.cuda.Event()
            consumed[b].record()

    for _ in range(args.warmup):
        step()
    fe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fe.synchronize()
    fe.set_profiling(True)
    fe.reset_kernel_time()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fe.synchronize()            # every internal stream of the context
    torch.cuda.synchronize()    # device-wide
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        te = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    fe.set_profiling(False)
    # Official per-kernel HIP-event times: measured inside the timed region (batches overlap there).
    k_match = KERNEL_SIFT_DOT if sift else KERNEL_HAMMING
    ham_ms, ham_launches, ham_pairs = fe.kernel_time(k_match)
    rsc_ms, rsc_launches, rsc_pairs = fe.kernel_time(KERNEL_RANSAC)
    fin_ms, fin_launches, _ = fe.kernel_time(KERNEL_SIFT_FINISH)
    # Extra, clearly separate pass: three NON-overlapped steps (one batch in flight at a time), so that
    # per-kernel durations are not inflated by the neighbouring batch sharing the CUs.  Reported as
    # roofline["isolated_*_ms"] next to the official numbers.
    iso = {}
    fe.reset_kernel_time()
    fe.set_profiling(True)
    for _ in range(3):
        if sift:
            tk = fe.submit_sift_pair_list(pq, pt, d_local[0].data_ptr())
        else:
            tk = fe.submit_pair_list(pq, pt, d_local[0].data_ptr())
        fe.wait_ticket(tk, None)
    fe.synchronize()
    fe.set_profiling(False)
    for nme, kk in (("match", k_match), ("ransac", KERNEL_RANSAC), ("sift_finish", KERNEL_SIFT_FINISH)):
        ms, nl, _ = fe.kernel_time(kk)
        if nl:
            iso["isolated_%s_ms" % nme] = round(ms / nl, 4)

    total_pairs = sum(counts) * args.steps
    value = total_pairs / elapsed

    # sanity: results of the last step are real (edges found)
    last = d_local[(state["k"] - 1) % NBUF]
    res = np.frombuffer(last.cpu().numpy().tobytes(), dtype=RESULT_DTYPE)[:n_local]
    edge_frac = float((res["id1"] >= 0).mean()) if n_local else 0.0
    mean_iters = float(res["real_iterations"].mean()) if n_local else 0.0

    if rank == 0:
        b_pair, b_ham, b_rsc = algorithmic_bytes(N, MAX_MATCHES)
        dominant = "hamming_nn" if ham_ms >= rsc_ms else "select_ransac"
        dom_ms, dom_launches, dom_bytes = ((ham_ms, ham_launches, b_ham) if dominant == "hamming_nn"
                                           else (rsc_ms, rsc_launches, b_rsc))
        avg_launch_ms = dom_ms / max(dom_launches, 1)
        achieved = (dom_bytes * n_local) / (avg_launch_ms * 1e-3) / 1e9 if dom_launches else 0.0
        ham_avg_ms = ham_ms / max(ham_launches, 1)
        valu_achieved = (16.0 * N * (N - 1) * n_local) / (ham_avg_ms * 1e-3) if ham_launches else 0.0
        traffic = valu_busy = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc_path) and args.ransac_path != "one_wave":  # collected on the default schedule
            try:
                pmc = json.load(open(pmc_path)).get(dominant, {})
                traffic = pmc.get("hbm_bytes_per_launch")
                valu_busy = pmc.get("valu_busy_frac")  # SQ_ACTIVE_INST_VALU over the SIMD time of the launch
            except Exception:
                traffic = valu_busy = None
        if sift:
            # configs[3]: the dense contraction.  FLOP per pair = 2 * Nq * Nt * 128 on the bf16 MFMA
            # (dense peak 2.5 PFLOP/s, MI355X_MICROARCH.md); reported for the MFMA kernel itself.
            flop = 2.0 * N * N * 128 * n_local
            tf = flop / (ham_avg_ms * 1e-3) / 1e12 if ham_launches else 0.0
            out = {
                "metric": "frame-pairs matched+RANSAC/sec, 640x480 SIFT-1000 (128-d float, bf16 MFMA)",
                "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16 MFMA (exact u8 dot products) + f32/f64 (RANSAC)", "data": "synthetic",
                "config": {"workload": "configs[3]: synthetic SIFT 128-d float descriptors, %d kp, %d candidate "
                                       "pairs/frame, %d frames" % (N, args.pairs_per_frame, F),
                           "pairs_per_gpu_per_step": n_local, "edge_fraction": round(edge_frac, 4),
                           "mean_ransac_iterations": round(mean_iters, 2)},
                "roofline": {"bound": "mfma", "kernel": "sift_dot_top2", "achieved": round(tf, 3),
                             "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5),
                             "traffic": None, "flop_per_pair": 2.0 * N * N * 128,
                             "pairs_per_launch": n_local, "avg_launch_ms": round(ham_avg_ms, 4),
                             "finish_ms_per_launch": round(fin_ms / max(fin_launches, 1), 4),
                             "ransac_ms_per_launch": round(rsc_ms / max(rsc_launches, 1), 4), **iso},
            }
            print(json.dumps(out), flush=True)
            fe.close()
            if world > 1:
                dist.destroy_process_group()
            return
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic,
            "valu_busy_frac_pmc": None if valu_busy is None else round(valu_busy, 3),
            "algorithmic_bytes_per_pair": dom_bytes, "pairs_per_launch": n_local,
            "avg_launch_ms": round(avg_launch_ms, 4),
            "hamming_ms_per_launch": round(ham_avg_ms, 4),
            "ransac_ms_per_launch": round(rsc_ms / max(rsc_launches, 1), 4),
            "ransac_schedule": "one wave per pair (1 kernel per batch)" if args.ransac_path == "one_wave" else
                               "record/replay (a batch's select+RANSAC stage = pair_prep_kernel + 4 x (recording "
                               "launch of select_ransac_kernel + replay_walk_kernel) + 1 result launch; avg_launch_ms "
                               "spans the whole stage)",
            "pair_path_GBs": round(value / world * b_pair / 1e9, 3),
            "valu_laneops_per_s": round(valu_achieved, 1), "valu_peak": VALU_PEAK_LANEOPS,
            "valu_frac": round(valu_achieved / VALU_PEAK_LANEOPS, 4), **iso,
            "note": "neither kernel of the ORB pair path is HBM bound: the match is integer-VALU bound (xor+bcnt), "
                    "select+RANSAC is VALU-issue/latency bound (valu_busy_frac_pmc = fraction of SIMD time with a VALU "
                    "instruction active, rocprofv3 PMC pass); see DESIGN.md 4.1-4.2",
        }
        out = {
            "metric": "frame-pairs matched+RANSAC/sec, 640x480 ORB-1000",
            "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 popcount (match) + f32/f64 (RANSAC)", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 640x480 RGB-D, ORB %d kp, %d candidate pairs/frame, %d frames"
                                   % (N, args.pairs_per_frame, F),
                       "pairs_per_gpu_per_step": n_local, "max_matches": MAX_MATCHES,
                       "ransac_iterations": 200, "parallelism": "pair-sharded x%d" % world,
                       "edge_fraction": round(edge_frac, 4), "mean_ransac_iterations": round(mean_iters, 2)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq, pq, pt, fe, args.cpu_seconds)
            ref = cpu_reference_code(seq, pq, pt, fe, 5.0)
            if ref is not None:
                out["cpu_baseline_reference_code"] = ref
        print(json.dumps(out), flush=True)

    fe.close()
    if world > 1:
        dist.destroy_process_group()


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


def cpu_baseline(seq, pq, pt, fe, budget_s):
    """The oracle (CPU restatement of the reference pair path, kind='port') timed pair-parallel
    on this host's cores over a bounded sample of the same pair list."""
    from oracle import pyoracle as po
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
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


def cpu_reference_code(seq, pq, pt, fe, budget_s):
    """The reference's OWN pair op (Node::matchNodePair and everything below it, compiled from the reference
    sources into oracle/_ref/libref_ransac.so with Eigen / PCL stand-ins) timed on one host
    thread over a bounded sample.  Reported next to cpu_baseline; None when the prebuilt pin is absent."""
    from oracle import pyoracle as po
    if po.ref_ransac_lib() is None:
        return None
    prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov)
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
