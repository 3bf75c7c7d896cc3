"""-m gpu: bench.py's N > 1 path end to end -- two ranks under torch.distributed.run as the driver launches them, on the
ONE GPU of the test box (RGBDFE_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device, so the collectives go through
host tensors; everything else -- sharding, count exchange, padding, per-step gather, barriers, max-over-ranks timing, the
one JSON line from rank 0 -- is the code the 8-GPU run executes)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_two_ranks(extra, **more_env):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RGBDFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **more_env)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra,
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    return json.loads(lines[0])


def test_bench_two_ranks_one_gpu():
    d = _run_two_ranks(["--steps", "4", "--warmup", "1", "--frames", "60", "--pairs-per-frame", "10", "--gather", "compact"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    # weak scaling: 10 candidates per frame and rank -> 20 per frame globally, sharded round robin
    assert d["config"]["pairs_per_gpu_per_step"] > 0 and d["config"]["parallelism"] == "pair-sharded x2"
    assert "sift" not in d and "cpu_baseline" not in d    # extras and the CPU leg belong to the N = 1 line
    # the per-step gather moves compact records here (--gather compact), and the line says how many ranks the collective saw
    assert d["gather"]["payload"] == "rgbdfe_compact_result" and d["gather"]["bytes_per_record"] == 144
    assert d["gather"]["rccl_ranks"] == 2 and d["gather"]["backend"] == "gloo"
    assert len(d["repeats"]["values"]) == 7 and d["value"] == sorted(d["repeats"]["values"])[3]   # bench.py REPEATS, median
    assert d["parity_check"]["checked"] is False            # a reduced workload has no oracle constants


@pytest.mark.parametrize("gather", ["inliers", "compact", "full"])
def test_bench_two_ranks_full_workload_is_checked_against_the_oracle(gather):
    """The N > 1 line of the driver's scaling run: configs[1] at world 2 = 2 x 4000 pairs; rank 0 checks the
    records the all-gather left behind -- both ranks' -- against the oracle's sums over the global pair list
    (tests/golden/bench_expected.json, orb / 0.01 / 2), for every gather payload."""
    d = _run_two_ranks(["--steps", "3", "--warmup", "1", "--gather", gather])
    assert d["config"]["pairs_per_gpu_per_step"] == 4000 and d["n_gpus"] == 2
    pc = d["parity_check"]
    assert pc["checked"] and pc["ok"] and "8000 records of 2 ranks" in pc["records"]
    import bench
    assert pc["oracle_aggregates"] == bench.expected("orb", 0.01, 2)
    assert d["gather"]["payload_option"] == gather


def test_inlier_lists_that_outgrow_the_collective_are_gathered_again():
    """The inlier gather is ONE collective sized before the ranks have counted their lists; a capacity that turns out too small
    (forced here: 1000 entries per rank against ~10^5) is noticed on every rank from the gathered headers and that step's
    gather is repeated at a size that fits -- the records rank 0 ends up with are the oracle's."""
    d = _run_two_ranks(["--steps", "3", "--warmup", "1", "--gather", "inliers"], RGBDFE_BENCH_INLIER_CAP="1000")
    pc, ig = d["parity_check"], d["gather"]["inlier_gather"]
    assert pc["checked"] and pc["ok"] and pc["inlier_list_entries"] == pc["oracle_aggregates"]["inliers"]
    assert ig["regathers"] >= 1 and ig["list_capacity_entries"] > 1000
    assert ig["collectives_issued"] == ig["gathers_issued"] + ig["regathers"]     # no exchange of lengths anywhere


@pytest.mark.parametrize("gather", ["inliers", "compact", "full"])
def test_bench_gather_path_over_rccl_with_one_rank(gather):
    """The RCCL branch of the N > 1 code (nccl process group, pack kernels on torch's stream, ncclAllGather of device
    buffers -- one collective per step for every payload --, the parity check on the gathered records) cannot run with two
    ranks on a one-GPU box; RGBDFE_BENCH_FORCE_GATHER=1 runs it with ONE rank under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RGBDFE_BENCH_FORCE_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RGBDFE_BENCH_BACKEND", None)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
         "--no-extras", "--no-cpu-baseline", "--gather", gather],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["gather"]["backend"] == "nccl" and d["gather"]["rccl_ranks"] == 1
    assert d["gather"]["payload_option"] == gather and d["gather"]["gathers_in_timed_regions"] == 7 * 4   # REPEATS x steps
    pc = d["parity_check"]
    assert pc["checked"] and pc["ok"] and "4000 records of 1 ranks" in pc["records"]
    if gather == "inliers":
        assert pc["inlier_list_entries"] == pc["oracle_aggregates"]["inliers"]
        assert d["gather"]["collectives_per_step"] == 1 and 200 < d["gather"]["bytes_per_record"] < 400
        ig = d["gather"]["inlier_gather"]    # one collective per gather; the first (warm-up) one exchanged the lengths before it
        assert ig["collectives_issued"] == ig["gathers_issued"] + 1 and ig["regathers"] == 0
        assert pc["inlier_list_entries"] < ig["list_capacity_entries"] <= pc["inlier_list_entries"] * 5 // 4 + 64


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run must create the two ranks itself (it re-executes
    under torch.distributed.run) -- invoked the way the driver invokes the N = 1 line it used to run ONE rank and print
    n_gpus 1.  Full configs[1] workload, so the line carries the oracle check of both ranks' records."""
    env = dict(os.environ, RGBDFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gather"]["rccl_ranks"] == 2
    assert d["parity_check"]["checked"] and d["parity_check"]["ok"]
    assert len(d["gather"]["rank_devices"]) == 2 and d["gather"]["rank_devices"][1].startswith("rank 1: cuda:")
    assert d["gather"]["distinct_devices"] == 1      # (two ranks on the one GPU of the test box: the line says so)
