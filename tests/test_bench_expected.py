"""CPU: the oracle constants bench.py checks its numbers against (tests/golden/bench_expected.json, made by
tools/make_bench_expected.py) are complete for every workload the driver's line can print, the checker raises on a
difference, and a sample of the constants is re-derived here with the oracle (the full re-derivation takes 3 minutes on 16
threads: `python tools/make_bench_expected.py` rewrites the file; git diff must stay empty)."""
import numpy as np
import pytest

import bench
from oracle import pyoracle as po
from rgbdslam_v2_amd._lib import COMPACT_DTYPE, RESULT_DTYPE


def test_constants_exist_for_every_printed_number():
    for world in (1, 2, 4, 8):
        e = bench.expected("orb", 0.01, world)
        assert e and set(e) == {"edges", "real_iterations", "inliers"}, world
        # weak scaling: N x 4000 pairs, 200 iterations each at this noise level
        assert e["real_iterations"] == world * 4000 * 200 and 0.97 * world * 4000 < e["edges"] <= world * 4000
    assert bench.expected("orb", 0.002, 1)["edges"] == 4000
    assert bench.expected("loop_closure", 0.01)["edges"] == 810
    assert bench.expected("sift", "0.01")["real_iterations"] == 2000 * 200
    for key in ("640x480_orb1000", "1280x960_orb4000"):
        d = bench.expected("detect", key)
        assert d["keypoints"] > 50000 and 0 <= d["crc32"] < 2 ** 32
    f = bench.expected("front_end", "640x480_orb1000")
    assert f["keypoints"] == bench.expected("detect", "640x480_orb1000")["keypoints"] and f["edges"] > 1900
    s = bench.expected("sift_extract", "640x480")
    assert len(s["features_per_frame"]) == 8 and s["descriptor_abs_sum"] > 0
    assert bench.expected("orb", 0.01, 3) is None and bench.expected("nothing") is None


def test_checker_raises_on_any_difference():
    exp = bench.expected("orb", 0.01, 1)
    assert bench.check_against(exp, dict(exp), "same", "test")["ok"]
    for k in exp:
        with pytest.raises(SystemExit):
            bench.check_against(exp, dict(exp, **{k: exp[k] - 1}), "doctored", "test")
    assert bench.check_against(None, dict(exp), "unknown workload", "test")["checked"] is False
    s = bench.expected("sift_extract", "640x480")
    near = dict(s, descriptor_abs_sum=s["descriptor_abs_sum"] * (1 + 0.5 * bench.SIFT_DESC_RTOL))
    assert bench.check_against(s, near, "libm tolerance", "test", approx=("descriptor_abs_sum",))["ok"]
    far = dict(s, descriptor_abs_sum=s["descriptor_abs_sum"] * 1.01)
    with pytest.raises(SystemExit):
        bench.check_against(s, far, "beyond the tolerance", "test", approx=("descriptor_abs_sum",))


def test_aggregates_read_full_and_compact_records_alike():
    rec = np.zeros(5, RESULT_DTYPE)
    rec["id1"] = [-1, 3, 0, -1, 7]
    rec["real_iterations"] = [200, 14, 70, 0, 200]
    rec["n_inl"] = [0, 120, 33, 0, 64]
    want = {"edges": 3, "real_iterations": 484, "inliers": 217}
    assert bench.pair_aggregates(rec) == want
    c = np.zeros(5, COMPACT_DTYPE)
    for f in ("id1", "real_iterations", "n_inl"):
        c[f] = rec[f]
    assert bench.pair_aggregates(c) == want


def test_a_sample_of_the_constants_is_rederived_by_the_oracle():
    """The first 300 pairs of the world-2 global pair list: the oracle's sums over them are reproducible here in a second
    (the whole file takes three minutes: tools/make_bench_expected.py), and the shard a rank owns is every world-th pair."""
    seq, pq, pt = bench.orb_workload(2)
    assert len(pq) == 8000
    prm = po.default_params(seed=bench.SEED, depth_cov=1e-4)
    recs = po.match_pairs_mt(list(seq["desc"]), list(seq["xyz1"]), np.arange(len(seq["desc"])), pq[:300], pt[:300], prm, 0)
    assert sum(r.real_iterations for r in recs) == 300 * 200 and sum(1 for r in recs if r.id1 >= 0) == 289
    # rank 0's shard of the world-2 list is every second pair (dist.shard_pairs): the bench's sharding
    from rgbdslam_v2_amd.dist import shard_pairs
    q0, t0 = shard_pairs(pq, pt, 0, 2)
    assert np.array_equal(q0, pq[0::2]) and len(q0) == 4000


@pytest.mark.parametrize("gpus,world", [(8, 1), (1, 2), (2, 4)])
def test_bench_refuses_a_world_size_that_disagrees_with_gpus(gpus, world):
    """a WORLD_SIZE that differs from --gpus in EITHER direction ends the run before anything is measured (with
    WORLD_SIZE unset and --gpus N > 1 bench.py starts its N ranks itself: tests/test_gpu_bench_ranks.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE=str(world), RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "1"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and ("--gpus %d but WORLD_SIZE=%d" % (gpus, world)) in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
