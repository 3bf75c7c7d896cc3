#!/usr/bin/env python3
"""tests/golden/bench_expected.json: the oracle's aggregates of every workload bench.py prints a number for.

CPU only (run here, in the build container; a few minutes on 16 threads):
  * pair workloads -- oracle/liboracle.so (orc_match_pairs_mt / orc_match_sift_node_pair) over bench.py's own workload
    builders: configs[1] at N = 1, 2, 4, 8 ranks (N x 4000 pairs of the global list), depth noise 0.002 z^2, the
    loop-closure search, configs[3] (2000 SIFT pairs);
  * detect / front_end -- oracle/orb_oracle.c frame after frame from a fresh detector state (+ misc2.h's projectTo3D), the
    front end's pairs over the oracle's features;
  * sift_extract -- SiftGPU's own pipeline compiled from the reference tree (oracle/_ref/libref_siftgpu.so).
bench.py compares what the GPU produced with these constants and refuses to print a number that differs;
tests/test_gpu_bench_parity.py compares the same workloads record by record where the oracle is fast enough.

    python tools/make_bench_expected.py [section ...]     (sections: orb heavy loop_closure sift detect front_end sift_extract)
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pyoracle as po, pyorb  # noqa: E402
from rgbdslam_v2_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, bench.EXPECTED_FILE)


def orc_aggregates(recs):
    return {"edges": int(sum(1 for r in recs if r.id1 >= 0)), "real_iterations": int(sum(r.real_iterations for r in recs)),
            "inliers": int(sum(r.n_inl for r in recs))}


def pairs_orb(descs, xyzs, pq, pt, seed=bench.SEED):
    prm = po.default_params(seed=seed, depth_cov=1e-4)
    return orc_aggregates(po.match_pairs_mt(list(descs), list(xyzs), np.arange(len(descs)), pq, pt, prm, 0))


def oracle_detect_run(grays, masks, depths, K, n_kp):
    """Node::Node's feature path frame after frame, the detector state carried along (what rgbdfe_detect_describe_batch returns)."""
    st = pyorb.grid_state(n_kp)
    out = []
    for g, m, d in zip(grays, masks, depths):
        kp, desc = pyorb.node_features(st, g, m, d, n_kp)
        kept, xyz = po.project_to_3d(np.stack([kp["x"], kp["y"]], 1), d, K[0], K[1], K[2], K[3], 1.0, n_kp)
        assert len(kept) == len(kp)
        out.append((kp, desc, xyz))
    return out


def main():
    want = set(sys.argv[1:]) or {"orb", "heavy", "loop_closure", "sift", "detect", "front_end", "sift_extract"}
    try:
        E = json.load(open(OUT))
    except Exception:  # noqa: BLE001
        E = {}
    t00 = time.time()
    if "orb" in want:
        for world in (1, 2, 4, 8):
            seq, pq, pt = bench.orb_workload(world)
            t0 = time.time()
            E.setdefault("orb", {}).setdefault("0.01", {})[str(world)] = pairs_orb(seq["desc"], seq["xyz1"], pq, pt)
            print("orb 0.01 x%d: %d pairs %s (%.1f s)" % (world, len(pq), E["orb"]["0.01"][str(world)], time.time() - t0), flush=True)
    if "heavy" in want:
        seq, pq, pt = bench.orb_workload(1, depth_noise=synth.DEPTH_NOISE_R1)
        E.setdefault("orb", {}).setdefault(str(synth.DEPTH_NOISE_R1), {})["1"] = pairs_orb(seq["desc"], seq["xyz1"], pq, pt)
        print("orb heavy:", E["orb"][str(synth.DEPTH_NOISE_R1)]["1"], flush=True)
    if "loop_closure" in want:
        desc, xyz, pq, pt = synth.loop_closure_places(depth_noise=0.01)
        prm = po.default_params()   # (the sub-record's FrontEnd runs with the library's default seed)
        E.setdefault("loop_closure", {})["0.01"] = orc_aggregates(po.match_pairs_mt(desc, xyz, np.arange(len(desc)), pq, pt, prm, 0))
        print("loop_closure:", E["loop_closure"]["0.01"], flush=True)
    if "sift" in want:
        seq, _, _ = bench.orb_workload(1)
        sd, pq, pt = bench.sift_workload(seq)
        prm = po.default_params(seed=bench.SEED, depth_cov=1e-4)
        t0 = time.time()

        def one(k):
            q, t = int(pq[k]), int(pt[k])
            r = po.match_sift_node_pair(sd[q], seq["xyz1"][q], q, sd[t], seq["xyz1"][t], t, prm)
            return (r["id1"] >= 0, r["real_iterations"], r["n_inl"])
        with ThreadPoolExecutor(po.usable_cpus()) as ex:
            rs = list(ex.map(one, range(len(pq))))
        E.setdefault("sift", {})["0.01"] = {"edges": int(sum(r[0] for r in rs)), "real_iterations": int(sum(r[1] for r in rs)),
                                            "inliers": int(sum(r[2] for r in rs))}
        print("sift: %d pairs %s (%.1f s)" % (len(pq), E["sift"]["0.01"], time.time() - t0), flush=True)
    if "detect" in want:
        for (w, h, n_kp, n_base, n_run) in bench.DETECT_WORKLOADS:
            _, _, grays, mks, depths, K = bench.detect_workload(w, h, n_base, n_run)
            t0 = time.time()
            key = "%dx%d_orb%d" % (w, h, n_kp)
            E.setdefault("detect", {})[key] = bench.features_checksum(oracle_detect_run(grays, mks, depths, K, n_kp))
            print("detect %s: %s (%.1f s)" % (key, E["detect"][key], time.time() - t0), flush=True)
    if "front_end" in want:
        fe = bench.FRONT_END
        _, _, grays, mks, depths, K = bench.detect_workload(640, 480, fe["n_base"], fe["n_run"])
        feats = oracle_detect_run(grays, mks, depths, K, fe["n_kp"])
        pq, pt = bench.front_end_pairs(fe["n_run"], fe["cand"])
        agg = pairs_orb([f[1] for f in feats], [f[2] for f in feats], pq, pt)
        E.setdefault("front_end", {})["640x480_orb1000"] = dict(bench.features_checksum(feats), **agg)
        print("front_end:", E["front_end"]["640x480_orb1000"], flush=True)
    if "sift_extract" in want:
        seq, _ = bench.sift_extract_workload()
        frames = []
        kpt = np.dtype([("x", "<f4"), ("y", "<f4")])
        for g in seq["gray"]:
            keys, desc, _ = po.ref_sift_detect(g, 1000)
            kp = np.zeros(len(keys), kpt)
            kp["x"], kp["y"] = keys[:, 0], keys[:, 1]
            frames.append((kp, desc))
        E.setdefault("sift_extract", {})["640x480"] = bench.sift_features_checksum(frames)
        print("sift_extract:", E["sift_extract"]["640x480"], flush=True)
    E["made_by"] = "tools/make_bench_expected.py (oracle/liboracle.so, oracle/orb_oracle.c, oracle/_ref/libref_siftgpu.so; CPU)"
    json.dump(E, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote %s (%.0f s)" % (OUT, time.time() - t00))


if __name__ == "__main__":
    main()
