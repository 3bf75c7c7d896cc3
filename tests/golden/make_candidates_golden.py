"""Generates tests/golden/candidates.json: inputs and outputs of the reference's own
GraphManager::getPotentialEdgeTargetsWithDijkstra (oracle/_ref/libref_graph.so, compiled from
/root/reference/src/graph_manager.cpp:204-324 with Qt / g2o stand-ins) on seeded random pose graphs; rand() is
glibc's, seeded per case.  Run in the build container (needs /root/reference):  python tests/golden/make_candidates_golden.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402


def random_case(rng):
    n = int(rng.choice([1, 2, 3, 5, 8, 15, 40, 120]))
    node_ids = list(range(n))
    stride = int(rng.choice([1, 1, 2]))
    vertex_ids = [i * stride + (3 if stride == 2 else 0) for i in node_ids]
    matchable = [int(rng.random() > 0.1) for _ in node_ids]
    edges = [(i, i - 1) for i in range(1, n) if rng.random() > 0.05]
    for _ in range(int(rng.integers(0, max(1, n // 3) + 1))):
        a, b = (int(v) for v in rng.integers(0, n, 2))
        if a != b:
            edges.append((a, b))
    keyframes = [i for i in node_ids if rng.random() < 0.4]
    return dict(node_ids=node_ids, vertex_ids=vertex_ids, matchable=matchable, keyframes=keyframes,
                edges=[list(e) for e in edges],
                sequential_targets=int(rng.integers(0, 6)), geodesic_targets=int(rng.integers(0, 6)),
                sampled_targets=int(rng.integers(0, 6)), geodesic_depth=int(rng.integers(1, 6)),
                predecessor_id=int(rng.choice([-1, -1, int(rng.integers(0, n))])),
                include_predecessor=bool(rng.random() < 0.3), srand_seed=int(rng.integers(0, 2**31)))


def run_reference(c):
    return po.ref_potential_edge_targets(c["node_ids"], c["vertex_ids"], c["matchable"], c["keyframes"], c["edges"],
                                         c["sequential_targets"], c["geodesic_targets"], c["sampled_targets"],
                                         c["geodesic_depth"], c["predecessor_id"], c["include_predecessor"],
                                         c["srand_seed"]).tolist()


if __name__ == "__main__":
    rng = np.random.default_rng(20260924)
    cases = []
    for _ in range(60):
        c = random_case(rng)
        c["expected"] = run_reference(c)
        cases.append(c)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "candidates.json")
    json.dump(cases, open(out, "w"))
    print("wrote", out, len(cases), "cases")
