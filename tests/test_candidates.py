"""Candidate selection for loop closure: rgbdfe_potential_edge_targets against the reference's
own GraphManager::getPotentialEdgeTargetsWithDijkstra (graph_manager.cpp:204-324) -- live when the reference tree is
present (oracle/_ref/libref_graph.so), and against outputs of that code frozen in tests/golden/candidates.json.
Host code only: runs without a GPU."""
import ctypes as C
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd.candidates import PoseGraph

HERE = os.path.dirname(os.path.abspath(__file__))
libc = C.CDLL(None)
libc.rand.restype = C.c_int


def product(c):
    g = PoseGraph()
    try:
        kf = set(c["keyframes"])
        for nid, vid, m in zip(c["node_ids"], c["vertex_ids"], c["matchable"]):
            g.add_node(nid, vid, matchable=bool(m), keyframe=nid in kf)
        for a, b in c["edges"]:
            g.add_edge(a, b)
        libc.srand(C.c_uint(c["srand_seed"]))
        return g.potential_edge_targets(c["sequential_targets"], c["geodesic_targets"], c["sampled_targets"],
                                        c["geodesic_depth"], c["predecessor_id"], c["include_predecessor"],
                                        rand=libc.rand).tolist()
    finally:
        g.close()


def test_frozen_reference_outputs():
    cases = json.load(open(os.path.join(HERE, "golden", "candidates.json")))
    assert len(cases) >= 50
    n_geo = n_samp = 0
    for c in cases:
        assert product(c) == c["expected"], c
        n_geo += c["geodesic_targets"] > 0 and len(c["node_ids"]) > 15
        n_samp += c["sampled_targets"] > 0 and len(c["node_ids"]) > 15
    assert n_geo >= 5 and n_samp >= 5  # the fixture exercises all three target classes


@pytest.mark.skipif(po.ref_graph_lib() is None, reason="reference tree not available (oracle/_ref/libref_graph.so)")
def test_live_reference_random_graphs():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_candidates_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    rng = np.random.default_rng(7)
    for _ in range(400):
        c = mk.random_case(rng)
        assert product(c) == mk.run_reference(c), c


def test_counter_based_generator_is_reproducible_and_valid():
    g = PoseGraph()
    n = 200
    for i in range(n):
        g.add_node(i, keyframe=(i % 3 == 0), matchable=(i % 17 != 5))
        if i:
            g.add_edge(i, i - 1)
    for a, b in ((150, 20), (199, 60), (100, 5)):
        g.add_edge(a, b)
    a = g.potential_edge_targets(3, 4, 5, geodesic_depth=3, seed=11)
    b = g.potential_edge_targets(3, 4, 5, geodesic_depth=3, seed=11)
    c = g.potential_edge_targets(3, 4, 5, geodesic_depth=3, seed=12)
    assert a.tolist() == b.tolist() and a.tolist() != c.tolist()
    assert len(a) == 12 and len(set(a.tolist())) == 12
    assert a[-3:].tolist() == [198, 197, 196]                      # sequential targets, nearest first (:223-226)
    # geodesic_depth 3 = graph distance < 3 from node 199 (:232): 198, 197 (sequential, excluded :265) and 60, 59, 61
    assert sorted(a[-6:-3].tolist()) == [59, 60, 61]
    # only three geodesic candidates exist, so the uniform sampling fills up to the requested total (:309)
    assert all(int(v) % 3 == 0 and int(v) % 17 != 5 for v in a[:6])  # sampled: matchable keyframes
    # fewer nodes than targets: everything sequential (:212-219)
    small = PoseGraph()
    for i in range(4):
        small.add_node(i, keyframe=True)
    assert small.potential_edge_targets(2, 2, 2).tolist() == [2, 1, 0]
    assert small.potential_edge_targets(2, 2, 2, include_predecessor=True).tolist() == [2, 1, 0, 3]


def test_argument_and_capacity_errors():
    import ctypes as C
    from rgbdslam_v2_amd import _lib
    L = _lib.load()
    g = PoseGraph()
    for i in range(30):
        g.add_node(i, keyframe=True)
        if i:
            g.add_edge(i, i - 1)
    with pytest.raises(_lib.RgbdfeError):
        g.add_edge(3, 3)              # an edge needs two different nodes
    with pytest.raises(_lib.RgbdfeError):
        g.add_edge(3, 99)             # ... that exist
    with pytest.raises(_lib.RgbdfeError):
        g.set_matchable(99, False)
    with pytest.raises(_lib.RgbdfeError):
        g.add_node(-1)
    # a too small output buffer: RGBDFE_ERR_CAPACITY and the needed size
    out = np.zeros(2, np.int32)
    n = C.c_int32(0)
    rc = L.rgbdfe_potential_edge_targets(g._g, 3, 2, 2, 3, -1, 0, L.rgbdfe_rand_fn(0), None, 5, out.ctypes.data, 2, C.byref(n))
    assert rc != 0 and n.value == 7
    assert L.rgbdfe_status_string(rc).decode().lower().find("capacity") >= 0
    # unmatchable nodes are never drawn (geodesic :264, sampled :302) but stay sequential targets (:223-226)
    for i in range(10, 25):
        g.set_matchable(i, False)
    ids = g.potential_edge_targets(2, 3, 4, geodesic_depth=30, seed=3).tolist()
    assert ids[-2:] == [28, 27]
    assert all(not (10 <= v < 25) for v in ids[:-2])
    g.close()
