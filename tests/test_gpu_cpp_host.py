"""-m gpu: the C++ host side (include/rgbdfe.hpp: Node::matchNodePair, GraphManager::nodeComparisons)
built with plain g++ against the C ABI gives the same results as the oracle."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "examples", "cpp", "match_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples", "cpp")])
    F = 5
    seq = synth.make_sequence(n_frames=F, n_kp=700, n_world=2800, seed=6)
    path = tmp_path / "nodes.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<i", F))
        for k in range(F):
            f.write(struct.pack("<i", 700))
            f.write(seq["desc"][k].tobytes())
            f.write(seq["xyz1"][k].tobytes())
    out = subprocess.check_output([exe, str(path)], text=True, timeout=120)
    lines = [json.loads(l) for l in out.strip().splitlines()]
    assert len(lines) == F + 1
    # getPotentialEdgeTargetsWithDijkstra(2 sequential, 1 geodesic, 1 sampled) over the 4 earlier nodes: no more nodes
    # than targets, so all of them, sequentially from the predecessor (graph_manager.cpp:212-227)
    assert lines[-2]["candidates"] == [2, 1, 0]
    prm = po.default_params()
    for t, rec in enumerate(lines[:-2]):
        ref = po.match_node_pair(seq["desc"][F - 1], seq["xyz1"][F - 1], F - 1, seq["desc"][t], seq["xyz1"][t], t, prm)
        assert (rec["id1"], rec["id2"]) == (ref["id1"], ref["id2"])
        assert rec["n_all"] == ref["n_all"] and rec["n_inl"] == ref["n_inl"]
        T = np.array(rec["T"], np.float32).reshape(4, 4).T
        assert np.array_equal(T, ref["T"])
        assert np.float32(rec["rmse"]) == ref["rmse"] and rec["info"] == ref["info_scale"]
    assert lines[-1]["single_id1"] == lines[0]["id1"] and lines[-1]["single_n_inl"] == lines[0]["n_inl"]
