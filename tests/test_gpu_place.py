"""-m gpu: the loop-closure prefilter (rgbdfe_place_recognition) against its oracle
(oracle/pyoracle.py::place_recognition, a numpy restatement of loop_closing.cpp:190-277 with exact neighbours):
integer votes, one float division per node -- ranking and scores identical."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    # 4 unrelated places x 10 frames each: frames of one place share world points, the others do not
    seqs = [synth.make_sequence(n_frames=10, n_kp=500, n_world=1500, seed=100 + p, motion_scale=0.6) for p in range(4)]
    desc = np.concatenate([s["desc"] for s in seqs])
    xyz = np.concatenate([s["xyz1"] for s in seqs])
    return desc, xyz


@pytest.mark.parametrize("k,max_hd", [(1, 128), (2, 128), (4, 257), (8, 100)])
def test_votes_and_ranking_match_oracle(world, k, max_hd):
    from rgbdslam_v2_amd.frontend import FrontEnd
    desc, xyz = world
    fe = FrontEnd(device_id=0, max_nodes=48, max_keypoints=512, max_pairs_per_batch=64)
    for f in range(len(desc)):
        fe.upload_node(f, desc[f], xyz[f])
    for q in (39, 17, 5):
        cands = np.array([c for c in range(len(desc)) if c != q], np.int32)[::-1].copy()
        ids, sc = fe.place_recognition(q, cands, k_neighbours=k, max_hd=max_hd)
        pos, rsc = po.place_recognition(desc[q], [desc[c] for c in cands], k, max_hd)
        assert np.array_equal(ids, cands[pos])
        assert np.array_equal(sc, rsc)
        # the frames of the query's own place come first
        same_place = (ids[:5] // 10) == (q // 10)
        assert same_place.all(), (q, ids[:8])
        top3, _ = fe.place_recognition(q, cands, k_neighbours=k, max_hd=max_hd, max_out=3)
        assert np.array_equal(top3, ids[:3])
    fe.close()


def test_batch_of_queries_equals_single_calls(world):
    """An offline sweep (every frame against all earlier frames) is one Hamming launch + one vote launch."""
    from rgbdslam_v2_amd.frontend import FrontEnd
    desc, xyz = world
    F = 24
    fe = FrontEnd(device_id=0, max_nodes=32, max_keypoints=512, max_pairs_per_batch=F * F)
    for f in range(F):
        fe.upload_node(f, desc[f], xyz[f])
    queries = list(range(1, F)) + [0]
    lists = [np.arange(q) for q in range(1, F)] + [np.zeros(0, np.int32)]
    ranked = fe.place_recognition_batch(queries, lists, k_neighbours=3, max_hd=128, max_out=6)
    assert len(ranked) == len(queries)
    for q, cands, (ids, sc) in zip(queries, lists, ranked):
        ids1, sc1 = fe.place_recognition(q, cands, k_neighbours=3, max_hd=128, max_out=6)
        assert np.array_equal(ids, ids1) and np.array_equal(sc, sc1)
        pos, rsc = po.place_recognition(desc[q], [desc[c] for c in cands], 3, 128)
        assert np.array_equal(ids, cands[pos][:6]) and np.array_equal(sc, rsc[:6])
    fe.close()


def test_edge_cases(world):
    from rgbdslam_v2_amd._lib import RgbdfeError
    from rgbdslam_v2_amd.frontend import FrontEnd
    desc, xyz = world
    fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=512, max_pairs_per_batch=8)
    for f in range(6):
        fe.upload_node(f, desc[f], xyz[f])
    fe.upload_node(6, desc[6][:1], xyz[6][:1])      # one row: bruteForceSearchORB never looks at the last row
    fe.upload_node(7, desc[7][:0], xyz[7][:0])      # empty node
    ids, sc = fe.place_recognition(0, [6, 7], k_neighbours=2)
    assert len(ids) == 0
    ids, sc = fe.place_recognition(7, [0, 1, 2], k_neighbours=2)
    assert len(ids) == 0
    ids, sc = fe.place_recognition(0, [], k_neighbours=2)
    assert len(ids) == 0
    ids, sc = fe.place_recognition(0, [1, 1, 2], k_neighbours=2)      # a candidate listed twice: the first wins ties
    pos, rsc = po.place_recognition(desc[0], [desc[1], desc[1], desc[2]], 2, 128)
    assert np.array_equal(ids, np.array([1, 1, 2])[pos]) and np.array_equal(sc, rsc)
    with pytest.raises(RgbdfeError):
        fe.place_recognition(0, [99], k_neighbours=2)                 # unknown node
    with pytest.raises(RgbdfeError):
        fe.place_recognition(0, [1], k_neighbours=9)                  # k beyond 8
    with pytest.raises(RgbdfeError):
        fe.place_recognition(0, list(range(1, 6)) * 2, k_neighbours=2)  # more candidates than max_pairs_per_batch
    # a malformed offsets array (an intermediate offset beyond the total that later descends) is refused
    # before anything indexed by it is written
    import ctypes as C
    qs = np.array([0, 1, 2], np.int32)
    cands = np.array([1, 2, 3, 4], np.int32)
    ids = np.zeros((3, 4), np.int32); sc = np.zeros((3, 4), np.float32); cnt = np.zeros(3, np.int32)
    for offs in ([0, 1 << 20, 2, 4], [0, 3, 2, 4], [0, -1, 2, 4], [1, 2, 3, 4]):
        o = np.array(offs, np.int32)
        st = fe._L.rgbdfe_place_recognition_batch(fe._ctx, qs.ctypes.data, 3, o.ctypes.data, cands.ctypes.data, 2, 128, 4,
                                                  ids.ctypes.data, sc.ctypes.data, cnt.ctypes.data)
        assert st == -1, offs                       # RGBDFE_ERR_INVALID_ARG
    fe.close()
