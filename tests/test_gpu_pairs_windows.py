"""-m gpu: the refinement kernel's WINDOWS (round 6).

A phased plan records a pair's iteration range in windows inside ONE persistent launch; between two windows the workgroup's
server runs the reference's in-order bookkeeping (node.cpp:1171-1190) over what the pair has recorded and decides whether the
loop has ended, and the result waves finish the walk.  A window holds at most 256 list entries, so iteration counts above
that chain windows inside a phase as well.  Everything here is compared with the one-wave-per-pair kernel byte for byte and
with the oracle: pairs that end early (a > 80 % hypothesis: the walk stops the recording), pairs that run every iteration,
pairs without RANSAC, at iteration counts below, at and above a window's capacity."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rgbdslam_v2_amd import synth
from test_gpu_pairs import check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth_noise", [0.002, 0.01])
def test_windows_equal_the_one_wave_kernel_and_the_oracle(depth_noise):
    from rgbdslam_v2_amd.frontend import FrontEnd
    rng = np.random.default_rng(606)
    seq = synth.make_sequence(n_frames=8, n_kp=900, seed=606, depth_noise=depth_noise)
    nodes = {f: (seq["desc"][f], seq["xyz1"][f]) for f in range(8)}
    nodes[8] = (rng.integers(0, 256, (400, 32), dtype=np.uint8), seq["xyz1"][1][:400])   # unrelated: every hypothesis is junk
    nodes[9] = (seq["desc"][0][:25], seq["xyz1"][0][:25])                                  # barely above min_matches
    nodes[10] = (seq["desc"][0][:3], seq["xyz1"][0][:3])                                   # no RANSAC
    fe = FrontEnd(device_id=0, max_nodes=16, max_keypoints=1024, max_pairs_per_batch=512)
    try:
        for k, (d, x) in nodes.items():
            fe.upload_node(k, d, x)
        pairs = [(a, b) for a in range(8) for b in range(a)] + [(8, 0), (0, 8), (9, 0), (0, 9), (10, 0), (8, 9)]
        pq = np.array([p[0] for p in pairs], np.int32)
        pt = np.array([p[1] for p in pairs], np.int32)
        big_q, big_t = np.tile(pq, 9)[:300], np.tile(pt, 9)[:300]   # more than 256 pairs: the library's own phased plan
        for iters in (255, 256, 257, 600, 1000):
            fe.set_params(ransac_iterations=iters)
            fe.set_latency_mode(0)                                   # one wave per pair: the byte reference
            ref = fe.match_pair_list(pq, pt)
            if iters in (257, 600):
                prm = po.default_params(seed=fe.params.seed, depth_cov=fe.params.depth_cov, ransac_iterations=iters)
                for rec, (q, t) in zip(ref, pairs):
                    check_against_oracle(rec, po.match_node_pair(nodes[q][0], nodes[q][1], q, nodes[t][0], nodes[t][1], t, prm))
            assert len({int(r["real_iterations"]) for r in ref}) > 2   # early exits, full runs and no-RANSAC pairs are all here
            fe.set_latency_mode(1 << 20, -4)                         # the phased plan forced onto the small batch
            assert fe.match_pair_list(pq, pt).tobytes() == ref.tobytes(), ("phased, small batch", iters)
            fe.set_latency_mode(1 << 20, 0)                          # the default plan of a large batch
            assert fe.match_pair_list(big_q, big_t).tobytes() == np.tile(ref, 9)[:300].tobytes(), ("phased, 300 pairs", iters)
            fe.set_latency_mode(64, 7)                               # full speculation in (pair, share) units
            assert fe.match_pair_list(pq, pt).tobytes() == ref.tobytes(), ("shares", iters)
    finally:
        fe.close()
