"""CPU checks of the oracle's SiftGPUWrapper::match restatement against an independent numpy
formulation (integer dot matrix + explicit tie rules)."""
import os

import numpy as np

from oracle import pyoracle as po


def _rand_sift(rng, n):
    v = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = np.minimum(v, 0.2)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def numpy_sift_match(d1, d2):
    q1 = (512 * d1 + np.float64(0.5)).astype(np.int64).astype(np.uint8).astype(np.int64)  # SiftMatchCU.cpp:98
    q2 = (512 * d2 + np.float64(0.5)).astype(np.int64).astype(np.uint8).astype(np.int64)
    dot = q1 @ q2.T
    n1, n2 = dot.shape

    def angle(v):
        p = (v.astype(np.float32) * np.float32(0.000003814697265625)).astype(np.float64)
        return np.arccos(np.minimum(p, 1.0)).astype(np.float32)

    def bitrev5(x):
        return int("{:05b}".format(x)[::-1], 2)

    row_match = np.full(n1, -1)
    for i in range(n1):
        r = dot[i]
        mx = r.max()
        if mx <= 0:
            continue
        cand = np.flatnonzero(r == mx)
        # RowMatch_Kernel: within a thread the smallest column wins, across threads the butterfly
        # prefers the smaller bit-reversed thread id
        j = min(cand, key=lambda c: (bitrev5(c & 31), c >> 5))
        rest = np.delete(r, j)
        nx = max(rest.max(), 0) if len(rest) else 0
        d, dn = angle(np.array([mx]))[0], angle(np.array([nx]))[0]
        if d < np.float32(0.9) and d < dn * np.float32(0.9):
            row_match[i] = j
    col_match = np.full(n2, -1)
    for j in range(n2):
        c = dot[:, j]
        mx = c.max()
        if mx <= 0:
            continue
        i = int(np.flatnonzero(c == mx)[0])  # lowest row wins
        rest = np.delete(c, i)
        nx = max(rest.max(), 0) if len(rest) else 0
        d, dn = angle(np.array([mx]))[0], angle(np.array([nx]))[0]
        if d < np.float32(0.9) and d < dn * np.float32(0.9):
            col_match[j] = i
    mq = [i for i in range(n1) if row_match[i] >= 0 and col_match[row_match[i]] == i]
    mt = [row_match[i] for i in mq]
    return np.array(mq, np.int32), np.array(mt, np.int32)


def test_sift_match_matches_numpy_formulation():
    rng = np.random.default_rng(1)
    for n1, n2 in [(90, 70), (33, 200), (64, 64)]:
        d2 = _rand_sift(rng, n2)
        d1 = _rand_sift(rng, n1)
        k = min(n1, n2) // 2
        d1[:k] = np.abs(d2[rng.permutation(n2)[:k]] + rng.normal(0, 0.01, (k, 128)).astype(np.float32))
        d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
        mq, mt, md = po.sift_match(d1, d2)
        nq, nt = numpy_sift_match(d1, d2)
        if len(nq) <= 3 and ((nq == 0).any() or (nt == 0).any()):
            continue  # the wrapper's "context error" heuristic may clear tiny lists
        assert np.array_equal(mq, nq) and np.array_equal(mt, nt)
        assert len(mq) >= k // 2
        ref = np.sqrt(((d1[mq].astype(np.float64) - d2[mt]) ** 2).sum(1))
        assert np.allclose(md, ref, rtol=1e-5)


def test_sift_match_tie_rules_and_context_heuristic():
    rng = np.random.default_rng(2)
    base = _rand_sift(rng, 12)
    d2 = base[rng.integers(0, 12, 150)]
    d1 = base[rng.integers(0, 12, 80)]
    d2 = d2 + rng.normal(0, 1e-4, d2.shape).astype(np.float32)
    mq, mt, _ = po.sift_match(d1, d2)
    nq, nt = numpy_sift_match(d1, d2)
    assert np.array_equal(mq, nq) and np.array_equal(mt, nt)
    # a single match that touches index 0 is discarded (sift_gpu_wrapper.cpp:199-209)
    one = _rand_sift(rng, 1)
    mq, mt, _ = po.sift_match(one, one)
    assert len(mq) == 0
    assert po.sift_match(one[:0], one)[0].size == 0


def test_sift_node_features_oracle():
    """a20: projectTo3DSiftGPU (truncating lookup, node.cpp:733) + descriptor re-pack + RootSIFT."""
    rng = np.random.default_rng(12)
    rows, cols, n = 48, 64, 300
    depth = rng.uniform(0.5, 4, (rows, cols)).astype(np.float32)
    depth[rng.random((rows, cols)) < 0.2] = np.nan
    kp = np.stack([rng.uniform(0, cols - 0.01, n), rng.uniform(0, rows - 0.01, n)], 1).astype(np.float32)
    kp[7] = [10.9999959, 20.5]  # the reference's own example: truncation reads column 10, round() would read 11
    desc = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
    desc[3] *= -1.0   # cv::abs
    desc[5] = 0.0     # zero row: left alone (node.cpp:1565)
    kept, xyz, raw, feat = po.sift_node_features(kp, desc, depth, 52.5, 52.5, 31.5, 23.5, 1.0, 1000)
    exp = [i for i, (x, y) in enumerate(kp) if not np.isnan(depth[int(y), int(x)])]
    assert list(kept) == exp
    assert (7 in exp) == (not np.isnan(depth[20, 10]))  # column 10, not 11
    assert np.array_equal(raw, desc[kept])
    i = kept[0]
    assert xyz[0, 2] == depth[int(kp[i, 1]), int(kp[i, 0])] and xyz[0, 3] == 1
    # RootSIFT: unit L2 norm, sqrt of the L1-normalised magnitudes
    a = np.abs(desc[kept]).astype(np.float64)
    ref = np.sqrt(a / np.maximum(a.sum(1, keepdims=True), 1e-300))
    nz = a.sum(1) > 0
    assert np.allclose(feat[nz], ref[nz], rtol=2e-6, atol=0)
    assert np.allclose((feat[nz].astype(np.float64) ** 2).sum(1), 1.0, atol=1e-5)
    assert np.array_equal(feat[~nz], np.zeros_like(feat[~nz]))
    # the L1 norm follows cv::reduce's two-accumulator order exactly
    d = np.abs(desc[kept[1]])
    a0, a1 = np.float32(d[0]), np.float32(d[1])
    for j in range(2, 125, 4):
        a0 = np.float32(a0 + d[j]); a1 = np.float32(a1 + d[j + 1])
        a0 = np.float32(a0 + d[j + 2]); a1 = np.float32(a1 + d[j + 3])
    a0 = np.float32(np.float32(a0 + d[126]) + d[127])
    s = np.float32(a0 + a1)
    assert np.array_equal(feat[1], np.sqrt((d / s).astype(np.float32)).astype(np.float32))
    # max_keypoints cut (node.cpp:748) and use_root_sift = false
    kept2, _, raw2, feat2 = po.sift_node_features(kp, desc, depth, 52.5, 52.5, 31.5, 23.5, 1.0, 9, use_root_sift=False)
    assert list(kept2) == exp[:9] and np.array_equal(raw2, feat2)


import pytest  # noqa: E402


@pytest.mark.skipif(po.ref_sift_lib() is None, reason="reference pin (oracle/_ref/libref_siftmatch.so) not built")
def test_sift_match_matches_the_reference_trees_own_matcher():
    """configs[3]'s behavioural reference itself: the SiftGPU matcher vendored in the reference tree --
    MultiplyDescriptor_Kernel, RowMatch_Kernel, ColMatch_Kernel (ProgramCU.cu), SiftMatchCU::SetDescriptors /
    GetSiftMatch / GetBestMatch -- and SiftGPUWrapper::match (src/sift_gpu_wrapper.cpp:169-227), compiled from where
    they lie on a CUDA-on-CPU emulation, against orc_sift_match: same (queryIdx, trainIdx) list and the same float L2
    distances, including exact duplicates (the 32-thread butterfly's tie rule), ragged sizes (not multiples of the
    8-row / 128-column / 32-column blocks) and the index-0 "context error" heuristic."""
    rng = np.random.default_rng(41)
    base = _rand_sift(rng, 24)
    cases = []
    for n1, n2 in ((200, 260), (33, 129), (130, 31), (8, 8), (1, 40), (40, 1)):
        d2 = _rand_sift(rng, n2)
        d1 = d2[rng.integers(0, n2, n1)] + rng.normal(0, 0.03, (n1, 128)).astype(np.float32)
        cases.append((np.abs(d1).astype(np.float32), d2))
    # heavy ties: both sides drawn from 24 prototypes, a few of them exact duplicates
    d2 = base[rng.integers(0, 24, 170)].copy()
    d1 = base[rng.integers(0, 24, 90)].copy()
    d2[::3] += rng.normal(0, 1e-4, d2[::3].shape).astype(np.float32)
    cases.append((d1, d2))
    one = _rand_sift(rng, 1)
    cases.append((one, one))                      # a single match touching index 0: discarded (:199-209)
    cases.append((_rand_sift(rng, 50), _rand_sift(rng, 60)))  # unrelated: (almost) nothing passes the ratio test
    n_total = 0
    for d1, d2 in cases:
        mq, mt, md = po.sift_match(d1, d2)
        rq, rt, rd = po.ref_sift_match(d1, d2)
        assert np.array_equal(mq, rq) and np.array_equal(mt, rt), (d1.shape, d2.shape)
        assert np.array_equal(md, rd)
        n_total += len(mq)
    assert n_total > 100


GOLD_SIFT = os.path.join(os.path.dirname(__file__), "golden", "sift_golden.npz")


def test_sift_golden_vectors_from_the_reference_matcher():
    """tests/golden/sift_golden.npz (tests/golden/make_golden.py): the 677 real SIFT descriptors of the reference tree's
    only golden feature file (external/SiftGPU/doc/evaluation/box.siftgpu) against a noisy permuted subset, against
    themselves and against a set with exact duplicates, matched by the reference's OWN matcher code (SiftGPU's CUDA
    kernels + SiftMatchCU + SiftGPUWrapper::match compiled into oracle/_ref/libref_siftmatch.so).  The oracle must
    return the same (queryIdx, trainIdx) list and the same float distances."""
    g = np.load(GOLD_SIFT)
    total = 0
    for name in ("view", "self", "dups"):
        d1 = g[name + "_d1"].astype(np.float32) / 512.0
        d2 = g[name + "_d2"].astype(np.float32) / 512.0
        mq, mt, md = po.sift_match(d1, d2)
        assert np.array_equal(mq, g[name + "_q"]) and np.array_equal(mt, g[name + "_t"]), name
        assert np.array_equal(md, g[name + "_dist"]), name
        total += len(mq)
    assert total == 520 + 677 + 37
