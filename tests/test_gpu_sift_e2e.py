"""-m gpu: SIFT extraction -> projectTo3DSiftGPU -> matcher -> RANSAC with features from the compiled reference pipeline
(SiftGPU's own kernels + host code on the CPU emulation, oracle/_ref/libref_siftgpu.so) on one side and from
rgbdfe_sift_detect on the other (VERDICT r3 "missing" 2; tools/sift_e2e.py holds the procedure and prints the full
report).  Views: two of the picture pairs SiftGPU ships as test data (tests/golden/sift_photo_pairs.npz) and seeded synthetic
views.  Matching and RANSAC run on the GPU pair path for both sides (bit-equal to the oracle elsewhere), so a difference
here is a difference of the extraction.

What must hold (asserted) and what is only reported is stated in DESIGN.md 4.11."""
import json
import os
import sys

import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

POSE_TOL = 1e-4              # north_star: RANSAC pose for float descriptors
BYTE_FLIP_FRACTION = 1e-3    # measured: 1 .. 29 of 3e5 .. 4e5 quantised bytes (profiles/r04/sift_e2e.json)


@pytest.fixture(scope="module")
def report():
    if po.ref_siftgpu_lib() is None:
        pytest.skip("oracle/_ref/libref_siftgpu.so (the compiled reference pipeline) is not present")
    import sift_e2e
    from rgbdslam_v2_amd.frontend import FrontEnd
    fe = FrontEnd(device_id=0, max_nodes=8, max_keypoints=4096, max_pairs_per_batch=8)
    try:
        rep = sift_e2e.evaluate(fe, po)
    finally:
        fe.close()
    print(json.dumps(rep, indent=1))
    return rep


def test_extraction_hands_the_matcher_the_same_features(report):
    """Feature lists (count, order, positions, the 3-D points projectTo3DSiftGPU makes of them) are identical; of the bytes
    the SiftGPU matcher quantises the descriptors to, at most one in a thousand differs (measured: 1e-5 .. 1e-4), by a few
    steps at most (a feature whose orientation histogram peak moved by a fraction of a bin)."""
    for name, r in report.items():
        assert r["feature_lists_identical"] and r["position_or_point_differences"] == 0, name
        assert r["features_a"][0] == r["features_a"][1] > 300 and r["features_b"][0] == r["features_b"][1] > 300, name
        assert r["quantised_bytes_that_differ"] <= BYTE_FLIP_FRACTION * r["quantised_bytes"], name
        assert r["largest_byte_step"] <= 8 and r["largest_descriptor_relative_l2_difference"] <= 5e-2, name


@pytest.mark.parametrize("variant", ["siftgpu_matcher_normalised", "flann_rootsift"])
def test_edges_and_poses_agree_with_reference_features(report, variant):
    """The same edge decision, the same SET of matches, the same inlier set and a pose within north_star's 1e-4 on every
    pair (measured: 0 .. 2.4e-7, profiles/r04/sift_e2e.json).  Where the match lists are identical in order too -- 6 of the
    8 (view, matcher) cases -- the pose is identical to the bit.  In the other two, descriptors that differ in the 4th digit
    swap neighbours of the distance-sorted list (DMatch.distance is an f32 L2 norm of the raw descriptors,
    sift_gpu_wrapper.cpp:211-217): the same matches in another order, the same consensus set, the final fit summed in
    another order."""
    identical = 0
    for name, r in report.items():
        v = r[variant]
        assert v["edge"] == [True, True], (name, v)
        assert min(v["matches"]) >= 100 and v["matches_only_on_one_side"] == 0, (name, v)
        assert v["inliers"][0] == v["inliers"][1] and v["inliers_only_on_one_side"] == 0, (name, v)
        assert v["pose_max_abs_diff"] <= POSE_TOL, (name, v)
        if v["match_lists_identical"]:
            identical += 1
            assert v["pose_max_abs_diff"] == 0.0 and v["rmse"][0] == v["rmse"][1], (name, v)
    assert identical >= len(report) // 2, identical


def test_the_wrapper_as_written_matches_nothing_on_either_side(report):
    """matcher_type SIFTGPU with the wrapper's "-unn" descriptors (norm ~2) and SiftMatchGPU's byte quantisation
    (SiftMatchCU.cpp:96-99): the bytes wrap around and no mutual-best pair passes the tests -- on both sides alike.
    Recorded as a property of the reference, not fixed."""
    for name, r in report.items():
        v = r["siftgpu_matcher_unn"]
        assert v["edge"] == [False, False] and v["matches"][0] == v["matches"][1] and v["matches"][0] <= 5, (name, v)
