"""Sensitivity of the pair op to the third-party arithmetic that cannot be pinned here (Eigen 3.2 JacobiSVD / LLT,
PCL 1.7 TransformationFromCorrespondences; VERDICT r1 "what's weak" 1).  tools/sensitivity.py re-runs bench-step pairs
under alternative roundings of every restated piece and under a fused-multiply-add build of the same code; this test
pins the conclusions DESIGN.md 3.1 draws from the full 4000-pair study (profiles/r02_sensitivity/) on a sample:

  * whenever nothing discrete flips (same final inlier set, pose fitted from the same inlier set) the pose stays
    within 1e-4 of the restatement -- the tolerance north_star states for the RANSAC pose -- in fact within 1e-5;
  * the edge decision (id1 >= 0) never changes;
  * at most 1 % of the pairs change their inlier set at all;
  * the double-precision Mahalanobis distance (LLT + solves, misc.cpp:763) is insensitive: no pair changes.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import sensitivity  # noqa: E402


@pytest.fixture(scope="module")
def study():
    return sensitivity.study(n_pairs=240, depth_noise=0.01)


def test_variants_cover_every_unpinned_piece(study):
    names = " | ".join(r["variant"] for r in study["rows"])
    for piece in ("LLT", "triangular", "Jacobi sweep", "Jacobi threshold", "pre-scaling", "PCL covariance", "R = U"):
        assert piece in names
    if sensitivity.has_fma():
        assert "fused multiply-adds" in names
    assert study["edges_in_baseline"] > 200


def test_pose_within_tolerance_when_nothing_flips(study):
    for r in study["rows"]:
        assert r["max_pose_dev_nothing_flipped"] <= 1e-4, r       # north_star tolerance
        assert r["max_pose_dev_nothing_flipped"] <= 1e-5, r       # what is actually observed: a few float ulps
        assert r["nothing_flipped_pct"] >= 99.0, r


def test_edge_decisions_do_not_depend_on_the_unpinned_arithmetic(study):
    for r in study["rows"]:
        assert r["edge_decision_kept_pct"] == 100.0, r


def test_double_precision_scoring_is_insensitive(study):
    for r in study["rows"]:
        if r["flags"] in (0x001, 0x002, 0x080):
            assert r["inlier_set_kept_pct"] == 100.0 and r["max_pose_dev_any_edge"] == 0.0, r
