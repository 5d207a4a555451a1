"""CPU tests of the PoseOptimization oracle (oracle/ba_oracle.cpp: orc_pose_optimization)."""
import numpy as np

from cube_slam_amd import synth


def _rot_err(qa, qb):
    d = abs(float(np.dot(qa, qb)))
    return 2 * np.arccos(min(1.0, d))


def test_recovers_pose_and_flags_outliers(oracle):
    fr = synth.pose_frame(3, n=600, outlier_frac=0.2)
    pose, flags, ninl = oracle.pose_optimization(fr["Xw"], fr["obs"], fr["inv_sigma2"], fr["intr"], fr["pose"])
    assert np.linalg.norm(pose[:3] - fr["pose_true"][:3]) < 0.05 < np.linalg.norm(fr["pose"][:3] - fr["pose_true"][:3]) + 0.05
    assert _rot_err(pose[3:], fr["pose_true"][3:]) < 2e-3
    assert abs(np.linalg.norm(pose[3:]) - 1) < 1e-12
    # gross outliers are flagged; hardly any true inlier is (chi2 5.991 = 95 % of a 2-dof chi-square)
    assert flags[fr["is_outlier"]].mean() > 0.97
    assert flags[~fr["is_outlier"]].mean() < 0.1
    assert ninl == len(flags) - int(flags.sum())


def test_stereo_edges_and_small_inputs(oracle):
    fr = synth.pose_frame(4, n=300, outlier_frac=0.1, stereo_frac=0.5)
    pose, flags, ninl = oracle.pose_optimization(fr["Xw"], fr["obs"], fr["inv_sigma2"], fr["intr"], fr["pose"])
    assert np.linalg.norm(pose[:3] - fr["pose_true"][:3]) < 0.05 and flags[fr["is_outlier"]].mean() > 0.9
    # fewer than 3 correspondences: pose untouched, 0 returned (Optimizer.cc:385-386)
    p2, f2, n2 = oracle.pose_optimization(fr["Xw"][:2], fr["obs"][:2], fr["inv_sigma2"][:2], fr["intr"], fr["pose"])
    assert n2 == 0 and np.allclose(p2, fr["pose"] / np.r_[1, 1, 1, [np.linalg.norm(fr["pose"][3:])] * 4])
    # fewer than 10 edges: a single round (`optimizer.edges().size() < 10`)
    p3, f3, n3 = oracle.pose_optimization(fr["Xw"][:8], fr["obs"][:8], fr["inv_sigma2"][:8], fr["intr"], fr["pose"])
    assert 0 <= n3 <= 8 and np.all(np.isfinite(p3))


def test_exact_data_gives_exact_pose(oracle):
    fr = synth.pose_frame(5, n=200, outlier_frac=0.0, noise_px=0.0)
    # undo the float rounding of the generator for this test: project the (float) points exactly
    from cube_slam_amd.synth import _quat_from_R  # noqa: F401
    pose, flags, ninl = oracle.pose_optimization(fr["Xw"], fr["obs"], fr["inv_sigma2"], fr["intr"], fr["pose"])
    assert ninl == 200 and flags.sum() == 0
    assert np.linalg.norm(pose[:3] - fr["pose_true"][:3]) < 1e-3
