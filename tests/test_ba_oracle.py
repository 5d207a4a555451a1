"""CPU known-answer / property tests of the object-BA oracle (oracle/ba_oracle.cpp): residual definitions, robust chi2,
the LM loop and the additivity of the reduced camera system over landmark shards (what the multi-GPU path relies on)."""
import numpy as np

from cube_slam_amd import synth
from cube_slam_amd.ba import shard_landmarks


def _tiny(n_kf=6, n_points=120, n_cuboids=2, seed=5, **kw):
    return synth.ba_problem(seed, n_kf=n_kf, n_points=n_points, n_cuboids=n_cuboids, **kw)


def test_reprojection_error_known_answer(oracle):
    """EdgeSE3ProjectXYZ::computeError = obs - project(T * X) (types_six_dof_expmap.h:164-193)."""
    d = _tiny()
    d = dict(d)
    d["cam_pose"] = np.array([[0, 0, 0, 0, 0, 0, 1.0], [1.0, 0, 0, 0, 0, 0, 1.0]])  # tx ty tz qx qy qz qw (SE3Quat::toVector)
    d["cam_fixed"] = np.array([1, 0], np.uint8)
    d["points"] = np.array([[0.0, 0.0, 2.0], [1.0, -0.5, 4.0]])
    d["obs_cam"] = np.array([0, 1, 1], np.int32); d["obs_point"] = np.array([0, 0, 1], np.int32)
    fx, fy, cx, cy = d["fx"], d["fy"], d["cx"], d["cy"]
    d["obs_uv"] = np.array([[cx + 1.0, cy], [cx, cy], [cx, cy]])
    d["obs_inv_sigma2"] = np.array([1.0, 1.0, 0.5])
    d["huber_mono"] = 0.0
    for k in ("cobs_cam", "cobs_cuboid"):
        d[k] = np.zeros(0, np.int32)
    d["cobs_bbox"] = np.zeros((0, 4)); d["cobs_info"] = np.zeros((0, 4))
    d["pc_cuboid"] = np.zeros(0, np.int32); d["pc_offsets"] = np.zeros(1, np.int32); d["pc_points"] = np.zeros((0, 3))
    chi, eo, _, _ = oracle.ba_errors(d)
    exp = np.array([[1.0, 0.0], [-fx * 1.0 / 2.0, 0.0], [-fx * 2.0 / 4.0, fy * 0.5 / 4.0]])
    assert np.allclose(eo, exp, rtol=0, atol=1e-12)
    assert np.isclose(chi, (exp[0] ** 2).sum() + (exp[1] ** 2).sum() + 0.5 * (exp[2] ** 2).sum(), rtol=1e-14)


def test_huber_chi2_matches_definition(oracle):
    """RobustKernelHuber (robust_kernel_impl.cpp:78-91): rho = e for e <= delta^2 else 2 delta sqrt(e) - delta^2."""
    d = _tiny(noise_px=6.0)
    chi, eo, ec, ep = oracle.ba_errors(d)
    e2 = (eo ** 2).sum(1) * d["obs_inv_sigma2"]
    dl = d["huber_mono"]
    rho = np.where(e2 <= dl * dl, e2, 2 * dl * np.sqrt(e2) - dl * dl)
    e2c = ((ec ** 2) * d["cobs_info"]).sum(1)
    do = d["huber_obj"]
    rhoc = np.where(e2c <= do * do, e2c, 2 * do * np.sqrt(e2c) - do * do)
    e2p = (ep ** 2).sum(1)  # EdgePointCuboidOnlyObject: identity information, no kernel
    assert (e2 > dl * dl).any(), "the sample must exercise the robust branch"
    assert np.isclose(chi, rho.sum() + rhoc.sum() + e2p.sum(), rtol=1e-12)


def test_lm_converges_on_noise_free_problem(oracle):
    d = _tiny(n_kf=8, n_points=200, n_cuboids=0, noise_px=0.0)
    cam, pts, cub, st = oracle.ba_optimize(d, 15)
    tr = st["chi2_trace"]
    assert st["chi2_init"] > 1.0 and all(b <= a * (1 + 1e-12) for a, b in zip([st["chi2_init"]] + tr[:-1], tr)), "chi2 never increases (rejected trials are rolled back)"
    assert st["chi2_final"] < 1e-6 * st["chi2_init"]
    assert st["lm_trials"] >= st["iterations"]
    d2 = dict(d); d2["cam_pose"], d2["points"] = cam, pts
    chi, _, _, _ = oracle.ba_errors(d2)
    assert np.isclose(chi, st["chi2_final"], rtol=1e-9, atol=1e-12)
    assert np.allclose(cam[0], d["cam_pose"][0]), "the fixed keyframe does not move"
    assert np.allclose(np.linalg.norm(cam[:, 3:], axis=1), 1.0, atol=1e-12)


def test_with_cuboids_decreases_and_keeps_fixed_scale(oracle):
    d = _tiny(n_kf=10, n_points=300, n_cuboids=3)
    cam, pts, cub, st = oracle.ba_optimize(d, 10)
    assert st["chi2_final"] < st["chi2_init"]
    assert cub.shape == (3, 7) and np.allclose(np.linalg.norm(cub[:, 3:], axis=1), 1.0, atol=1e-12)


def test_reduced_system_is_additive_over_landmark_shards(oracle):
    """sum_r S_r (pose edges and lambda on shard 0 only) == S: the quantity the ranks all-reduce (DESIGN.md multi-GPU)."""
    d = _tiny(n_kf=7, n_points=150, n_cuboids=2)
    L = len(d["points"]); lam = 0.37
    H, b = oracle.ba_reduced_dense(d, 0, L, True, lam)
    assert np.allclose(H, H.T, rtol=1e-12, atol=1e-9)
    for world in (2, 3):
        Hs = np.zeros_like(H); bs = np.zeros_like(b)
        for r in range(world):
            lo, hi = shard_landmarks(L, r, world)
            Hr, br = oracle.ba_reduced_dense(d, lo, hi, r == 0, lam)
            Hs += Hr; bs += br
        assert np.allclose(Hs, H, rtol=1e-10, atol=1e-8 * np.abs(H).max()) and np.allclose(bs, b, rtol=1e-10, atol=1e-8 * np.abs(b).max())
    w = np.linalg.eigvalsh(H)
    assert w.min() > 0, "damped reduced camera system is positive definite"
