"""(Runs late on purpose -- file name -- so that `pytest -x` reaches every bit-exact path first: this is the one engine with run-dependent low bits.)
GPU parity of the dynamic-object bundle adjustment (cs_ba_dyn_*, cube_slam_amd/csrc/badyn.hip) against the oracle (oracle/badyn_oracle.cpp):
residuals bit-for-bit up to libm, the reduced pose system, the LM trace of optimize() and the two-stage flow of
Optimizer::LocalBACameraPointObjectsDynamic (Optimizer.cc:2353-2415).  Tolerance: 1e-5 relative on chi2 / residuals / estimates (BASELINE
north_star's floating-point bar); the fp64 atomics reorder sums, so bit-equality is not expected beyond the residuals."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.ba_dynamic import DynamicBundleAdjuster, optimize_two_stages as LocalBACameraPointObjectsDynamic, second_stage_problem

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-5):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-12) if a.size else True


@pytest.mark.parametrize("kw", [dict(), dict(fix_points=True), dict(fix_cams=True), dict(stereo_frac=0.0, n_objects=1)])
def test_residuals_and_reduced_system(ctx, oracle, kw):
    kw = dict(dict(n_kf=8, n_points=200, n_objects=2, pts_per_obj=16), **kw)
    d = synth.ba_dyn_problem(31, **kw)
    ba = DynamicBundleAdjuster(d, ctx=ctx)
    chi, e = ba.errors()
    chi_o, e_o = oracle.badyn_errors(d)
    assert np.isclose(chi, chi_o, rtol=1e-10)
    for k in e_o:
        assert e[k].shape == e_o[k].shape and np.allclose(e[k], e_o[k], rtol=1e-9, atol=1e-9), k
    H, b = ba.reduced_dense(2e-3)
    H_o, b_o = oracle.badyn_reduced_dense(d, 2e-3)
    assert H.shape == H_o.shape and H.shape[0] > 0
    assert np.abs(H - H_o).max() <= 1e-8 * np.abs(H_o).max() and np.abs(b - b_o).max() <= 1e-8 * np.abs(b_o).max()
    ba.close()


@pytest.mark.parametrize("seed,kw", [(41, dict()), (42, dict(fix_points=True)), (43, dict(n_kf=14, n_points=500, n_objects=4, pts_per_obj=25))])
def test_optimize_matches_oracle(ctx, oracle, seed, kw):
    kw = dict(dict(n_kf=8, n_points=200, n_objects=2, pts_per_obj=16), **kw)
    d = synth.ba_dyn_problem(seed, **kw)
    ba = DynamicBundleAdjuster(d, ctx=ctx)
    st = ba.optimize(6)
    res = ba.read()
    res_o, st_o = oracle.badyn_optimize(d, 6)
    assert st["iterations"] == st_o["iterations"] and st["lm_trials"] == st_o["lm_trials"]
    assert np.isclose(st["chi2_init"], st_o["chi2_init"], rtol=1e-10) and _close(st["chi2_trace"], st_o["chi2_trace"]) and np.isclose(st["lambda_final"], st_o["lambda_final"], rtol=1e-4)
    assert st["chi2_final"] < 0.3 * st["chi2_init"]
    for k in res_o:  # the build adds the edges' pieces in edge order, like the oracle: no atomics, reproducible; 1e-5 like chi2
        assert _close(res[k], res_o[k], 1e-5), k
    chi, _ = ba.errors()
    assert np.isclose(chi, st["chi2_final"], rtol=1e-9), "the residuals on the device are those of the accepted state"
    ba.close()


def test_two_stage_local_ba(ctx, oracle):
    d = dict(synth.ba_dyn_problem(51, n_kf=10, n_points=300, n_objects=3, pts_per_obj=20))
    d["obs_uv"] = d["obs_uv"].copy(); d["obs_uv"][::31] += 40.0
    d["dobs_uv"] = d["dobs_uv"].copy(); d["dobs_uv"][::19] -= 35.0
    res, d2, (st1, st2) = LocalBACameraPointObjectsDynamic(d, ctx=ctx)
    r1, s1 = oracle.badyn_optimize(d, 5)
    assert st1["iterations"] == s1["iterations"] and _close(st1["chi2_trace"], s1["chi2_trace"])
    d1 = dict(d); d1.update(r1)
    o2 = second_stage_problem(d1, oracle.badyn_errors(d1)[1])
    # the levels come from chi2 thresholds at two estimates that agree to ~1e-6: an edge sitting on a threshold may differ
    for k in ("obs_level", "dobs_level", "cobs_level"):
        assert (d2[k] != o2[k]).sum() <= 2, k
    assert d2["obs_level"][::31].mean() > 0.9 and d2["dobs_level"][::19].mean() > 0.8
    # second stage: the oracle on the problem the GPU path built (its own first-stage estimates and levels)
    r2, s2 = oracle.badyn_optimize(d2, 10)
    assert st2["iterations"] == s2["iterations"] and st2["lm_trials"] == s2["lm_trials"]
    assert _close(st2["chi2_trace"], s2["chi2_trace"])
    for k in r2:
        assert _close(res[k], r2[k], 1e-5), k
    err0 = np.abs(d["cam_pose"][:, :3] - d["cam_true"][:, :3]).max(); err2 = np.abs(res["cam_pose"][:, :3] - d["cam_true"][:, :3]).max()
    assert err2 < 0.5 * err0


def test_rejects_bad_graph(ctx):
    d = dict(synth.ba_dyn_problem(61, n_kf=5, n_points=40, n_objects=1, pts_per_obj=8))
    d["dobs_obj"] = d["dobs_obj"].copy(); d["dobs_obj"][0] = len(d["obj_pose"])
    with pytest.raises(RuntimeError):
        DynamicBundleAdjuster(d, ctx=ctx)


@pytest.mark.parametrize("kw", [dict(objects=False), dict(dynamic=False), dict(static=False), dict(static=False, dynamic=False)])
def test_empty_edge_classes(ctx, oracle, kw):
    """Windows without cars (a plain local BA), without dynamic points, without static points: empty arrays on every class."""
    d = synth.ba_dyn_strip(synth.ba_dyn_problem(23, n_kf=6, n_points=80, n_objects=2, pts_per_obj=10), **kw)
    ba = DynamicBundleAdjuster(d, ctx=ctx)
    chi, _ = ba.errors()
    assert np.isclose(chi, oracle.badyn_errors(d)[0], rtol=1e-10)
    H, b = ba.reduced_dense(1e-2)
    H_o, b_o = oracle.badyn_reduced_dense(d, 1e-2)
    assert H.shape == H_o.shape and np.abs(H - H_o).max() <= 1e-8 * np.abs(H_o).max() and np.abs(b - b_o).max() <= 1e-8 * max(np.abs(b_o).max(), 1e-300)
    st = ba.optimize(4)
    res = ba.read()
    res_o, st_o = oracle.badyn_optimize(d, 4)
    assert st["iterations"] == st_o["iterations"] and _close(st["chi2_trace"], st_o["chi2_trace"])
    for k in res_o:
        assert res[k].shape == res_o[k].shape and _close(res[k], res_o[k], 1e-5), k
    ba.close()


def test_two_runs_are_bit_identical(ctx):
    """No floating-point atomics anywhere in the build: the same graph gives the same bits."""
    d = synth.ba_dyn_problem(47, n_kf=10, n_points=300, n_objects=3, pts_per_obj=20)
    out = []
    for _ in range(2):
        ba = DynamicBundleAdjuster(d, ctx=ctx)
        st = ba.optimize(5)
        res = ba.read()
        ba.close()
        out.append((st["chi2_trace"], {k: v.tobytes() for k, v in res.items()}))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert out[0][1][k] == out[1][1][k], k
