"""CPU: the per-item bodies the dynamic-BA kernels wrap (cube_slam_amd/csrc/badyn_math.h), compiled with g++ and run serially by
tests/cpp/badyn_items.cpp, against the oracle: residuals, robust chi2, the reduced pose system and one damped step.  (The product runs these
bodies only inside HIP kernels; this harness is test infrastructure.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.ba_dynamic import problem_struct, second_stage_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def items(tmp_path_factory):
    so = tmp_path_factory.mktemp("badyn") / "badyn_items.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "badyn_items.cpp"), "-o", str(so)])
    return C.CDLL(str(so))


def _reduced(items, d, lam):
    p = problem_struct(d)
    chi = C.c_double()
    n_e = p.n_obs * 3 + p.n_dobs * 2 + p.n_mot * 3 + p.n_cobs * 4 + p.n_pc * 3 + p.n_dpoints * 3
    errs = np.zeros(max(n_e, 1))
    n = items.badyn_items_reduced(C.byref(p), C.c_double(lam), C.byref(chi), errs.ctypes.data_as(C.c_void_p), None, None)
    S = np.zeros((max(n, 1), max(n, 1))); bs = np.zeros(max(n, 1))
    items.badyn_items_reduced(C.byref(p), C.c_double(lam), C.byref(chi), errs.ctypes.data_as(C.c_void_p), S.ctypes.data_as(C.c_void_p), bs.ctypes.data_as(C.c_void_p))
    return chi.value, errs[:n_e], S[:n, :n], bs[:n]


def _step(items, d, lam):
    p = problem_struct(d)
    n = p.n_cams * 7 + p.n_objs * 7 + p.n_vels * 2 + p.n_points * 3 + p.n_dpoints * 3
    st = np.zeros(n)
    rc = items.badyn_items_step(C.byref(p), C.c_double(lam), st.ctypes.data_as(C.c_void_p))
    o = np.cumsum([0, p.n_cams * 7, p.n_objs * 7, p.n_vels * 2, p.n_points * 3, p.n_dpoints * 3])
    shapes = (7, 7, 2, 3, 3)
    return rc, {k: st[o[i]:o[i + 1]].reshape(-1, shapes[i]) for i, k in enumerate(("cam_pose", "obj_pose", "vel", "points", "dpoints"))}


@pytest.mark.parametrize("kw", [dict(), dict(fix_points=True), dict(fix_cams=True), dict(stereo_frac=0.0)])
def test_items_match_oracle(items, oracle, kw):
    d = synth.ba_dyn_problem(21, n_kf=7, n_points=120, n_objects=2, pts_per_obj=14, **kw)
    chi_o, e_o = oracle.badyn_errors(d)
    chi, errs, S, bs = _reduced(items, d, 3e-3)
    cat = np.concatenate([e_o[k].reshape(-1) for k in ("obs", "dobs", "mot", "cobs", "pc", "ulp")])
    assert np.allclose(errs, cat, rtol=1e-12, atol=1e-12) and np.isclose(chi, chi_o, rtol=1e-12)
    H_o, b_o = oracle.badyn_reduced_dense(d, 3e-3)
    assert S.shape == H_o.shape and S.shape[0] > 0
    assert np.abs(S - H_o).max() <= 1e-9 * np.abs(H_o).max() and np.abs(bs - b_o).max() <= 1e-9 * np.abs(b_o).max()
    rc, st = _step(items, d, 3e-3)
    so, rco = oracle.badyn_step(d, 3e-3)
    assert rc == rco == 0
    for k in st:
        assert np.allclose(st[k], so[k], rtol=1e-8, atol=1e-9), k
    assert not np.array_equal(st["obj_pose"], d["obj_pose"])


def test_items_second_stage(items, oracle):
    d = dict(synth.ba_dyn_problem(22, n_kf=7, n_points=120, n_objects=2, pts_per_obj=14))
    d["obs_uv"] = d["obs_uv"].copy(); d["obs_uv"][::29] += 35.0; d["dobs_uv"] = d["dobs_uv"].copy(); d["dobs_uv"][::17] -= 30.0  # wrong matches
    res, _ = oracle.badyn_optimize(d, 3)
    d1 = dict(d); d1.update(res)
    d2 = second_stage_problem(d1, oracle.badyn_errors(d1)[1])
    assert d2["obs_level"].sum() > 0 and d2["dobs_level"].sum() > 0
    chi, errs, S, bs = _reduced(items, d2, 1e-2)
    chi_o, _ = oracle.badyn_errors(d2)
    H_o, b_o = oracle.badyn_reduced_dense(d2, 1e-2)
    assert np.isclose(chi, chi_o, rtol=1e-12) and np.abs(S - H_o).max() <= 1e-9 * np.abs(H_o).max() and np.abs(bs - b_o).max() <= 1e-9 * np.abs(b_o).max()


@pytest.mark.parametrize("kw", [dict(objects=False), dict(dynamic=False), dict(static=False), dict(static=False, dynamic=False)])
def test_items_empty_classes(items, oracle, kw):
    """Windows without cars (a static local BA), without dynamic points, without static points: every edge class may be empty."""
    d = synth.ba_dyn_strip(synth.ba_dyn_problem(23, n_kf=6, n_points=80, n_objects=2, pts_per_obj=10), **kw)
    chi_o, _ = oracle.badyn_errors(d)
    chi, errs, S, bs = _reduced(items, d, 1e-2)
    H_o, b_o = oracle.badyn_reduced_dense(d, 1e-2)
    assert np.isclose(chi, chi_o, rtol=1e-12) and S.shape == H_o.shape and S.shape[0] > 0
    assert np.abs(S - H_o).max() <= 1e-9 * np.abs(H_o).max() and np.abs(bs - b_o).max() <= 1e-9 * max(np.abs(b_o).max(), 1e-300)
    rc, st = _step(items, d, 1e-2)
    so, rco = oracle.badyn_step(d, 1e-2)
    assert rc == rco == 0
    for k in st:
        assert st[k].shape == so[k].shape and np.allclose(st[k], so[k], rtol=1e-8, atol=1e-9), k
    res, stt = oracle.badyn_optimize(d, 4)
    assert stt["chi2_final"] < stt["chi2_init"]
