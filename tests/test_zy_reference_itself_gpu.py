"""The HIP paths against the REFERENCE'S OWN functions, without the oracle in between: Optimizer::BundleAdjustment and Optimizer::PoseOptimization as text, running on the
reference's vendored g2o compiled whole (oracle/_ref/libref_graph.so, built here from /root/reference and carried to the GPU box; tests/ref_graph.py).  The
two-stage local BA with objects has the same kind of test in tests/test_local_ba_objects.py.  The reference stores poses and points as float cv::Mat: that is the
precision of the comparison."""
import math
import os

import numpy as np
import pytest

from cube_slam_amd.ba import BundleAdjuster
from cube_slam_amd.optimizer import PoseOptimization
from tests import local_map
from tests import ref_graph as rg
from tests.test_ref_graph_pins import _all_frames_problem, _float_close

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(rg.SO), reason="oracle/_ref/libref_graph.so is built from /root/reference")]


@pytest.mark.parametrize("iterations,robust", [(10, True), (20, False)])
def test_bundle_adjustment_equals_the_reference_itself(ctx, iterations, robust):
    cur, params, extra = local_map.build(1, n_kf=10, n_points=80, n_cuboids=3)
    rg.quantize(cur, params, extra)
    for k in extra["kfs"]:
        k.bad = False; k.local_cuboids, k.cuboids_landmark = [], []
    kfs, mps = extra["kfs"], [m for m in extra["mps"] if m.observations]
    extra["mps"], extra["mos"] = mps, []
    d = _all_frames_problem(kfs, mps, params, float(np.float32(math.sqrt(5.99))) if robust else 0.0)
    ba = BundleAdjuster(d, ctx=ctx)
    st = ba.optimize(iterations)
    cam, pts, _ = ba.read()
    ba.close()
    assert st["iterations"] >= 5
    G = rg.Graph(cur, params, extra)
    try:
        G.bundle_adjustment(iterations, robust=robust)
        moved = 0.0
        for i, k in enumerate(kfs):
            T, n, _ = G.kf_pose(k)
            assert n == 1 and _float_close(T, rg.cvmat_from_pose(cam[i]), ulps=8), k.mnId
            moved = max(moved, float(np.abs(T - k.T_f32).max()))
        assert moved > 1e-2
        for j, m in enumerate(mps):
            got, nw, _ = G.mp_pos(m)
            assert nw == 1 and np.abs(got.astype(np.float64) - pts[j]).max() <= 2e-6 * max(1.0, float(np.linalg.norm(pts[j])))
    finally:
        G.close()


def test_pose_optimization_equals_the_reference_itself(ctx):
    cur, params, extra = local_map.build(2)
    rg.quantize(cur, params, extra)
    K = params["K"]
    frames, starts, rows_of = [], [], []
    for k in extra["kfs"][2:8]:
        rows = [i for i, m in enumerate(k.map_point_matches) if m is not None]
        T0 = k.T_f32.copy(); T0[:3, 3] += np.float32([0.05, -0.02, 0.08])
        frames.append({"Xw": np.array([k.map_point_matches[i].pos for i in rows]), "obs": np.array([[k.mvKeysUn[i][0], k.mvKeysUn[i][1], k.mvuRight[i] if k.mvuRight[i] >= 0 else -1.0] for i in rows]),
                       "inv_sigma2": np.array([k.mvInvLevelSigma2[k.octave[i]] for i in rows]), "intr": (K[0, 0], K[1, 1], K[0, 2], K[1, 2], params["bf"]), "pose": rg.pose_from_cvmat(T0)})
        starts.append(T0); rows_of.append(rows)
    got = PoseOptimization(frames, ctx=ctx)
    for k, T0, rows, (pose, flags, ninl) in zip(extra["kfs"][2:8], starts, rows_of, got):
        keep = k.T_f32; k.T_f32 = T0
        H = rg.Graph(cur, params, extra)
        k.T_f32 = keep
        try:
            n_ref, T, out_ref = H.pose_optimization(k)
        finally:
            H.close()
        assert n_ref == ninl and np.array_equal(out_ref[rows], np.asarray(flags, bool))
        assert _float_close(T, rg.cvmat_from_pose(pose), ulps=8), k.mnId


@pytest.mark.parametrize("seed", [1, 3])
def test_dynamic_local_ba_equals_the_reference_itself(ctx, seed):
    """The HIP dynamic-object BA from the map to the values written back, all of it product code -- the window as flat arrays (tests/local_map_dynamic.flatten_window: the pointer walk
    an adapter does), cube_slam_amd.ba_dynamic.LocalBACameraPointObjectsDynamic (graph construction, two stages over cs_ba_dyn_*, erase list) -- against the reference's own
    Optimizer::LocalBACameraPointObjectsDynamic on the same window: the observations erased, key-frame poses, per-frame object poses, velocities, dynamic points."""
    from cube_slam_amd.ba_dynamic import LocalBACameraPointObjectsDynamic
    from tests import local_map_dynamic as lmd
    cur, params, extra = lmd.build(seed)
    rg.quantize(cur, params, extra)
    G = rg.Graph(cur, params, extra)
    w, rows = lmd.flatten_window(cur)
    res = LocalBACameraPointObjectsDynamic(w, params, ctx=ctx)
    try:
        G.local_ba_dynamic(cur)
        erase = sorted((rows["kfs"][k].mnId, rows["points"][r].mnId) for k, r in res["erase"])
        assert erase == sorted(G.erased()) and len(erase) > 50
        for i, k in enumerate(rows["kfs"][:int(w["n_local"])]):
            T, n, _ = G.kf_pose(k)
            To = rg.cvmat_from_pose(res["kf_pose"][i]).astype(np.float64)
            assert n == 1 and np.abs(T[:3, :3] - To[:3, :3]).max() <= 3e-5 and np.abs(T[:3, 3] - To[:3, 3]).max() <= 3e-5 * max(1.0, np.abs(To[:3, 3]).max()), k.mnId
        for r, p in res["point_pos"].items():
            got, nw, _ = G.mp_pos(rows["points"][r])
            assert nw == (0 if r in res["point_unwritten"] else 1), rows["points"][r].mnId
            if nw:
                assert np.abs(got.astype(np.float64) - p).max() <= 2e-4 * max(1.0, float(np.linalg.norm(p)))
        for v, (mo, kf) in enumerate(rows["ov_key"]):
            got, baed = G.mo_dynamic_pose(mo, kf)
            assert baed and np.allclose(got, res["vertex_pose"][v], rtol=0, atol=1e-3), (mo.mnId, kf.mnId, np.abs(got - res["vertex_pose"][v]).max())
        for i, v in res["object_latest"].items():
            assert np.allclose(G.mo_dynamic_state(rows["objects"][i])["latest"], res["vertex_pose"][v], rtol=0, atol=1e-3)
        assert len(res["velocity"]) >= 2
        for i, vel in res["velocity"].items():
            assert np.allclose(G.mo_dynamic_state(rows["objects"][i])["velocity"], vel, rtol=0, atol=1e-3)
        assert len(res["dpoint_local"]) > 30
        for r, p in res["dpoint_local"].items():
            s = G.mp_dynamic(rows["points"][r])
            assert s["is_optimized"] and np.abs(s["PosToObj"].astype(np.float64) - p).max() <= 1e-2, rows["points"][r].mnId   # (tests/test_ref_graph_pins.py: several times what the reference's own result moves with the heap layout)
            assert np.abs(s["latest"].astype(np.float64) - res["dpoint_world"][r]).max() <= 1e-2
    finally:
        G.close()
