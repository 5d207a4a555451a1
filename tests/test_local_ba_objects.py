"""a44: the graph-level flow of Optimizer::LocalBACameraPointObjects (Optimizer.cc:826-1534).
CPU: the mirror's graph construction (cube_slam_amd/ba_objects.build_graph, arrays) equals the oracle's (oracle/local_ba_objects, pointer
graph) array for array, and the branches of the flow are all taken by the synthetic window.
GPU: the two-stage optimisation through the C-ABI ends at the oracle's poses / points / objects, levels and erase list."""
import numpy as np
import pytest

from cube_slam_amd import ba_objects
from oracle import local_ba_objects as lo
from tests import local_map


def _oracle_graph(cur, params):
    """The oracle's graph before any optimisation (stage-1 problem and levels) -- optimisation replaced by the identity."""
    import oracle.pyoracle as po
    real = (po.ba_optimize, po.ba_errors)
    try:
        def ident(d, it):
            return d["cam_pose"], d["points"], d["cuboid_pose"], {"iterations": 0}
        po.ba_optimize = ident
        po.ba_errors = lambda d: (0.0, np.zeros((len(d["obs_cam"]), 3)), np.zeros((len(d["cobs_cam"]), 4)), np.zeros((len(d["pc_cuboid"]), 3)))
        return lo.local_ba_camera_point_objects(cur, params)
    finally:
        po.ba_optimize, po.ba_errors = real


@pytest.mark.parametrize("seed", [1, 2])
def test_graph_construction_matches_oracle(seed):
    cur, params, _ = local_map.build(seed)
    ref = _oracle_graph(cur, params)
    w = lo.flatten_window(cur)
    g = ba_objects.build_graph(w, params)
    d, r = g["problem"], ref["problem"]
    for k in ("cam_pose", "cam_fixed", "points", "cuboid_pose", "cuboid_scale", "cuboid_flags", "obs_cam", "obs_point", "obs_uv", "obs_inv_sigma2", "obs_ur",
              "cobs_cam", "cobs_cuboid", "cobs_bbox", "cobs_info", "pc_cuboid", "pc_offsets", "pc_points"):
        assert np.array_equal(np.asarray(d[k]), np.asarray(r[k])), k
    for k in ("fx", "fy", "cx", "cy", "huber_mono", "huber_stereo", "huber_obj", "bf", "max_outside_margin_ratio"):
        assert d[k] == r[k], k
    assert np.array_equal(g["cobs_level"], ref["cobs_level"])
    # the window takes every branch
    assert (d["cam_fixed"][:w["n_local"]] == 0).sum() >= 5 and len(d["cam_pose"]) > w["n_local"], "free and fixed key frames"
    assert (w["mp_nobs"] == 1).any() and len(d["points"]) < len(w["mp_id"]), "points with one observation are skipped"
    assert (d["obs_ur"] >= 0).any() and (d["obs_ur"] < 0).any(), "mono and stereo edges"
    assert len(d["cuboid_pose"]) > 5 and np.allclose(d["cobs_info"].max(), 0.25 * w["mo_meas_quality"].max() ** 2, rtol=0.5), "more than five objects halve the weight"
    assert (g["cobs_level"] == 1).sum() >= 1, "an object seen once sits at level 1"
    assert len(g["det_rows"]) < len(w["det_mo"]), "a detection outside the margin is dropped"
    assert 0 < len(d["pc_cuboid"]) < len(d["cuboid_pose"]), "objects with at most ten good points get no unary edge"
    n_pts = np.diff(d["pc_offsets"])
    assert (n_pts <= 42).all() and (n_pts > 10).all(), "the 6 m point is filtered"
    reset = d["cuboid_pose"][:, 1] == np.float32(w["cur_cam_center"][1]) + 1.0
    assert reset.any() and not reset.all(), "height reset from the camera, overridden where more than five points give a centroid"
    assert (d["cuboid_scale"] == np.array(ba_objects.KITTI_OBJECT_HALF_SIZE)).all()


def test_left_right_balancing():
    cur, params, _ = local_map.build(3, left_heavy=True)
    w = lo.flatten_window(cur)
    g = ba_objects.build_graph(w, params)
    lr = w["det_left_right_to_car"][g["det_rows"]]
    assert (lr == 1).sum() > 2 * ((lr != 1).sum())
    q = w["mo_meas_quality"][w["det_mo"][g["det_rows"]]]
    expect = np.where(lr == 1, 0.125, 0.25) * q * q
    assert np.allclose(g["problem"]["cobs_info"][:, 0], expect, rtol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_two_stage_flow_matches_oracle(ctx, seed):
    cur, params, _ = local_map.build(seed)
    ref = lo.local_ba_camera_point_objects(cur, params)
    w = lo.flatten_window(cur)
    got = ba_objects.LocalBACameraPointObjects(w, params, ctx=ctx)
    assert np.array_equal(got["cobs_level"], ref["cobs_level"]) and np.array_equal(got["cobs_level2"], ref["cobs_level2"])
    assert np.array_equal(got["obs_level"], ref["obs_level"]) and 0 < got["obs_level"].sum() < len(got["obs_level"]) // 4
    kf_id, mp_id = w["kf_id"], w["mp_id"]
    assert [(int(kf_id[a]), int(mp_id[b])) for a, b in got["erase"]] == ref["erase"] and len(ref["erase"]) > 0
    assert [int(mp_id[r]) for r in got["point_unwritten"]] == ref["point_unwritten"] and len(ref["point_unwritten"]) > 0, "points the erasures leave with one observation are not written back"
    for i in range(w["n_local"]):
        assert np.allclose(got["kf_pose"][i], ref["kf_pose"][int(kf_id[i])], atol=1e-6, rtol=0)
    for r, x in got["point_pos"].items():
        assert np.allclose(x, ref["point_pos"][int(mp_id[r])], atol=1e-5, rtol=1e-6)  # (the depth of a 30 m point with 3 m of baseline is the loosest number here)
    for i, mid in enumerate(w["mo_id"]):
        assert np.allclose(got["object_pose"][i], ref["object_pose"][int(mid)], atol=1e-5, rtol=0)
    for a, b in zip(got["stats"], ref["stats"]):
        assert a["iterations"] == b["iterations"]
        assert np.allclose(a["chi2_trace"], b["chi2_trace"], rtol=1e-6)
    moved = max(np.abs(got["kf_pose"][i] - w["kf_pose"][i]).max() for i in range(w["n_local"]))
    assert moved > 1e-3, "the optimisation moved the key frames"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_two_stage_flow_matches_the_reference_itself(ctx, seed):
    """The HIP path against the REFERENCE'S OWN Optimizer::LocalBACameraPointObjects (its text, running on its vendored g2o: oracle/_ref/libref_graph.so,
    tests/ref_graph.py) on the same window, without the oracle in between: erase list, which points are written back, poses, points, objects."""
    import os
    from tests import ref_graph as rg
    if not os.path.exists(rg.SO):
        pytest.skip("oracle/_ref/libref_graph.so is built from /root/reference")
    cur, params, extra = local_map.build(seed)
    rg.quantize(cur, params, extra)
    w = lo.flatten_window(cur)
    got = ba_objects.LocalBACameraPointObjects(w, params, ctx=ctx)
    G = rg.Graph(cur, params, extra)
    try:
        G.local_ba_objects(cur)
        kf_id, mp_id = w["kf_id"], w["mp_id"]
        assert sorted((int(kf_id[a]), int(mp_id[b])) for a, b in got["erase"]) == sorted(G.erased()) and len(got["erase"]) > 0
        kid = {k.mnId: k for k in extra["kfs"]}
        for i in range(w["n_local"]):
            T, n, _ = G.kf_pose(kid[int(kf_id[i])])
            To = rg.cvmat_from_pose(got["kf_pose"][i]).astype(np.float64)
            assert n == 1 and np.abs(T[:3, :3] - To[:3, :3]).max() <= 5e-6 and np.abs(T[:3, 3] - To[:3, 3]).max() <= 5e-6 * max(1.0, np.abs(To[:3, 3]).max()), int(kf_id[i])
        mid = {m.mnId: m for m in G.mps}
        unwritten = set(got["point_unwritten"])
        assert len(unwritten) > 0
        for r, x in got["point_pos"].items():
            ref_p, nw, _ = G.mp_pos(mid[int(mp_id[r])])
            assert nw == (0 if r in unwritten else 1), int(mp_id[r])
            if nw:
                assert np.abs(ref_p.astype(np.float64) - x).max() <= 1e-5 * max(1.0, float(np.linalg.norm(x))), (int(mp_id[r]), ref_p, x)
        oid = {o.mnId: o for o in extra["mos"]}
        for i, mn in enumerate(w["mo_id"]):
            st = G.mo_state(oid[int(mn)])
            assert st["writes"] == 1 and np.allclose(st["pose"], got["object_pose"][i], rtol=0, atol=2e-5), (int(mn), np.abs(st["pose"] - got["object_pose"][i]).max())
            assert np.array_equal(st["scale"], got["object_scale"][i])
    finally:
        G.close()


@pytest.mark.gpu
def test_window_without_objects(ctx):
    """Most local windows of a sequence hold no object at all: the flow is then the point-only local BA with the same two stages."""
    cur, params, _ = local_map.build(4, with_objects=False)
    ref = lo.local_ba_camera_point_objects(cur, params)
    w = lo.flatten_window(cur)
    assert len(w["mo_id"]) == 0 and len(w["det_mo"]) == 0
    got = ba_objects.LocalBACameraPointObjects(w, params, ctx=ctx)
    assert np.array_equal(got["obs_level"], ref["obs_level"]) and len(got["cobs_level"]) == 0
    kf_id, mp_id = w["kf_id"], w["mp_id"]
    assert [(int(kf_id[a]), int(mp_id[b])) for a, b in got["erase"]] == ref["erase"]
    for i in range(w["n_local"]):
        assert np.allclose(got["kf_pose"][i], ref["kf_pose"][int(kf_id[i])], atol=1e-6, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("with_objects", [True, False])
def test_cpp_mirror_equals_python_mirror(ctx, tmp_path, with_objects):
    """cube_slam_amd/host/local_ba_objects.hpp (what adapters/Optimizer_hip.cc calls) against cube_slam_amd/ba_objects.py on the same window."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cur, params, _ = local_map.build(2, with_objects=with_objects)
    w = lo.flatten_window(cur)
    types = {"kf_id": np.int64, "kf_pose": np.float64, "mp_pos": np.float64, "mp_nobs": np.int32, "obs_mp": np.int32, "obs_kf": np.int32, "obs_uv": np.float64, "obs_ur": np.float64,
             "obs_inv_sigma2": np.float64, "mo_pose": np.float64, "mo_scale": np.float64, "mo_meas_quality": np.float64, "mo_largest_point_observations": np.int32,
             "up_mo": np.int32, "up_count": np.int32, "up_pos": np.float64, "det_mo": np.int32, "det_kf": np.int32, "det_bbox_2d": np.int32, "det_left_right_to_car": np.int32,
             "det_bbox_vec": np.float64}
    for name, dt in types.items():
        (tmp_path / (name + ".bin")).write_bytes(np.ascontiguousarray(w[name], dt).tobytes())
    sc = [w["n_local"], *w["cur_cam_center"], *np.asarray(params["K"]).reshape(-1), params["img_width"], params["img_height"], params["bf"], params["camera_object_BA_weight"]]
    (tmp_path / "scalars.bin").write_bytes(np.asarray(sc, np.float64).tobytes())
    exe = tmp_path / "local_ba_objects"
    lib_dir = os.path.join(root, "cube_slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", root, os.path.join(root, "tests", "cpp", "local_ba_objects.cpp"), "-o", str(exe), "-L", lib_dir, "-lcubeslam_hip",
                           "-Wl,-rpath," + lib_dir])
    out = subprocess.check_output([str(exe), str(tmp_path)], timeout=300).decode().splitlines()
    tok = {ln.split()[0]: ln.split()[1:] for ln in out}
    got = ba_objects.LocalBACameraPointObjects(w, params, ctx=ctx)

    def fnv(b):
        h = 1469598103934665603
        for x in bytes(b):
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    assert int(tok["levels"][0]) == len(got["obs_level"]) and int(tok["levels"][1], 16) == fnv(got["obs_level"].astype(np.uint8).tobytes())
    assert int(tok["levels"][3], 16) == fnv(got["cobs_level"].astype(np.uint8).tobytes()) and int(tok["levels"][4], 16) == fnv(got["cobs_level2"].astype(np.uint8).tobytes())
    assert [tuple(int(v) for v in e.split(":")) for e in tok["erase"][1:]] == got["erase"] and int(tok["erase"][0]) == len(got["erase"])
    assert int(tok["stats"][0]) == got["stats"][0]["iterations"] and int(tok["stats"][1]) == got["stats"][1]["iterations"]
    assert [int(v) for v in tok["unwritten"][1:]] == got["point_unwritten"] and int(tok["unwritten"][0]) == len(got["point_unwritten"])
    # two runs of the solver differ in the last digits (fp64 atomics in the Hessian accumulation): the same bars as against the oracle
    assert np.allclose([float(v) for v in tok["kf"]], got["kf_pose"].reshape(-1), atol=1e-6, rtol=0)
    assert np.allclose([float(v) for v in tok["points"]], np.concatenate([got["point_pos"][int(r)] for r in got["graph"]["point_rows"]]), atol=1e-5, rtol=1e-6)
    assert np.allclose([float(v) for v in tok["objects"]], got["object_pose"].reshape(-1), atol=1e-5, rtol=0)
