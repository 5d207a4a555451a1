"""The drop-in adapters, run through the REFERENCE'S class interfaces on the MI355X (oracle/_ref/libadapters.so = adapters/*.cc compiled
against the reference's headers, linked with libcubeslam_hip.so), next to the reference's own translation units (oracle/_ref/libref.so):
ORB_SLAM2::ORBextractor::operator() and line_lbd_detect::detect_raw_lines / detect_filter_lines give the same bytes."""
import ctypes as C
import os

import numpy as np
import pytest

from cube_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs(ctx, oracle):
    a, r = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("libadapters.so", "libref.so"))
    if not (os.path.exists(a) and os.path.exists(r)):
        pytest.skip("oracle/_ref/*.so are built where /root/reference exists and travel with the snapshot")
    return C.CDLL(a), C.CDLL(r)


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_orbextractor_adapter_equals_reference_class(libs, oracle):
    adp, ref = libs
    for gray in (synth.cuboid_scene(synth.SEED, n_boxes=3, bg_texture=0.5)["gray"], synth.texture_image(5, 1241, 376)):
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        nf, nl = 1000, 8
        cap = 2 * nf + 64
        out = []
        for lib, fn in ((adp, "adp_orb_extract"), (ref, "ref_orb_extract")):
            kps = np.zeros(cap, oracle.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
            levels = np.zeros(4 * W * H, np.uint8); dims = np.zeros(2 * nl, np.int32)
            args = [nf, C.c_float(1.2), nl, 20, 7, _dp(gray), W, H, _dp(kps), _dp(desc), cap, _dp(levels), _dp(dims)]
            if fn.startswith("adp"):
                tables = np.zeros(4 * nl, np.float32)
                args.append(_dp(tables))
            n = getattr(lib, fn)(*args)
            out.append((n, kps[:n].tobytes(), desc[:n].copy(), levels.copy(), dims.copy()))
        (na, ka, da, la, dima), (nr, kr, dr, lr, dimr) = out
        assert na == nr > 500 and ka == kr and np.array_equal(da, dr)
        assert np.array_equal(dima, dimr) and np.array_equal(la, lr)  # mvImagePyramid
        ext = oracle.ORBextractor(nf, 1.2, nl, 20, 7)
        assert np.allclose(tables[:nl], [1.2 ** i for i in range(nl)], rtol=1e-6)


def test_line_lbd_detect_adapter_equals_reference_class(libs, oracle):
    adp, ref = libs
    for gray in (synth.cuboid_scene(synth.SEED, n_boxes=3, bg_texture=0.5)["gray"], synth.texture_image(9, 200, 150)):
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        cap = 20000
        kl_a = np.zeros(cap, oracle.KEYLINE_DTYPE); kl_r = np.zeros(cap, oracle.KEYLINE_DTYPE)
        filt = np.zeros((cap, 4), np.float32); nfilt = C.c_int()
        na = adp.adp_lsd_keylines(_dp(gray), W, H, C.c_float(15.0), _dp(kl_a), cap, _dp(filt), C.byref(nfilt))
        nr = ref.ref_lsd_keylines(_dp(gray), W, H, _dp(kl_r), cap)
        assert na == nr > 20 and kl_a[:na].tobytes() == kl_r[:nr].tobytes()
        want = oracle.lsd_detect_filter_lines(gray, 15.0)
        assert nfilt.value == len(want) and np.array_equal(filt[:nfilt.value], want)


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# b3: adapters/Optimizer_hip.cc RUN -- its two member functions over a stand-in map (oracle/_ref/libadapter_graph.so: the adapter compiled against the runnable
# stand-ins of KeyFrame / MapPoint / MapObject / Map and linked with libcubeslam_hip.so) against the reference's own function text running on the reference's own
# g2o over the same map (oracle/_ref/libref_graph.so).  Map in, map out: what SetPose / SetWorldPos / EraseMapPointMatch left behind.
ADP_GRAPH_SO = os.path.join(ROOT, "oracle", "_ref", "libadapter_graph.so")


def _adapter_graph():
    from tests import ref_graph as rg
    if not (os.path.exists(ADP_GRAPH_SO) and os.path.exists(rg.SO)):
        pytest.skip("oracle/_ref/libadapter_graph.so / libref_graph.so are built from /root/reference")
    return rg, C.CDLL(ADP_GRAPH_SO)


@pytest.mark.parametrize("seed", [1, 2])
def test_optimizer_adapter_local_ba_equals_the_reference(seed):
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(seed)
    rg.quantize(cur, params, extra)
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        Gr.local_ba_objects(cur)
        A.adp_graph_set_params(1, 0, C.c_double(params["camera_object_BA_weight"]))
        err = C.create_string_buffer(512)
        assert A.adp_graph_local_ba_objects(Ga.h, Ga.kf[id(cur)], 0, None, err, 512) == 0, err.value
        assert sorted(Ga.erased()) == sorted(Gr.erased()) and len(Gr.erased()) > 0
        moved = 0.0
        for k in extra["kfs"]:
            Tr, nr, _ = Gr.kf_pose(k); Ta, na, _ = Ga.kf_pose(k)
            assert na == nr and Ga.kf_markers(k) == Gr.kf_markers(k)   # (a bad covisible key frame keeps its mnBALocalForKF mark in the reference: it is in no list that is reset)
            assert np.abs(Tr[:3, :3] - Ta[:3, :3]).max() <= 5e-6 and np.abs(Tr[:3, 3] - Ta[:3, 3]).max() <= 5e-6 * max(1.0, float(np.abs(Tr[:3, 3]).max())), k.mnId
            moved = max(moved, float(np.abs(Ta - k.T_f32).max()))
        assert moved > 1e-3
        n_unwritten = 0
        for m in Gr.mps:
            pr, nr, ur = Gr.mp_pos(m); pa, na, ua = Ga.mp_pos(m)
            assert (na, ua) == (nr, ur), m.mnId   # written (or left alone: no vertex, or one observation left after the erasures) on both sides
            n_unwritten += int(nr == 0 and len(m.observations) > 1)
            assert np.abs(pr.astype(np.float64) - pa.astype(np.float64)).max() <= 1e-5 * max(1.0, float(np.linalg.norm(pr))), (m.mnId, pr, pa)
        assert n_unwritten > 0
        for o in extra["mos"]:
            sr, sa = Gr.mo_state(o), Ga.mo_state(o)
            assert (sa["writes"], sa["been_optimized"], sa["n_used"], sa["n_filtered"]) == (sr["writes"], sr["been_optimized"], sr["n_used"], sr["n_filtered"]), o.mnId
            assert np.allclose(sa["pose"], sr["pose"], rtol=0, atol=2e-5) and np.array_equal(sa["scale"], sr["scale"])
    finally:
        Gr.close(); Ga.close()


@pytest.mark.parametrize("seed", [1, 3])
def test_optimizer_adapter_dynamic_local_ba_equals_the_reference(seed):
    """Optimizer::LocalBACameraPointObjectsDynamic of adapters/Optimizer_hip.cc -- window gathering from the map, cube_slam_amd/host/local_ba_dynamic.hpp, cs_ba_dyn_*, write-back --
    against the reference's own function text on its own g2o over the same map (tests/local_map_dynamic.py: every branch but the aliased vertex id).  Identical: the observations erased,
    the dynamic points set bad while the window is gathered, which poses / points / velocities were written and how often, every marker field, the bookkeeping fields of the objects.
    Numbers: the tolerances of tests/test_ref_graph_pins.py::test_dynamic_local_ba_equals_reference (several times what the reference's own result moves with its heap layout)."""
    rg, A = _adapter_graph()
    from tests import local_map_dynamic as lmd
    cur, params, extra = lmd.build(seed)
    rg.quantize(cur, params, extra)
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        Gr.local_ba_dynamic(cur)
        A.adp_graph_set_params(1, int(params["build_worldframe_on_ground"]), C.c_double(params["camera_object_BA_weight"]))
        A.adp_graph_set_dyn_params(int(params["ba_dyna_pt_obj_cam"]), int(params["ba_dyna_obj_velo"]), int(params["ba_dyna_obj_cam"]), C.c_double(params["object_velocity_BA_weight"]), 1)
        err = C.create_string_buffer(512)
        assert A.adp_graph_local_ba_dynamic(Ga.h, Ga.kf[id(cur)], 0, 0, None, err, 512) == 0, err.value
        assert sorted(Ga.erased()) == sorted(Gr.erased()) and len(Gr.erased()) > 50
        moved = 0.0
        for k in extra["kfs"]:
            Tr, nr, _ = Gr.kf_pose(k); Ta, na, _ = Ga.kf_pose(k)
            assert na == nr and Ga.kf_markers(k) == Gr.kf_markers(k), k.mnId
            assert np.abs(Tr[:3, :3] - Ta[:3, :3]).max() <= 3e-5 and np.abs(Tr[:3, 3] - Ta[:3, 3]).max() <= 3e-5 * max(1.0, float(np.abs(Tr[:3, 3]).max())), k.mnId
            moved = max(moved, float(np.abs(Ta - k.T_f32).max()))
        assert moved > 1e-3
        n_static = n_dyn = n_bad = 0
        for m in Gr.mps:
            pr, nr, ur = Gr.mp_pos(m); pa, na, ua = Ga.mp_pos(m)
            assert (na, ua) == (nr, ur), m.mnId
            sr, sa = Gr.mp_dynamic(m), Ga.mp_dynamic(m)
            assert (sa["is_optimized"], sa["bad"], sa["local_for"]) == (sr["is_optimized"], sr["bad"], sr["local_for"]), m.mnId
            n_bad += int(sr["bad"])
            if getattr(m, "is_dynamic", False):
                tol = 1e-2 if sr["is_optimized"] else 0.0
                n_dyn += int(sr["is_optimized"])
                assert np.abs(sr["PosToObj"].astype(np.float64) - sa["PosToObj"]).max() <= tol and np.abs(sr["latest"].astype(np.float64) - sa["latest"]).max() <= tol, m.mnId
                assert np.abs(pr.astype(np.float64) - pa).max() <= tol
            else:
                n_static += int(nr)
                assert np.abs(pr.astype(np.float64) - pa.astype(np.float64)).max() <= 2e-4 * max(1.0, float(np.linalg.norm(pr))), (m.mnId, pr, pa)
        assert n_static > 100 and n_dyn > 30 and n_bad >= 1
        n_vel = 0
        for o in extra["mos"]:
            sr, sa = Gr.mo_state(o), Ga.mo_state(o)
            assert (sa["writes"], sa["been_optimized"], sa["n_used"], sa["n_filtered"], sa["point_threshold"]) == (sr["writes"], sr["been_optimized"], sr["n_used"], sr["n_filtered"], sr["point_threshold"]), o.mnId
            assert np.allclose(sa["pose"], sr["pose"], rtol=0, atol=1e-3) and np.array_equal(sa["scale"], sr["scale"])
            dr, da = Gr.mo_dynamic_state(o), Ga.mo_dynamic_state(o)
            assert (da["n_history"], da["local_for"]) == (dr["n_history"], dr["local_for"]), o.mnId
            assert np.allclose(da["latest"], dr["latest"], rtol=0, atol=1e-3) and np.allclose(da["afterba"], dr["afterba"], rtol=0, atol=1e-3)
            assert np.allclose(da["velocity"], dr["velocity"], rtol=0, atol=1e-3) and np.allclose(da["history"], dr["history"], rtol=0, atol=1e-3)
            n_vel += dr["n_history"]
            for kf in getattr(o, "allDynamicPoses", {}):
                (pr, br), (pa, ba) = Gr.mo_dynamic_pose(o, kf), Ga.mo_dynamic_pose(o, kf)
                assert br == ba and np.allclose(pa, pr, rtol=0, atol=1e-3 if br else 0.0), (o.mnId, kf.mnId)
        assert n_vel >= 2
    finally:
        Gr.close(); Ga.close()


@pytest.mark.parametrize("loop_kf", [0, 7])
def test_optimizer_adapter_bundle_adjustment_equals_the_reference(loop_kf):
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(1, n_kf=10, n_points=80, n_cuboids=3)
    rg.quantize(cur, params, extra)
    for k in extra["kfs"]:
        k.bad = False; k.local_cuboids, k.cuboids_landmark = [], []
    extra["mps"], extra["mos"] = [m for m in extra["mps"] if m.observations], []
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        Gr.bundle_adjustment(10, loop_kf=loop_kf)
        err = C.create_string_buffer(512)
        assert A.adp_graph_bundle_adjustment(Ga.h, 10, C.c_ulong(loop_kf), 1, None, err, 512) == 0, err.value
        for k in extra["kfs"]:
            Tr, nr, Gr_gba = Gr.kf_pose(k); Ta, na, Ga_gba = Ga.kf_pose(k)
            assert na == nr == (0 if loop_kf else 1)
            a, b = (Gr_gba, Ga_gba) if loop_kf else (Tr, Ta)
            assert np.abs(a - b).max() <= 2e-6 * max(1.0, float(np.abs(a[:3, 3]).max())), k.mnId
        for m in Gr.mps:
            pr, nr, _ = Gr.mp_pos(m); pa, na, _ = Ga.mp_pos(m)
            assert na == nr and np.abs(pr.astype(np.float64) - pa.astype(np.float64)).max() <= 1e-5 * max(1.0, float(np.linalg.norm(pr)))
    finally:
        Gr.close(); Ga.close()


def test_optimizer_adapter_local_bundle_adjustment_equals_the_reference():
    """Optimizer::LocalBundleAdjustment (Optimizer.cc:474-825, the no-object fallback of the mapping thread, LocalMapping.cc:74) through the adapter against the
    reference's own text on the same window: erase list, key-frame poses, points written or left alone, marker fields."""
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(4, with_objects=False)
    rg.quantize(cur, params, extra)
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        Gr.local_ba(cur)
        err = C.create_string_buffer(512)
        assert A.adp_graph_local_ba(Ga.h, Ga.kf[id(cur)], None, err, 512) == 0, err.value
        assert sorted(Ga.erased()) == sorted(Gr.erased()) and len(Gr.erased()) > 0
        moved = 0.0
        for k in extra["kfs"]:
            Tr, nr, _ = Gr.kf_pose(k); Ta, na, _ = Ga.kf_pose(k)
            assert na == nr and Ga.kf_markers(k) == Gr.kf_markers(k), k.mnId
            assert np.abs(Tr[:3, :3] - Ta[:3, :3]).max() <= 5e-6 and np.abs(Tr[:3, 3] - Ta[:3, 3]).max() <= 5e-6 * max(1.0, float(np.abs(Tr[:3, 3]).max())), k.mnId
            moved = max(moved, float(np.abs(Ta - k.T_f32).max()))
        assert moved > 1e-3
        n_written = 0
        for m in Gr.mps:
            pr, nr, ur = Gr.mp_pos(m); pa, na, ua = Ga.mp_pos(m)
            assert (na, ua) == (nr, ur), m.mnId
            n_written += nr
            assert np.abs(pr.astype(np.float64) - pa.astype(np.float64)).max() <= 1e-5 * max(1.0, float(np.linalg.norm(pr))), (m.mnId, pr, pa)
        assert n_written > 50
    finally:
        Gr.close(); Ga.close()


@pytest.mark.parametrize("loop_kf", [0, 9])
def test_optimizer_adapter_global_bundle_adjustment_equals_the_reference(loop_kf):
    """Optimizer::GlobalBundleAdjustemnt (:57-62; LoopClosing.cc:641, Tracking.cc:1073): the map's key frames and points through BundleAdjustment."""
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(2, n_kf=10, n_points=80, n_cuboids=3)
    rg.quantize(cur, params, extra)
    for k in extra["kfs"]:
        k.bad = False; k.local_cuboids, k.cuboids_landmark = [], []
    extra["mps"], extra["mos"] = [m for m in extra["mps"] if m.observations], []
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        Gr.global_ba(10, loop_kf=loop_kf)
        err = C.create_string_buffer(512)
        assert A.adp_graph_global_ba(Ga.h, 10, C.c_ulong(loop_kf), 1, None, err, 512) == 0, err.value
        for k in extra["kfs"]:
            Tr, nr, Gr_gba = Gr.kf_pose(k); Ta, na, Ga_gba = Ga.kf_pose(k)
            assert na == nr == (0 if loop_kf else 1)
            a, b = (Gr_gba, Ga_gba) if loop_kf else (Tr, Ta)
            assert np.abs(a - b).max() <= 2e-6 * max(1.0, float(np.abs(a[:3, 3]).max())), k.mnId
        for m in Gr.mps:
            pr, nr, _ = Gr.mp_pos(m); pa, na, _ = Ga.mp_pos(m)
            assert na == nr and np.abs(pr.astype(np.float64) - pa.astype(np.float64)).max() <= 1e-5 * max(1.0, float(np.linalg.norm(pr)))
    finally:
        Gr.close(); Ga.close()


def test_optimizer_adapter_pose_optimization_equals_the_reference():
    """int Optimizer::PoseOptimization(Frame *) (:253-472) through the adapter against the reference's own text: a frame made of a key frame's key points and matches,
    started off its pose -- the same inlier count, the same mvbOutlier flags, the pose to a few float ulps."""
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(2)
    rg.quantize(cur, params, extra)
    n_frames = 0
    for k in extra["kfs"][2:8]:
        keep = k.T_f32
        T0 = keep.copy(); T0[:3, 3] += np.float32([0.05, -0.02, 0.08])
        k.T_f32 = T0
        Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
        k.T_f32 = keep
        try:
            n_ref, Tr, out_ref = Gr.pose_optimization(k)
            Ta = np.zeros(16, np.float32); out = np.zeros(len(k.mvKeysUn), np.uint8); err = C.create_string_buffer(512)
            n_adp = A.adp_graph_pose_optimization(Ga.h, Ga.kf[id(k)], Ta.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_ubyte)), err, 512)
            assert n_adp == n_ref > 10, (err.value, n_adp, n_ref)
            assert np.array_equal(out.astype(bool), out_ref) and out_ref.any()
            Ta = Ta.reshape(4, 4)
            assert np.abs(Ta - Tr).max() <= 4e-6 * max(1.0, float(np.abs(Tr[:3, 3]).max())), k.mnId
            assert np.abs(Ta - T0).max() > 1e-3
            n_frames += 1
        finally:
            Gr.close(); Ga.close()
    assert n_frames == 6


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# b1: adapters/detect_3d_cuboid_hip.cpp RUN through the reference's own class (oracle/_ref/libadapter_cuboid.so: the adapter's three member functions under the class
# definition of detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h) next to the reference's own detect_cuboid text (libref.so::ref_detect_cuboid).
@pytest.mark.parametrize("mode", ["default", "height", "config1", "top3", "rollpitch"])
def test_detect_cuboid_adapter_equals_reference_class(libs, oracle, mode):
    import oracle.pyoracle as po
    so = os.path.join(ROOT, "oracle", "_ref", "libadapter_cuboid.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libadapter_cuboid.so is built from /root/reference")
    adp, ref = C.CDLL(so), libs[1]
    total = 0
    for seed in ((1, 19, 36) if mode == "rollpitch" else (synth.SEED, 5, 9)):   # (1, 19, 36: frames on which the box-to-box chain changes a later box)
        s = synth.cuboid_scene(seed, n_boxes=3, bg_texture=0.0 if seed != 9 else 0.5)
        opts = po.cuboid_opts()
        if mode == "height":
            opts.whether_sample_bbox_height = 1
        if mode == "config1":
            opts.consider_config_2 = 0
        if mode == "top3":
            opts.max_cuboid_num = 3
        gray = np.ascontiguousarray(s["gray"], np.uint8); H, W = gray.shape
        K = np.ascontiguousarray(s["K"], np.float64); Twc = np.ascontiguousarray(s["Twc"], np.float64)
        boxes = np.ascontiguousarray(s["boxes"], np.float64).reshape(-1, 5); lines = np.ascontiguousarray(s["lines"], np.float64).reshape(-1, 4)
        if mode == "rollpitch":   # the reference carries the sampled camera pose from box to box (pin D1 of DESIGN.md); cs_cuboid_detect chains the boxes of its frame the same way
            opts.whether_sample_cam_roll_pitch = 1; opts.stateful_cam_pose = 1
        nb = len(boxes)
        want = np.zeros((nb, opts.max_cuboid_num), po.CUBOID_DTYPE); cnt_r = np.zeros(nb, np.int32)
        got = np.zeros((nb, opts.max_cuboid_num), po.CUBOID_DTYPE); cnt_a = np.zeros(nb, np.int32)
        args = [_dp(gray), W, H, _dp(K), _dp(Twc), _dp(boxes), nb, _dp(lines), len(lines), C.byref(opts)]
        assert ref.ref_detect_cuboid(*args, _dp(want), _dp(cnt_r)) == 0
        euler = np.zeros(3); err = C.create_string_buffer(512)
        assert adp.adp_detect_cuboid(*args, _dp(got), _dp(cnt_a), _dp(euler), err, 512) == 0, err.value
        assert np.array_equal(np.minimum(cnt_a, opts.max_cuboid_num), np.minimum(cnt_r, opts.max_cuboid_num)), (seed, cnt_a, cnt_r)
        for b in range(nb):
            for k in range(min(cnt_r[b], opts.max_cuboid_num)):
                g, w = got[b, k], want[b, k]
                for f in po.CUBOID_DTYPE.names:
                    if f in ("box_corners_3d_world", "pos", "scale"):
                        assert np.allclose(g[f], w[f], rtol=1e-12, atol=1e-12), (seed, b, k, f)   # (device sin / cos against libm's in the back-projection)
                    elif f in ("edge_distance_error", "edge_angle_error", "normalized_error", "skew_ratio", "rotY", "box_corners_2d", "camera_roll_delta", "camera_pitch_delta"):
                        assert np.allclose(g[f], w[f], rtol=1e-12, atol=1e-12), (seed, b, k, f, g[f], w[f])
                    else:
                        assert np.array_equal(g[f], w[f]), (seed, b, k, f, g[f], w[f])
                total += 1
        assert np.isfinite(euler).all()
    assert total >= (9 if mode != "top3" else 20), total


def test_optimizer_adapter_honours_a_raised_stop_flag_like_the_reference():
    """*pbStopFlag already true: the reference builds its graph and returns before optimising (Optimizer.cc:1386-1388) -- nothing is written, nothing erased, the marker
    fields stay as the gathering left them; the adapter returns at the same place."""
    rg, A = _adapter_graph()
    from tests import local_map
    cur, params, extra = local_map.build(1)
    rg.quantize(cur, params, extra)
    Gr, Ga = rg.Graph(cur, params, extra), rg.Graph(cur, params, extra)
    try:
        stop = C.c_bool(True)
        Gr.L.ref_graph_local_ba_objects(Gr.h, Gr.kf[id(cur)], 0, 0, C.byref(stop))
        A.adp_graph_set_params(1, 0, C.c_double(params["camera_object_BA_weight"]))
        err = C.create_string_buffer(512)
        assert A.adp_graph_local_ba_objects(Ga.h, Ga.kf[id(cur)], 0, C.byref(stop), err, 512) == 0, err.value
        assert Gr.erased() == [] and Ga.erased() == []
        marked = 0
        for k in extra["kfs"]:
            assert Gr.kf_pose(k)[1] == 0 and Ga.kf_pose(k)[1] == 0
            assert Ga.kf_markers(k) == Gr.kf_markers(k)
            marked += int(Gr.kf_markers(k) != (0, 0))
        assert marked >= 5, "local and fixed key frames keep their marks when the function returns early"
        for m in Gr.mps:
            assert Gr.mp_pos(m)[1] == 0 and Ga.mp_pos(m)[1] == 0
        for o in extra["mos"]:
            assert Gr.mo_state(o)["writes"] == 0 and Ga.mo_state(o)["writes"] == 0
    finally:
        Gr.close(); Ga.close()
