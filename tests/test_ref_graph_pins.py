"""Graph-level pins of the BA oracles against the REFERENCE'S OWN FUNCTIONS run on the reference's own g2o (oracle/_ref/libref_graph.so, oracle/Makefile.ref):
  Optimizer::LocalBACameraPointObjects (orb_object_slam/src/Optimizer.cc:826-1534), Optimizer::BundleAdjustment (:64-251) and Optimizer::PoseOptimization
  (:253-472) are cut out of the reference at build time and compiled against the vendored g2o WHOLE (SparseOptimizer, BlockSolver_6_3 with its Schur complement
  block_solver.hpp:354-486, OptimizationAlgorithmLevenberg, the robust kernels, LinearSolverDense), the reference's g2o_Object.{h,cpp}, stand-ins for Eigen and for
  Eigen's sparse Cholesky (absent from the image) and stand-ins for the map classes (oracle/ref_shim/slam_graph_standins.hpp).  The oracle's restatements
  (oracle/local_ba_objects.py, orc_ba_optimize, orc_pose_optimization) are held to what the reference's functions leave in the map.

Tolerances.  The reference stores key-frame poses and map points as FLOAT cv::Mat, so those are compared as floats (a few units in the last place of a float);
object poses are doubles on both sides.  The two sides factor the same reduced system in different elimination orders and sum edges in different orders (the
reference iterates std::map<KeyFrame *> in pointer order), so nothing here is bit for bit: the discrete outcome (which observations are erased, which points are
written back, thresholds, counters) must be IDENTICAL, the numbers agree to round-off amplified by fifteen LM iterations."""
import math

import numpy as np
import pytest

from oracle import local_ba_objects as lo
from oracle import pyoracle as po
from tests import local_map
from tests import ref_graph as rg

pytestmark = pytest.mark.skipif(not __import__("os").path.exists(rg.SO), reason="oracle/_ref/libref_graph.so is built from /root/reference, which is not present here")


def _float_close(a, b, ulps=4):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool((np.abs(a.astype(np.float64) - b.astype(np.float64)) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64) + 1e-7).all())


def _pose_close(T, To, tol=5e-6):
    """4 x 4 float poses: rotation entries and translation within `tol` (x the translation's size beyond a metre)."""
    T, To = np.asarray(T, np.float64), np.asarray(To, np.float64)
    return bool(np.abs(T[:3, :3] - To[:3, :3]).max() <= tol and np.abs(T[:3, 3] - To[:3, 3]).max() <= tol * max(1.0, float(np.abs(To[:3, 3]).max())))


@pytest.mark.parametrize("seed,kwargs", [(1, {}), (2, {}), (3, {"left_heavy": True}), (4, {"with_objects": False})])
def test_local_ba_camera_point_objects_equals_reference(seed, kwargs):
    cur, params, extra = local_map.build(seed, **kwargs)
    rg.quantize(cur, params, extra)
    ref = lo.local_ba_camera_point_objects(cur, params)
    G = rg.Graph(cur, params, extra)
    try:
        G.local_ba_objects(cur)
        kid = {k.mnId: k for k in extra["kfs"]}
        # the erase list: the same (key frame, map point) pairs (the reference walks its edge vectors in std::map pointer order: compared as sets)
        assert sorted(G.erased()) == sorted(ref["erase"]) and (len(ref["erase"]) > 0)
        # key frames: every local key frame written once, to the oracle's pose as a float matrix; markers reset (:1500-1507, :1519-1524)
        moved = 0.0
        for mn, pose in ref["kf_pose"].items():
            T, n, _ = G.kf_pose(kid[mn])
            assert n == 1 and _pose_close(T, rg.cvmat_from_pose(pose)), mn
            moved = max(moved, float(np.abs(T - kid[mn].T_f32).max()))
            assert G.kf_markers(kid[mn]) == (0, 0)
        assert any(G.kf_markers(k)[0] == cur.mnId for k in cur.covisible if k.bad), "a bad covisible key frame keeps its mark (it is in no list that :1500-1524 resets)"
        assert moved > 1e-3
        for k in extra["kfs"]:
            if k.mnId not in ref["kf_pose"]:
                T, n, _ = G.kf_pose(k)
                assert n == 0 and np.array_equal(T, k.T_f32), "fixed key frames and key frames outside the window are not written"
        # points: written (SetWorldPos + UpdateNormalAndDepth) unless the erasures left exactly one observation (:1511-1512) or the point was no vertex (:1052)
        mid = {m.mnId: m for m in G.mps}
        unwritten = set(ref["point_unwritten"])
        assert len(unwritten) > 0
        n_written = 0
        for mn, p in ref["point_pos"].items():
            got, nw, nu = G.mp_pos(mid[mn])
            if mn in unwritten:
                assert nw == 0 and nu == 0 and np.array_equal(got, np.float32(mid[mn].pos)), mn
            else:
                assert nw == 1 and nu == 1, mn
                # the depth of a far point seen over a short baseline is the loosest number here: 5e-6 relative to its distance (the reference's own spread over heap layouts: 6e-7)
                assert np.abs(got.astype(np.float64) - p).max() <= 5e-6 * max(1.0, float(np.linalg.norm(p))), (mn, got, p)
                n_written += 1
        assert n_written > 100
        for m in G.mps:
            if m.mnId not in ref["point_pos"]:
                assert G.mp_pos(m)[1] == 0
        # objects: pose written for every local object, fixed KITTI half size, the association counters of :1149-1209
        oid = {o.mnId: o for o in extra["mos"]}
        for mn, p in ref["object_pose"].items():
            s = G.mo_state(oid[mn])
            assert s["writes"] == 1 and s["been_optimized"]
            assert np.allclose(s["pose"], p, rtol=0, atol=2e-5), (mn, np.abs(s["pose"] - p).max())   # (the reference's own result moves by 3e-6 with the heap layout: its edge order follows pointer values)
            assert np.array_equal(s["scale"], ref["object_scale"][mn])
            assert s["point_threshold"] == max(int(oid[mn].largest_point_observations * 0.4), 2)
        for o in extra["mos"]:
            if o.mnId not in ref["object_pose"]:
                assert G.mo_state(o)["writes"] == 0
        if kwargs.get("with_objects", True):
            assert len(ref["object_pose"]) > 5
    finally:
        G.close()


def _all_frames_problem(kfs, mps, params, huber_mono):
    ki = {id(k): i for i, k in enumerate(kfs)}
    oc, op, uv, w, ur = [], [], [], [], []
    for j, m in enumerate(mps):
        for k, i in m.observations.items():
            oc.append(ki[id(k)]); op.append(j); uv.append(k.mvKeysUn[i]); w.append(k.mvInvLevelSigma2[k.octave[i]]); ur.append(k.mvuRight[i] if k.mvuRight[i] >= 0 else -1.0)
    K = params["K"]
    return {"cam_pose": np.stack([k.Tcw for k in kfs]), "cam_fixed": np.array([k.mnId == 0 for k in kfs], np.uint8), "points": np.stack([m.pos for m in mps]),
            "cuboid_pose": np.zeros((0, 7)), "cuboid_scale": np.zeros((0, 3)), "cuboid_flags": np.zeros(0, np.uint8),
            "obs_cam": np.array(oc, np.int32), "obs_point": np.array(op, np.int32), "obs_uv": np.array(uv, float).reshape(-1, 2), "obs_inv_sigma2": np.array(w, float),
            "obs_ur": np.array(ur, float), "fx": K[0, 0], "fy": K[1, 1], "cx": K[0, 2], "cy": K[1, 2], "huber_mono": huber_mono,
            "huber_stereo": float(np.float32(math.sqrt(7.815))) if huber_mono else 0.0, "bf": params["bf"],
            "cobs_cam": np.zeros(0, np.int32), "cobs_cuboid": np.zeros(0, np.int32), "cobs_bbox": np.zeros((0, 4)), "cobs_info": np.zeros((0, 4)), "K": K, "huber_obj": 0.0,
            "pc_cuboid": np.zeros(0, np.int32), "pc_offsets": np.zeros(1, np.int32), "pc_points": np.zeros((0, 3)), "max_outside_margin_ratio": 1.0}


@pytest.mark.parametrize("iterations,robust,loop_kf", [(1, True, 0), (10, True, 0), (20, False, 7)])
def test_bundle_adjustment_equals_reference(iterations, robust, loop_kf):
    """Optimizer::BundleAdjustment: BlockSolver_6_3 over the whole map -- the Schur complement, the landmark back-substitution and the LM loop of the reference's g2o
    against orc_ba_optimize (ba_oracle.cpp), iteration counts included.  With nLoopKF != 0 the result goes to mTcwGBA / mPosGBA and the poses stay (:222-250)."""
    cur, params, extra = local_map.build(1, n_kf=10, n_points=80, n_cuboids=3)
    rg.quantize(cur, params, extra)
    for k in extra["kfs"]:
        k.bad = False; k.local_cuboids, k.cuboids_landmark = [], []
    kfs, mps = extra["kfs"], [m for m in extra["mps"] if m.observations]
    extra["mps"], extra["mos"] = mps, []
    d = _all_frames_problem(kfs, mps, params, float(np.float32(math.sqrt(5.99))) if robust else 0.0)
    cam, pts, _, st = po.ba_optimize(d, iterations)
    assert st["iterations"] >= min(iterations, 5)
    G = rg.Graph(cur, params, extra)
    try:
        G.bundle_adjustment(iterations, loop_kf=loop_kf, robust=robust)
        moved = 0.0
        for i, k in enumerate(kfs):
            T, n, Tg = G.kf_pose(k)
            if loop_kf:
                assert n == 0 and np.array_equal(T, k.T_f32)
                T = Tg
            else:
                assert n == 1
            assert _float_close(T, rg.cvmat_from_pose(cam[i])), k.mnId
            moved = max(moved, float(np.abs(T - k.T_f32).max()))
        assert moved > 1e-2
        if not loop_kf:
            for j, m in enumerate(mps):
                got, nw, nu = G.mp_pos(m)
                assert nw == 1 and nu == 1 and np.abs(got.astype(np.float64) - pts[j]).max() <= 2e-6 * max(1.0, float(np.linalg.norm(pts[j])))
    finally:
        G.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_pose_optimization_equals_reference(seed):
    """Optimizer::PoseOptimization: the four rounds of ten iterations with the outlier re-classification between them (:401-461), LinearSolverDense under
    BlockSolver_6_3, against orc_pose_optimization: inlier count, outlier flags, pose."""
    cur, params, extra = local_map.build(seed)
    rg.quantize(cur, params, extra)
    G = rg.Graph(cur, params, extra)
    K = params["K"]
    try:
        n_checked = 0
        for k in extra["kfs"][2:8]:
            rows = [i for i, m in enumerate(k.map_point_matches) if m is not None]
            Xw = np.array([k.map_point_matches[i].pos for i in rows]); obs = np.array([[k.mvKeysUn[i][0], k.mvKeysUn[i][1], k.mvuRight[i] if k.mvuRight[i] >= 0 else -1.0] for i in rows])
            w = np.array([k.mvInvLevelSigma2[k.octave[i]] for i in rows])
            # a start a few centimetres off, as a float pose
            T0 = k.T_f32.copy(); T0[:3, 3] += np.float32([0.05, -0.02, 0.08])
            start = rg.pose_from_cvmat(T0)
            pose, outl, n_in = po.pose_optimization(Xw, obs, w, (K[0, 0], K[1, 1], K[0, 2], K[1, 2], params["bf"]), start)
            k_T = k.T_f32; k.T_f32 = T0
            H = rg.Graph(cur, params, extra)
            k.T_f32 = k_T
            try:
                n_ref, T, out_ref = H.pose_optimization(k)
            finally:
                H.close()
            assert n_ref == n_in and np.array_equal(out_ref[rows], outl.astype(bool)) and not out_ref[[i for i in range(len(out_ref)) if i not in rows]].any()
            assert _float_close(T, rg.cvmat_from_pose(pose)), k.mnId
            assert outl.any() and not outl.all()
            n_checked += 1
        assert n_checked == 6
    finally:
        G.close()


def _assoc_scene(rng, n_points=400, n_land=5, n_cand=8):
    votes = [dict() for _ in range(n_points)]
    land_pts = [rng.choice(n_points, 60, replace=False) for _ in range(n_land)]
    for o, pts in enumerate(land_pts):
        for p in pts:
            votes[int(p)][100 + o] = int(rng.integers(1, 4))
    cands = []
    for i in range(n_cand):
        if i % 4 == 2:
            pts = rng.choice(n_points, 30, replace=False)            # mostly unseen points: a new landmark
        elif i % 4 == 3 and cands:
            pts = np.array(cands[-1][:24] + [int(p) for p in rng.choice(n_points, 6, replace=False)])  # shares the points of the landmark just created
        else:
            base = land_pts[i % n_land]
            pts = np.concatenate([rng.choice(base, int(rng.integers(8, 30)), replace=False), rng.choice(n_points, 10, replace=False)])
        cands.append(sorted(set(int(p) for p in pts)))
    return votes, cands


@pytest.mark.parametrize("seed", range(6))
def test_associate_cuboids_equals_reference(seed):
    """Tracking::AssociateCuboids (Tracking.cc:1848-2043) with the vote bookkeeping it drives (MapObject::SetAsLandmark / MergeIntoLandmark / addObservation,
    MapPoint::AddObjectObservation: the reference's text) against pyoracle.associate_cuboids: which landmark every candidate ends up in, which candidates become
    landmarks, every point's votes, best object and best vote afterwards.  Also what the oracle leaves to the caller: the gathering of candidates and landmarks
    from the local key frames (:1858-1876) and the ageing of rarely seen objects (:1992-2043)."""
    import ctypes as C
    rng = np.random.default_rng(100 + seed)
    votes, cands = _assoc_scene(rng)
    n_land = 5
    land_bad = [0, seed % 2, 0, 0, 0]
    L = rg.lib()
    h = C.c_void_p(L.ref_graph_open())
    try:
        K = np.eye(3)
        L.ref_graph_set_params(h, 1, 0, C.c_double(1.0), 0, 1241, 376, rg._p(K.reshape(-1), C.c_double), 0)
        eye = np.eye(4, dtype=np.float32).reshape(-1); z3 = np.zeros(3, np.float32)
        one = np.ones(1, np.float32); zi = np.zeros(1, np.int32)

        def add_kf(mnid):
            return L.ref_graph_add_kf(h, C.c_long(mnid), 0, rg._p(eye, C.c_float), rg._p(z3, C.c_float), 0, rg._p(one, C.c_float), rg._p(one, C.c_float), rg._p(zi, C.c_int), 1, rg._p(one, C.c_float),
                                      C.c_float(1), C.c_float(1), C.c_float(0), C.c_float(0), C.c_float(0))
        local = [add_kf(30), add_kf(31), add_kf(32)]
        cur, old = add_kf(40), add_kf(5)
        mps = [L.ref_graph_add_mp(h, C.c_long(p), 0, rg._p(z3, C.c_float), 0) for p in range(len(votes))]
        pose = np.array([1.0, 2.0, 3.0, 0, 0, 0, 1.0]); scale = np.array([2.0, 1.0, 0.9]); bv = np.zeros(4); b2 = np.zeros(4, np.int32)
        mos = [L.ref_graph_add_mo(h, C.c_long(100 + i), land_bad[i], rg._p(pose, C.c_double), rg._p(scale, C.c_double), C.c_double(1.0), 0) for i in range(n_land)]
        # landmarks hang on the local key frames (KeyFrame::cuboids_landmark), landmark 0 on two of them (gathered once: association_refid_in_tracking)
        refs0 = [0] * n_land
        gathered = []
        attach = {0: [0, 3, 0], 1: [1, 0], 2: [2, 4]}
        for kf_i, ls in attach.items():
            for li in ls:
                det = L.ref_graph_kf_detection(h, local[kf_i], rg._p(bv, C.c_double), rg._p(b2, C.c_int), 0, C.c_double(1.0), li)
                L.ref_graph_mo_observe(h, mos[li], local[kf_i], det)
                refs0[li] += 1
                if not land_bad[li] and li not in gathered:
                    gathered.append(li)
        # two objects the local key frames do not hold, last seen 35 key frames ago: one seen once (-> bad), one seen three times (-> good)
        aged = [L.ref_graph_add_mo(h, C.c_long(900 + i), 0, rg._p(pose, C.c_double), rg._p(scale, C.c_double), C.c_double(1.0), 0) for i in range(2)]
        L.ref_graph_mo_observe(h, aged[0], old, 0)
        for k in (old, add_kf(3), add_kf(4)):
            L.ref_graph_mo_observe(h, aged[1], k, 0)
        # votes and best objects as MapPoint::AddObjectObservation would have left them
        best = np.full(len(votes), -1, np.int32); mv = np.zeros(len(votes), np.int32)
        for p, d in enumerate(votes):
            for o, c in d.items():
                L.ref_graph_mp_vote(h, mps[p], mos[o - 100], c)
                if c > mv[p]:
                    best[p] = o; mv[p] = c
            L.ref_graph_mp_best(h, mps[p], -1 if best[p] < 0 else mos[best[p] - 100], int(mv[p]))
        # candidates: detections with potential points, spread over the local key frames; one detection that is no candidate, one already associated
        cand_at = []
        for i, pts in enumerate(cands):
            kf_i = i % 3
            det = L.ref_graph_kf_detection(h, local[kf_i], rg._p(bv, C.c_double), rg._p(b2, C.c_int), 0, C.c_double(1.0), -2)
            L.ref_graph_det_candidate(h, local[kf_i], det, 1, 0, rg._p(pose, C.c_double), rg._p(scale, C.c_double))
            for p in pts:
                L.ref_graph_det_potential_point(h, local[kf_i], det, mps[p])
            cand_at.append((kf_i, det))
        for flags in ((0, 0), (1, 1)):
            det = L.ref_graph_kf_detection(h, local[0], rg._p(bv, C.c_double), rg._p(b2, C.c_int), 0, C.c_double(1.0), -2)
            L.ref_graph_det_candidate(h, local[0], det, flags[0], flags[1], rg._p(pose, C.c_double), rg._p(scale, C.c_double))
            for p in range(40 if not flags[1] else 0):
                L.ref_graph_det_potential_point(h, local[0], det, mps[p])
        # the reference walks key frames, then their detections: that is the candidate order
        order = sorted(range(len(cands)), key=lambda i: (cand_at[i][0], cand_at[i][1]))
        cand_id = [200 + i for i in order]
        v = [dict(d) for d in votes]
        assoc, created = po.associate_cuboids(cand_id, [cands[i] for i in order], [100 + li for li in gathered], [0] * len(gathered), v, 10, best, mv)
        assert created.sum() >= 2 and (created == 0).sum() >= 2, "both branches"
        L.ref_graph_associate_cuboids(h, cur, rg._p(np.array(local, np.int32), C.c_int), 3, C.c_long(500), 0)
        gidx = {L.ref_graph_det_global_index(h, local[cand_at[i][0]], cand_at[i][1]): 200 + i for i in range(len(cands))}

        def oid(kind, index):
            return -1 if kind < 0 else (100 + index if kind == 0 and index < n_land else (900 + index - n_land if kind == 0 else gidx[index]))
        merged_into = [0] * n_land
        n_created = 0
        if __import__("os").environ.get("ASSOC_DEBUG"):
            for k, i in enumerate(order):
                ak, ai, mnid, aa, nobs = C.c_int(), C.c_int(), C.c_long(), C.c_int(), C.c_int()
                sc = np.zeros(3)
                L.ref_graph_det_state(h, local[cand_at[i][0]], cand_at[i][1], C.byref(ak), C.byref(ai), C.byref(mnid), C.byref(aa), C.byref(nobs), rg._p(sc, C.c_double))
                print(k, i, "oracle", assoc[k], created[k], "ref", oid(ak.value, ai.value), mnid.value, "gathered", gathered)
        for k, i in enumerate(order):
            ak, ai, mnid, aa, nobs = C.c_int(), C.c_int(), C.c_long(), C.c_int(), C.c_int()
            sc = np.zeros(3)
            L.ref_graph_det_state(h, local[cand_at[i][0]], cand_at[i][1], C.byref(ak), C.byref(ai), C.byref(mnid), C.byref(aa), C.byref(nobs), rg._p(sc, C.c_double))
            assert oid(ak.value, ai.value) == assoc[k] and aa.value == 1, (k, i)
            if created[k]:
                n_created += 1
                assert assoc[k] == 200 + i and mnid.value == 500 + n_created and nobs.value >= 1 and np.array_equal(sc, [1.9420, 0.8143, 0.7631])
            elif assoc[k] < 200:
                merged_into[assoc[k] - 100] += 1
        # votes, best object, best vote of every point
        tri = np.zeros(3 * 64, np.int32)
        for p in range(len(votes)):
            bk, bi, m = C.c_int(), C.c_int(), C.c_int()
            n = L.ref_graph_mp_votes(h, mps[p], rg._p(tri, C.c_int), 64, C.byref(bk), C.byref(bi), C.byref(m))
            got = {oid(int(tri[3 * j]), int(tri[3 * j + 1])): int(tri[3 * j + 2]) for j in range(n)}
            assert got == v[p], p
            assert oid(bk.value, bi.value) == best[p] and m.value == mv[p], p
        # landmarks: one more KeyFrame::cuboids_landmark entry per merged candidate; the aged objects
        for li in range(n_land):
            f = [C.c_int() for _ in range(6)]
            L.ref_graph_mo_flags(h, mos[li], *[C.byref(x) for x in f])
            assert f[5].value == refs0[li] + merged_into[li], li
        f = [C.c_int() for _ in range(6)]
        L.ref_graph_mo_flags(h, aged[0], *[C.byref(x) for x in f])
        assert f[1].value == 1 and f[2].value == 0, "seen once, not for more than 15 key frames: bad (:2021-2028)"
        L.ref_graph_mo_flags(h, aged[1], *[C.byref(x) for x in f])
        assert f[1].value == 0 and f[2].value == 1, "seen three times: good (:2029-2032)"
    finally:
        L.ref_graph_close(h)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_local_ba_dynamic_equals_reference(seed):
    """Optimizer::LocalBACameraPointObjectsDynamic (Optimizer.cc:1537-2573) on BlockSolverX + LinearSolverDense + Levenberg, its own text, against
    oracle/local_ba_dynamic.py (graph construction) over orc_badyn_optimize (the solver): the window of tests/local_map_dynamic.py takes every branch but the
    point-object association's aliased vertex id (see that module).  Identical: the observations erased, the dynamic points set bad while the window is gathered,
    which vertices exist (per-frame object poses flagged as optimised, velocities, dynamic points), the zero velocity that is initialised and written.  Numbers:
    key frames as float matrices, objects / velocities / dynamic points to the round-off that fifteen LM iterations leave on weakly constrained vertices
    (a car with three vertices has no motion edges: its points and poses are the loosest numbers).  The tolerances are five times and more what the REFERENCE'S OWN result moves
    when nothing but the heap layout changes (its maps are keyed by pointers, so the order in which it sums edges follows addresses): measured over twelve layouts per
    window, key frames 7e-6, static points 4e-5 of their distance, per-frame object poses 1e-4, velocities 1.5e-4, dynamic points 1.8e-3; never a discrete difference."""
    from oracle import local_ba_dynamic as ld
    from tests import local_map_dynamic as lmd
    cur, params, extra = lmd.build(seed)
    rg.quantize(cur, params, extra)
    ref = ld.local_ba_dynamic(cur, params)
    g = ref["graph"]; d = g["problem"]
    # the window takes the branches
    n_vert = {mo.mnId: len(g["vertex_of"][id(mo)]) for mo in g["objects"]}
    assert min(n_vert.values()) < 4 <= max(n_vert.values()) and 0 < len(d["vel"]) < len(g["objects"]), "a car without a velocity vertex"
    assert len(ref["velocity_written"]) == 1 and len(ref["set_bad"]) >= 1 and len(ref["erase"]) > 50
    assert len(d["cam_pose"]) > g["n_local"] and (d["obs_ur"] >= 0).any() and (d["obs_ur"] < 0).any() and (d["cobs_level"] == 0).sum() > 10
    assert len(d["dpoints"]) < sum(1 for m in extra["mps"] if getattr(m, "is_dynamic", False)), "dynamic points without a vertex"
    assert ref["dobs_level"].sum() > 0 and ref["obs_level"].sum() > 0
    G = rg.Graph(cur, params, extra)
    try:
        G.local_ba_dynamic(cur)
        kid = {k.mnId: k for k in extra["kfs"]}
        oid = {o.mnId: o for o in extra["mos"]}
        mid = {m.mnId: m for m in G.mps}
        assert sorted(G.erased()) == sorted(ref["erase"])
        for mn, pose in ref["kf_pose"].items():
            T, n, _ = G.kf_pose(kid[mn])
            assert n == 1 and _pose_close(T, rg.cvmat_from_pose(pose), tol=3e-5), mn
        for k in extra["kfs"]:
            if k.mnId not in ref["kf_pose"]:
                assert G.kf_pose(k)[1] == 0
        # static points: written unless the erasures left one observation; dynamic points never through this loop
        unwritten = set(ref["point_unwritten"])
        for mn, p in ref["point_pos"].items():
            got, nw, _ = G.mp_pos(mid[mn])
            assert nw == (0 if mn in unwritten else 1), mn
            if nw:
                assert np.abs(got.astype(np.float64) - p).max() <= 2e-4 * max(1.0, float(np.linalg.norm(p))), (mn, got, p)  # (a point 45 m away seen over 1 m of baseline is the loosest)
        # objects: every vertex pose back in allDynamicPoses with its flag set, the newest observing key frame's pose as the world pose, velocities
        for (mo, kf), p in ref["object_frame_pose"].items():
            got, baed = G.mo_dynamic_pose(oid[mo], kid[kf])
            assert baed and np.allclose(got, p, rtol=0, atol=1e-3), (mo, kf, np.abs(got - p).max())
        for o in g["objects"]:
            for kf in o.allDynamicPoses:
                if (o.mnId, kf.mnId) not in ref["object_frame_pose"]:
                    got, baed = G.mo_dynamic_pose(o, kf)
                    assert not baed and np.allclose(got, o.allDynamicPoses[kf], atol=1e-12), "a pose without a vertex stays"
        for o in g["objects"]:
            st = G.mo_dynamic_state(o)
            assert np.allclose(st["latest"], ref["object_latest"][o.mnId], rtol=0, atol=1e-3) and np.array_equal(st["latest"], st["afterba"])
            assert st["local_for"] == 0
            if o.mnId in ref["velocity"]:
                assert np.allclose(st["velocity"], ref["velocity"][o.mnId], rtol=0, atol=1e-3) and st["n_history"] == 1 and np.array_equal(st["history"], st["velocity"])
            else:
                assert st["n_history"] == 0 and np.array_equal(st["velocity"], o.velocityPlanar)
        # dynamic points: PosToObj and the world position under the newest object pose; those without a vertex untouched; the ones set bad on the way
        for mn, p in ref["dpoint_local"].items():
            s = G.mp_dynamic(mid[mn])
            tol = 1e-2   # (millimetres in the car's frame: numeric Jacobians with delta = 1e-9 under two edge orders through fifteen LM iterations)
            assert s["is_optimized"] and np.abs(s["PosToObj"].astype(np.float64) - p).max() <= tol, (mn, s["PosToObj"], p)
            assert np.abs(s["latest"].astype(np.float64) - ref["dpoint_world"][mn]).max() <= tol and G.mp_pos(mid[mn])[1] == 1
        for m in G.mps:
            if getattr(m, "is_dynamic", False) and m.mnId not in ref["dpoint_local"]:
                s = G.mp_dynamic(m)
                assert not s["is_optimized"] and np.array_equal(s["PosToObj"], np.float32(m.PosToObj)) and G.mp_pos(m)[1] == 0
        assert sorted(m.mnId for m in G.mps if G.mp_dynamic(m)["bad"]) == sorted(ref["set_bad"])
    finally:
        G.close()


def test_local_ba_with_fixed_cameras_equals_reference():
    """fixCamera = true (Optimizer.h:47; every key-frame vertex fixed, :955-956): points and objects move, poses are written back unchanged."""
    cur, params, extra = local_map.build(2)
    rg.quantize(cur, params, extra)
    ref = lo.local_ba_camera_point_objects(cur, params, fixCamera=True)
    G = rg.Graph(cur, params, extra)
    try:
        G.local_ba_objects(cur, fix_camera=True)
        assert sorted(G.erased()) == sorted(ref["erase"]) and len(ref["erase"]) > 0
        kid = {k.mnId: k for k in extra["kfs"]}
        for mn, pose in ref["kf_pose"].items():
            T, n, _ = G.kf_pose(kid[mn])
            assert n == 1 and _pose_close(T, rg.cvmat_from_pose(pose)) and np.abs(T - kid[mn].T_f32).max() <= 2e-7, mn   # (SetPose of the estimate it started from: float -> SE3Quat -> float)
        mid = {m.mnId: m for m in G.mps}
        unwritten = set(ref["point_unwritten"])
        moved = 0.0
        for mn, p in ref["point_pos"].items():
            got, nw, _ = G.mp_pos(mid[mn])
            assert nw == (0 if mn in unwritten else 1)
            if nw:
                d_p = float(np.linalg.norm(p))   # every point on its own here: the depth of a far point is barely held.  The reference's own result moves with its heap layout (2e-5 ... 7e-5 of the distance within 30 m, 1.2e-4 beyond, over fifty layouts): the bar is several times that; the strict part of this test is the discrete outcome
                assert np.abs(got.astype(np.float64) - p).max() <= (3e-4 if d_p <= 30 else 1e-3) * max(1.0, d_p), mn
                moved = max(moved, float(np.abs(got - np.float32(mid[mn].pos)).max()))
        assert moved > 1e-3
        oid = {o.mnId: o for o in extra["mos"]}
        for mn, p in ref["object_pose"].items():
            assert np.allclose(G.mo_state(oid[mn])["pose"], p, rtol=0, atol=5e-4), mn   # (an object held by its bounding boxes alone: looser than with free cameras, and moving with the reference's edge order)
    finally:
        G.close()


def test_local_ba_again_on_its_own_result_equals_reference():
    """The window after a local BA, optimised again (and again): the second stage stops early (three or four iterations), a few more observations cross the
    thresholds.  The oracle's early stops and its outlier classification (pin D4 of DESIGN.md: it reads the residuals at the accepted estimate, g2o the edges'
    stored errors) stay on the reference's: the same observations erased, round after round."""
    cur, params, extra = local_map.build(1)
    rg.quantize(cur, params, extra)
    early = 0
    for rnd in range(3):
        ref = lo.local_ba_camera_point_objects(cur, params)
        G = rg.Graph(cur, params, extra)
        try:
            G.local_ba_objects(cur)
            assert sorted(G.erased()) == sorted(ref["erase"]), rnd
            kid = {k.mnId: k for k in extra["kfs"]}
            for mn, pose in ref["kf_pose"].items():
                assert _pose_close(G.kf_pose(kid[mn])[0], rg.cvmat_from_pose(pose)), (rnd, mn)
        finally:
            G.close()
        early += int(ref["stats"][1]["iterations"] < 10)
        # the result becomes the next round's window
        mid = {m.mnId: m for m in extra["mps"]}; oid = {o.mnId: o for o in extra["mos"]}
        for mn, p in ref["kf_pose"].items():
            kid[mn].Tcw = p
        un = set(ref["point_unwritten"])
        for mn, p in ref["point_pos"].items():
            if mn not in un:
                mid[mn].pos = p
        for mn, p in ref["object_pose"].items():
            oid[mn].pose = p
        for kfid, mpid in ref["erase"]:
            k, m = kid[kfid], mid[mpid]
            k.map_point_matches[m.observations.pop(k)] = None
        rg.quantize(cur, params, extra)
    assert early >= 1, "a later round ends its second stage before the tenth iteration"



def test_local_bundle_adjustment_is_the_object_window_without_objects():
    """Optimizer::LocalBundleAdjustment (Optimizer.cc:474-825), the reference's own text, on a window without objects against oracle/local_ba_objects.py -- the restatement
    of LocalBACameraPointObjects: without object vertices the two functions build the same graph (same windows, same skip of points seen once, same two stages), so one
    restatement serves both.  What differs is bookkeeping: the fixed key frames keep their mnBALocalForKF mark here (:820-824 resets mnBAFixedForKF only)."""
    cur, params, extra = local_map.build(4, with_objects=False)
    rg.quantize(cur, params, extra)
    ref = lo.local_ba_camera_point_objects(cur, params)
    G = rg.Graph(cur, params, extra)
    try:
        G.local_ba(cur)
        kid = {k.mnId: k for k in extra["kfs"]}
        assert sorted(G.erased()) == sorted(ref["erase"]) and len(ref["erase"]) > 0
        for mn, pose in ref["kf_pose"].items():
            T, n, _ = G.kf_pose(kid[mn])
            assert n == 1 and _pose_close(T, rg.cvmat_from_pose(pose)), mn
        unwritten = set(ref["point_unwritten"])
        for m in extra["mps"]:
            if m.mnId in ref["point_pos"]:
                p, n, _ = G.mp_pos(m)
                assert n == (0 if m.mnId in unwritten else 1), m.mnId
    finally:
        G.close()
