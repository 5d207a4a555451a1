"""Python host-side mirror of Optimizer::LocalBACameraPointObjectsDynamic (reference orb_object_slam/include/Optimizer.h:57-58,
src/Optimizer.cc:1537-2573) over the C-ABI (cs_ba_dyn_*).

Two levels, like cube_slam_amd/ba_objects.py:

* `LocalBACameraPointObjectsDynamic(window, params)`: the graph-level flow from the local window as flat arrays -- what adapters/Optimizer_hip.cc gathers from
  KeyFrame* / MapPoint* / MapObject* in the reference's iteration order (the window selection of :1540-1665 walks pointer containers and stays on the caller's
  side) -- through `build_graph` (:1667-2340: one cuboid vertex per (object, observing key frame), static and dynamic points, the point-object association with
  the reference's vertex-id aliasing, velocity vertices and motion edges, camera-object edges), the two stages and the erase list, to the values the caller
  writes back (:2446-2572).  cube_slam_amd/host/local_ba_dynamic.hpp is the same flow in C++.
* `optimize_two_stages(problem)` / `DynamicBundleAdjuster`: the solver level over an already flattened graph (cs_ba_dyn_problem; the dict layout is the one
  cube_slam_amd.synth.ba_dyn_problem produces).

The window (rows in the reference's iteration order):
  kf_id, kf_pose (n,7 world-to-camera [t q]), kf_stamp (n, mTimeStamp), kf_cam_center (n,3: GetCameraCenter(), float), n_local (lLocalKeyFrames first, then lFixedCameras)
  mp_pos (m,3), mp_nobs (MapPoint::Observations(): nObs), mp_dynamic (m), mp_pos_to_obj (m,3: PosToObj), mp_best_mo (m: row of mo_* of GetBelongedObject() when that
      object is in lLocalMapObjects, else -1)
  obs_mp, obs_kf, obs_uv (o,2), obs_ur (o; < 0 monocular), obs_inv_sigma2 (o)       GetObservations() of every local point, by key frames that are not bad
  mo_id (c: mnId), mo_meas_quality, mo_largest_point_observations, mo_velocity (c,2: velocityPlanar)
  ov_mo, ov_kf, ov_pose (v,7: allDynamicPoses[kf]), ov_bbox_vec (v,4), ov_bbox_2d (v,4 x y w h), ov_left_right_to_car (v)
      GetObservations() of every local object by key frames that are in the window and not bad: one cuboid vertex each, in this order
  seq_mo, seq_kf          GetObserveFramesSequential() of every local object (key frames of the window that are not bad)
  up_mo, up_pos (u,3), up_count (u)      GetUniqueMapPoints() (not bad) with MapObjObservations[object]
params: K, img_width, img_height, bf, camera_object_BA_weight, object_velocity_BA_weight, kitti, build_worldframe_on_ground, ba_dyna_pt_obj_cam, ba_dyna_obj_velo,
ba_dyna_obj_cam."""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import check, lib
from .ba import BAStats


class BADynProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("cam_pose", C.c_void_p), ("cam_fixed", C.c_void_p),
                ("n_objs", C.c_int), ("obj_pose", C.c_void_p), ("obj_scale", C.c_void_p), ("obj_flags", C.c_void_p),
                ("n_vels", C.c_int), ("vel", C.c_void_p),
                ("n_points", C.c_int), ("points", C.c_void_p),
                ("n_dpoints", C.c_int), ("dpoints", C.c_void_p),
                ("fix_points", C.c_int),
                ("n_obs", C.c_int), ("obs_cam", C.c_void_p), ("obs_point", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_ur", C.c_void_p), ("obs_inv_sigma2", C.c_void_p),
                ("obs_level", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double), ("huber_mono", C.c_double), ("huber_stereo", C.c_double),
                ("ulp_info", C.c_double), ("ulp_scale", C.c_double * 3), ("ulp_ratio", C.c_double),
                ("n_dobs", C.c_int), ("dobs_cam", C.c_void_p), ("dobs_obj", C.c_void_p), ("dobs_point", C.c_void_p), ("dobs_uv", C.c_void_p), ("dobs_inv_sigma2", C.c_void_p),
                ("dobs_level", C.c_void_p),
                ("K", C.c_double * 9), ("huber_dyn", C.c_double),
                ("n_mot", C.c_int), ("mot_from", C.c_void_p), ("mot_to", C.c_void_p), ("mot_vel", C.c_void_p), ("mot_dt", C.c_void_p), ("mot_info", C.c_double * 3),
                ("n_cobs", C.c_int), ("cobs_cam", C.c_void_p), ("cobs_obj", C.c_void_p), ("cobs_bbox", C.c_void_p), ("cobs_info", C.c_void_p), ("cobs_level", C.c_void_p),
                ("huber_obj", C.c_double),
                ("n_pc", C.c_int), ("pc_obj", C.c_void_p), ("pc_offsets", C.c_void_p), ("pc_points", C.c_void_p), ("pc_ratio", C.c_double)]


_INT = ("obs_cam", "obs_point", "dobs_cam", "dobs_obj", "dobs_point", "mot_from", "mot_to", "mot_vel", "cobs_cam", "cobs_obj", "pc_obj", "pc_offsets")
_U8 = ("cam_fixed", "obj_flags", "obs_level", "dobs_level", "cobs_level")
_F64 = ("cam_pose", "obj_pose", "obj_scale", "vel", "points", "dpoints", "obs_uv", "obs_ur", "obs_inv_sigma2", "dobs_uv", "dobs_inv_sigma2", "mot_dt", "cobs_bbox",
        "cobs_info", "pc_points")
_COUNTS = (("n_cams", "cam_pose"), ("n_objs", "obj_pose"), ("n_vels", "vel"), ("n_points", "points"), ("n_dpoints", "dpoints"), ("n_obs", "obs_cam"),
           ("n_dobs", "dobs_cam"), ("n_mot", "mot_from"), ("n_cobs", "cobs_cam"), ("n_pc", "pc_obj"))


def problem_struct(d):
    keep = {}
    p = BADynProblem()
    for names, dt in ((_INT, np.int32), (_U8, np.uint8), (_F64, np.float64)):
        for name in names:
            if d.get(name) is None:
                continue
            a = np.ascontiguousarray(d[name], dt)
            if a.size == 0:
                a = np.zeros(2, dt)
            keep[name] = a
            setattr(p, name, a.ctypes.data)
    for cnt, name in _COUNTS:
        setattr(p, cnt, len(d[name]))
    p.fix_points = int(d["fix_points"])
    for k in ("fx", "fy", "cx", "cy", "bf", "huber_mono", "huber_stereo", "ulp_info", "ulp_ratio", "huber_dyn", "huber_obj", "pc_ratio"):
        setattr(p, k, float(d[k]))
    for i in range(3):
        p.ulp_scale[i] = d["ulp_scale"][i]; p.mot_info[i] = d["mot_info"][i]
    for i, v in enumerate(np.asarray(d["K"], np.float64).reshape(-1)):
        p.K[i] = v
    p._keep = keep
    return p


def _rot_t(p7):
    x, y, z, w = p7[..., 3], p7[..., 4], p7[..., 5], p7[..., 6]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
    return R, p7[..., :3]


def second_stage_problem(d, errors):
    """The graph of the second optimize() (Optimizer.cc:2366-2411) from the estimates in `d` (those after the first five iterations) and
    the residuals at them: reprojection edges with chi2 > 5.991 (mono) / 7.815 (stereo) or a non-positive depth and dynamic-point edges with
    chi2 > 8 go to level 1, all three lose their robust kernel; camera-object edges with |error| > 80 go to level 1 and keep theirs."""
    d2 = dict(d)
    eo = np.asarray(errors["obs"]); w = np.asarray(d["obs_inv_sigma2"])
    ur = d.get("obs_ur")
    st = np.zeros(len(eo), bool) if ur is None else np.asarray(ur) >= 0
    chi = np.where(st, (eo ** 2).sum(1), (eo[:, :2] ** 2).sum(1)) * w
    R, t = _rot_t(np.asarray(d["cam_pose"])[np.asarray(d["obs_cam"], int)])
    z = np.einsum("nj,nj->n", R[:, 2, :], np.asarray(d["points"])[np.asarray(d["obs_point"], int)]) + t[:, 2]  # isDepthPositive
    lvl = np.asarray(d["obs_level"], np.uint8).copy() if d.get("obs_level") is not None else np.zeros(len(eo), np.uint8)
    lvl[(chi > np.where(st, 7.815, 5.991)) | ~(z > 0)] = 1
    d2["obs_level"] = lvl
    ed = np.asarray(errors["dobs"])
    dl = np.asarray(d["dobs_level"], np.uint8).copy() if d.get("dobs_level") is not None else np.zeros(len(ed), np.uint8)
    dl[(ed ** 2).sum(1) * np.asarray(d["dobs_inv_sigma2"]) > 8] = 1
    d2["dobs_level"] = dl
    ec = np.asarray(errors["cobs"])
    cl = np.asarray(d["cobs_level"], np.uint8).copy() if d.get("cobs_level") is not None else np.zeros(len(ec), np.uint8)
    cl[np.sqrt((ec ** 2).sum(1)) > 80] = 1
    d2["cobs_level"] = cl
    d2["huber_mono"] = d2["huber_stereo"] = d2["huber_dyn"] = 0.0
    return d2


class DynamicBundleAdjuster:
    """g2o::SparseOptimizer + OptimizationAlgorithmLevenberg + BlockSolverX / LinearSolverDense on the GPU for one stage of the dynamic BA."""

    def __init__(self, problem, ctx=None, device=0):
        self.ctx = ctx or _lib.Context(device)
        self.d = problem
        self.p = problem_struct(problem)
        self._b = C.c_void_p()
        check(self.ctx.ptr, lib().cs_ba_dyn_create(self.ctx.ptr, C.byref(self.p), C.byref(self._b)), "cs_ba_dyn_create")

    def optimize(self, iterations):
        st = BAStats()
        check(self.ctx.ptr, lib().cs_ba_dyn_optimize(self.ctx.ptr, self._b, int(iterations), None, C.byref(st)), "cs_ba_dyn_optimize")
        return {"iterations": st.iterations, "lm_trials": st.lm_trials, "chi2_init": st.chi2_init, "chi2_final": st.chi2_final, "lambda_final": st.lambda_final,
                "chi2_trace": list(st.chi2_trace)[:st.iterations]}

    def _shapes(self):
        p = self.p
        return (p.n_cams, 7), (p.n_objs, 7), (p.n_vels, 2), (p.n_points, 3), (p.n_dpoints, 3)

    def read(self):
        out = [np.zeros((max(n, 1), k)) for n, k in self._shapes()]
        check(self.ctx.ptr, lib().cs_ba_dyn_read(self.ctx.ptr, self._b, *[a.ctypes.data_as(C.c_void_p) for a in out]), "cs_ba_dyn_read")
        return dict(zip(("cam_pose", "obj_pose", "vel", "points", "dpoints"), [a[:n] for a, (n, _) in zip(out, self._shapes())]))

    def errors(self):
        p = self.p
        sh = ((p.n_obs, 3), (p.n_dobs, 2), (p.n_mot, 3), (p.n_cobs, 4), (p.n_pc, 3), (p.n_dpoints, 3))
        e = [np.zeros((max(n, 1), k)) for n, k in sh]
        chi = C.c_double()
        check(self.ctx.ptr, lib().cs_ba_dyn_errors(self.ctx.ptr, self._b, C.byref(chi), *[a.ctypes.data_as(C.c_void_p) for a in e]), "cs_ba_dyn_errors")
        return chi.value, dict(zip(("obs", "dobs", "mot", "cobs", "pc", "ulp"), [a[:n] for a, (n, _) in zip(e, sh)]))

    def reduced_dense(self, lam):
        n = C.c_int()
        check(self.ctx.ptr, lib().cs_ba_dyn_reduced_dense(self.ctx.ptr, self._b, C.c_double(lam), None, None, C.byref(n)), "cs_ba_dyn_reduced_dense")
        H = np.zeros((max(n.value, 1), max(n.value, 1))); b = np.zeros(max(n.value, 1))
        check(self.ctx.ptr, lib().cs_ba_dyn_reduced_dense(self.ctx.ptr, self._b, C.c_double(lam), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(n)),
              "cs_ba_dyn_reduced_dense")
        return H[:n.value, :n.value], b[:n.value]

    def close(self):
        if self._b:
            lib().cs_ba_dyn_destroy(self.ctx.ptr, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def erase_rows(d1, e1, d2, fin, e2):
    """vToErase (Optimizer.cc:2417-2444) as rows of obs_*: the static point edges whose stored chi2 exceeds 5.991 (mono) / 7.815 (stereo) or whose point lies behind
    the camera at the final estimates, monocular edges first.  An edge at level 1 in the second stage keeps the error it had after the first."""
    eo1, eo2 = np.asarray(e1["obs"]), np.asarray(e2["obs"])
    w = np.asarray(d1["obs_inv_sigma2"])
    ur = d1.get("obs_ur")
    st = np.zeros(len(eo1), bool) if ur is None else np.asarray(ur) >= 0
    chi1 = np.where(st, (eo1 ** 2).sum(1), (eo1[:, :2] ** 2).sum(1)) * w
    chi2 = np.where(st, (eo2 ** 2).sum(1), (eo2[:, :2] ** 2).sum(1)) * w
    lvl = np.asarray(d2["obs_level"])
    first = np.zeros(len(eo1), bool) if d1.get("obs_level") is None else np.asarray(d1["obs_level"]) != 0   # never in the graph: not in vpEdgesMono / vpEdgesStereo
    chi = np.where(lvl == 0, chi2, chi1)
    R, t = _rot_t(np.asarray(fin["cam_pose"])[np.asarray(d1["obs_cam"], int)])
    z = np.einsum("nj,nj->n", R[:, 2, :], np.asarray(fin["points"])[np.asarray(d1["obs_point"], int)]) + t[:, 2]
    bad = ((chi > np.where(st, 7.815, 5.991)) | ~(z > 0)) & ~first
    return np.array([k for k in np.nonzero(~st)[0] if bad[k]] + [k for k in np.nonzero(st)[0] if bad[k]], int)


def optimize_two_stages(problem, ctx=None):
    """Optimizer.cc:2353-2444: optimize(5), outlier levels + kernel removal, optimize(10), the observations to erase.  Returns the final estimates (with
    `erase_obs`: the rows of obs_* the caller erases, in the reference's order), the stage-2 problem and both stages' statistics.  The caller writes back what
    :2446-2572 writes: poses, static points unless the erasures leave them one observation, every object vertex into allDynamicPoses, velocities, dynamic points."""
    ba = DynamicBundleAdjuster(problem, ctx=ctx)
    st1 = ba.optimize(5)
    d1 = dict(problem); d1.update(ba.read())
    _, e1 = ba.errors()
    ba.close()
    d2 = second_stage_problem(d1, e1)
    ba2 = DynamicBundleAdjuster(d2, ctx=ctx or ba.ctx)
    st2 = ba2.optimize(10)
    res = ba2.read()
    _, e2 = ba2.errors()
    ba2.close()
    res["erase_obs"] = erase_rows(d1, e1, d2, res, e2)
    return res, d2, (st1, st2)


KITTI_OBJECT_HALF_SIZE = (1.9420, 0.8143, 0.7631)  # Optimizer.cc:1733


def build_graph(w, params, fixCamera=False, fixPoint=False):
    """The g2o graph of :1667-2340 as cs_ba_dyn_problem arrays plus the bookkeeping that maps the result back to rows of the window:
    point_rows / dpoint_rows (rows of mp_*), obs_rows / dobs_rows (rows of obs_*), cobs_rows (rows of ov_*), vel_mo (rows of mo_*), velocity_init {row of mo_*: the
    value :2223-2233 writes into the object before the solve}, up_used / up_filtered (rows of up_*)."""
    if not params.get("kitti", True):
        raise ValueError("LocalBACameraPointObjectsDynamic: the reference fixes the object size for scene_unique_id == kitti only (Optimizer.cc:1762-1765)")
    n_kf, n_local = len(w["kf_id"]), int(w["n_local"])
    stamp = np.asarray(w["kf_stamp"], float); now = stamp[0]
    cam_fixed = np.ones(n_kf, np.uint8)
    cam_fixed[:n_local] = (np.asarray(w["kf_id"][:n_local]) == 0) | bool(fixCamera)                      # :1693-1696
    # ---- one cuboid vertex per row of ov_* :1727-1786
    ov_mo, ov_kf = np.asarray(w["ov_mo"], int), np.asarray(w["ov_kf"], int)
    ov_pose = np.array(w["ov_pose"], float).reshape(-1, 7)
    n_v, n_obj = len(ov_mo), len(w["mo_id"])
    obj_pose = ov_pose.copy()
    if not params.get("build_worldframe_on_ground", False) and n_v:
        obj_pose[:, 1] = np.asarray(w["kf_cam_center"], np.float32).reshape(-1, 3)[ov_kf, 1].astype(np.float64) + 1.0   # :1771-1772
    vertex = {(int(m), int(k)): v for v, (m, k) in enumerate(zip(ov_mo, ov_kf))}
    n_vert_of = np.bincount(ov_mo, minlength=n_obj) if n_v else np.zeros(n_obj, int)
    # ---- static points :1808-1906, dynamic points :1919-2001
    nobs, dyn = np.asarray(w["mp_nobs"], int), np.asarray(w["mp_dynamic"], bool)
    best = np.asarray(w["mp_best_mo"], int)
    point_rows = np.nonzero((nobs != 1) & ~dyn)[0]
    prow = -np.ones(len(nobs), int); prow[point_rows] = np.arange(len(point_rows))
    obs_mp, obs_kf = np.asarray(w["obs_mp"], int), np.asarray(w["obs_kf"], int)
    obs_rows = np.nonzero(prow[obs_mp] >= 0)[0]
    dpoint_rows = np.nonzero(dyn & (nobs >= 4) & (best >= 0))[0] if params.get("ba_dyna_pt_obj_cam", True) else np.zeros(0, int)
    drow = -np.ones(len(nobs), int); drow[dpoint_rows] = np.arange(len(dpoint_rows))
    dobs_rows = [o for o in np.nonzero(drow[obs_mp] >= 0)[0] if (int(best[obs_mp[o]]), int(obs_kf[o])) in vertex]     # :1965-1966
    dobs_rows = np.array(dobs_rows, int)
    dobs_obj = np.array([vertex[(int(best[obs_mp[o]]), int(obs_kf[o]))] for o in dobs_rows], np.int32)
    # ---- point-object association :2008-2115.  `optimizer.vertex(pMObj->mnId + maxKFid + 1)` is the id scheme of the STATIC function: maxKFid was incremented before
    # the cuboid vertices took maxKFid + 1, + 2, ... (:1730), so the id names the (mnId + 1)-th cuboid vertex created, whichever object it belongs to
    up_mo, up_pos, up_count = np.asarray(w["up_mo"], int), np.asarray(w["up_pos"], float).reshape(-1, 3), np.asarray(w["up_count"], int)
    pc_obj, pc_off, pc_pts, up_used, up_filtered = [], [0], [], [], []
    for i in range(n_obj):
        thr = max(int(int(w["mo_largest_point_observations"][i]) * 0.4), 2)
        rows = np.nonzero((up_mo == i) & (up_count > thr))[0]
        up_used += [int(r) for r in rows]
        P = up_pos[rows]
        good = np.zeros((0, 3))
        if len(P):
            mean = np.zeros(3)
            for p in P:
                mean = mean + p
            mean = mean / float(len(P))
            near = np.linalg.norm(mean - P, axis=1) < 4.0
            mean2 = np.zeros(3)
            for p in P[near]:
                mean2 = mean2 + p
            mean2 = mean2 / float(near.sum()) if near.any() else np.full(3, np.nan)
            keep = np.linalg.norm(mean2 - P, axis=1) < 3.0
            good = P[keep]; up_filtered += [int(r) for r in rows[keep]]
        named = int(w["mo_id"][i])
        if len(good) > 5:
            if not 0 <= named < n_v:
                raise ValueError("LocalBACameraPointObjectsDynamic: object %d names cuboid vertex %d of %d (the reference dereferences a null vertex here, Optimizer.cc:2075)" % (i, named, n_v))
            acc = np.zeros(3)
            for p in good:
                acc = acc + p
            obj_pose[named, :3] = acc / float(len(good))
        if len(good) > 10:
            pc_obj.append(named); pc_pts.append(good); pc_off.append(pc_off[-1] + len(good))
    # ---- velocity vertices, motion edges :2137-2237
    vel, vel_mo, mot_from, mot_to, mot_vel, mot_dt, velocity_init = [], [], [], [], [], [], {}
    velo = params.get("ba_dyna_obj_velo", True)
    if velo:
        seq_mo, seq_kf = np.asarray(w["seq_mo"], int), np.asarray(w["seq_kf"], int)
        for i in range(n_obj):
            if n_vert_of[i] < 4:
                continue
            vi = len(vel); v0 = np.array(w["mo_velocity"][i], float); vel.append(v0); vel_mo.append(i)
            first = last = prev = None
            for k in seq_kf[seq_mo == i]:
                k = int(k)
                if (i, k) not in vertex or (now - stamp[k]) > 5.0:
                    continue
                if prev is None:
                    prev = first = k
                else:
                    mot_from.append(vertex[(i, prev)]); mot_to.append(vertex[(i, k)]); mot_vel.append(vi); mot_dt.append(stamp[k] - stamp[prev])
                    prev = last = k
            if v0[0] == 0 and v0[1] == 0 and first is not None and last is not None:
                a, b = ov_pose[vertex[(i, first)], :3], ov_pose[vertex[(i, last)], :3]     # allDynamicPoses as they are stored, not the height-reset estimates
                vel[vi] = np.array([float(np.linalg.norm(b - a)) / (stamp[last] - stamp[first]), 0.0]); velocity_init[i] = vel[vi].copy()
    # ---- camera-object edges :2243-2340 (no halving for more than five objects here)
    w_co = 1.0 * params.get("camera_object_BA_weight", 1.0)
    m, Wimg, Himg = 10, params["img_width"], params["img_height"]
    cobs_rows, cobs_info, cobs_level, lr = [], [], [], []
    if params.get("ba_dyna_obj_cam", True):
        rect = np.asarray(w["ov_bbox_2d"], int).reshape(-1, 4)
        for i in range(n_obj):
            mine = []
            for v in np.nonzero(ov_mo == i)[0]:
                if velo and (now - stamp[ov_kf[v]]) > 5.0:
                    continue
                x, y, ww, hh = rect[v]
                if x > m and y > m and x + ww < Wimg - m and y + hh < Himg - m:
                    q = float(w["mo_meas_quality"][i])
                    mine.append(len(cobs_rows)); cobs_rows.append(int(v)); cobs_info.append(np.full(4, w_co * w_co) * q * q); cobs_level.append(0)
                    lr.append(int(w["ov_left_right_to_car"][v]))
            if len(mine) == 1:
                cobs_level[mine[0]] = 1
        lr = np.array(lr, int)
        tl, tr, tm = int((lr == 1).sum()), int((lr == 2).sum()), int((lr == 0).sum())
        if tl > 2 * (tr + tm):
            for k in np.nonzero(lr == 1)[0]:
                cobs_info[k] = cobs_info[k] / 2.0
        if tr > 2 * (tl + tm):
            for k in np.nonzero(lr == 2)[0]:
                cobs_info[k] = cobs_info[k] / 2.0
    cobs_rows = np.array(cobs_rows, int)
    K = np.asarray(params["K"], float)
    wv = params.get("object_velocity_BA_weight", 1.0)
    mp_pos, uv, ur, isg = np.asarray(w["mp_pos"], float).reshape(-1, 3), np.asarray(w["obs_uv"], float).reshape(-1, 2), np.asarray(w["obs_ur"], float), np.asarray(w["obs_inv_sigma2"], float)
    d = {"cam_pose": np.array(w["kf_pose"], float).reshape(-1, 7), "cam_fixed": cam_fixed,
         "obj_pose": obj_pose, "obj_scale": np.tile(np.array(KITTI_OBJECT_HALF_SIZE), (n_v, 1)).reshape(-1, 3), "obj_flags": np.full(n_v, 2 | 8, np.uint8),   # whether_fixrotation, fixed scale
         "vel": np.array(vel, float).reshape(-1, 2), "points": mp_pos[point_rows], "dpoints": np.asarray(w["mp_pos_to_obj"], float).reshape(-1, 3)[dpoint_rows], "fix_points": int(fixPoint),
         "obs_cam": obs_kf[obs_rows].astype(np.int32), "obs_point": prow[obs_mp[obs_rows]].astype(np.int32), "obs_uv": uv[obs_rows], "obs_ur": ur[obs_rows],
         "obs_inv_sigma2": isg[obs_rows], "obs_level": np.zeros(len(obs_rows), np.uint8),
         "fx": K[0, 0], "fy": K[1, 1], "cx": K[0, 2], "cy": K[1, 2], "bf": params.get("bf", 0.0),
         "huber_mono": float(np.float32(math.sqrt(5.991))), "huber_stereo": float(np.float32(math.sqrt(7.815))),            # `const float` widths :1802-1803
         "ulp_info": 10.0, "ulp_scale": np.array(KITTI_OBJECT_HALF_SIZE), "ulp_ratio": 2.0,                                 # UnaryLocalPoint :1950-1956
         "dobs_cam": obs_kf[dobs_rows].astype(np.int32), "dobs_obj": dobs_obj, "dobs_point": drow[obs_mp[dobs_rows]].astype(np.int32) if len(dobs_rows) else np.zeros(0, np.int32),
         "dobs_uv": uv[dobs_rows] if len(dobs_rows) else np.zeros((0, 2)), "dobs_inv_sigma2": isg[dobs_rows] if len(dobs_rows) else np.zeros(0),
         "dobs_level": np.zeros(len(dobs_rows), np.uint8), "K": K, "huber_dyn": float(np.float32(math.sqrt(5.991))),
         "mot_from": np.array(mot_from, np.int32), "mot_to": np.array(mot_to, np.int32), "mot_vel": np.array(mot_vel, np.int32), "mot_dt": np.array(mot_dt, float),
         "mot_info": (np.array([1.0, 1.0, 5.0]) * wv) ** 2,
         "cobs_cam": ov_kf[cobs_rows].astype(np.int32) if len(cobs_rows) else np.zeros(0, np.int32), "cobs_obj": cobs_rows.astype(np.int32),
         "cobs_bbox": np.asarray(w["ov_bbox_vec"], float).reshape(-1, 4)[cobs_rows] if len(cobs_rows) else np.zeros((0, 4)),
         "cobs_info": np.array(cobs_info, float).reshape(-1, 4), "cobs_level": np.array(cobs_level, np.uint8), "huber_obj": float(np.float32(math.sqrt(900.0))),
         "pc_obj": np.array(pc_obj, np.int32), "pc_offsets": np.array(pc_off, np.int32), "pc_points": np.concatenate(pc_pts).reshape(-1, 3) if pc_pts else np.zeros((0, 3)),
         "pc_ratio": 2.0}
    return {"problem": d, "point_rows": point_rows, "obs_rows": obs_rows, "dpoint_rows": dpoint_rows, "dobs_rows": dobs_rows, "cobs_rows": cobs_rows, "vel_mo": np.array(vel_mo, int),
            "velocity_init": velocity_init, "up_used": up_used, "up_filtered": up_filtered}


def LocalBACameraPointObjectsDynamic(window, params, ctx=None, fixCamera=False, fixPoint=False):
    """-> what the caller writes back (:2446-2572), over rows of the window: kf_pose (n_local,7); point_pos {row of mp_*: xyz} and point_unwritten (rows the erasures leave
    with one observation: not written, :2478); erase [(row of kf_*, row of mp_*)] + erase_stereo; vertex_pose (v,7: allDynamicPoses[key frame] of every row of ov_*, flag true);
    object_latest {row of mo_*: row of ov_* whose key frame has the largest mnId}: pose_Twc_latestKF / SetWorldPos / pose_Twc_afterba; velocity {row of mo_*: v}
    (velocityPlanar, velocityhistory[pKF]) and velocity_init (written before the solve even when the function is stopped); dpoint_local {row of mp_*: PosToObj} and
    dpoint_world {row: pose_Twc_latestKF.pose * estimate}: mWorldPos_latestKF / SetWorldPos / is_optimized; up_used / up_filtered."""
    g = build_graph(window, params, fixCamera, fixPoint)
    res, d2, stats = optimize_two_stages(g["problem"], ctx=ctx)
    rows = g["obs_rows"]
    st = np.asarray(g["problem"]["obs_ur"]) >= 0
    erase = [(int(window["obs_kf"][rows[k]]), int(window["obs_mp"][rows[k]])) for k in res["erase_obs"]]
    erase_stereo = [bool(st[k]) for k in res["erase_obs"]]
    left = np.asarray(window["mp_nobs"], int).copy()
    for (_, r), s_ in zip(erase, erase_stereo):
        left[r] -= 2 if s_ else 1
    ov_mo, ov_kf, kf_id = np.asarray(window["ov_mo"], int), np.asarray(window["ov_kf"], int), np.asarray(window["kf_id"])
    latest = {}
    for v in range(len(ov_mo)):
        i = int(ov_mo[v])
        if i not in latest or kf_id[ov_kf[v]] > kf_id[ov_kf[latest[i]]]:
            latest[i] = v
    dlocal, dworld = {}, {}
    for j, r in enumerate(g["dpoint_rows"]):
        dlocal[int(r)] = res["dpoints"][j]
        i = int(window["mp_best_mo"][r])
        if i in latest:
            R, t = _rot_t(res["obj_pose"][latest[i]])
            dworld[int(r)] = R @ res["dpoints"][j] + t
    return {"kf_pose": res["cam_pose"][:int(window["n_local"])], "point_pos": {int(r): res["points"][j] for j, r in enumerate(g["point_rows"])},
            "point_unwritten": [int(r) for r in g["point_rows"] if left[r] == 1], "erase": erase, "erase_stereo": erase_stereo, "vertex_pose": res["obj_pose"],
            "object_latest": latest, "velocity": {int(i): res["vel"][k] for k, i in enumerate(g["vel_mo"])}, "velocity_init": g["velocity_init"],
            "dpoint_local": dlocal, "dpoint_world": dworld, "up_used": g["up_used"], "up_filtered": g["up_filtered"], "stats": stats, "graph": g, "stage2": d2}
