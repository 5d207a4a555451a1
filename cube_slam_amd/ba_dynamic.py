"""Python host-side mirror of Optimizer::LocalBACameraPointObjectsDynamic (reference orb_object_slam/include/Optimizer.h:57-58,
src/Optimizer.cc:1537-2573) over the C-ABI (cs_ba_dyn_*).  The map objects of the reference (KeyFrame*, MapPoint*, MapObject* with one
cuboid vertex per observing key frame, the velocity vertex) are flattened into the SoA arrays of cs_ba_dyn_problem; the dict layout is the
one cube_slam_amd.synth.ba_dyn_problem produces."""
import ctypes as C

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


def LocalBACameraPointObjectsDynamic(problem, ctx=None):
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
