"""Python host-side mirror of the object bundle adjustment entry points (reference orb_object_slam/include/Optimizer.h:39-62:
Optimizer::BundleAdjustment / LocalBACameraPointObjects) over the C-ABI.  The map objects of the reference (KeyFrame*,
MapPoint*, MapObject*) are flattened into the SoA arrays of cs_ba_problem."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class BAProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("cam_pose", C.c_void_p), ("cam_fixed", C.c_void_p),
                ("n_points", C.c_int), ("points", C.c_void_p),
                ("n_cuboids", C.c_int), ("cuboid_pose", C.c_void_p), ("cuboid_scale", C.c_void_p), ("cuboid_flags", C.c_void_p),
                ("n_obs", C.c_int), ("obs_cam", C.c_void_p), ("obs_point", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_inv_sigma2", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("huber_mono", C.c_double),
                ("n_cobs", C.c_int), ("cobs_cam", C.c_void_p), ("cobs_cuboid", C.c_void_p), ("cobs_bbox", C.c_void_p), ("cobs_info", C.c_void_p),
                ("K", C.c_double * 9), ("huber_obj", C.c_double),
                ("n_pc", C.c_int), ("pc_cuboid", C.c_void_p), ("pc_offsets", C.c_void_p), ("pc_points", C.c_void_p),
                ("max_outside_margin_ratio", C.c_double),
                ("obs_ur", C.c_void_p), ("bf", C.c_double), ("huber_stereo", C.c_double)]


class BAStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("lm_trials", C.c_int), ("chi2_init", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("chi2_trace", C.c_double * 64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long)


def problem_struct(d):
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(d[name], dt)
        keep[name] = a
        return a.ctypes.data

    p = BAProblem()
    p.n_cams = len(d["cam_pose"]); p.cam_pose = arr("cam_pose", np.float64); p.cam_fixed = arr("cam_fixed", np.uint8)
    p.n_points = len(d["points"]); p.points = arr("points", np.float64)
    p.n_cuboids = len(d["cuboid_pose"]); p.cuboid_pose = arr("cuboid_pose", np.float64); p.cuboid_scale = arr("cuboid_scale", np.float64)
    p.cuboid_flags = arr("cuboid_flags", np.uint8)
    p.n_obs = len(d["obs_cam"]); p.obs_cam = arr("obs_cam", np.int32); p.obs_point = arr("obs_point", np.int32); p.obs_uv = arr("obs_uv", np.float64)
    p.obs_inv_sigma2 = arr("obs_inv_sigma2", np.float64)
    p.fx, p.fy, p.cx, p.cy, p.huber_mono = d["fx"], d["fy"], d["cx"], d["cy"], d["huber_mono"]
    p.n_cobs = len(d["cobs_cam"]); p.cobs_cam = arr("cobs_cam", np.int32); p.cobs_cuboid = arr("cobs_cuboid", np.int32)
    p.cobs_bbox = arr("cobs_bbox", np.float64); p.cobs_info = arr("cobs_info", np.float64)
    for i, v in enumerate(np.asarray(d["K"], np.float64).reshape(-1)):
        p.K[i] = v
    p.huber_obj = d["huber_obj"]
    p.n_pc = len(d["pc_cuboid"]); p.pc_cuboid = arr("pc_cuboid", np.int32); p.pc_offsets = arr("pc_offsets", np.int32); p.pc_points = arr("pc_points", np.float64)
    p.max_outside_margin_ratio = d["max_outside_margin_ratio"]
    if d.get("obs_ur") is not None:  # stereo observations (EdgeStereoSE3ProjectXYZ): u_right >= 0
        p.obs_ur = arr("obs_ur", np.float64)
    p.bf, p.huber_stereo = d.get("bf", 0.0), d.get("huber_stereo", 0.0)
    p._keep = keep
    return p


def shard_landmarks(n_points, rank, world):
    """Landmark range owned by `rank` (contiguous blocks; cameras and cuboids are replicated)."""
    return n_points * rank // world, n_points * (rank + 1) // world


class BundleAdjuster:
    """g2o::SparseOptimizer + OptimizationAlgorithmLevenberg + BlockSolver_6_3 on the GPU."""

    def __init__(self, problem, ctx=None, device=0, rank=0, world=1, allreduce=None):
        self.ctx = ctx or _lib.Context(device)
        self.d = problem
        self.p = problem_struct(problem)
        self._b = C.c_void_p()
        check(self.ctx.ptr, lib().cs_ba_create(self.ctx.ptr, C.byref(self.p), rank, world, C.byref(self._b)), "cs_ba_create")
        self.rank, self.world = rank, world
        self._cb = None
        if allreduce is not None:
            def _cb(user, dev_ptr, n):
                try:
                    allreduce(int(dev_ptr), int(n))
                    return 0
                except Exception:  # pragma: no cover
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLREDUCE_FN(_cb)
            check(self.ctx.ptr, lib().cs_ba_set_allreduce(self._b, self._cb, None), "cs_ba_set_allreduce")

    def set_stop_flag_bool(self, flag):
        """flag: a ctypes c_ubyte another thread raises (the reference's `bool *pbStopFlag`, setForceStopFlag), or None; polled during every later optimize()."""
        self._stop8 = flag  # keep it alive
        check(self.ctx.ptr, lib().cs_ba_set_stop_flag_bool(self._b, None if flag is None else C.byref(flag)), "cs_ba_set_stop_flag_bool")

    def optimize(self, iterations, stop_flag=None):
        st = BAStats()
        check(self.ctx.ptr, lib().cs_ba_optimize(self.ctx.ptr, self._b, iterations, stop_flag, C.byref(st)), "cs_ba_optimize")
        return {"iterations": st.iterations, "lm_trials": st.lm_trials, "chi2_init": st.chi2_init, "chi2_final": st.chi2_final,
                "lambda_final": st.lambda_final, "chi2_trace": list(st.chi2_trace)[:min(st.iterations, 64)]}

    def read(self):
        cam = np.zeros((self.p.n_cams, 7)); pts = np.array(self.d["points"], np.float64).copy().reshape(-1, 3); cub = np.zeros((max(self.p.n_cuboids, 1), 7))
        check(self.ctx.ptr, lib().cs_ba_read(self.ctx.ptr, self._b, cam.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), cub.ctypes.data_as(C.c_void_p)), "cs_ba_read")
        return cam, pts, cub[:self.p.n_cuboids]

    def errors(self):
        chi = C.c_double()
        eo = np.zeros((max(self.p.n_obs, 1), 3)); ec = np.zeros((max(self.p.n_cobs, 1), 4)); ep = np.zeros((max(self.p.n_pc, 1), 3))
        check(self.ctx.ptr, lib().cs_ba_errors(self.ctx.ptr, self._b, C.byref(chi), eo.ctypes.data_as(C.c_void_p), ec.ctypes.data_as(C.c_void_p),
                                               ep.ctypes.data_as(C.c_void_p)), "cs_ba_errors")
        return chi.value, eo[:self.p.n_obs], ec[:self.p.n_cobs], ep[:self.p.n_pc]

    def reduced_dense(self, lam):
        P = int((1 - np.asarray(self.d["cam_fixed"])).sum()) + self.p.n_cuboids
        H = np.zeros((6 * P, 6 * P)); b = np.zeros(6 * P); Pout = C.c_int()
        check(self.ctx.ptr, lib().cs_ba_reduced_dense(self.ctx.ptr, self._b, C.c_double(lam), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(Pout)),
              "cs_ba_reduced_dense")
        assert Pout.value == P
        return H, b

    def close(self):
        if self._b:
            lib().cs_ba_destroy(self.ctx.ptr, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
