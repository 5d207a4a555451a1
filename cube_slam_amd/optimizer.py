"""Python host-side mirror of ORB_SLAM2::Optimizer (reference orb_object_slam/include/Optimizer.h:36-52): PoseOptimization here,
BundleAdjustment / LocalBACameraPointObjects in cube_slam_amd.ba."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def PoseOptimization(frames, ctx=None, device=0):
    """frames: list of dicts {Xw (n,3), obs (n,3: u, v, u_right or -1), inv_sigma2 (n,), intr (fx, fy, cx, cy, bf), pose (7,)}.
    Returns a list of (pose (7,), outlier flags (n,) u8, n_inliers), one per frame -- what Optimizer::PoseOptimization(Frame*)
    writes into pFrame->mTcw / mvbOutlier and returns (Optimizer.cc:253-472)."""
    ctx = ctx or _lib.Context(device)
    F = len(frames)
    off = np.zeros(F + 1, np.int32)
    for f, fr in enumerate(frames):
        off[f + 1] = off[f] + len(fr["Xw"])
    ne = int(off[F])
    cat = lambda k, w: (np.concatenate([np.asarray(fr[k], np.float64).reshape(-1, w) for fr in frames]) if ne else np.zeros((1, w)))
    Xw, obs = np.ascontiguousarray(cat("Xw", 3)), np.ascontiguousarray(cat("obs", 3))
    w = np.ascontiguousarray(np.concatenate([np.asarray(fr["inv_sigma2"], np.float64).reshape(-1) for fr in frames])) if ne else np.zeros(1)
    intr = np.ascontiguousarray(np.stack([np.asarray(fr["intr"], np.float64) for fr in frames])) if F else np.zeros((1, 5))
    pin = np.ascontiguousarray(np.stack([np.asarray(fr["pose"], np.float64) for fr in frames])) if F else np.zeros((1, 7))
    pout = np.zeros((max(F, 1), 7)); flags = np.zeros(max(ne, 1), np.uint8); ninl = np.zeros(max(F, 1), np.int32)
    check(ctx.ptr, lib().cs_pose_optimization(ctx.ptr, F, _p(off, C.c_int), _p(Xw, C.c_double), _p(obs, C.c_double), _p(w, C.c_double), _p(intr, C.c_double),
                                              _p(pin, C.c_double), _p(pout, C.c_double), _p(flags, C.c_uint8), _p(ninl, C.c_int)), "cs_pose_optimization")
    return [(pout[f].copy(), flags[off[f]:off[f + 1]].copy(), int(ninl[f])) for f in range(F)]


def cuboid9_oplus(cub, upd, ctx=None, device=0):
    """g2o::VertexCuboid::oplusImpl for a batch (object_slam/include/object_slam/g2o_Object.h:193-204); cuboid = [t, q, half scale]."""
    ctx = ctx or _lib.Context(device)
    cub = np.ascontiguousarray(cub, np.float64).reshape(-1, 10); upd = np.ascontiguousarray(upd, np.float64).reshape(-1, 9)
    out = np.zeros_like(cub)
    check(ctx.ptr, lib().cs_cuboid9_oplus(ctx.ptr, len(cub), _p(cub, C.c_double), _p(upd, C.c_double), _p(out, C.c_double)), "cs_cuboid9_oplus")
    return out


def cuboid9_edge_linearize(cam_Tcw, cub_global, cub_meas, jac=True, ctx=None, device=0):
    """g2o::EdgeSE3Cuboid::computeError (+ g2o's numeric Jacobians) for a batch of camera-object edges (g2o_Object.h:227-252)."""
    ctx = ctx or _lib.Context(device)
    T = np.ascontiguousarray(cam_Tcw, np.float64).reshape(-1, 7); g = np.ascontiguousarray(cub_global, np.float64).reshape(-1, 10)
    m = np.ascontiguousarray(cub_meas, np.float64).reshape(-1, 10)
    n = len(T)
    err = np.zeros((n, 9)); Jc = np.zeros((n, 9, 6)); Jq = np.zeros((n, 9, 9))
    check(ctx.ptr, lib().cs_cuboid9_edge_linearize(ctx.ptr, n, _p(T, C.c_double), _p(g, C.c_double), _p(m, C.c_double), _p(err, C.c_double),
                                                   _p(Jc, C.c_double) if jac else None, _p(Jq, C.c_double) if jac else None), "cs_cuboid9_edge_linearize")
    return (err, Jc, Jq) if jac else err
