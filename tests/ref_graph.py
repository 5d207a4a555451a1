"""Test helper: the reference's own graph-level functions (oracle/_ref/libref_graph.so: Optimizer::LocalBACameraPointObjects, BundleAdjustment,
PoseOptimization cut out of the reference and run on its vendored g2o, oracle/ref_shim/ref_graph_api.cpp) over the pointer graphs of
tests/local_map.py.  The reference keeps poses and points as FLOAT cv::Mat: `quantize` moves a synthetic window onto values that survive that
storage and hands the oracle's objects exactly the estimates the reference's vertices start from."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_graph.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(SO)
        L.ref_graph_open.restype = C.c_void_p
        L.ref_graph_pose_optimization.restype = C.c_int
        for name in ("ref_graph_add_kf", "ref_graph_add_mp", "ref_graph_add_mo", "ref_graph_kf_detection", "ref_graph_erased"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, np.float32))


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_from_cvmat(T):
    out = np.zeros(7)
    lib().ref_graph_pose_from_cvmat(_p(_f32(T).reshape(-1), C.c_float), _p(out, C.c_double))
    return out


def cvmat_from_pose(p):
    out = np.zeros(16, np.float32)
    lib().ref_graph_cvmat_from_pose(_p(np.ascontiguousarray(p, np.float64), C.c_double), _p(out, C.c_float))
    return out.reshape(4, 4)


def quantize(cur, params, extra):
    """In place: poses -> 4 x 4 float matrices and the SE3Quat the reference builds from them, camera centres as KeyFrame::SetPose stores them (cv::Mat product
    of floats: accumulated in double, rounded once), key points / right coordinates / points / intrinsics -> float values."""
    for kf in extra["kfs"]:
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = _rot(kf.Tcw[3:]).astype(np.float32); T[:3, 3] = kf.Tcw[:3].astype(np.float32)
        kf.T_f32 = T
        kf.Tcw = pose_from_cvmat(T)
        kf.Ow = (-(T[:3, :3].astype(np.float64).T @ T[:3, 3].astype(np.float64))).astype(np.float32)
        kf.mvKeysUn = np.asarray(kf.mvKeysUn, np.float32).astype(np.float64)
        kf.mvuRight = np.asarray(kf.mvuRight, np.float32).astype(np.float64)
        kf.mvInvLevelSigma2 = np.asarray(kf.mvInvLevelSigma2, np.float32).astype(np.float64)
    pts = list(extra["mps"])
    for mo in extra["mos"]:
        pts += [mp for mp in mo.unique_points if mp is not None]
    for mp in pts:
        mp.pos = np.asarray(mp.pos, np.float32).astype(np.float64)
        if getattr(mp, "PosToObj", None) is not None:
            mp.PosToObj = np.asarray(mp.PosToObj, np.float32).astype(np.float64)
    params["K"] = np.asarray(params["K"], np.float32).astype(np.float64)
    params["bf"] = float(np.float32(params["bf"]))
    return cur, params, extra


class Graph:
    """The window as the reference's functions see it."""

    def __init__(self, cur, params, extra, verbose=False):
        L = lib()
        self.L, self.h = L, C.c_void_p(L.ref_graph_open())
        K = np.ascontiguousarray(params["K"], np.float64)
        L.ref_graph_set_params(self.h, int(params.get("kitti", True)), int(params.get("build_worldframe_on_ground", False)), C.c_double(params.get("camera_object_BA_weight", 1.0)),
                               0, int(params["img_width"]), int(params["img_height"]), _p(K.reshape(-1), C.c_double), int(verbose))
        self.kf, self.mp, self.mo = {}, {}, {}
        kfs = list(extra["kfs"])
        mps = list(extra["mps"])
        for mo in extra["mos"]:
            mps += [m for m in mo.unique_points if m is not None and all(m is not x for x in mps)]
        for k in kfs:
            n = len(k.mvKeysUn)
            keys, ur, octv, sig = _f32(k.mvKeysUn).reshape(-1), _f32(k.mvuRight), np.ascontiguousarray(k.octave, np.int32), _f32(k.mvInvLevelSigma2)
            self.kf[id(k)] = L.ref_graph_add_kf(self.h, C.c_long(k.mnId), int(k.bad), _p(_f32(k.T_f32).reshape(-1), C.c_float), _p(_f32(k.Ow), C.c_float), n, _p(keys, C.c_float), _p(ur, C.c_float),
                                                _p(octv, C.c_int), len(sig), _p(sig, C.c_float), C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]), C.c_float(K[1, 2]), C.c_float(params["bf"]))
        for m in mps:
            self.mp[id(m)] = L.ref_graph_add_mp(self.h, C.c_long(m.mnId), int(m.bad), _p(_f32(m.pos), C.c_float), 0)
        for o in extra["mos"]:
            self.mo[id(o)] = L.ref_graph_add_mo(self.h, C.c_long(o.mnId), int(o.bad), _p(np.ascontiguousarray(o.pose, np.float64), C.c_double), _p(np.ascontiguousarray(o.scale, np.float64), C.c_double),
                                                C.c_double(o.meas_quality), int(o.largest_point_observations))
        for k in kfs:
            for other in k.covisible:
                L.ref_graph_kf_covisible(self.h, self.kf[id(k)], self.kf[id(other)])
            for i, m in enumerate(k.map_point_matches):
                if m is not None:
                    L.ref_graph_kf_match(self.h, self.kf[id(k)], i, self.mp[id(m)])
            for det, lm in zip(k.local_cuboids, list(k.cuboids_landmark) + [Ellipsis] * (len(k.local_cuboids) - len(k.cuboids_landmark))):
                L.ref_graph_kf_detection(self.h, self.kf[id(k)], _p(np.ascontiguousarray(det["bbox_vec"], np.float64), C.c_double), _p(np.ascontiguousarray(det["bbox_2d"], np.int32), C.c_int),
                                         int(det["left_right_to_car"]), C.c_double(1.0), -2 if lm is Ellipsis else (-1 if lm is None else self.mo[id(lm)]))
        for m in mps:
            for k, i in m.observations.items():
                L.ref_graph_mp_observe(self.h, self.mp[id(m)], self.kf[id(k)], int(i))
        for o in extra["mos"]:
            for k, i in o.observations.items():
                L.ref_graph_mo_observe(self.h, self.mo[id(o)], self.kf[id(k)], int(i))
            for m in o.unique_points:
                L.ref_graph_mo_unique_point(self.h, self.mo[id(o)], -1 if m is None else self.mp[id(m)], 0 if m is None else int(m.MapObjObservations.get(o, 0)))
        self.kfs, self.mps, self.mos = kfs, mps, list(extra["mos"])
        # the dynamic-object BA's extra state: time stamps, dynamic points, per-frame object poses, velocities
        for k in kfs:
            if hasattr(k, "mTimeStamp"):
                L.ref_graph_kf_stamp(self.h, self.kf[id(k)], C.c_double(k.mTimeStamp))
        for m in mps:
            if getattr(m, "is_dynamic", False):
                L.ref_graph_mp_dynamic(self.h, self.mp[id(m)], _p(_f32(m.PosToObj), C.c_float), -1 if m.best_object is None else self.mo[id(m.best_object)])
        for o in self.mos:
            for k, pose in getattr(o, "allDynamicPoses", {}).items():
                L.ref_graph_mo_dynamic_pose(self.h, self.mo[id(o)], self.kf[id(k)], _p(np.ascontiguousarray(pose, np.float64), C.c_double), _p(np.ascontiguousarray(o.scale, np.float64), C.c_double))
            if hasattr(o, "velocityPlanar"):
                L.ref_graph_mo_velocity(self.h, self.mo[id(o)], _p(np.ascontiguousarray(o.velocityPlanar, np.float64), C.c_double))
        if "ba_dyna_obj_velo" in params:
            L.ref_graph_set_dyn_params(self.h, int(params["ba_dyna_pt_obj_cam"]), int(params["ba_dyna_obj_velo"]), int(params["ba_dyna_obj_cam"]),
                                       C.c_double(params.get("object_velocity_BA_weight", 1.0)), 1)

    def close(self):
        if self.h:
            self.L.ref_graph_close(self.h); self.h = None

    def local_ba_objects(self, cur, fix_camera=False, fix_point=False):
        self.L.ref_graph_local_ba_objects(self.h, self.kf[id(cur)], int(fix_camera), int(fix_point), None)

    def local_ba_dynamic(self, cur, fix_camera=False, fix_point=False):
        self.L.ref_graph_local_ba_dynamic(self.h, self.kf[id(cur)], int(fix_camera), int(fix_point), None)

    def mo_dynamic_pose(self, mo, kf):
        pose = np.zeros(7); baed = C.c_int(0)
        self.L.ref_graph_mo_dynamic_pose_out.restype = C.c_int
        ok = self.L.ref_graph_mo_dynamic_pose_out(self.h, self.mo[id(mo)], self.kf[id(kf)], _p(pose, C.c_double), C.byref(baed))
        return (pose, bool(baed.value)) if ok else (None, False)

    def mo_dynamic_state(self, mo):
        a, b, v, hst = np.zeros(7), np.zeros(7), np.zeros(2), np.zeros(2)
        n, lf = C.c_int(0), C.c_long(0)
        self.L.ref_graph_mo_dynamic_state(self.h, self.mo[id(mo)], _p(a, C.c_double), _p(b, C.c_double), _p(v, C.c_double), C.byref(n), _p(hst, C.c_double), C.byref(lf))
        return {"latest": a, "afterba": b, "velocity": v, "n_history": n.value, "history": hst, "local_for": lf.value}

    def mp_dynamic(self, mp):
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        o, bad, lf = C.c_int(0), C.c_int(0), C.c_long(0)
        self.L.ref_graph_mp_dynamic_out(self.h, self.mp[id(mp)], _p(a, C.c_float), _p(b, C.c_float), C.byref(o), C.byref(bad), C.byref(lf))
        return {"PosToObj": a, "latest": b, "is_optimized": bool(o.value), "bad": bool(bad.value), "local_for": lf.value}

    def bundle_adjustment(self, iterations, loop_kf=0, robust=True):
        self.L.ref_graph_bundle_adjustment(self.h, int(iterations), C.c_ulong(loop_kf), int(robust), None)

    def local_ba(self, kf):
        """Optimizer::LocalBundleAdjustment (Optimizer.cc:474-825)."""
        self.L.ref_graph_local_ba(self.h, self.kf[id(kf)], None)

    def global_ba(self, iterations, loop_kf=0, robust=True):
        """Optimizer::GlobalBundleAdjustemnt (Optimizer.cc:57-62)."""
        self.L.ref_graph_global_ba(self.h, int(iterations), C.c_ulong(loop_kf), int(robust), None)

    def pose_optimization(self, kf):
        T = np.zeros(16, np.float32); out = np.zeros(len(kf.mvKeysUn), np.uint8)
        n = self.L.ref_graph_pose_optimization(self.h, self.kf[id(kf)], _p(T, C.c_float), _p(out, C.c_ubyte))
        return n, T.reshape(4, 4), out.astype(bool)

    def kf_pose(self, kf):
        T = np.zeros(16, np.float32); n = C.c_int(0); G = np.full(16, np.nan, np.float32)
        self.L.ref_graph_kf_pose(self.h, self.kf[id(kf)], _p(T, C.c_float), C.byref(n), _p(G, C.c_float))
        return T.reshape(4, 4), n.value, G.reshape(4, 4)

    def kf_markers(self, kf):
        a, b = C.c_long(0), C.c_long(0)
        self.L.ref_graph_kf_markers(self.h, self.kf[id(kf)], C.byref(a), C.byref(b))
        return a.value, b.value

    def mp_pos(self, mp):
        p = np.zeros(3, np.float32); n, u = C.c_int(0), C.c_int(0)
        self.L.ref_graph_mp_pos(self.h, self.mp[id(mp)], _p(p, C.c_float), C.byref(n), C.byref(u))
        return p, n.value, u.value

    def mo_state(self, mo):
        pose, scale = np.zeros(7), np.zeros(3)
        v = [C.c_int(0) for _ in range(5)]
        self.L.ref_graph_mo_state(self.h, self.mo[id(mo)], _p(pose, C.c_double), _p(scale, C.c_double), *[C.byref(x) for x in v])
        return {"pose": pose, "scale": scale, "writes": v[0].value, "been_optimized": bool(v[1].value), "point_threshold": v[2].value, "n_used": v[3].value, "n_filtered": v[4].value}

    def erased(self):
        buf = np.zeros(2 * 65536, np.int64)
        n = self.L.ref_graph_erased(self.h, _p(buf, C.c_long), 65536)
        return [tuple(x) for x in buf[:2 * n].reshape(-1, 2).tolist()]
