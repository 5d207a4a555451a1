"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by cube_slam_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_box_edge_sum_dists.restype = C.c_double
        _LIB.orc_box_edge_angle_error.restype = C.c_double
    return _LIB


class CuboidOpts(C.Structure):
    _fields_ = [
        ("consider_config_1", C.c_int), ("consider_config_2", C.c_int),
        ("whether_sample_cam_roll_pitch", C.c_int), ("whether_sample_bbox_height", C.c_int),
        ("max_cuboid_num", C.c_int),
        ("nominal_skew_ratio", C.c_double), ("max_cut_skew", C.c_double),
        ("yaw_range_deg", C.c_double), ("yaw_step_deg", C.c_double),
        ("canny_low", C.c_int), ("canny_high", C.c_int), ("stateful_cam_pose", C.c_int),
    ]


class Cuboid(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3), ("scale", C.c_double * 3), ("rotY", C.c_double),
        ("box_config_type", C.c_double * 2), ("box_corners_2d", C.c_int32 * 16),
        ("box_corners_3d_world", C.c_double * 24), ("rect_detect_2d", C.c_double * 4),
        ("edge_distance_error", C.c_double), ("edge_angle_error", C.c_double),
        ("normalized_error", C.c_double), ("skew_ratio", C.c_double),
        ("down_expand_height", C.c_double), ("camera_roll_delta", C.c_double),
        ("camera_pitch_delta", C.c_double),
    ]


CUBOID_DTYPE = np.dtype([
    ("pos", "f8", 3), ("scale", "f8", 3), ("rotY", "f8"), ("box_config_type", "f8", 2),
    ("box_corners_2d", "i4", (2, 8)), ("box_corners_3d_world", "f8", (3, 8)), ("rect_detect_2d", "f8", 4),
    ("edge_distance_error", "f8"), ("edge_angle_error", "f8"), ("normalized_error", "f8"),
    ("skew_ratio", "f8"), ("down_expand_height", "f8"), ("camera_roll_delta", "f8"),
    ("camera_pitch_delta", "f8")], align=True)
assert CUBOID_DTYPE.itemsize == C.sizeof(Cuboid)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def cuboid_opts(**kw):
    o = CuboidOpts()
    lib().orc_cuboid_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    g = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(bgr, C.c_uint8), w, h, _p(g, C.c_uint8))
    return g


def canny_roi(gray, x0, y0, w, h, low=80, high=200):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    e = np.empty((h, w), np.uint8)
    lib().orc_canny_roi(_p(gray, C.c_uint8), W, H, x0, y0, w, h, low, high, _p(e, C.c_uint8))
    return e


def dist_transform(src):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    d = np.empty((h, w), np.float32)
    lib().orc_dist_transform_3x3(_p(src, C.c_uint8), w, h, _p(d, C.c_float))
    return d


def canny_dt_roi(gray, x0, y0, w, h, low=80, high=200):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    d = np.empty((h, w), np.float32)
    lib().orc_canny_dt_roi(_p(gray, C.c_uint8), W, H, x0, y0, w, h, low, high, _p(d, C.c_float))
    return d


def merge_break_lines(lines, dist_thre=20.0, angle_thre_deg=5.0, len_thre=30.0):
    lines = np.ascontiguousarray(lines, np.float64).reshape(-1, 4)
    out = np.empty_like(lines)
    n = lib().orc_merge_break_lines(_p(lines, C.c_double), len(lines), C.c_double(dist_thre),
                                    C.c_double(angle_thre_deg), C.c_double(len_thre), _p(out, C.c_double))
    return out[:n].copy()


def fuse_normalize_scores(dist_err, angle_err, weight_vp_angle=0.8, normalize=True):
    d = np.ascontiguousarray(dist_err, np.float64)
    a = np.ascontiguousarray(angle_err, np.float64)
    keep = np.empty(len(d), np.int32)
    sc = np.empty(len(d), np.float64)
    n = lib().orc_fuse_normalize_scores(_p(d, C.c_double), _p(a, C.c_double), len(d), C.c_double(weight_vp_angle),
                                        int(normalize), _p(keep, C.c_int), _p(sc, C.c_double))
    return keep[:n].copy(), sc[:n].copy()


def detect_cuboid(gray, K, Twc, boxes, lines, opts=None, debug=False, rows_cap=400000):
    """Returns (list per box of structured arrays, debug dict or None)."""
    opts = opts or cuboid_opts()
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    Twc = np.ascontiguousarray(Twc, np.float64).reshape(4, 4)
    boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 5)
    lines = np.ascontiguousarray(lines, np.float64).reshape(-1, 4)
    nb = len(boxes)
    out = np.zeros((nb, max(1, opts.max_cuboid_num)), CUBOID_DTYPE)
    counts = np.zeros(nb, np.int32)
    rows = np.zeros((rows_cap, 25), np.float64) if debug else None
    rc = np.zeros(nb * 3 + 1, np.int32)
    r = lib().orc_detect_cuboid(_p(gray, C.c_uint8), W, H, _p(K, C.c_double), _p(Twc, C.c_double),
                                _p(boxes, C.c_double), nb, _p(lines, C.c_double), len(lines), C.byref(opts),
                                out.ctypes.data_as(C.c_void_p), _p(counts, C.c_int),
                                _p(rows, C.c_double) if debug else None, rows_cap, _p(rc, C.c_int))
    if r != 0:
        raise RuntimeError("orc_detect_cuboid failed: %d" % r)
    res = [out[i, :counts[i]].copy() for i in range(nb)]
    dbg = None
    if debug:
        dbg = {"row_count": rc, "rows": rows[: int(rc.sum())].copy()}
    return res, dbg


# ----------------------------------------------------------------------------------------------- ORB extractor
KEYPOINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                           ("octave", "i4"), ("class_id", "i4")])
assert KEYPOINT_DTYPE.itemsize == 28


class ORBextractor:
    """Oracle ORB_SLAM2::ORBextractor (orb_object_slam/include/ORBextractor.h:51-61)."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        l = lib()
        l.orc_orb_create.restype = C.c_void_p
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self._e = C.c_void_p(l.orc_orb_create(nfeatures, C.c_float(scaleFactor), nlevels, iniThFAST, minThFAST))

    def __call__(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        cap = self.nfeatures * 2 + 64
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_orb_extract(self._e, _p(gray, C.c_uint8), W, H, kps.ctypes.data_as(C.c_void_p), _p(desc, C.c_uint8), cap)
        return kps[:n].copy(), desc[:n].copy()

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        lib().orc_orb_features_per_level(self._e, _p(out, C.c_int))
        return out

    def level(self, l, blurred=False):
        w, h = C.c_int(), C.c_int()
        lib().orc_orb_level_dims(self._e, l, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        r = lib().orc_orb_get_level(self._e, l, int(blurred), _p(out, C.c_uint8))
        return out if r == 0 else None

    def candidates(self, l, cap=200000):
        out = np.zeros((cap, 3), np.float32)
        n = lib().orc_orb_get_candidates(self._e, l, _p(out, C.c_float), cap)
        return out[:n].copy()

    def __del__(self):
        try:
            lib().orc_orb_destroy(self._e)
        except Exception:
            pass


def fast(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((w * h, 3), np.float32)
    n = lib().orc_fast(_p(img, C.c_uint8), w, w, h, threshold, _p(out, C.c_float), w * h)
    return out[:n].copy()


def fast_atan2(y, x):
    lib().orc_fast_atan2.restype = C.c_float
    return lib().orc_fast_atan2(C.c_float(y), C.c_float(x))


def sincos_f(a):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_f(C.c_float(a), C.byref(s), C.byref(c))
    return s.value, c.value


# ----------------------------------------------------------------------------------------------- ORB matcher
class Frame(C.Structure):
    _fields_ = [("N", C.c_int), ("keysUn", C.c_void_p), ("desc", C.c_void_p),
                ("minX", C.c_float), ("maxX", C.c_float), ("minY", C.c_float), ("maxY", C.c_float)]


def make_frame(keys, desc, bounds):
    keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    f = Frame(len(keys), keys.ctypes.data, desc.ctypes.data, *[float(b) for b in bounds])
    f._keep = (keys, desc)
    return f


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a, C.c_uint8), _p(b, C.c_uint8))


def get_features_in_area(frame, x, y, r, minLevel=-1, maxLevel=-1):
    out = np.zeros(max(frame.N, 1), np.int32)
    n = lib().orc_get_features_in_area(C.byref(frame), C.c_float(x), C.c_float(y), C.c_float(r), minLevel, maxLevel, _p(out, C.c_int), len(out))
    return out[:n].copy()


def search_by_projection_frame(cur, world_pos, valid, blocks, mp_desc, last_octave, last_angle, Tcw, fx, fy, cx, cy, scale_factors, th, check_ori=True, train_blocked=None):
    wp = np.ascontiguousarray(world_pos, np.float32); va = np.ascontiguousarray(valid, np.uint8); bl = np.ascontiguousarray(blocks, np.uint8)
    md = np.ascontiguousarray(mp_desc, np.uint8); lo = np.ascontiguousarray(last_octave, np.int32); la = np.ascontiguousarray(last_angle, np.float32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy(); sf = np.ascontiguousarray(scale_factors, np.float32)
    tm = np.zeros(max(cur.N, 1), np.int32)
    n = lib().orc_search_by_projection_frame(C.byref(cur), len(va), _p(wp, C.c_float), _p(va, C.c_uint8), _p(bl, C.c_uint8), _p(md, C.c_uint8),
                                             _p(lo, C.c_int), _p(la, C.c_float), _p(T, C.c_float), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                             C.c_float(cy), _p(sf, C.c_float), C.c_float(th), int(check_ori),
                                             None if train_blocked is None else _p(np.ascontiguousarray(train_blocked, np.uint8), C.c_uint8), _p(tm, C.c_int))
    return tm[:cur.N].copy(), n


def search_local_map(F, proj_xy, view_cos, pred_level, in_view, blocks, mp_desc, scale_factors, th, nnratio, train_blocked=None):
    pxy = np.ascontiguousarray(proj_xy, np.float32); vc = np.ascontiguousarray(view_cos, np.float32); pl = np.ascontiguousarray(pred_level, np.int32)
    iv = np.ascontiguousarray(in_view, np.uint8); bl = np.ascontiguousarray(blocks, np.uint8); md = np.ascontiguousarray(mp_desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    tb = None if train_blocked is None else np.ascontiguousarray(train_blocked, np.uint8)
    tm = np.zeros(max(F.N, 1), np.int32)
    n = lib().orc_search_local_map(C.byref(F), len(iv), _p(pxy, C.c_float), _p(vc, C.c_float), _p(pl, C.c_int), _p(iv, C.c_uint8), _p(bl, C.c_uint8),
                                   _p(md, C.c_uint8), _p(sf, C.c_float), C.c_float(th), C.c_float(nnratio), None if tb is None else _p(tb, C.c_uint8),
                                   _p(tm, C.c_int))
    return tm[:F.N].copy(), n


def search_for_initialization(F1, F2, prev_matched, window_size=100, nnratio=0.9, check_ori=True):
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(F1.N, 1), np.int32)
    n = lib().orc_search_for_initialization(C.byref(F1), C.byref(F2), _p(prev, C.c_float), int(window_size), C.c_float(nnratio), int(check_ori), _p(m12, C.c_int))
    return m12[:F1.N].copy(), prev, n


def undistort_points(xy, K4, dist5):
    """cv::undistortPoints(src, dst, K, D, Mat(), K) in its classic five-iteration form (OpenCV 2.4 - 3.2 cvUndistortPoints), as
    Frame::UndistortKeyPoints / ComputeImageBounds use it (orb_object_slam/src/Frame.cc:546-609).  The intrinsics and coefficients are
    floats widened to double, the result is rounded to float.  dist5 = k1 k2 p1 p2 k3; dist5[0] == 0 means `mvKeysUn = mvKeys`.
    THIRD-PARTY algorithm restated from memory (the OpenCV sources are not in the reference tree): parity unpinned for this function."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    if dist5 is None or np.float32(dist5[0]) == 0:
        return xy.copy()
    fx, fy, cx, cy = [np.float64(np.float32(v)) for v in K4]
    k = [np.float64(np.float32(v)) for v in dist5]
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = (xy[:, 0].astype(np.float64) - cx) * ifx
    y = (xy[:, 1].astype(np.float64) - cy) * ify
    x0, y0 = x.copy(), y.copy()
    for _ in range(5):
        r2 = x * x + y * y
        icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (x0 - dX) * icdist
        y = (y0 - dY) * icdist
    xx = fx * x + 0.0 * y + cx
    yy = 0.0 * x + fy * y + cy
    ww = 1.0 / (0.0 * x + 0.0 * y + 1.0)
    return np.stack([(xx * ww).astype(np.float32), (yy * ww).astype(np.float32)], axis=1)


def image_bounds(cols, rows, K4, dist5):
    """Frame::ComputeImageBounds (Frame.cc:578-609) -> (mnMinX, mnMaxX, mnMinY, mnMaxY) as float32."""
    if dist5 is None or np.float32(dist5[0]) == 0:
        return np.array([0, cols, 0, rows], np.float32)
    c = undistort_points([[0, 0], [cols, 0], [0, rows], [cols, rows]], K4, dist5)
    return np.array([min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])], np.float32)


def fuse(F, u_right, inv_level_sigma2, n_mp_uv, ur, pred_level, valid, mp_desc, scale_factors, th, keys_static=None):
    uv = np.ascontiguousarray(n_mp_uv, np.float32); urr = np.ascontiguousarray(ur, np.float32); pl = np.ascontiguousarray(pred_level, np.int32)
    va = np.ascontiguousarray(valid, np.uint8); md = np.ascontiguousarray(mp_desc, np.uint8); sf = np.ascontiguousarray(scale_factors, np.float32)
    kr = np.ascontiguousarray(u_right, np.float32); isg = np.ascontiguousarray(inv_level_sigma2, np.float32)
    ks = None if keys_static is None else np.ascontiguousarray(keys_static, np.uint8)
    bi = np.zeros(max(len(va), 1), np.int32); bd = np.zeros(max(len(va), 1), np.int32)
    n = lib().orc_fuse(C.byref(F), _p(kr, C.c_float), _p(isg, C.c_float), None if ks is None else _p(ks, C.c_uint8), len(va), _p(uv, C.c_float), _p(urr, C.c_float),
                       _p(pl, C.c_int), _p(va, C.c_uint8), _p(md, C.c_uint8), _p(sf, C.c_float), C.c_float(th), _p(bi, C.c_int), _p(bd, C.c_int))
    return bi[:len(va)].copy(), bd[:len(va)].copy(), n


def search_for_triangulation(F1, node1, skip1, ur1, F2, node2, skip2, ur2, F12, ex, ey, scale_factors2, level_sigma2_2, only_stereo=False, check_ori=True):
    n1 = np.ascontiguousarray(node1, np.int32); s1 = np.ascontiguousarray(skip1, np.uint8); u1 = np.ascontiguousarray(ur1, np.float32)
    n2 = np.ascontiguousarray(node2, np.int32); s2 = np.ascontiguousarray(skip2, np.uint8); u2 = np.ascontiguousarray(ur2, np.float32)
    Fm = np.ascontiguousarray(F12, np.float32).reshape(-1); sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
    m12 = np.zeros(max(F1.N, 1), np.int32)
    n = lib().orc_search_for_triangulation(C.byref(F1), _p(n1, C.c_int), _p(s1, C.c_uint8), _p(u1, C.c_float), None, C.byref(F2), _p(n2, C.c_int), _p(s2, C.c_uint8),
                                           _p(u2, C.c_float), None, _p(Fm, C.c_float), C.c_float(ex), C.c_float(ey), _p(sf, C.c_float), _p(sg, C.c_float),
                                           int(only_stereo), int(check_ori), _p(m12, C.c_int))
    return m12[:F1.N].copy(), n


def search_by_bow(KF, nodeKF, skipKF, F, nodeF, skipF, nnratio, check_ori=True):
    nk = np.ascontiguousarray(nodeKF, np.int32); sk = np.ascontiguousarray(skipKF, np.uint8); nf = np.ascontiguousarray(nodeF, np.int32)
    sf = None if skipF is None else np.ascontiguousarray(skipF, np.uint8)
    mf = np.zeros(max(F.N, 1), np.int32)
    n = lib().orc_search_by_bow(C.byref(KF), _p(nk, C.c_int), _p(sk, C.c_uint8), C.byref(F), _p(nf, C.c_int), None if sf is None else _p(sf, C.c_uint8), C.c_float(nnratio),
                                int(check_ori), _p(mf, C.c_int))
    return mf[:F.N].copy(), n


def search_by_bow_kf(K1, node1, skip1, K2, node2, skip2, nnratio, check_ori=True):
    n1 = np.ascontiguousarray(node1, np.int32); s1 = np.ascontiguousarray(skip1, np.uint8)
    n2 = np.ascontiguousarray(node2, np.int32); s2 = np.ascontiguousarray(skip2, np.uint8)
    m12 = np.zeros(max(K1.N, 1), np.int32)
    n = lib().orc_search_by_bow_kf(C.byref(K1), _p(n1, C.c_int), _p(s1, C.c_uint8), C.byref(K2), _p(n2, C.c_int), _p(s2, C.c_uint8), C.c_float(nnratio),
                                   int(check_ori), _p(m12, C.c_int))
    return m12[:K1.N].copy(), n


def hamming_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    bi = np.zeros(len(q), np.int32); bd = np.zeros(len(q), np.int32); sd = np.zeros(len(q), np.int32)
    lib().orc_hamming_knn2(_p(q, C.c_uint8), len(q), _p(t, C.c_uint8), len(t), _p(bi, C.c_int), _p(bd, C.c_int), _p(sd, C.c_int))
    return bi, bd, sd


# ----------------------------------------------------------------------------------------------- object BA
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


def ba_struct(d, cls=BAProblem):
    """Builds the C problem struct from the dict made by cube_slam_amd.synth.ba_problem (keeps the arrays alive)."""
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(d[name], dt)
        keep[name] = a
        return a.ctypes.data

    p = cls()
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


def ba_optimize(d, iterations):
    p = ba_struct(d)
    cam = np.zeros((p.n_cams, 7)); pts = np.zeros((p.n_points, 3)); cub = np.zeros((max(p.n_cuboids, 1), 7))
    st = BAStats()
    lib().orc_ba_optimize(C.byref(p), iterations, _p(cam, C.c_double), _p(pts, C.c_double), _p(cub, C.c_double), C.byref(st))
    return cam, pts, cub[:p.n_cuboids], {"iterations": st.iterations, "lm_trials": st.lm_trials, "chi2_init": st.chi2_init, "chi2_final": st.chi2_final,
                                          "lambda_final": st.lambda_final, "chi2_trace": list(st.chi2_trace)[:st.iterations]}


def ba_errors(d):
    p = ba_struct(d)
    eo = np.zeros((p.n_obs, 3)); ec = np.zeros((max(p.n_cobs, 1), 4)); ep = np.zeros((max(p.n_pc, 1), 3))
    lib().orc_ba_errors.restype = C.c_double
    chi = lib().orc_ba_errors(C.byref(p), _p(eo, C.c_double), _p(ec, C.c_double), _p(ep, C.c_double))
    return chi, eo, ec[:p.n_cobs], ep[:p.n_pc]


def ba_reduced_dense(d, lm_begin, lm_end, with_pose_edges, lam):
    p = ba_struct(d)
    P = int((1 - np.asarray(d["cam_fixed"])).sum()) + p.n_cuboids
    Hm = np.zeros((6 * P, 6 * P)); b = np.zeros(6 * P)
    lib().orc_ba_reduced_dense(C.byref(p), lm_begin, lm_end, int(with_pose_edges), C.c_double(lam), _p(Hm, C.c_double), _p(b, C.c_double))
    return Hm, b


# ----------------------------------------------------------------------------------------------- dynamic-object BA
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


def badyn_struct(d, cls=BADynProblem):
    """C struct from the dict made by cube_slam_amd.synth.ba_dyn_problem (keeps the arrays alive)."""
    keep = {}

    def arr(name, dt, n_min=1):
        a = np.ascontiguousarray(d[name], dt)
        if a.size == 0:
            a = np.zeros(n_min, dt)
        keep[name] = a
        return a.ctypes.data

    p = cls()
    p.n_cams = len(d["cam_pose"]); p.cam_pose = arr("cam_pose", np.float64); p.cam_fixed = arr("cam_fixed", np.uint8)
    p.n_objs = len(d["obj_pose"]); p.obj_pose = arr("obj_pose", np.float64); p.obj_scale = arr("obj_scale", np.float64); p.obj_flags = arr("obj_flags", np.uint8)
    p.n_vels = len(d["vel"]); p.vel = arr("vel", np.float64)
    p.n_points = len(d["points"]); p.points = arr("points", np.float64)
    p.n_dpoints = len(d["dpoints"]); p.dpoints = arr("dpoints", np.float64)
    p.fix_points = int(d["fix_points"])
    p.n_obs = len(d["obs_cam"]); p.obs_cam = arr("obs_cam", np.int32); p.obs_point = arr("obs_point", np.int32); p.obs_uv = arr("obs_uv", np.float64)
    if d.get("obs_ur") is not None:
        p.obs_ur = arr("obs_ur", np.float64)
    p.obs_inv_sigma2 = arr("obs_inv_sigma2", np.float64)
    if d.get("obs_level") is not None:
        p.obs_level = arr("obs_level", np.uint8)
    p.fx, p.fy, p.cx, p.cy, p.bf, p.huber_mono, p.huber_stereo = d["fx"], d["fy"], d["cx"], d["cy"], d["bf"], d["huber_mono"], d["huber_stereo"]
    p.ulp_info, p.ulp_ratio = d["ulp_info"], d["ulp_ratio"]
    for i in range(3):
        p.ulp_scale[i] = d["ulp_scale"][i]; p.mot_info[i] = d["mot_info"][i]
    p.n_dobs = len(d["dobs_cam"]); p.dobs_cam = arr("dobs_cam", np.int32); p.dobs_obj = arr("dobs_obj", np.int32); p.dobs_point = arr("dobs_point", np.int32)
    p.dobs_uv = arr("dobs_uv", np.float64); p.dobs_inv_sigma2 = arr("dobs_inv_sigma2", np.float64)
    if d.get("dobs_level") is not None:
        p.dobs_level = arr("dobs_level", np.uint8)
    for i, v in enumerate(np.asarray(d["K"], np.float64).reshape(-1)):
        p.K[i] = v
    p.huber_dyn = d["huber_dyn"]
    p.n_mot = len(d["mot_from"]); p.mot_from = arr("mot_from", np.int32); p.mot_to = arr("mot_to", np.int32); p.mot_vel = arr("mot_vel", np.int32); p.mot_dt = arr("mot_dt", np.float64)
    p.n_cobs = len(d["cobs_cam"]); p.cobs_cam = arr("cobs_cam", np.int32); p.cobs_obj = arr("cobs_obj", np.int32); p.cobs_bbox = arr("cobs_bbox", np.float64)
    p.cobs_info = arr("cobs_info", np.float64)
    if d.get("cobs_level") is not None:
        p.cobs_level = arr("cobs_level", np.uint8)
    p.huber_obj = d["huber_obj"]
    p.n_pc = len(d["pc_obj"]); p.pc_obj = arr("pc_obj", np.int32); p.pc_offsets = arr("pc_offsets", np.int32, 2); p.pc_points = arr("pc_points", np.float64)
    p.pc_ratio = d["pc_ratio"]
    p._keep = keep
    return p


def badyn_errors(d):
    p = badyn_struct(d)
    e = [np.zeros((max(n, 1), k)) for n, k in ((p.n_obs, 3), (p.n_dobs, 2), (p.n_mot, 3), (p.n_cobs, 4), (p.n_pc, 3), (p.n_dpoints, 3))]
    lib().orc_badyn_errors.restype = C.c_double
    chi = lib().orc_badyn_errors(C.byref(p), *[_p(a, C.c_double) for a in e])
    ns = (p.n_obs, p.n_dobs, p.n_mot, p.n_cobs, p.n_pc, p.n_dpoints)
    return chi, dict(zip(("obs", "dobs", "mot", "cobs", "pc", "ulp"), [a[:n] for a, n in zip(e, ns)]))


def badyn_reduced_dense(d, lam):
    p = badyn_struct(d)
    n = lib().orc_badyn_reduced_dense(C.byref(p), C.c_double(lam), None, None)
    Hm = np.zeros((max(n, 1), max(n, 1))); b = np.zeros(max(n, 1))
    lib().orc_badyn_reduced_dense(C.byref(p), C.c_double(lam), _p(Hm, C.c_double), _p(b, C.c_double))
    return Hm[:n, :n], b[:n]


def badyn_step(d, lam):
    p = badyn_struct(d)
    out = [np.zeros((max(n, 1), k)) for n, k in ((p.n_cams, 7), (p.n_objs, 7), (p.n_vels, 2), (p.n_points, 3), (p.n_dpoints, 3))]
    rc = lib().orc_badyn_step(C.byref(p), C.c_double(lam), *[_p(a, C.c_double) for a in out])
    ns = (p.n_cams, p.n_objs, p.n_vels, p.n_points, p.n_dpoints)
    return dict(zip(("cam_pose", "obj_pose", "vel", "points", "dpoints"), [a[:n] for a, n in zip(out, ns)])), rc


def badyn_optimize(d, iterations):
    p = badyn_struct(d)
    out = [np.zeros((max(n, 1), k)) for n, k in ((p.n_cams, 7), (p.n_objs, 7), (p.n_vels, 2), (p.n_points, 3), (p.n_dpoints, 3))]
    st = BAStats()
    lib().orc_badyn_optimize(C.byref(p), iterations, *[_p(a, C.c_double) for a in out], C.byref(st))
    ns = (p.n_cams, p.n_objs, p.n_vels, p.n_points, p.n_dpoints)
    res = dict(zip(("cam_pose", "obj_pose", "vel", "points", "dpoints"), [a[:n] for a, n in zip(out, ns)]))
    return res, {"iterations": st.iterations, "lm_trials": st.lm_trials, "chi2_init": st.chi2_init, "chi2_final": st.chi2_final, "lambda_final": st.lambda_final,
                 "chi2_trace": list(st.chi2_trace)[:st.iterations]}


# ----------------------------------------------------------------------------------------------- LSD
KEYLINE_DTYPE = np.dtype([("angle", "f4"), ("class_id", "i4"), ("octave", "i4"), ("pt", "f4", 2), ("response", "f4"), ("size", "f4"),
                          ("startPointX", "f4"), ("startPointY", "f4"), ("endPointX", "f4"), ("endPointY", "f4"),
                          ("sPointInOctaveX", "f4"), ("sPointInOctaveY", "f4"), ("ePointInOctaveX", "f4"), ("ePointInOctaveY", "f4"),
                          ("lineLength", "f4"), ("numOfPixels", "i4")])
assert KEYLINE_DTYPE.itemsize == 68


def lsd_detect(gray, cap=20000):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    out = np.zeros(cap, KEYLINE_DTYPE)
    n = lib().orc_lsd_detect(_p(gray, C.c_uint8), W, H, out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def lsd_detect_filter_lines(gray, length_thres=15.0, cap=20000):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    out = np.zeros((cap, 4), np.float32)
    n = lib().orc_lsd_detect_filter_lines(_p(gray, C.c_uint8), W, H, C.c_float(length_thres), _p(out, C.c_float), cap)
    return out[:n].copy()


def lsd_maps(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    sw, sh, no = C.c_int(), C.c_int(), C.c_int()
    lib().orc_lsd_maps(_p(gray, C.c_uint8), W, H, C.byref(sw), C.byref(sh), None, None, None, None, None)
    n = sw.value * sh.value
    sc = np.zeros(n); mg = np.zeros(n); an = np.zeros(n); order = np.zeros(n, np.int32)
    lib().orc_lsd_maps(_p(gray, C.c_uint8), W, H, C.byref(sw), C.byref(sh), _p(sc, C.c_double), _p(mg, C.c_double), _p(an, C.c_double), _p(order, C.c_int), C.byref(no))
    shape = (sh.value, sw.value)
    return sc.reshape(shape), mg.reshape(shape), an.reshape(shape), order[:no.value].copy()


def lbd_compute(gray, keylines, want_float=False):
    """BinaryDescriptor::compute for one octave -> (n, 32) u8 (and (n, 72) f32)."""
    gray = np.ascontiguousarray(gray, np.uint8); H, W = gray.shape
    kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE); n = len(kl)
    desc = np.zeros((n, 32), np.uint8); fd = np.zeros((n, 72), np.float32)
    lib().orc_lbd_compute(_p(gray, C.c_uint8), W, H, kl.ctypes.data_as(C.c_void_p), n, _p(desc, C.c_uint8), _p(fd, C.c_float))
    return (desc, fd) if want_float else desc


def lbd_maps(gray):
    gray = np.ascontiguousarray(gray, np.uint8); H, W = gray.shape
    b = np.zeros((H, W), np.uint8); dx = np.zeros((H, W), np.int16); dy = np.zeros((H, W), np.int16)
    lib().orc_lbd_maps(_p(gray, C.c_uint8), W, H, _p(b, C.c_uint8), _p(dx, C.c_int16), _p(dy, C.c_int16))
    return b, dx, dy


def pose_optimization(Xw, obs, inv_sigma2, intr, pose):
    """Optimizer::PoseOptimization for one frame -> (pose_out[7], outlier[n] u8, n_inliers)."""
    Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3); obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 3)
    w = np.ascontiguousarray(inv_sigma2, np.float64); pose = np.ascontiguousarray(pose, np.float64)
    n = len(Xw)
    out = np.zeros(7); flags = np.zeros(max(n, 1), np.uint8)
    fx, fy, cx, cy, bf = [float(v) for v in intr]
    lib().orc_pose_optimization.restype = C.c_int
    r = lib().orc_pose_optimization(n, _p(Xw, C.c_double), _p(obs, C.c_double), _p(w, C.c_double), C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), C.c_double(bf),
                                    _p(pose, C.c_double), _p(out, C.c_double), _p(flags, C.c_uint8))
    return out, flags[:n], r


def cuboid9_oplus(cub, upd):
    cub = np.ascontiguousarray(cub, np.float64).reshape(-1, 10); upd = np.ascontiguousarray(upd, np.float64).reshape(-1, 9)
    out = np.zeros_like(cub)
    lib().orc_cuboid9_oplus(len(cub), _p(cub, C.c_double), _p(upd, C.c_double), _p(out, C.c_double))
    return out


def cuboid9_edge_linearize(cam_Tcw, cub_global, cub_meas, jac=True):
    T = np.ascontiguousarray(cam_Tcw, np.float64).reshape(-1, 7); g = np.ascontiguousarray(cub_global, np.float64).reshape(-1, 10)
    m = np.ascontiguousarray(cub_meas, np.float64).reshape(-1, 10)
    n = len(T)
    err = np.zeros((n, 9)); Jc = np.zeros((n, 9, 6)); Jq = np.zeros((n, 9, 9))
    if jac:
        lib().orc_cuboid9_edge_linearize(n, _p(T, C.c_double), _p(g, C.c_double), _p(m, C.c_double), _p(err, C.c_double), _p(Jc, C.c_double), _p(Jq, C.c_double))
        return err, Jc, Jq
    lib().orc_cuboid9_edge_error(n, _p(T, C.c_double), _p(g, C.c_double), _p(m, C.c_double), _p(err, C.c_double))
    return err


# ----------------------------------------------------------------------------------------------- object association
def bbox_overlap_ratio(r1, r2):
    """bboxOverlapratio (detect_3d_cuboid/src/object_3d_util.cpp:650-654): int areas, float ratio; cv::Rect & cv::Rect is empty when the
    intersection has no positive width and height."""
    x1, y1 = max(r1[0], r2[0]), max(r1[1], r2[1])
    w, h = min(r1[0] + r1[2], r2[0] + r2[2]) - x1, min(r1[1] + r1[3], r2[1] + r2[3]) - y1
    ov = w * h if (w > 0 and h > 0) else 0
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.float32(ov) / np.float32(r1[2] * r1[3] + r2[2] * r2[3] - ov)


def associate_keypoints(kp_xy, boxes, enable_ground_height_scale=False):
    """Keypoint -> local cuboid association of Tracking::DetectCuboid (orb_object_slam/src/Tracking.cc:1717-1775) for one keyframe.
    Point2f -> Point2i inside Rect::contains is cv::saturate_cast<int>(float) = cvRound (round half to even).  Pure Python: small cases."""
    boxes = [tuple(int(v) for v in b) for b in boxes]
    nb = len(boxes)
    ov = [False] * nb
    for i in range(nb):
        if not ov[i]:
            for j in range(i + 1, nb):
                if not ov[j] and float(bbox_overlap_ratio(boxes[i], boxes[j])) > 0.15:
                    ov[i] = True; ov[j] = True
    assoc = np.full(len(kp_xy), -1, np.int32)
    inany = np.zeros(len(kp_xy), np.uint8)
    for k, (x, y) in enumerate(np.asarray(kp_xy, np.float32).reshape(-1, 2)):
        px, py = int(np.rint(np.float32(x))), int(np.rint(np.float32(y)))
        times = 0
        for j, (bx, by, bw, bh) in enumerate(boxes):
            inside = bx <= px < bx + bw and by <= py < by + bh
            if enable_ground_height_scale:
                if inside:
                    inany[k] = 1
                    if not ov[j]:
                        times += 1
                        assoc[k] = j if times == 1 else -1
            elif (not ov[j]) and inside:
                times += 1
                assoc[k] = j if times == 1 else -1
    return assoc, inany, np.array(ov, np.uint8)


def associate_cuboids(cand_id, cand_pts, landmark_id, landmark_bad, point_votes, thres, best_object=None, max_vote=None):
    """Tracking::AssociateCuboids (Tracking.cc:1848-1990, use_truth_trackid off) on ids.  PINNED to the reference's own text (that function with
    MapObject::SetAsLandmark / MergeIntoLandmark and MapPoint::AddObjectObservation, tests/test_ref_graph_pins.py).  cand_pts: list of point-id lists
    (GetPotentialMapPoints); point_votes: list of dicts object id -> count (MapPoint::MapObjObservations), updated in place like
    SetAsLandmark / MergeIntoLandmark -> MapPoint::AddObjectObservation do (MapObject.cc:100-115, MapPoint.cc:219-242)."""
    L = list(landmark_id)
    bad = {o: bool(b) for o, b in zip(landmark_id, landmark_bad)}
    assoc, created = [], []
    last_new = None

    def add(p, obj):
        point_votes[p][obj] = point_votes[p].get(obj, 0) + 1
        if best_object is not None and point_votes[p][obj] > max_vote[p]:
            best_object[p] = obj; max_vote[p] = point_votes[p][obj]

    for cid, pts in zip(cand_id, cand_pts):
        if last_new is not None:
            L.append(last_new); bad[last_new] = False
        last_new = None
        best = None
        if L:
            counter = {}
            for p in pts:
                for o in point_votes[p]:
                    counter[o] = counter.get(o, 0) + 1
            largest = thres
            for o in L:
                if not bad[o] and o in counter and counter[o] > largest:
                    largest = counter[o]; best = o
        if best is None:
            for p in pts:
                add(p, cid)
            assoc.append(cid); created.append(1); last_new = cid
        else:
            for p in pts:
                add(p, best)
            assoc.append(best); created.append(0)
    return np.array(assoc, np.int32), np.array(created, np.uint8)
