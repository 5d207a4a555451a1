"""Python host-side mirror of detect_3d_cuboid (reference detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:53-79)
over the C-ABI.  Same member names and argument meaning as the reference class; the work runs in HIP kernels."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib



def _boxes5(x):
    """obj_bbox_coors as the reference reads it: a matrix with one row per box of which columns 0-3 (x y w h) are used (box_proposal_detail.cpp:102-108;
    a fifth column is the detection probability of the txt files).  -> (n, 5) float64."""
    b = np.asarray(x, np.float64)
    if b.ndim == 1:
        if b.size % 5 != 0:
            raise ValueError("obj_bbox_coors: a flat array must hold 5 numbers per box (x y w h prob), got %d numbers" % b.size)
        b = b.reshape(-1, 5)
    if b.ndim != 2 or (len(b) and b.shape[1] < 4):
        raise ValueError("obj_bbox_coors: expected an (n, >= 4) matrix, got shape %r" % (b.shape,))
    out = np.zeros((len(b), 5))
    out[:, :min(5, b.shape[1])] = b[:, :5]
    return np.ascontiguousarray(out)


class CuboidOpts(C.Structure):
    _fields_ = [
        ("consider_config_1", C.c_int), ("consider_config_2", C.c_int),
        ("whether_sample_cam_roll_pitch", C.c_int), ("whether_sample_bbox_height", C.c_int),
        ("max_cuboid_num", C.c_int),
        ("nominal_skew_ratio", C.c_double), ("max_cut_skew", C.c_double),
        ("yaw_range_deg", C.c_double), ("yaw_step_deg", C.c_double),
        ("canny_low", C.c_int), ("canny_high", C.c_int),
    ]


CUBOID_DTYPE = np.dtype([
    ("pos", "f8", 3), ("scale", "f8", 3), ("rotY", "f8"), ("box_config_type", "f8", 2),
    ("box_corners_2d", "i4", (2, 8)), ("box_corners_3d_world", "f8", (3, 8)), ("rect_detect_2d", "f8", 4),
    ("edge_distance_error", "f8"), ("edge_angle_error", "f8"), ("normalized_error", "f8"),
    ("skew_ratio", "f8"), ("down_expand_height", "f8"), ("camera_roll_delta", "f8"),
    ("camera_pitch_delta", "f8")], align=True)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class detect_3d_cuboid:
    """Drop-in for the reference class: set_calibration(), then detect_cuboid(img, transToWolrd, obj_bbox_coors, edges)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or _lib.Context(device)
        self.consider_config_1 = True
        self.consider_config_2 = True
        self.whether_sample_cam_roll_pitch = False
        self.whether_sample_bbox_height = False
        self.max_cuboid_num = 1
        self.nominal_skew_ratio = 1.0
        self.max_cut_skew = 3.0
        self.yaw_range_deg = 45.0
        self.yaw_step_deg = 6.0
        self.Kalib = None

    def set_calibration(self, Kalib):
        self.Kalib = np.ascontiguousarray(Kalib, np.float64).reshape(3, 3)

    def opts(self):
        o = CuboidOpts()
        lib().cs_cuboid_default_opts(C.byref(o))
        o.consider_config_1 = int(self.consider_config_1)
        o.consider_config_2 = int(self.consider_config_2)
        o.whether_sample_cam_roll_pitch = int(self.whether_sample_cam_roll_pitch)
        o.whether_sample_bbox_height = int(self.whether_sample_bbox_height)
        o.max_cuboid_num = int(self.max_cuboid_num)
        o.nominal_skew_ratio = float(self.nominal_skew_ratio)
        o.max_cut_skew = float(self.max_cut_skew)
        o.yaw_range_deg = float(self.yaw_range_deg)
        o.yaw_step_deg = float(self.yaw_step_deg)
        return o

    def detect_cuboid(self, rgb_img, transToWolrd, obj_bbox_coors, edges):
        """Returns all_object_cuboids: one structured array (<= max_cuboid_num records, best first) per 2-D box."""
        if self.Kalib is None:
            raise ValueError("set_calibration() first")
        img = np.ascontiguousarray(rgb_img, np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        H, W = img.shape[:2]
        T = np.ascontiguousarray(transToWolrd, np.float64).reshape(4, 4)
        boxes = _boxes5(obj_bbox_coors)
        lines = np.ascontiguousarray(edges, np.float64).reshape(-1, 4)
        nb = len(boxes)
        out = np.zeros((max(nb, 1), self.max_cuboid_num), CUBOID_DTYPE)
        counts = np.zeros(max(nb, 1), np.int32)
        o = self.opts()
        r = lib().cs_cuboid_detect(self.ctx.ptr, _p(img, C.c_uint8), W, H, ch, W * ch, _p(self.Kalib, C.c_double), _p(T, C.c_double),
                                   _p(boxes, C.c_double), nb, _p(lines, C.c_double), len(lines), C.byref(o),
                                   out.ctypes.data_as(C.c_void_p), _p(counts, C.c_int))
        check(self.ctx.ptr, r, "cs_cuboid_detect")
        return [out[i, :counts[i]].copy() for i in range(nb)]


class CuboidBatch:
    """Device-resident batch (BASELINE config 4): frames stay in HBM, run() only launches kernels."""

    def __init__(self, ctx, grays, K, Twcs, boxes_list, lines_list, opts):
        self.ctx = ctx
        grays = np.ascontiguousarray(grays, np.uint8)
        self.F, self.H, self.W = grays.shape
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        T = np.ascontiguousarray(Twcs, np.float64).reshape(self.F, 16)
        bo = np.zeros(self.F + 1, np.int32)
        lo = np.zeros(self.F + 1, np.int32)
        for f in range(self.F):
            bo[f + 1] = bo[f] + len(boxes_list[f])
            lo[f + 1] = lo[f] + len(lines_list[f])
        boxes = np.ascontiguousarray(np.concatenate([_boxes5(b) for b in boxes_list] + [np.zeros((0, 5))]))
        lines = np.ascontiguousarray(np.concatenate([np.asarray(l, np.float64).reshape(-1, 4) for l in lines_list] + [np.zeros((1, 4))]))
        self.n_boxes = int(bo[-1])
        self.max_cuboid_num = opts.max_cuboid_num
        self._b = C.c_void_p()
        r = lib().cs_cuboid_batch_create(ctx.ptr, self.F, self.W, self.H, _p(grays, C.c_uint8), _p(K, C.c_double), _p(T, C.c_double),
                                         _p(bo, C.c_int), _p(boxes, C.c_double), _p(lo, C.c_int), _p(lines, C.c_double),
                                         C.byref(opts), C.byref(self._b))
        check(ctx.ptr, r, "cs_cuboid_batch_create")

    def set_lines(self, lines_list):
        """New edge lists (one (m, 4) array per frame) for the resident frames: the hand-over of detect_filter_lines -> detect_cuboid."""
        lo = np.zeros(self.F + 1, np.int32)
        for f in range(self.F):
            lo[f + 1] = lo[f] + len(lines_list[f])
        lines = np.ascontiguousarray(np.concatenate([np.asarray(l, np.float64).reshape(-1, 4) for l in lines_list] + [np.zeros((1, 4))]))
        check(self.ctx.ptr, lib().cs_cuboid_batch_set_lines(self.ctx.ptr, self._b, _p(lo, C.c_int), _p(lines, C.c_double)), "cs_cuboid_batch_set_lines")

    @staticmethod
    def pack_scene(Twcs, boxes_list, lines_list=None):
        """(Twc (F, 16), box_offsets, boxes (n, 5), line_offsets | None, lines | None) as cs_cuboid_batch_set_scene / cs_frontend_stream_push_scene take them."""
        F = len(boxes_list)
        T = np.ascontiguousarray(Twcs, np.float64).reshape(F, 16)
        bo = np.zeros(F + 1, np.int32)
        for f in range(F):
            bo[f + 1] = bo[f] + len(boxes_list[f])
        boxes = np.ascontiguousarray(np.concatenate([_boxes5(b) for b in boxes_list] + [np.zeros((0, 5))]))
        lo = lines = None
        if lines_list is not None:
            lo = np.zeros(F + 1, np.int32)
            for f in range(F):
                lo[f + 1] = lo[f] + len(lines_list[f])
            lines = np.ascontiguousarray(np.concatenate([np.asarray(l, np.float64).reshape(-1, 4) for l in lines_list] + [np.zeros((1, 4))]))
        return T, bo, boxes, lo, lines

    def set_scene(self, Twcs, boxes_list, lines_list=None, packed=None):
        """Other boxes, poses and (lines_list given) edge lists for the frames of the batch: what every detect_cuboid call brings with its pixels (cs_cuboid_batch_set_scene)."""
        T, bo, boxes, lo, lines = packed if packed is not None else self.pack_scene(Twcs, boxes_list, lines_list)
        assert len(bo) == self.F + 1
        check(self.ctx.ptr, lib().cs_cuboid_batch_set_scene(self.ctx.ptr, self._b, _p(T, C.c_double), _p(bo, C.c_int), _p(boxes, C.c_double), None if lo is None else _p(lo, C.c_int),
                                                            None if lines is None else _p(lines, C.c_double)), "cs_cuboid_batch_set_scene")
        self.n_boxes = int(bo[-1])

    def set_shared_gpu(self, shared):
        """Speed hint: long-running kernels of other streams hold most CUs while this batch runs (cs_cuboid_batch_set_shared_gpu)."""
        check(self.ctx.ptr, lib().cs_cuboid_batch_set_shared_gpu(self._b, 1 if shared else 0), "cs_cuboid_batch_set_shared_gpu")

    def run(self):
        check(self.ctx.ptr, lib().cs_cuboid_batch_run(self.ctx.ptr, self._b), "cs_cuboid_batch_run")

    def read(self):
        out = np.zeros((max(self.n_boxes, 1), self.max_cuboid_num), CUBOID_DTYPE)
        counts = np.zeros(max(self.n_boxes, 1), np.int32)
        check(self.ctx.ptr, lib().cs_cuboid_batch_read(self.ctx.ptr, self._b, out.ctypes.data_as(C.c_void_p), _p(counts, C.c_int)),
              "cs_cuboid_batch_read")
        return [out[i, :counts[i]].copy() for i in range(self.n_boxes)]

    def stats(self):
        v = [C.c_long() for _ in range(4)]
        check(self.ctx.ptr, lib().cs_cuboid_batch_stats(self.ctx.ptr, self._b, *[C.byref(x) for x in v]), "cs_cuboid_batch_stats")
        return dict(zip(("n_units", "roi_pixels", "n_hypotheses", "n_valid"), [x.value for x in v]))

    def score_stats(self):
        v = (C.c_long * 6)()
        check(self.ctx.ptr, lib().cs_cuboid_batch_score_stats(self.ctx.ptr, self._b, v), "cs_cuboid_batch_score_stats")
        return dict(zip(("code_units", "code_pixels", "code_valid", "float_units", "float_pixels", "float_valid"), list(v)))

    def unit(self, u, rows_cap=400000):
        dims = (C.c_int * 12)()
        check(self.ctx.ptr, lib().cs_cuboid_batch_unit(self.ctx.ptr, self._b, u, dims, None, None, None, 0, None, 0), "cs_cuboid_batch_unit")
        x, y, w, h, cap, nvalid, nmerged, nyaw, frame, box, hs, n_hs = list(dims)
        edges = np.zeros((h, w), np.uint8)
        dist = np.zeros((h, w), np.float32)
        rows = np.zeros((max(nvalid, 1), 25), np.float64)
        merged = np.zeros((max(nmerged, 1), 4), np.float64)
        check(self.ctx.ptr, lib().cs_cuboid_batch_unit(self.ctx.ptr, self._b, u, dims, _p(edges, C.c_uint8), _p(dist, C.c_float),
                                                       _p(rows, C.c_double), max(nvalid, 1), _p(merged, C.c_double), max(nmerged, 1)),
              "cs_cuboid_batch_unit")
        return {"roi": (x, y, w, h), "frame": frame, "box": box, "hs": hs, "n_hs": n_hs, "n_valid": nvalid, "n_yaw": nyaw, "edges": edges, "dist": dist, "rows": rows[:nvalid],
                "merged": merged[:nmerged]}

    def close(self):
        if self._b:
            lib().cs_cuboid_batch_destroy(self.ctx.ptr, self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
