"""Python host-side mirror of line_lbd_detect (reference line_lbd/include/line_lbd/line_lbd_allclass.h:22-70), LSD path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

KEYLINE_DTYPE = np.dtype([("angle", "f4"), ("class_id", "i4"), ("octave", "i4"), ("pt", "f4", 2), ("response", "f4"), ("size", "f4"),
                          ("startPointX", "f4"), ("startPointY", "f4"), ("endPointX", "f4"), ("endPointY", "f4"),
                          ("sPointInOctaveX", "f4"), ("sPointInOctaveY", "f4"), ("ePointInOctaveX", "f4"), ("ePointInOctaveY", "f4"),
                          ("lineLength", "f4"), ("numOfPixels", "i4")])


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class line_lbd_detect:
    """use_LSD = True path: detect_raw_lines / detect_filter_lines (numoctaves = 1, octaveratio = 1)."""

    def __init__(self, width, height, max_frames=1, ctx=None, device=0):
        self.ctx = ctx or _lib.Context(device)
        self.use_LSD = True
        self.line_length_thres = 50.0  # line_lbd_allclass.cpp:121
        self.W, self.H, self.max_frames = width, height, max_frames
        self.cap = 8192
        self._l = C.c_void_p()
        check(self.ctx.ptr, lib().cs_lsd_create(self.ctx.ptr, width, height, max_frames, C.byref(self._l)), "cs_lsd_create")

    def _imgs(self, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        if g.ndim == 2:
            g = g[None]
        assert g.shape[1:] == (self.H, self.W) and len(g) <= self.max_frames
        return g

    # ---- resident-batch form (bench.py): upload once, run many times
    def upload(self, gray):
        g = self._imgs(gray)
        self._n = len(g)
        check(self.ctx.ptr, lib().cs_lsd_upload(self.ctx.ptr, self._l, _p(g, C.c_uint8), len(g), self.W), "cs_lsd_upload")

    def run(self, with_lbd=True):
        check(self.ctx.ptr, lib().cs_lsd_run(self.ctx.ptr, self._l, int(with_lbd)), "cs_lsd_run")

    def set_region_stage(self, stage):
        """'auto' | 'host' | 'wave_per_frame' | 'backlog' (cs_lsd_set_region_stage): the formulation of region_grow ... rect_improve of the next runs."""
        check(self.ctx.ptr, lib().cs_lsd_set_region_stage(self._l, {"auto": 0, "host": 1, "wave_per_frame": 2, "backlog": 3}[stage]), "cs_lsd_set_region_stage")

    def region_stats(self):
        """Region stage of the last batch (see cs_lsd_region_stats): device = the one-wave-per-frame stage ran, grows = region_grow calls,
        candidates = rectangles at rect_improve, host_fallback, fetches = pixel-window fetches.  All zero for the host stage."""
        st = (C.c_long * 5)()
        check(self.ctx.ptr, lib().cs_lsd_region_stats(self.ctx.ptr, self._l, st), "cs_lsd_region_stats")
        return {"device": bool(st[0]), "grows": st[1], "candidates": st[2], "host_fallback": bool(st[3]), "fetches": st[4]}

    def read(self, frame, with_desc=True):
        n = C.c_int()
        check(self.ctx.ptr, lib().cs_lsd_read(self.ctx.ptr, self._l, frame, None, 0, C.byref(n), None), "cs_lsd_read")
        kl = np.zeros(n.value, KEYLINE_DTYPE); desc = np.zeros((n.value, 32), np.uint8)
        check(self.ctx.ptr, lib().cs_lsd_read(self.ctx.ptr, self._l, frame, kl.ctypes.data_as(C.c_void_p), n.value, C.byref(n),
                                              _p(desc, C.c_uint8) if with_desc else None), "cs_lsd_read")
        return (kl, desc) if with_desc else kl

    def read_filter_lines(self, n_frames, cap=1024):
        """detect_filter_lines' rows (x1 y1 x2 y2, float32) for the resident frames, from the KeyLines of the last run()."""
        out = np.zeros((n_frames, cap, 4), np.float32); counts = np.zeros(n_frames, np.int32)
        check(self.ctx.ptr, lib().cs_lsd_read_filter_lines(self.ctx.ptr, self._l, C.c_float(self.line_length_thres), _p(out, C.c_float), cap, _p(counts, C.c_int), int(n_frames)), "cs_lsd_read_filter_lines")
        return [out[f, :counts[f]] for f in range(n_frames)]

    def detect_raw_lines(self, gray):
        g = self._imgs(gray)
        out = np.zeros((len(g), self.cap), KEYLINE_DTYPE); counts = np.zeros(len(g), np.int32)
        check(self.ctx.ptr, lib().cs_lsd_detect(self.ctx.ptr, self._l, _p(g, C.c_uint8), len(g), self.W, out.ctypes.data_as(C.c_void_p), self.cap, _p(counts, C.c_int)), "cs_lsd_detect")
        res = [out[f, :counts[f]].copy() for f in range(len(g))]
        return res if np.ndim(gray) == 3 else res[0]

    def detect_filter_lines(self, gray):
        g = self._imgs(gray)
        out = np.zeros((len(g), self.cap, 4), np.float32); counts = np.zeros(len(g), np.int32)
        check(self.ctx.ptr, lib().cs_lsd_detect_filter_lines(self.ctx.ptr, self._l, _p(g, C.c_uint8), len(g), self.W, C.c_float(self.line_length_thres), _p(out, C.c_float),
                                                             self.cap, _p(counts, C.c_int)), "cs_lsd_detect_filter_lines")
        res = [out[f, :counts[f]].copy() for f in range(len(g))]
        return res if np.ndim(gray) == 3 else res[0]

    def maps(self, frame=0):
        sw, sh = C.c_int(), C.c_int()
        check(self.ctx.ptr, lib().cs_lsd_get_maps(self.ctx.ptr, self._l, frame, None, None, None, C.byref(sw), C.byref(sh)), "cs_lsd_get_maps")
        n = sw.value * sh.value
        sc, mg, an = np.zeros(n), np.zeros(n), np.zeros(n)
        check(self.ctx.ptr, lib().cs_lsd_get_maps(self.ctx.ptr, self._l, frame, _p(sc, C.c_double), _p(mg, C.c_double), _p(an, C.c_double), C.byref(sw), C.byref(sh)), "cs_lsd_get_maps")
        shape = (sh.value, sw.value)
        return sc.reshape(shape), mg.reshape(shape), an.reshape(shape)

    # ---- LBD descriptors (use_LSD path of detect_descrip_lines, line_lbd_allclass.cpp:222-269) and their matcher (:339-356)
    def get_line_descriptors(self, gray, keylines, want_float=False):
        g = np.ascontiguousarray(gray, np.uint8)
        kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE); n = len(kl)
        desc = np.zeros((n, 32), np.uint8); fd = np.zeros((n, 72), np.float32) if want_float else None
        check(self.ctx.ptr, lib().cs_lbd_compute(self.ctx.ptr, _p(g, C.c_uint8), g.shape[1], g.shape[0], g.shape[1], kl.ctypes.data_as(C.c_void_p), n,
                                                 _p(desc, C.c_uint8), _p(fd, C.c_float) if want_float else None), "cs_lbd_compute")
        return (desc, fd) if want_float else desc

    def detect_descrip_lines(self, gray):
        """-> (keylines, descriptors) of one gray image: descriptors of all raw lines, then the octave-0 lines longer
        than line_length_thres are kept (line_lbd_allclass.cpp:222-269)."""
        kl = self.detect_raw_lines(gray)
        desc = self.get_line_descriptors(gray, kl)
        keep = (kl["octave"] == 0) & (kl["lineLength"] > self.line_length_thres)
        return kl[keep], desc[keep]

    def match_line_descrip(self, desc_q, desc_t, matching_dist_thres=25.0):
        """-> (query_idx, train_idx, distance) of the good matches (BinaryDescriptorMatcher::match + distance threshold)."""
        q = np.ascontiguousarray(desc_q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(desc_t, np.uint8).reshape(-1, 32)
        qi = np.zeros(len(q), np.int32); ti = np.zeros(len(q), np.int32); d = np.zeros(len(q), np.int32); m = C.c_int()
        check(self.ctx.ptr, lib().cs_lbd_match(self.ctx.ptr, _p(q, C.c_uint8), len(q), _p(t, C.c_uint8), len(t), C.c_float(matching_dist_thres),
                                               _p(qi, C.c_int), _p(ti, C.c_int), _p(d, C.c_int), C.byref(m)), "cs_lbd_match")
        return qi[:m.value], ti[:m.value], d[:m.value]

    def lbd_maps(self, gray):
        g = np.ascontiguousarray(gray, np.uint8); H, W = g.shape
        b = np.zeros((H, W), np.uint8); dx = np.zeros((H, W), np.int16); dy = np.zeros((H, W), np.int16)
        check(self.ctx.ptr, lib().cs_lbd_maps(self.ctx.ptr, _p(g, C.c_uint8), W, H, W, _p(b, C.c_uint8), _p(dx, C.c_int16), _p(dy, C.c_int16)), "cs_lbd_maps")
        return b, dx, dy

    def close(self):
        if self._l:
            lib().cs_lsd_destroy(self.ctx.ptr, self._l)
            self._l = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
