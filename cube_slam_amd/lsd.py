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

    def close(self):
        if self._l:
            lib().cs_lsd_destroy(self.ctx.ptr, self._l)
            self._l = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
