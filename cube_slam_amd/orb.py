"""Python host-side mirror of ORB_SLAM2::ORBextractor (reference orb_object_slam/include/ORBextractor.h:46-117) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

KEYPOINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                           ("octave", "i4"), ("class_id", "i4")])


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); __call__(image) -> (keypoints, descriptors).
    The image size is fixed at construction (device buffers are planned once)."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, max_frames=1, ctx=None, device=0):
        self.ctx = ctx or _lib.Context(device)
        self.nfeatures, self.nlevels, self.W, self.H, self.max_frames = nfeatures, nlevels, width, height, max_frames
        self._e = C.c_void_p()
        check(self.ctx.ptr, lib().cs_orb_create(self.ctx.ptr, nfeatures, C.c_float(scaleFactor), nlevels, iniThFAST, minThFAST, width, height,
                                                max_frames, C.byref(self._e)), "cs_orb_create")
        self.cap = nfeatures + 4 * nlevels + 64

    def _table(self, which, dtype):
        out = np.zeros(self.nlevels, dtype)
        check(self.ctx.ptr, lib().cs_orb_get_table(self._e, which, out.ctypes.data_as(C.c_void_p)), "cs_orb_get_table")
        return out

    def GetScaleFactors(self):
        return self._table(0, np.float32)

    def GetInverseScaleFactors(self):
        return self._table(1, np.float32)

    def GetScaleSigmaSquares(self):
        return self._table(2, np.float32)

    def GetInverseScaleSigmaSquares(self):
        return self._table(3, np.float32)

    def features_per_level(self):
        return self._table(4, np.int32)

    def upload(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 2:
            images = images[None]
        assert images.shape[1:] == (self.H, self.W)
        self.n_frames = images.shape[0]
        check(self.ctx.ptr, lib().cs_orb_upload(self.ctx.ptr, self._e, _p(images, C.c_uint8), self.n_frames, self.W), "cs_orb_upload")

    def run(self):
        check(self.ctx.ptr, lib().cs_orb_run(self.ctx.ptr, self._e), "cs_orb_run")

    def read(self):
        F = self.n_frames
        kps = np.zeros((F, self.cap), KEYPOINT_DTYPE)
        desc = np.zeros((F, self.cap, 32), np.uint8)
        counts = np.zeros(F, np.int32)
        check(self.ctx.ptr, lib().cs_orb_read(self.ctx.ptr, self._e, kps.ctypes.data_as(C.c_void_p), _p(desc, C.c_uint8), self.cap, _p(counts, C.c_int)),
              "cs_orb_read")
        return [(kps[f, :counts[f]].copy(), desc[f, :counts[f]].copy()) for f in range(F)]

    def read_packed(self, kps=None, desc=None):
        """(key points of every frame one behind the other, descriptors likewise, first[n_frames + 1]) -- two device-to-host copies for the whole batch (cs_orb_read_packed).
        kps / desc: caller's buffers to fill (e.g. pinned), at least n_frames * cap entries."""
        F = self.n_frames
        first = np.zeros(F + 1, np.int32)
        total = C.c_long()
        if kps is None:
            kps = np.zeros(F * self.cap, KEYPOINT_DTYPE)
        if desc is None:
            desc = np.zeros((F * self.cap, 32), np.uint8)
        check(self.ctx.ptr, lib().cs_orb_read_packed(self.ctx.ptr, self._e, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.POINTER(C.c_uint8)), len(kps), _p(first, C.c_int), C.byref(total)),
              "cs_orb_read_packed")
        return kps[:total.value], desc[:total.value], first

    def extract_batch(self, images):
        self.upload(images)
        self.run()
        return self.read()

    def __call__(self, image, mask=None):
        return self.extract_batch(image)[0]

    def level(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        check(self.ctx.ptr, lib().cs_orb_get_level(self.ctx.ptr, self._e, frame, level, int(blurred), None, C.byref(w), C.byref(h)), "cs_orb_get_level")
        out = np.zeros((h.value, w.value), np.uint8)
        check(self.ctx.ptr, lib().cs_orb_get_level(self.ctx.ptr, self._e, frame, level, int(blurred), _p(out, C.c_uint8), C.byref(w), C.byref(h)), "cs_orb_get_level")
        return out

    def candidates(self, frame, level):
        n = C.c_int()
        check(self.ctx.ptr, lib().cs_orb_get_candidates(self.ctx.ptr, self._e, frame, level, None, 0, C.byref(n)), "cs_orb_get_candidates")
        out = np.zeros((max(n.value, 1), 3), np.float32)
        check(self.ctx.ptr, lib().cs_orb_get_candidates(self.ctx.ptr, self._e, frame, level, _p(out, C.c_float), n.value, C.byref(n)), "cs_orb_get_candidates")
        return out[:n.value]

    def close(self):
        if self._e:
            lib().cs_orb_destroy(self.ctx.ptr, self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
