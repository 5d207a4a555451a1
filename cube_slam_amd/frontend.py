"""ctypes wrapper of the batch front-end runner (cs_frontend_*, cube_slam_amd/csrc/frontend.hip): ORB + cuboid on the caller's
context, LSD + LBD on worker threads with their own contexts.  Plumbing for bench.py and the tests."""
import ctypes as C

from ._lib import check, lib


class Frontend:
    def __init__(self, ctx, orb=None, batch=None, line_detectors=(), phased=False):
        """line_detectors: line_lbd_detect objects, each created on its own Context and holding the same uploaded frames.
        phased: the detectors' device region stages run together after one pass per detector, with the caller's stream idle
        (cs_frontend_set_phased)."""
        self.ctx, self.orb, self.batch, self.lines = ctx, orb, batch, list(line_detectors)
        n = len(self.lines)
        ctxs = (C.c_void_p * max(n, 1))(*[d.ctx.ptr for d in self.lines])
        lsds = (C.c_void_p * max(n, 1))(*[d._l for d in self.lines])
        self._fe = C.c_void_p()
        check(ctx.ptr, lib().cs_frontend_create(ctx.ptr, orb._e if orb is not None else None, batch._b if batch is not None else None, n, ctxs, lsds, C.byref(self._fe)),
              "cs_frontend_create")
        if phased:
            self.set_phased(True)

    def set_phased(self, on):
        check(self.ctx.ptr, lib().cs_frontend_set_phased(self._fe, 1 if on else 0), "cs_frontend_set_phased")

    def set_chain(self, on, length_thres=15.0):
        """The reference's chain, pipelined: the cuboid pass of step k takes the lines of line pass k - W (cs_frontend_set_chain)."""
        check(self.ctx.ptr, lib().cs_frontend_set_chain(self._fe, 1 if on else 0, C.c_float(length_thres)), "cs_frontend_set_chain")

    def set_cuboid_ctx(self, ctx):
        """The cuboid batch on its own Context (stream), beside the ORB pass of the same step (cs_frontend_set_cuboid_ctx); None: the caller's stream."""
        check(self.ctx.ptr, lib().cs_frontend_set_cuboid_ctx(self._fe, ctx.ptr if ctx is not None else None), "cs_frontend_set_cuboid_ctx")
        self._cub_ctx = ctx

    def set_backlog(self, n_steps):
        """n_steps more step() calls follow on the same frames: their line passes may start as soon as a worker is free (cs_frontend_set_backlog)."""
        check(self.ctx.ptr, lib().cs_frontend_set_backlog(self._fe, int(n_steps)), "cs_frontend_set_backlog")

    def queues(self):
        """(enough, streams the runner keeps busy, hardware queues of this process) -- cs_frontend_queues."""
        a, b = C.c_int(0), C.c_int(0)
        r = lib().cs_frontend_queues(self._fe, C.byref(a), C.byref(b))
        if r < 0:
            check(self.ctx.ptr, r, "cs_frontend_queues")
        return bool(r), a.value, b.value

    def stream_begin(self, n_frames, width, height, n_slots=3):
        """From now on every step takes its frames from the host through a ring of device slots (cs_frontend_stream_begin)."""
        check(self.ctx.ptr, lib().cs_frontend_stream_begin(self._fe, int(n_frames), int(width), int(height), int(n_slots)), "cs_frontend_stream_begin")

    def stream_push(self, gray):
        """The frames (uint8, C-contiguous; pinned memory makes this asynchronous) of the next step that has none yet.  The array must stay alive until that step has run."""
        assert gray.dtype.name == "uint8" and gray.flags["C_CONTIGUOUS"]
        check(self.ctx.ptr, lib().cs_frontend_stream_push(self._fe, gray.ctypes.data_as(C.POINTER(C.c_uint8))), "cs_frontend_stream_push")

    def stream_push_scene(self, gray, packed):
        """stream_push with the frames' poses, boxes and (optionally) edge lists: packed = CuboidBatch.pack_scene(...) (cs_frontend_stream_push_scene; the runner copies them)."""
        assert gray.dtype.name == "uint8" and gray.flags["C_CONTIGUOUS"]
        T, bo, boxes, lo, lines = packed
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
        check(self.ctx.ptr, lib().cs_frontend_stream_push_scene(self._fe, gray.ctypes.data_as(C.POINTER(C.c_uint8)), dp(T), ip(bo), dp(boxes), None if lo is None else ip(lo),
                                                                None if lines is None else dp(lines)), "cs_frontend_stream_push_scene")
        if self.batch is not None:
            self.batch.n_boxes = int(bo[-1])

    def stream_read_async(self, kps=None, desc=None, cuboids=None, counts=None):
        """The results of the step just enqueued into the caller's (pinned) arrays, on a copy stream behind the step's kernels (cs_frontend_stream_read_async): kps
        (KEYPOINT_DTYPE) / desc ((n, 32) uint8) packed over the frames, cuboids ((n_boxes, max_cuboid_num) CUBOID_DTYPE) / counts (int32).  Returns (first, total) of the
        packed key points (or None).  stream_read_wait() before the arrays are read."""
        import numpy as np
        first, total = None, C.c_long(0)
        if kps is not None:
            first = np.zeros(self.orb.n_frames + 1, np.int32)
        check(self.ctx.ptr, lib().cs_frontend_stream_read_async(self._fe, None if kps is None else kps.ctypes.data_as(C.c_void_p), None if desc is None else desc.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                                0 if kps is None else len(kps), None if first is None else first.ctypes.data_as(C.POINTER(C.c_int)), C.byref(total),
                                                                None if cuboids is None else cuboids.ctypes.data_as(C.c_void_p), None if counts is None else counts.ctypes.data_as(C.POINTER(C.c_int))),
              "cs_frontend_stream_read_async")
        return first, total.value

    def stream_read_wait(self):
        check(self.ctx.ptr, lib().cs_frontend_stream_read_wait(self._fe), "cs_frontend_stream_read_wait")

    def stream_end(self):
        check(self.ctx.ptr, lib().cs_frontend_stream_end(self._fe), "cs_frontend_stream_end")

    def step(self):
        check(self.ctx.ptr, lib().cs_frontend_step(self._fe), "cs_frontend_step")

    def drain(self):
        check(self.ctx.ptr, lib().cs_frontend_drain(self._fe), "cs_frontend_drain")

    def close(self):
        if self._fe:
            lib().cs_frontend_destroy(self._fe)
            self._fe = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
