#!/usr/bin/env python3
"""Experiment: per-workgroup wall time of cuboid_sweep_score on the bench batch (CUBESLAM_SCORE_MODE bit 8 makes the kernel record it)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CUBESLAM_SCORE_MODE"] = str(int(os.environ.get("CUBESLAM_SCORE_MODE", "0")) | 8)
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid  # noqa: E402

ctx = _lib.Context(0)
scenes = bench.make_frames(128, 3, seed0=1000)
det = detect_3d_cuboid(ctx)
det.set_calibration(scenes[0]["K"])
det.yaw_step_deg = 0.5
b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]),
                [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
for _ in range(3):
    b.run()
ctx.sync()
G = int(os.environ.get("CUBESLAM_SCORE_SEGMENTS", "256"))
out = np.zeros((G, 4), np.int64)
_lib.lib().cs_debug_score_dbg(ctx._ctx, out.ctypes.data_as(C.POINTER(C.c_long)), G)
t0 = out[:, 0].min()
st, en = (out[:, 0] - t0) / 100.0, (out[:, 1] - t0) / 100.0  # 100 MHz clock -> us
dur = en - st
print("start us: min %.1f max %.1f | end us: min %.1f med %.1f max %.1f | dur: min %.1f med %.1f max %.1f" % (st.min(), st.max(), en.min(), np.median(en), en.max(), dur.min(), np.median(dur), dur.max()))
print("tasks per WG: min %d med %d max %d; units per WG: %s" % (out[:, 2].min(), np.median(out[:, 2]), out[:, 2].max(), np.bincount(out[:, 3].astype(int))))
order = np.argsort(-dur)[:8]
print("slowest:", [(int(i), round(float(dur[i]), 1), int(out[i, 2]), int(out[i, 3])) for i in order])
print("us per task: ", np.round(np.percentile(dur / np.maximum(out[:, 2], 1), [5, 50, 95]), 2))
b.close()
