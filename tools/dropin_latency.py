#!/usr/bin/env python3
"""Latency of the drop-in single-frame calls (host buffers in, host results out: H2D + kernels + host stages + D2H), the numbers
DESIGN.md 7.6 quotes.  python tools/dropin_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.cuboid import detect_3d_cuboid
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor
from cube_slam_amd.optimizer import PoseOptimization

ctx = _lib.Context(0)
s = synth.cuboid_scene(1000)
det = detect_3d_cuboid(ctx); det.set_calibration(s["K"]); det.yaw_step_deg = 0.5
orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, ctx=ctx)
lsd = line_lbd_detect(640, 480, ctx=ctx)


def med(f, n=30):
    for _ in range(3):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


print("detect_cuboid (3 boxes, 180 yaws)   %.2f ms" % med(lambda: det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"])))
print("ORBextractor::operator() 1000 feat  %.2f ms" % med(lambda: orb(s["gray"])))
print("detect_descrip_lines (LSD + LBD)    %.2f ms" % med(lambda: lsd.detect_descrip_lines(s["gray"])))
f = synth.pose_frame(5, n=800)
print("PoseOptimization (800 points)       %.2f ms" % med(lambda: PoseOptimization([f], ctx=ctx)))
