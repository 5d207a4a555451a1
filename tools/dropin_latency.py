#!/usr/bin/env python3
"""Latency of the drop-in single-frame calls (host buffers in, host results out: H2D + kernels + host stages + D2H) on the bench's textured
frames, call by call and kernel by kernel -- the breakdown behind bench.py's `pcie_inclusive`.  python tools/dropin_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cube_slam_amd import _lib
from cube_slam_amd.cuboid import detect_3d_cuboid
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor

ctx = _lib.Context(0)
scenes = bench.make_frames(8, 3, seed0=1000)
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, ctx=ctx)
lsd = line_lbd_detect(640, 480, ctx=ctx)


def med(f, n=24):
    for i in range(3):
        f(scenes[i % 8])
    t = []
    for i in range(n):
        t0 = time.perf_counter(); f(scenes[i % 8]); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


kl = [lsd.detect_raw_lines(s["gray"]) for s in scenes]
calls = [("ORBextractor::operator() 1000 feat", lambda s: orb(s["gray"])),
         ("detect_raw_lines (LSD)", lambda s: lsd.detect_raw_lines(s["gray"])),
         ("get_line_descriptors (LBD)", lambda s: lsd.get_line_descriptors(s["gray"], kl[0])),
         ("detect_cuboid (3 boxes, 180 yaws)", lambda s: det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"]))]
for name, f in calls:
    ctx.timing(False)
    wall = med(f)
    ctx.timing(True); ctx.timing_reset()
    n = 8
    for i in range(n):
        f(scenes[i])
    ctx.sync()
    ks = {}
    for k in ("orb_resize", "orb_fast_score", "orb_cells", "orb_scan", "orb_quadtree", "orb_blur", "orb_angle", "orb_desc", "host_orb_quadtree", "host_lsd_regions", "lsd_blur_hv", "lsd_resize", "lsd_gradient",
              "lsd_emit", "lsd_rg_seq", "lsd_rg_improve", "lbd_blur5", "lbd_sobel", "lbd_line_desc", "cuboid_frame_prep", "cuboid_unit_lines", "cuboid_canny_nms", "cuboid_canny_cc", "cuboid_dt", "cuboid_vp",
              "cuboid_sweep_filter", "cuboid_sweep_score", "cuboid_select"):
        ms, cnt = ctx.timing_get(k)
        if cnt:
            ks[k] = round(1e3 * ms / n, 1)
    ctx.timing(False)
    print("%-36s %6.2f ms per call; per call in us: %s  (sum %.0f us)" % (name, wall, ks, sum(ks.values())), flush=True)
