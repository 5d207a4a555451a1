import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from concurrent.futures import ThreadPoolExecutor
from cube_slam_amd import _lib
from cube_slam_amd.cuboid import detect_3d_cuboid
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor
scenes = bench.make_frames(16, 3, seed0=1000)
T = 16
ctxs = [_lib.Context(0) for _ in range(T)]
def objs(c):
    det = detect_3d_cuboid(c); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
    return det, ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, ctx=c), line_lbd_detect(640, 480, ctx=c)
O = [objs(c) for c in ctxs]
def run(which, threads, per=12):
    def one(t, i):
        s = scenes[(t + i) % 16]; det, ext, ll = O[t]
        if "orb" in which: ext(s["gray"])
        if "lsd" in which: ll.detect_raw_lines(s["gray"])
        if "cub" in which: det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"])
    for t in range(threads): one(t, 0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda t: [one(t, i) for i in range(per)], range(threads)))
    dt = time.perf_counter() - t0
    print("%-14s threads %2d: %7.0f calls/s  (%.2f ms per call per thread)" % (which, threads, threads * per / dt, 1e3 * dt / per), flush=True)
for which in ("orb", "lsd", "cub", "orb+lsd+cub"):
    for th in (1, 4, 16):
        run(which, th)
