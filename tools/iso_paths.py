#!/usr/bin/env python3
"""Each front-end path ALONE on the GPU on the bench batch (textured 640x480 frames, 3 boxes, yaw step 0.5): ORB, the line path (LSD with the
device region stage + LBD), the cuboid path -- one after the other, so that a rocprofv3 --kernel-trace of this process gives per-kernel
durations free of the three-stream overlap of bench.py.  python tools/iso_paths.py [frames] [runs]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402
from cube_slam_amd.orb import ORBextractor  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.Context(0)
scenes = bench.make_frames(F, 3, seed0=1000)
gray = np.stack([s["gray"] for s in scenes])


def timed(name, fn):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    ctx.sync()
    print("%-8s %8.2f ms per %d frames" % (name, 1e3 * (time.perf_counter() - t0) / R, F), flush=True)


orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_frames=F, ctx=ctx)
orb.upload(gray)
timed("orb", orb.run)
os.environ.setdefault("CUBESLAM_LSD_REGIONS", "seq")
lsd = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
lsd.upload(gray)
timed("lines", lambda: lsd.run(True))
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
b = CuboidBatch(ctx, gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
timed("cuboid", b.run)
st = b.stats()
print("stats", st, "keypoints/frame %.0f" % (sum(len(k) for k, _ in orb.read()) / F), "keylines/frame %.0f" % (sum(len(lsd.read(f, with_desc=False)) for f in range(F)) / F), flush=True)
