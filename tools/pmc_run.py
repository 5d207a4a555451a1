import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
import bench
ctx = _lib.Context(0)
scenes = bench.make_frames(128, 3, 1000)
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
batch = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
for _ in range(3):
    batch.run()
ctx.sync()
