"""The cuboid path on the bench batch, a few runs, nothing else: the process rocprofv3 --pmc passes wrap (bench.py measure_traffic,
tools/run_pmc.sh).  python tools/pmc_run.py [frames boxes yaw_step bg_texture seed0]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
boxes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
yaw = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
tex = float(sys.argv[4]) if len(sys.argv) > 4 else bench.BG_TEXTURE
seed0 = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
ctx = _lib.Context(0)
scenes = bench.make_frames(frames, boxes, seed0, bg_texture=tex)
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = yaw
batch = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
for _ in range(3):
    batch.run()
ctx.sync()
