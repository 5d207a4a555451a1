import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
import bench
ctx = _lib.Context(0)
scenes = bench.make_frames(128, 3, 1000)
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
batch = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
for dbg in [0]:
    if hasattr(_lib.lib(), "cs_debug_set"):
        _lib.lib().cs_debug_set(dbg)
    for _ in range(3): batch.run()
    ctx.timing(True); ctx.timing_reset()
    for _ in range(10): batch.run()
    for k in ("cuboid_sweep_corners", "cuboid_sweep_score", "cuboid_select"):
        ms, n = ctx.timing_get(k)
        print(k, "avg %.1f us" % (1e3 * ms / max(n, 1)))
    ctx.timing(False)
