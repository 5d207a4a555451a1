#!/usr/bin/env python3
"""cuboid_sweep_score variants on the bench workload (128 frames x 3 boxes, 180 yaws): the default global-gather kernel against
CUBESLAM_SCORE=lds (16-bit chamfer codes resident in LDS).  Prints the isolated per-kernel times and checks that both produce byte-identical
cuboids.  python tools/score_variants.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
torch.cuda.is_available()
import numpy as np
from cube_slam_amd import _lib, synth
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = _lib.Context(0)
scenes = [synth.cuboid_scene(1000 + i, n_boxes=3) for i in range(F)]
det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"]); det.yaw_step_deg = 0.5
out = {}
for mode in ("global", "lds"):
    if mode == "lds":
        os.environ["CUBESLAM_SCORE"] = "lds"
    else:
        os.environ.pop("CUBESLAM_SCORE", None)
    batch = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes],
                        [s["lines"] for s in scenes], det.opts())
    for _ in range(2):
        batch.run()
    ctx.sync(); ctx.timing(True); ctx.timing_reset()
    for _ in range(5):
        batch.run()
    ctx.sync()
    t = {k: ctx.timing_get(k) for k in ("cuboid_sweep_score", "cuboid_sweep_score_lds", "cuboid_sweep_corners", "cuboid_dt", "cuboid_select")}
    ctx.timing(False)
    print(mode, {k: round(v[0] / v[1] * 1e3, 1) for k, v in t.items() if v[1]})
    got = batch.read()
    out[mode] = np.concatenate([np.asarray(g).view(np.uint8).reshape(-1) for g in got if len(g)])
    print(mode, "cuboids", sum(len(g) for g in got), "stats", batch.stats())
print("identical:", out["global"].shape == out["lds"].shape and bool(np.array_equal(out["global"], out["lds"])))
