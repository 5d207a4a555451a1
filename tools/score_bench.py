#!/usr/bin/env python3
"""Isolated per-kernel times of the cuboid path on the bench batch (frames x 3 boxes, yaw step 0.5 deg) for several settings of the score
kernel's knobs: python tools/score_bench.py [frames] [setting ...], a setting is `default` or KEY=VALUE[,KEY=VALUE] over
CUBESLAM_SCORE_THREADS (512 | 1024), CUBESLAM_SCORE_SEGMENTS (workgroups), CUBESLAM_SCORE_SLICES (items per unit)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    segs = sys.argv[2:] or ["default"]
    ctx = _lib.Context(0)
    scenes = bench.make_frames(frames, 3, seed0=1000)
    det = detect_3d_cuboid(ctx)
    det.set_calibration(scenes[0]["K"])
    det.yaw_step_deg = 0.5
    ref = None
    for sg in segs:
        for k in ("CUBESLAM_SCORE_THREADS", "CUBESLAM_SCORE_SEGMENTS", "CUBESLAM_SCORE_SLICES"):
            os.environ.pop(k, None)
        if sg != "default":
            for kv in sg.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
        b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]),
                        [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
        for _ in range(2):
            b.run()
        ctx.sync()
        ctx.timing(True); ctx.timing_reset()
        for _ in range(5):
            b.run()
        ctx.sync()
        names = ("cuboid_frame_prep", "cuboid_unit_lines", "cuboid_canny_nms", "cuboid_canny_cc_local", "cuboid_canny_cc_border", "cuboid_canny_cc", "cuboid_dt", "cuboid_vp", "cuboid_sweep_filter", "cuboid_sweep_score", "cuboid_select")
        t = {}
        for k in names:
            ms, n = ctx.timing_get(k)
            if n:
                t[k] = round(1e3 * ms / n, 1)
        ctx.timing(False)
        st = b.stats()
        b.score_stats()  # (prints the in-kernel phase profile under CUBESLAM_SCORE_PROF)
        alg = 4.0 * st["roi_pixels"] + 200.0 * st["n_valid"]  # SURVEY 8d, corner construction fused into the score kernel
        us = t.get("cuboid_sweep_score", 0)
        got = b.read()
        raw = b"".join(np.asarray(g).tobytes() for g in got)
        if ref is None:
            ref = raw
        print("setting=%s  %s  alg %.1f MB -> %.2f TB/s (%.3f of 8)  identical=%s" % (sg, t, alg / 1e6, alg / (us * 1e-6) / 1e12 if us else 0,
                                                                                    alg / (us * 1e-6) / 8e12 if us else 0, raw == ref), flush=True)
        b.close()


if __name__ == "__main__":
    main()
