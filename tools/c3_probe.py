#!/usr/bin/env python3
"""Config-3 shaped probe: 1241x376 textured stream, 2000 ORB features + LSD/LBD lines, frames resident; per-stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
from cube_slam_amd import _lib, synth
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1241, 376
ctx = _lib.Context(0)
imgs = np.stack([synth.texture_image(77, W, H, shift=3 * i) for i in range(F)])
orb = ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_frames=F, ctx=ctx); orb.upload(imgs)
lsd = line_lbd_detect(W, H, max_frames=F, ctx=ctx); lsd.upload(imgs)
for _ in range(2):
    orb.run(); lsd.run(True)
ctx.timing(True); ctx.timing_reset()
t0 = time.perf_counter()
for _ in range(5):
    orb.run()
t1 = time.perf_counter()
for _ in range(5):
    lsd.run(True)
t2 = time.perf_counter()
nk = sum(len(k) for k, _ in orb.read()); nl = sum(len(lsd.read(f, with_desc=False)) for f in range(F))
print("F=%d  orb.run %.2f ms (%d keypoints)  lsd+lbd %.2f ms (%d lines)" % (F, (t1 - t0) / 5 * 1e3, nk, (t2 - t1) / 5 * 1e3, nl))
for k in ("host_orb_quadtree", "orb_quadtree", "orb_compact_sel", "orb_resize", "orb_fast_score", "orb_cells", "orb_blur", "orb_angle", "orb_desc", "host_lsd_regions", "host_lsd_cpu_sort", "host_lsd_cpu_grow",
          "host_lsd_cpu_rect", "host_lsd_n_def", "lsd_blur_hv", "lsd_resize", "lsd_gradient", "lsd_emit", "lbd_blur5", "lbd_sobel", "lbd_line_desc"):
    ms, n = ctx.timing_get(k)
    print("  %-22s %10.3f ms total / 5 runs = %8.3f" % (k, ms, ms / 5))
