#!/usr/bin/env python3
"""Host-stage scaling probe: time of the LSD / ORB host stages per batch for the current OMP_NUM_THREADS."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = _lib.Context(0)
imgs = np.stack([synth.cuboid_scene(1000 + i)["gray"] for i in range(F)])
lsd = line_lbd_detect(640, 480, max_frames=F, ctx=ctx); lsd.upload(imgs)
orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_frames=F, ctx=ctx); orb.upload(imgs)
for _ in range(2):
    lsd.run(True); orb.run()
ctx.timing(True); ctx.timing_reset()
t0 = time.perf_counter()
for _ in range(5):
    lsd.run(True)
t1 = time.perf_counter()
for _ in range(5):
    orb.run()
t2 = time.perf_counter()
print("OMP", os.environ.get("OMP_NUM_THREADS"), "F", F, "lsd.run %.1f ms (host %.1f)  orb.run %.1f ms (host %.1f)" % (
    (t1 - t0) / 5 * 1e3, ctx.timing_get("host_lsd_regions")[0] / 5, (t2 - t1) / 5 * 1e3, ctx.timing_get("host_orb_quadtree")[0] / 5))

for k in ("host_lsd_cpu_sort", "host_lsd_cpu_grow", "host_lsd_cpu_rect", "host_lsd_n_seeds", "host_lsd_n_regions", "host_lsd_n_pix", "host_lsd_n_def"):
    print(k, ctx.timing_get(k)[0] / 5)
for k in ("lsd_blur_hv", "lsd_resize", "lsd_gradient", "lsd_scan_blocks", "lsd_scan_top", "lsd_scan_add", "lsd_emit", "lbd_blur5", "lbd_sobel", "lbd_line_desc",
          "orb_resize", "orb_fast_score", "orb_cells", "orb_scan", "orb_quadtree", "orb_compact_sel", "orb_blur", "orb_angle", "orb_desc"):
    t = ctx.timing_get(k)
    if t[1]:
        print("%-18s %8.1f us/call" % (k, t[0] / t[1] * 1e3))
