"""Host region stage of the line detector on a bench-sized batch: wall time per batch and CPU milliseconds by part (run on the GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = _lib.Context(0)
g = np.stack([synth.cuboid_scene(100 + i, n_boxes=3, bg_texture=0.5)["gray"] for i in range(F)])
det = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
det.upload(g)
det.run(with_lbd=False)
timed = len(sys.argv) > 2 and sys.argv[2] == "timed"
if timed:
    ctx.timing(True); ctx.timing_reset()
t0 = time.time()
R = 8
for _ in range(R):
    det.run(with_lbd=False)
dt = (time.time() - t0) / R
print("frames %d  ms/batch %.2f  frames/s %.0f" % (F, dt * 1e3, F / dt))
for k in () if not timed else ("host_lsd_regions", "host_lsd_cpu_sort", "host_lsd_cpu_grow", "host_lsd_cpu_rect", "host_lsd_cpu_improve", "host_lsd_n_seeds", "host_lsd_n_regions", "host_lsd_n_pix", "host_lsd_n_def"):
    t = ctx.timing_get(k)
    print("  %-20s total %12.2f  per batch %10.2f" % (k, t[0], t[0] / R))
