"""(development) the line path alone through the runner: W detectors of 1 024 frames, no ORB, no cuboid batch -- how the line path scales with detectors in flight.
python tools/lines_only_probe.py W [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from cube_slam_amd import _lib
from cube_slam_amd.frontend import Frontend
from cube_slam_amd.lsd import line_lbd_detect
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
F = 1024
scenes = bench.make_frames(F, 3, seed0=1000)
gray = np.stack([s["gray"] for s in scenes])
ctx = _lib.Context(0, priority=1)
lctx = [_lib.Context(0, priority=1) for _ in range(W)]
lsds = [line_lbd_detect(640, 480, max_frames=F, ctx=c) for c in lctx]
for d in lsds:
    d.upload(gray)
fe = Frontend(ctx, orb=None, batch=None, line_detectors=lsds)
def barrier():
    fe.drain(); ctx.sync()
    for c in lctx: c.sync()
fe.set_backlog(W + 1)
for _ in range(W + 1): fe.step()
barrier()
t0 = time.perf_counter()
fe.set_backlog(steps)
for _ in range(steps): fe.step()
barrier()
dt = time.perf_counter() - t0
print("workers %d: %.2f ms per step, %.0f frames/s (lines only)" % (W, 1e3 * dt / steps, F * steps / dt))
