#!/usr/bin/env python3
"""bench.py's pcie_inclusive block in a process of its own: the drop-in calls frame by frame from one caller thread and from sixteen.  A process that makes
these calls is not the batch runner's, so it runs with the HIP runtime's default hardware queues (bench.py raises GPU_MAX_HW_QUEUES for the runner's ten
streams; with sixteen queues sixteen callers' tiny kernels lose half their rate: 1.3 k -> 0.66 k frames/s).  usage: tools/pcie_probe.py [default|N] [yaw_step] [nfeat]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (sets GPU_MAX_HW_QUEUES for the runner; undone below, before the HIP runtime starts)

q = sys.argv[1] if len(sys.argv) > 1 else "default"
if q == "default":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
else:
    os.environ["GPU_MAX_HW_QUEUES"] = q
yaw_step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
nfeat = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
import torch  # noqa: E402,F401  (HIP runtime order, see tests/conftest.py)
from cube_slam_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
scenes = bench.make_frames(192, 3, seed0=1000)
out = bench.pcie_inclusive(ctx, scenes, yaw_step, nfeat)
out["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")
print(json.dumps(out))
