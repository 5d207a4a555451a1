#!/usr/bin/env python3
"""(development) bench.py's c3 block alone: python tools/c3_only.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
from cube_slam_amd import _lib
ctx = _lib.Context(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = bench.c3_bench(ctx, 2 * _lib.lib().cs_host_thread_count(), steps, with_cpu=False, with_traffic=False, with_small_window=bw)
print(json.dumps({k: out[k] for k in ("value", "ms_per_frame", "frames", "keypoints_per_frame", "keylines_per_frame", "kernels_us", "runner", "small_window") if k in out}))
