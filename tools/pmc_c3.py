"""Config 3's stream (ORB 2000 + lines + frame-to-frame SearchByProjection on 1241x376 frames) on a few frames, nothing else: the process the rocprofv3 --pmc passes of
bench.py's measure_traffic wrap for `match_candidates`.  python tools/pmc_c3.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = _lib.Context(0)
bench.c3_bench(ctx, frames, 1, False)
ctx.sync()
