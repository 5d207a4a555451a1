"""The BA block of bench.py alone (1 000 key frames, 100 k points, 500 cuboids): LM iterations/s and the per-kernel times.  `python tools/ba_only.py [iterations] [repeats]`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.Context(0)
best = None
for _ in range(reps):
    out = bench.ba_bench(ctx, 0, 1, iters, False)
    if best is None or out["value"] > best["value"]:
        best = out
print(json.dumps({k: best[k] for k in ("value", "ms_per_iteration", "iterations", "lm_trials", "chi2_final", "kernels_us")}))
