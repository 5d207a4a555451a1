"""(development) BASELINE config 4's share of one GPU (64 frames x 8 boxes) under the score kernel's launch knobs: python tools/c4_probe.py [setting ...], a setting is
`default` or KEY=VALUE[,KEY=VALUE] over CUBESLAM_SCORE_THREADS / CUBESLAM_SCORE_SEGMENTS / CUBESLAM_SCORE_SLICES; CUBESLAM_SCORE_PROF=1 prints the in-kernel phase profile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cube_slam_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
for sg in sys.argv[1:] or ["default"]:
    for k in ("CUBESLAM_SCORE_THREADS", "CUBESLAM_SCORE_SEGMENTS", "CUBESLAM_SCORE_SLICES"):
        os.environ.pop(k, None)
    if sg != "default":
        for kv in sg.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
    r = bench.c4_bench(ctx, 64, 8, 0.5, 10, with_cpu=False)
    print(sg, "frames/s %.0f  ms/batch %.2f  score %.1f us  filter %.1f us  frac %.3f  valid %d roi px %d" % (r["value"], r["ms_per_batch"], r["roofline"]["avg_kernel_us"], r["roofline"]["filter_kernel_us"],
                                                                                                      r["roofline"]["frac"], r["valid_proposals_per_batch"], r["roi_pixels_per_batch"]), flush=True)
