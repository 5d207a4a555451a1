#!/usr/bin/env python3
"""Dynamic-object BA (cs_ba_dyn_*): LM iterations/s of a KITTI-sized local window on the GPU against the single-threaded oracle, plus the
per-kernel times the context's timing hooks collect.  python tools/badyn_bench.py [n_kf n_points n_objects pts_per_obj]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (same HIP runtime copy as the library, see tests/conftest.py)
torch.cuda.is_available()
import numpy as np
from cube_slam_amd import _lib, synth
from cube_slam_amd.ba_dynamic import DynamicBundleAdjuster
from oracle import pyoracle

nums = [int(x) for x in sys.argv[1:] if not x.startswith('-')]
a = nums + [20, 2000, 6, 40][len(nums):]
d = synth.ba_dyn_problem(7, n_kf=a[0], n_points=a[1], n_objects=a[2], pts_per_obj=a[3])
ctx = _lib.Context(0)
ba = DynamicBundleAdjuster(d, ctx=ctx)
H, _ = ba.reduced_dense(1e-3)
print("pose scalars", H.shape[0], "edges", {k: len(d[k]) for k in ("obs_cam", "dobs_cam", "mot_from", "cobs_cam", "pc_obj")}, "dyn points", len(d["dpoints"]))
ba.optimize(2); ba.close()
ba = DynamicBundleAdjuster(d, ctx=ctx)  # un-instrumented: the rate
t0 = time.perf_counter(); st = ba.optimize(10); t1 = time.perf_counter()
ba.close()
print("gpu: %d iterations, %d trials in %.2f ms -> %.1f it/s; chi2 %.1f -> %.1f" % (st["iterations"], st["lm_trials"], (t1 - t0) * 1e3, st["iterations"] / (t1 - t0), st["chi2_init"], st["chi2_final"]))
if "--kernels" in sys.argv:  # event pairs around every launch: they add the launch gap to short kernels, use rocprofv3 for those
    ctx.timing(True); ctx.timing_reset()
    ba = DynamicBundleAdjuster(d, ctx=ctx)
    ba.optimize(10)
    for k in ("badyn_errors", "badyn_linearize", "badyn_schur_init", "badyn_dinv", "badyn_bd", "badyn_schur_blocks", "badyn_schur_rhs", "badyn_chol_panel", "badyn_chol_update",
              "badyn_chol_tri", "badyn_chol_solve", "badyn_backsub", "badyn_update", "badyn_diag"):
        t = ctx.timing_get(k)
        if t[1]:
            print("  %-18s %8.1f us/call x %d" % (k, t[0] / t[1] * 1e3, t[1]))
    ctx.timing(False)
pyoracle.build()
t0 = time.perf_counter(); _, so = pyoracle.badyn_optimize(d, 10); t1 = time.perf_counter()
print("cpu oracle: %d iterations in %.1f ms -> %.1f it/s; chi2 final %.1f" % (so["iterations"], (t1 - t0) * 1e3, so["iterations"] / (t1 - t0), so["chi2_final"]))
