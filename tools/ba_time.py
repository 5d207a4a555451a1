"""Object BA at BASELINE config 5 (1000 key frames, 100 k points, 500 cuboids): LM iterations per second and the solver kernels, for the band solvers."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.ba import BundleAdjuster
ctx = _lib.Context(0)
d = synth.ba_problem(5, n_kf=1000, n_points=100000, n_cuboids=500)
for solver in ("band", "cr"):
    os.environ["CUBESLAM_BA_SOLVER"] = solver
    ba = BundleAdjuster(d, ctx=ctx)
    ba.optimize(2)
    ba.close()
    ba = BundleAdjuster(d, ctx=ctx)
    t0 = time.time(); st = ba.optimize(20); dt = time.time() - t0
    print(solver, "untimed: it/s %.1f" % (st["iterations"] / dt), "trials", st["lm_trials"])
    ba.close()
    ba = BundleAdjuster(d, ctx=ctx)
    ctx.timing(True); ctx.timing_reset()
    t0 = time.time(); st = ba.optimize(10); dt = time.time() - t0
    print(solver, "it/s %.1f" % (st["iterations"] / dt), "trials", st["lm_trials"], "chi2", st["chi2_final"])
    for k in ("ba_band_twist_factor", "ba_band_mid", "ba_band_twist_back", "ba_cr_assemble", "ba_cr_eliminate", "ba_cr_back", "ba_band_assemble", "ba_schur_slots", "ba_lin_pose"):
        t = ctx.timing_get(k)
        if t[1]: print("   %-22s %8.3f ms total %5d launches  %7.1f us each" % (k, t[0], t[1], 1e3 * t[0] / t[1]))
    ctx.timing(False)
    ba.close()
