import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cube_slam_amd import _lib, synth
from cube_slam_amd.ba import BundleAdjuster
ctx = _lib.Context(0)
d = synth.ba_problem(20260923, n_kf=1000, n_points=100000, n_cuboids=500)
ba = BundleAdjuster(d, ctx=ctx); ba.optimize(2); ba.close()
os.environ["CUBESLAM_CR_PROF"] = "1"
ba = BundleAdjuster(d, ctx=ctx); ba.optimize(1); ctx.sync(); ba.close()
