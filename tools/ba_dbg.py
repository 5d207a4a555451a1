import sys; sys.path.insert(0,'.')
import numpy as np
from cube_slam_amd import synth, _lib
from cube_slam_amd.ba import BundleAdjuster
from oracle import pyoracle as po
ctx=_lib.Context(0)
for kw in [dict(n_kf=60, n_points=2500, n_cuboids=15), dict(n_kf=60, n_points=2500, n_cuboids=0)]:
    d = synth.ba_problem(11, **kw)
    ba = BundleAdjuster(d, ctx=ctx)
    st = ba.optimize(15)
    _,_,_, rst = po.ba_optimize(d, 15)
    a=np.array(st['chi2_trace']); b=np.array(rst['chi2_trace'])
    print(kw, st['iterations'], rst['iterations'], st['lm_trials'], rst['lm_trials'])
    n=min(len(a),len(b)); print(np.abs(a[:n]-b[:n])/b[:n])
