#!/usr/bin/env python3
"""Object BA alone (the benchmark graph, 3 LM iterations) for rocprofv3 --pmc passes: python tools/pmc_ba.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.ba import BundleAdjuster
ctx = _lib.Context(0)
d = synth.ba_problem(20260923, n_kf=1000, n_points=100000, n_cuboids=500)
ba = BundleAdjuster(d, ctx=ctx)
ba.optimize(3)
ctx.sync()
ba.close()
