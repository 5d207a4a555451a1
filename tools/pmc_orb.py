#!/usr/bin/env python3
"""ORB path alone (128 frames) for rocprofv3 --pmc passes: python tools/pmc_orb.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth
from cube_slam_amd.orb import ORBextractor
F = 128
ctx = _lib.Context(0)
imgs = np.stack([synth.cuboid_scene(1000 + i)["gray"] for i in range(F)])
orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_frames=F, ctx=ctx); orb.upload(imgs)
for _ in range(3):
    orb.run()
ctx.sync()
