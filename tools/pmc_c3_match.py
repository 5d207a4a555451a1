"""Config 3's window search alone (ORB on `frames` frames of the 1241x376 stream, then ONE cs_match_by_projection_stream over the window's pairs), nothing else: the process the
rocprofv3 --pmc passes of bench.py's measure_traffic wrap for `match_candidates` (one launch per window since round 6: count, slice from a cursor, fill).
python tools/pmc_c3_match.py [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.matcher import ORBmatcherStream  # noqa: E402
from cube_slam_amd.orb import ORBextractor  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, H = 1241, 376
fx, fy, cx, cy = 721.5377, 721.5377, 609.5593, 172.854
ctx = _lib.Context(0)
orb = ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_frames=frames, ctx=ctx)
orb.upload(synth.texture_stream(77, W, H, frames, step=3))
orb.run()
kall, _, first = orb.read_packed()
pk = kall[:first[frames - 1]]
z = np.full(len(pk), 10.0, np.float32)
wp = np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32)
ones = np.ones(len(pk), np.uint8)
ms = ORBmatcherStream(True, ctx=ctx)
sf = np.array([1.2 ** i for i in range(8)], np.float32)
ms.search(orb, 0, frames - 1, np.array([fx, fy, cx, cy], np.float32), None, (0.0, float(W), 0.0, float(H)), wp, ones, ones, np.broadcast_to(np.eye(4, dtype=np.float32)[:3], (frames - 1, 3, 4)),
          fx, fy, cx, cy, sf, 15.0, int(first[frames] - first[1]))
ctx.sync()
