"""Device region stage of the line detector against the oracle and against the host stage, with timings (run on the GPU box).
usage: python tools/lsd_regions_check.py [frames] [texture]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tex = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ctx = _lib.Context(0)
imgs = [synth.cuboid_scene(100 + i, n_boxes=3, bg_texture=tex)["gray"] for i in range(F - 2)]
imgs += [np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "orb_cabinet.npz"))["gray"], synth.texture_image(8, 640, 480)]
g = np.stack(imgs)
det = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
res = {}
for mode in ("host", "device"):
    os.environ["CUBESLAM_LSD_REGIONS"] = mode
    det.upload(g)
    det.run(with_lbd=False)
    t0 = time.time()
    for _ in range(3):
        det.run(with_lbd=False)
    dt = (time.time() - t0) / 3
    res[mode] = [det.read(f, with_desc=False) for f in range(F)]
    print(mode, "ms/batch %.2f" % (dt * 1e3), "lines/frame %.1f" % np.mean([len(k) for k in res[mode]]), det.region_stats(), flush=True)
bad = 0
for f in range(F):
    same = res["host"][f].tobytes() == res["device"][f].tobytes()
    if not same:
        bad += 1
        a, b = res["host"][f], res["device"][f]
        print("frame", f, "host", len(a), "device", len(b))
        if len(a) == len(b):
            d = [i for i in range(len(a)) if a[i].tobytes() != b[i].tobytes()]
            print("  differing lines", d[:10], a[d[0]], b[d[0]])
print("frames differing host/device:", bad, "of", F)
nchk = min(F, 4)
okc = sum(res["device"][f].tobytes() == po.lsd_detect(imgs[f]).tobytes() for f in list(range(nchk - 2)) + [F - 2, F - 1])
print("device == oracle on", okc, "of", nchk)
