"""Region stage of the line detector, one wave per frame (CUBESLAM_LSD_REGIONS=seq), against the host stage and the oracle, with timings at
several batch sizes (run on the GPU box).  usage: python tools/lsd_seq_check.py [max_frames] [distinct_scenes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

FMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = _lib.Context(0)
base = [synth.cuboid_scene(100 + i, n_boxes=3, bg_texture=0.5)["gray"] for i in range(D - 2)]
base += [np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "orb_cabinet.npz"))["gray"], synth.texture_image(8, 640, 480)]
sizes = [s for s in (D, 128, 512, 1024, 2048) if s <= FMAX]
for F in sizes:
    g = np.stack([base[i % D] for i in range(F)])
    det = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
    res = {}
    for mode in ("host", "seq"):
        os.environ["CUBESLAM_LSD_REGIONS"] = mode
        det.upload(g)
        det.run(with_lbd=False)
        ctx.timing(True); ctx.timing_reset()
        R = 2
        t0 = time.time()
        for _ in range(R):
            det.run(with_lbd=False)
        dt = (time.time() - t0) / R
        ks = {k: ctx.timing_get(k)[0] / R for k in ("lsd_rg_seq", "lsd_rg_improve", "lsd_rg_fill", "lsd_rg_scatter", "host_lsd_regions")}
        ctx.timing(False)
        res[mode] = [det.read(f, with_desc=False) for f in range(min(F, D))]
        print("F", F, mode, "ms/batch %.2f  frames/s %.0f" % (dt * 1e3, F / dt), "lines/frame %.1f" % np.mean([len(k) for k in res[mode]]), det.region_stats(),
              {k: round(v, 2) for k, v in ks.items() if v}, flush=True)
    bad = sum(res["host"][f].tobytes() != res["seq"][f].tobytes() for f in range(min(F, D)))
    print("F", F, "frames differing host/seq:", bad, "of", min(F, D), flush=True)
    if F == sizes[0]:
        chk = [0, 1, D - 2, D - 1]
        print("seq == oracle on", sum(res["seq"][f].tobytes() == po.lsd_detect(base[f]).tobytes() for f in chk), "of", len(chk), flush=True)
    det.close()
