"""Region stage of the line detector with walker + rectangle waves (CUBESLAM_LSD_REGIONS=wlk: lsd_rg_wlk.h) against the host stage, the one-wave-per-frame
stage and the oracle, with kernel timings (run on the GPU box).  usage: python tools/lsd_wlk_check.py [frames] [distinct_scenes] [modes,comma-separated]
WLK_SHAPES="1,8,1;2,8,1;..." = walkers, waves per workgroup, accepted pixels per iteration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = _lib.Context(0)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
with ThreadPoolExecutor(16) as ex:  # D distinct scenes (frames that repeat walk in lockstep inside a wave and flatter the several-frames-per-wave stages)
    base = list(ex.map(lambda i: synth.cuboid_scene(100 + i, n_boxes=3, bg_texture=0.5)["gray"], range(D - 2)))
base += [np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "orb_cabinet.npz"))["gray"], synth.texture_image(8, 640, 480)]
g = np.stack([base[i % D] for i in range(F)])
det = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
variants = [("host", None), ("seq", None)]
variants += [("wlk", shape) for shape in os.environ.get("WLK_SHAPES", "1,8,1;1,8,2;1,4,1;2,8,1;2,16,1;4,16,1;8,16,1").split(";")]  # walkers, waves per workgroup, accepts per iteration
if len(sys.argv) > 3:
    variants = [v for v in variants if v[0] in sys.argv[3].split(",")]
res = {}
for mode, wpb in variants:
    os.environ["CUBESLAM_LSD_REGIONS"] = mode
    if mode == "wlk":
        os.environ["CUBESLAM_LSD_WLK"] = wpb
    det.upload(g)
    det.run(with_lbd=False)
    ctx.timing(True); ctx.timing_reset()
    R = 2
    t0 = time.time()
    for _ in range(R):
        det.run(with_lbd=False)
    dt = (time.time() - t0) / R
    ks = {k: ctx.timing_get(k)[0] / R for k in ("lsd_rg_seq", "lsd_rg_wlk", "lsd_rg_improve", "lsd_rg_fill", "lsd_rg_scatter", "host_lsd_regions")}
    ctx.timing(False)
    key = mode + ("/%s" % wpb if wpb else "")
    res[key] = [det.read(f, with_desc=False) for f in range(min(F, D))]
    print("F", F, key, "ms/batch %.2f  frames/s %.0f" % (dt * 1e3, F / dt), det.region_stats(), {k: round(v, 2) for k, v in ks.items() if v}, flush=True)
ref = res.get("host") or next(iter(res.values()))
for key, r in res.items():
    bad = sum(ref[f].tobytes() != r[f].tobytes() for f in range(min(F, D)))
    print(key, "frames differing from", "host" if "host" in res else "first", ":", bad, "of", min(F, D), flush=True)
chk = [0, 1, D - 2, D - 1] if os.environ.get("CHECK_ORACLE", "1") == "1" else []
for key, r in res.items():
    print(key, "== oracle on", sum(r[f].tobytes() == po.lsd_detect(base[f]).tobytes() for f in chk), "of", len(chk), flush=True)
det.close()
