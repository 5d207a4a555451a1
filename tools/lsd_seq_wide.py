"""Device region stage against the host stage on many different frames (textures 0 ... 1, two sizes): KeyLines byte for byte (run on the GPU box).
usage: python tools/lsd_seq_wide.py [frames_per_size]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

from cube_slam_amd import _lib, synth  # noqa: E402
from cube_slam_amd.lsd import line_lbd_detect  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = _lib.Context(0)
bad_total = 0
for (W, H) in ((640, 480), (1241, 376)):
    tex = [0.0, 0.1, 0.25, 0.5, 0.75, 1.0, 1.5]
    with ThreadPoolExecutor(16) as ex:
        imgs = list(ex.map(lambda i: synth.cuboid_scene(5000 + i, W=W, H=H, n_boxes=3, bg_texture=tex[i % len(tex)])["gray"] if i % 9 else synth.texture_image(700 + i, W, H), range(N)))
    g = np.stack(imgs).astype(np.uint8)
    det = line_lbd_detect(W, H, max_frames=N, ctx=ctx)
    res = {}
    for mode in ("host", "seq"):
        os.environ["CUBESLAM_LSD_REGIONS"] = mode
        det.upload(g)
        det.run(with_lbd=True)
        res[mode] = [det.read(f) for f in range(N)]
        print(W, H, mode, det.region_stats(), "lines/frame %.1f" % np.mean([len(k) for k, _ in res[mode]]), flush=True)
    bad = sum(res["host"][f][0].tobytes() != res["seq"][f][0].tobytes() or res["host"][f][1].tobytes() != res["seq"][f][1].tobytes() for f in range(N))
    print(W, H, "frames differing host/seq (KeyLines or descriptors):", bad, "of", N, flush=True)
    bad_total += bad
    det.close()
sys.exit(1 if bad_total else 0)
