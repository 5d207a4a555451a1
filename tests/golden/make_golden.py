#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

* cuboid_ref_0000.npz : the reference's own demo input for detect_3d_cuboid (detect_3d_cuboid/data/0000_rgb_raw.jpg
  decoded with PIL and converted with the oracle's BGR2GRAY, data/edge_detection/LSD/0000_edge.txt, and the constants
  hard-coded in detect_3d_cuboid/src/main.cpp:35-48) + the oracle's output on it.  The reference stores NO expected
  output for this demo (it only draws it), so this pins the oracle against regressions, not against the reference:
  parity stays "unpinned" (DESIGN.md).
* cuboid_synth.npz    : oracle outputs for seeded synthetic scenes (regression vectors for CPU + GPU tests).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from cube_slam_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REF = "/root/reference"


def cuboid_ref():
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(REF, "detect_3d_cuboid/data/0000_rgb_raw.jpg")).convert("RGB"))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    gray = po.bgr2gray(bgr)
    lines = np.loadtxt(os.path.join(REF, "detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt")).reshape(-1, 4)
    K = np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]])           # main.cpp:35-38
    Twc = np.array([[1, 0.0011, 0.0004, 0], [0, -0.3376, 0.9413, 0], [0.0011, -0.9413, -0.3376, 1.35], [0, 0, 0, 1.0]])  # :40-44
    box = np.array([[188, 189, 201, 311, 0.88]])                                  # :46-48
    box[:, :2] -= 1
    res, dbg = po.detect_cuboid(gray, K, Twc, box, lines, opts=po.cuboid_opts(max_cuboid_num=3), debug=True)
    np.savez_compressed(os.path.join(HERE, "cuboid_ref_0000.npz"), gray=gray, lines=lines, K=K, Twc=Twc, box=box,
                        cuboids=res[0], n_valid=dbg["row_count"][:1], rows_head=dbg["rows"][:20])
    print("cuboid_ref_0000: valid proposals", dbg["row_count"][0], "best pos", res[0]["pos"][0], "scale", res[0]["scale"][0])


def cuboid_synth():
    out = {}
    for i, (seed, kw) in enumerate([(11, {}), (12, {"yaw_step_deg": 0.5}), (13, {"whether_sample_cam_roll_pitch": 1, "max_cuboid_num": 2}),
                                    (14, {"whether_sample_bbox_height": 1, "max_cuboid_num": 2})]):
        s = synth.cuboid_scene(seed, n_boxes=3)
        res, dbg = po.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=po.cuboid_opts(**kw), debug=True)
        out["case%d_seed" % i] = seed
        out["case%d_kw" % i] = np.array(repr(kw))
        out["case%d_counts" % i] = dbg["row_count"][:9]
        out["case%d_cuboids" % i] = np.concatenate(res)
    np.savez_compressed(os.path.join(HERE, "cuboid_synth.npz"), **out)
    print("cuboid_synth written")


def orb_cabinet():
    """line_lbd/data/cabinet.png (640x480, the reference's LSD demo image) -> gray fixture + oracle ORB output digest."""
    import hashlib
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(REF, "line_lbd/data/cabinet.png")).convert("RGB"))
    gray = po.bgr2gray(np.ascontiguousarray(rgb[:, :, ::-1]))
    k, d = po.ORBextractor(1000, 1.2, 8, 20, 7)(gray)
    np.savez_compressed(os.path.join(HERE, "orb_cabinet.npz"), gray=gray, n=len(k), kp_sha=hashlib.sha256(k.tobytes()).hexdigest(),
                        desc_sha=hashlib.sha256(d.tobytes()).hexdigest(), kp_head=k[:16], desc_head=d[:16])
    print("orb_cabinet:", len(k), "keypoints")


def lines_cabinet():
    """LSD KeyLines + LBD descriptors of the cabinet fixture (gray array already decoded in orb_cabinet.npz)."""
    import hashlib
    gray = np.load(os.path.join(HERE, "orb_cabinet.npz"))["gray"]
    kl = po.lsd_detect(gray)
    desc = po.lbd_compute(gray, kl)
    np.savez_compressed(os.path.join(HERE, "lines_cabinet.npz"), n_lines=len(kl), keylines_sha256=hashlib.sha256(kl.tobytes()).hexdigest(),
                        desc=desc, filter15=po.lsd_detect_filter_lines(gray, 15.0), keylines_head=kl[:16])
    print("lines_cabinet:", len(kl), "lines")


if __name__ == "__main__":
    cuboid_ref()
    cuboid_synth()
    orb_cabinet()
    lines_cabinet()
