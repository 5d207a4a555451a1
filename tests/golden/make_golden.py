#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

* cuboid_ref_0000.npz : the reference's own demo input for detect_3d_cuboid (detect_3d_cuboid/data/0000_rgb_raw.jpg
  decoded with PIL and converted with the oracle's BGR2GRAY, data/edge_detection/LSD/0000_edge.txt, and the constants
  hard-coded in detect_3d_cuboid/src/main.cpp:35-48) + the oracle's output on it.  The reference stores NO expected
  output for this demo (it only draws it), so this pins the oracle against regressions, not against the reference:
  parity stays "unpinned" (DESIGN.md).
* cuboid_synth.npz    : oracle outputs for seeded synthetic scenes (regression vectors for CPU + GPU tests).
* badyn_synth.npz     : the dynamic-object BA oracle on one seeded window (chi2 trace, estimates, reduced system): regression vector.
* object_slam_seq.npz : the reference's bundled TUM-cabinet sequence (object_slam/data: 58 frames, YOLO boxes, pop_cam_poses_saved.txt) with
  the ONE set of expected outputs the reference ships for this path: detect_cuboids_saved.txt, the author's offline (MATLAB) detections
  `frame x y z yaw l w h err` that object_slam consumes when online_detect_mode is off (main_obj.cpp:475-497).  The C++ detector "differs
  slightly from MATLAB due to different canny edge and distance transform" (detect_3d_cuboid/README.md), so this is a LOOSE pin of the whole
  chain (BGR2GRAY -> LSD + length filter -> Canny / chamfer map -> proposal sweep -> scoring -> selection -> 3-D box) against an independent
  implementation by the reference's author, not a bit-level one: the oracle's best cuboid per frame is stored next to the saved rows
  (tests/test_cuboid_oracle.py checks their agreement), and four frames are stored as images so that the oracle is re-run in the test.
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


def object_slam_seq():
    import math
    from PIL import Image
    base = os.path.join(REF, "object_slam/data")
    K = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1.0]])  # main_obj.cpp:347-349
    rows = np.loadtxt(os.path.join(base, "detect_cuboids_saved.txt")).reshape(-1, 9)
    poses = np.loadtxt(os.path.join(base, "pop_cam_poses_saved.txt")).reshape(-1, 8)

    def twc(p):  # `t x y z qx qy qz qw` -> 4x4 (g2o::SE3Quat(Vector7d) normalises the quaternion)
        x, y, z, w = p[4:8] / np.linalg.norm(p[4:8])
        T = np.eye(4)
        T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        T[:3, 3] = p[1:4]
        return T
    opts = dict(whether_sample_bbox_height=0, nominal_skew_ratio=2.0, max_cuboid_num=1)  # main_obj.cpp:359-360
    keep = set(range(len(rows)))  # every frame of the sequence (round 3: the chain detect_filter_lines -> detect_cuboid is run on all of them on the device)
    ours, boxes, Twcs, grays, n_lines = [], [], [], {}, []
    for r, row in enumerate(rows):
        f = int(row[0])
        rgb = np.asarray(Image.open(os.path.join(base, "raw_imgs/%04d_rgb_raw.jpg" % f)).convert("RGB"))
        gray = po.bgr2gray(np.ascontiguousarray(rgb[:, :, ::-1]))
        lines = po.lsd_detect_filter_lines(gray, 15.0)  # line_lbd_obj.use_LSD = true, line_length_thres = 15 (:364-366)
        box = np.loadtxt(os.path.join(base, "filter_2d_obj_txts/%04d_yolo2_0.15.txt" % f)).reshape(-1, 5)[:1].copy()
        box[:, :2] -= 1  # MATLAB -> C++ coordinates (:441)
        T = twc(poses[f])
        res, _ = po.detect_cuboid(gray, K, T, box, lines, opts=po.cuboid_opts(**opts), debug=True)
        c = res[0][0] if len(res[0]) else None
        ours.append([np.nan] * 8 if c is None else [*c["pos"], c["rotY"], *c["scale"], c["normalized_error"]])
        boxes.append(box[0]); Twcs.append(T); n_lines.append(len(lines))
        if r in keep:
            grays[r] = gray
    np.savez_compressed(os.path.join(HERE, "object_slam_seq.npz"), K=K, matlab_rows=rows, ours=np.array(ours), boxes=np.array(boxes), Twc=np.array(Twcs),
                        n_lines=np.array(n_lines), kept=np.array(sorted(keep)), **{"gray_%d" % r: g for r, g in grays.items()})
    o = np.array(ours)
    print("object_slam_seq: %d rows, median position difference %.3f m" % (len(rows), np.nanmedian(np.linalg.norm(o[:, :3] - rows[:, 1:4], axis=1))))


def badyn_synth():
    """Regression vector of the dynamic-object BA oracle: chi2 trace and a digest of the estimates for one seeded window."""
    d = synth.ba_dyn_problem(101, n_kf=8, n_points=150, n_objects=2, pts_per_obj=14)
    res, st = po.badyn_optimize(d, 6)
    H, b = po.badyn_reduced_dense(d, 1e-3)
    np.savez_compressed(os.path.join(HERE, "badyn_synth.npz"), chi2_init=st["chi2_init"], chi2_trace=np.array(st["chi2_trace"]), lm_trials=st["lm_trials"],
                        cam_pose=res["cam_pose"], obj_pose=res["obj_pose"], vel=res["vel"], H_diag=np.diag(H).copy(), b=b)
    print("badyn_synth: chi2", st["chi2_init"], "->", st["chi2_final"])


if __name__ == "__main__":
    badyn_synth()
    object_slam_seq()
    cuboid_ref()
    cuboid_synth()
    orb_cabinet()
    lines_cabinet()
