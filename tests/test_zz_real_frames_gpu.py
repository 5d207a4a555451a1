"""GPU parity on real frames: the reference's bundled TUM-cabinet sequence (object_slam/data: the 51 frames detect_cuboids_saved.txt lists;
tests/golden/object_slam_seq.npz, see tests/golden/make_golden.py) through the reference's chain detect_filter_lines -> detect_cuboid
(main_obj.cpp:428-449) on the device -- frame by frame through the drop-in calls, and all frames at once as a resident batch whose edge
lists are this run's LSD output -- against the oracle results stored in the fixture (which tests/test_cuboid_oracle.py ties to the author's
saved detections).  Runs last on purpose (file name)."""
import os

import numpy as np
import pytest

from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
from cube_slam_amd.lsd import line_lbd_detect

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_object_slam_sequence_frames(ctx, oracle):
    g = np.load(os.path.join(GOLD, "object_slam_seq.npz"))
    lsd = line_lbd_detect(640, 480, ctx=ctx)
    lsd.line_length_thres = 15.0  # main_obj.cpp:366
    det = detect_3d_cuboid(ctx)
    det.set_calibration(g["K"])
    det.whether_sample_bbox_height = False; det.nominal_skew_ratio = 2.0; det.max_cuboid_num = 1  # :359-360
    for r in g["kept"][::6]:  # (every sixth frame through the per-frame calls; all of them in the batch test below)
        gray = g["gray_%d" % r]
        lines = lsd.detect_filter_lines(gray)
        ref_lines = oracle.lsd_detect_filter_lines(gray, 15.0)
        assert np.array_equal(np.asarray(lines, np.float32), np.asarray(ref_lines, np.float32)) and len(lines) == g["n_lines"][r]
        got = det.detect_cuboid(gray, g["Twc"][r], g["boxes"][r][None], np.asarray(lines, np.float64))
        c = got[0][0]
        ours = g["ours"][r]
        assert np.allclose([*c["pos"], c["rotY"], *c["scale"], c["normalized_error"]], ours, rtol=1e-5, atol=1e-9)
        row = g["matlab_rows"][r]
        assert np.linalg.norm(np.asarray(c["pos"]) - row[1:4]) < 0.30


def test_object_slam_sequence_chain_as_a_resident_batch(ctx, oracle):
    """All 51 frames resident: cs_lsd_run over the batch, the filtered lines of THAT run handed to the cuboid batch (cs_lsd_read_filter_lines ->
    cs_cuboid_batch_set_lines), one cs_cuboid_batch_run -- every frame's best cuboid equals the oracle's chain on the same frame."""
    g = np.load(os.path.join(GOLD, "object_slam_seq.npz"))
    rows = [int(r) for r in g["kept"]]
    assert len(rows) == len(g["ours"]) == 51
    grays = np.stack([g["gray_%d" % r] for r in rows])
    lsd = line_lbd_detect(640, 480, max_frames=len(rows), ctx=ctx)
    lsd.line_length_thres = 15.0
    lsd.upload(grays)
    lsd.run(False)
    lines = lsd.read_filter_lines(len(rows))
    assert [len(l) for l in lines] == [int(g["n_lines"][r]) for r in rows]
    ref0 = oracle.lsd_detect_filter_lines(grays[0], 15.0)
    assert np.array_equal(np.asarray(lines[0], np.float32), np.asarray(ref0, np.float32))
    det = detect_3d_cuboid(ctx)
    det.set_calibration(g["K"])
    det.whether_sample_bbox_height = False; det.nominal_skew_ratio = 2.0; det.max_cuboid_num = 1
    b = CuboidBatch(ctx, grays, g["K"], np.stack([g["Twc"][r] for r in rows]), [g["boxes"][r][None] for r in rows], [np.zeros((0, 4)) for _ in rows], det.opts())
    b.run()  # (no edges at all: a different result)
    empty = b.read()
    b.set_lines([np.asarray(l, np.float64) for l in lines])
    b.run()
    got = b.read()
    n_same = 0
    for k, r in enumerate(rows):
        ours = g["ours"][r]
        if np.isnan(ours[0]):
            assert len(got[k]) == 0
            continue
        c = got[k][0]
        assert np.allclose([*c["pos"], c["rotY"], *c["scale"], c["normalized_error"]], ours, rtol=1e-5, atol=1e-9), r
        n_same += len(empty[k]) > 0 and np.allclose(empty[k][0]["pos"], c["pos"])
    assert n_same <= len(rows) - 10  # the edge lists matter: without them a third of the frames pick another cuboid
    b.close()
