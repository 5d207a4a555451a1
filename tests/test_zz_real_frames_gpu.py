"""GPU parity on real frames: four images of the reference's bundled TUM-cabinet sequence (tests/golden/object_slam_seq.npz, see
tests/golden/make_golden.py) through the HIP line detector and detect_3d_cuboid, against the oracle results stored in the fixture (which
tests/test_cuboid_oracle.py ties to the author's saved detections).  Runs last on purpose (file name)."""
import os

import numpy as np
import pytest

from cube_slam_amd.cuboid import detect_3d_cuboid
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
    for r in g["kept"]:
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
