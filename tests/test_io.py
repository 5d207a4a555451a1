"""CPU tests of the text feeders (cube_slam_amd/io.py and the C++ twin cube_slam_amd/host/txt_io.hpp)."""
import os
import subprocess

import numpy as np
import pytest

from cube_slam_amd import io as cio

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_number_matrix_semantics(tmp_path):
    p = tmp_path / "m.txt"
    p.write_text("1 2 3\n\n4.5\t-6e-1 7 trailing 9\n   \n8 9 10 11\n")
    m = cio.read_all_number_txt(str(p), cols=4)
    # blank line skipped; parsing stops at the first non-number; a whitespace-only line is a row of zeros (the reference counts it)
    assert np.array_equal(m, [[1, 2, 3, 0], [4.5, -0.6, 7, 0], [0, 0, 0, 0], [8, 9, 10, 11]])
    assert cio.read_all_number_txt(str(p)).shape == (4, 10)
    # more numbers than columns: the reference writes past the row (undefined); kept: the first `cols` numbers
    assert np.array_equal(cio.read_all_number_txt(str(p), cols=3), [[1, 2, 3], [4.5, -0.6, 7], [0, 0, 0], [8, 9, 10]])
    with pytest.raises(FileNotFoundError):
        cio.read_all_number_txt(str(tmp_path / "missing.txt"))


def test_detection_files(tmp_path):
    a = tmp_path / "a.txt"
    a.write_text("chair 10 20 30 40 0.9\nmonitor 1 2 3 4 0.5\n")
    m, names = cio.read_obj_detection_txt(str(a), cols=5)
    assert names == ["chair", "monitor"] and np.array_equal(m[0], [10, 20, 30, 40, 0.9])
    b = tmp_path / "b.txt"
    b.write_text("10 20 30 40 0.9 chair\n1 2 3 4 0.5 monitor\n")
    m2, names2 = cio.read_obj_detection2_txt(str(b), cols=5)
    assert names2 == names and np.array_equal(m2, m)
    y = tmp_path / "y.txt"
    y.write_text("175\t24\t385\t373\t0.42\n")
    assert np.array_equal(cio.read_yolo_boxes(str(y)), [[174, 23, 385, 373, 0.42]])


def test_edge_round_trip_and_poses(tmp_path):
    lines = np.array([[467.435, 0.526885, 453.015, 285.683], [1, 2, 3, 4]], np.float32)
    p = tmp_path / "e.txt"
    cio.write_edge_txt(str(p), lines)
    assert p.read_text().splitlines()[0] == "467.435\t0.526885\t453.015\t285.683"
    assert np.allclose(cio.read_edge_txt(str(p)), lines, rtol=1e-6)
    T = cio.pose_row_to_Twc([1341841278.8427, 0.0, 0.0, 1.1019, -0.9089, 0.0002, 0.0004, 0.4171])
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12) and np.allclose(T[:3, 3], [0, 0, 1.1019])
    assert T[2, 2] < -0.6  # camera looks down (pitch about the x axis)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_data_files():
    e = cio.read_edge_txt(os.path.join(REF, "detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt"))
    assert e.shape == (271, 4) and np.isclose(e[0, 0], 467.435)
    b = cio.read_yolo_boxes(os.path.join(REF, "object_slam/data/filter_2d_obj_txts/0000_yolo2_0.15.txt"))
    assert b.shape[1] == 5 and b[0, 0] == 174
    poses = cio.read_cam_poses(os.path.join(REF, "object_slam/data/pop_cam_poses_saved.txt"))
    assert poses.shape[1] == 8 and len(poses) >= 50
    cub = cio.read_all_number_txt(os.path.join(REF, "object_slam/data/detect_cuboids_saved.txt"), cols=9)
    assert cub.shape == (51, 9)


def test_cpp_twin_matches_python(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "cube_slam_amd/host/txt_io.hpp"\n#include <cstdio>\nint main(int argc, char** argv) { cubeslam::NumMat m; std::vector<std::string> n;\n'
                   ' if (!cubeslam::read_obj_detection_txt(argv[1], m, n, 5)) return 1; for (int r = 0; r < m.rows; r++) { printf("%s", n[r].c_str()); for (int c = 0; c < m.cols; c++) printf(" %.17g", m(r, c)); printf("\\n"); }\n'
                   ' cubeslam::NumMat k; if (cubeslam::read_all_number_txt("/nonexistent/x.txt", k)) return 2; float l[4] = {467.435f, 0.526885f, 453.015f, 285.683f}; cubeslam::write_edge_txt(argv[2], l, 1); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, str(src), "-o", str(exe)])
    a = tmp_path / "a.txt"
    a.write_text("chair 10 20 30 40 0.9\n\nmonitor 1 2.5 3\n")
    out = subprocess.check_output([str(exe), str(a), str(tmp_path / "e.txt")]).decode().splitlines()
    m, names = cio.read_obj_detection_txt(str(a), cols=5)
    for line, nm, row in zip(out, names, m):
        tok = line.split()
        assert tok[0] == nm and np.array_equal(np.array(tok[1:], float), row)
    assert (tmp_path / "e.txt").read_text() == "467.435\t0.526885\t453.015\t285.683\n"


def test_boxes_matrix_with_four_or_more_columns():
    """detect_cuboid reads columns 0-3 of obj_bbox_coors (box_proposal_detail.cpp:102-108): an (n, 4) matrix is as good as the (n, 5) rows of the txt files."""
    from cube_slam_amd.cuboid import _boxes5
    b4 = np.array([[10, 20, 30, 40], [1, 2, 3, 4]], float)
    assert np.array_equal(_boxes5(b4), [[10, 20, 30, 40, 0], [1, 2, 3, 4, 0]])
    assert np.array_equal(_boxes5(np.hstack([b4, [[0.9, 7], [0.5, 8]]])), [[10, 20, 30, 40, 0.9], [1, 2, 3, 4, 0.5]])
    assert np.array_equal(_boxes5([10, 20, 30, 40, 0.9]), [[10, 20, 30, 40, 0.9]])
    assert _boxes5(np.zeros((0, 5))).shape == (0, 5)
    with pytest.raises(ValueError):
        _boxes5([1, 2, 3, 4])
    with pytest.raises(ValueError):
        _boxes5(np.zeros((2, 3)))
