"""Text feeders of the reference: the number-matrix readers every node uses, the 2-D detection / edge /
pose / offline-cuboid files that sit on either side of the hot path.  Host-only plumbing; the C++ twin is
cube_slam_amd/host/txt_io.hpp.

read_all_number_txt / read_obj_detection_txt / read_obj_detection2_txt follow detect_3d_cuboid/src/matrix_utils.cpp:195-314:
blank lines are skipped, a line is parsed number by number until the first token that is not a number, the column count is the
caller's (10 when the caller passes an empty matrix, :209-210), missing trailing columns stay 0 (the reference leaves Eigen's
uninitialised storage there; zero is the documented intent ":195 if more cols given, will be zero").
"""
import numpy as np


def _numbers(tokens):
    out = []
    for t in tokens:
        try:
            out.append(float(t))
        except ValueError:
            break
    return out


def read_all_number_txt(path, cols=10, dtype=np.float64):
    """-> (rows, cols) matrix; raises FileNotFoundError where the reference prints an error and returns false."""
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip("\n"):
                continue  # `if (!line.empty())`; a line of blanks parses to zero numbers and still makes a row, like the reference
            v = _numbers(line.split())
            # more numbers than columns: the reference keeps writing read_number_mat(row, colu++) without a bound (matrix_utils.cpp:217-221), which is
            # outside the matrix (undefined).  Kept: the first `cols` numbers.
            rows.append((v + [0.0] * cols)[:cols])
    return np.array(rows, dtype).reshape(-1, cols)


def read_obj_detection_txt(path, cols=10):
    """class name first, then numbers (matrix_utils.cpp:235-270) -> (matrix, [class names])."""
    rows, names = [], []
    with open(path) as f:
        for line in f:
            if not line.strip("\n"):
                continue
            tok = line.split()
            names.append(tok[0] if tok else "")
            v = _numbers(tok[1:])
            if len(v) > cols:
                raise ValueError("%s: too many numbers for %d columns" % (path, cols))
            rows.append(v + [0.0] * (cols - len(v)))
    return np.array(rows, np.float64).reshape(-1, cols), names


def read_obj_detection2_txt(path, cols):
    """`cols` numbers first, class name after them (matrix_utils.cpp:272-313) -> (matrix, [class names])."""
    rows, names = [], []
    with open(path) as f:
        for line in f:
            if not line.strip("\n"):
                continue
            tok = line.split()
            v = _numbers(tok[:cols])
            names.append(tok[len(v)] if len(tok) > len(v) else "")
            rows.append(v + [0.0] * (cols - len(v)))
    return np.array(rows, np.float64).reshape(-1, cols), names


def read_yolo_boxes(path):
    """object_slam/data/filter_2d_obj_txts/%04d_yolo2_0.15.txt: rows `x y w h prob` (1-based pixels in the MATLAB pipeline;
    main_obj.cpp:403-409 reads 5 columns and subtracts 1 from x, y) -> (n, 5) with 0-based x, y."""
    m = read_all_number_txt(path, cols=5)
    m[:, 0] -= 1
    m[:, 1] -= 1
    return m


def read_edge_txt(path):
    """LSD dumps `x1 y1 x2 y2` per row (line_lbd/src/detect_lines.cpp:85-96, detect_3d_cuboid/data/edge_detection/LSD)."""
    return read_all_number_txt(path, cols=4)


def write_edge_txt(path, lines):
    """Same format the reference's detect_lines node writes (tab-separated, default ostream precision 6)."""
    with open(path, "w") as f:
        for x1, y1, x2, y2 in np.asarray(lines, np.float64).reshape(-1, 4):
            f.write("%s\t%s\t%s\t%s\n" % tuple("%.6g" % v for v in (x1, y1, x2, y2)))


def read_cam_poses(path):
    """truth_cam_poses.txt / pop_cam_poses_saved.txt: `t x y z qx qy qz qw` per row (main_obj.cpp:382-391) -> (n, 8)."""
    return read_all_number_txt(path, cols=8)


def pose_row_to_Twc(row):
    """(x y z qx qy qz qw) -> 4x4, the conversion main_obj.cpp:440-447 does with Eigen::Quaterniond(qw, qx, qy, qz)."""
    x, y, z, qx, qy, qz, qw = [float(v) for v in row[-7:]]
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [x, y, z]
    return T


def read_offline_cuboids(path, use_truth_trackid=False):
    """pred_3d_obj_matched_txt/%04d_3d_cuboids.txt rows: cuboid centre (3), yaw, scale (3), 2-D box x1 y1 w h, prob
    [, track id] (orb_object_slam/src/Tracking_util.cc:25-69) -> (n, 12 or 13)."""
    return read_all_number_txt(path, cols=13 if use_truth_trackid else 12)
