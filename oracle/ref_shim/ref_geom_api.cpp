// TEST INFRASTRUCTURE (never linked into the product): the geometry of the cuboid proposals as the reference wrote it -- getVanishingPoints,
// VP_support_edge_infos (+ smooth_jump_angles, normalize_to_pi), the predicates of the corner construction (check_inside_box, seg_hit_boundary,
// lineSegmentIntersect) and the way from eight 2D corners to a 3D cuboid (change_2d_corner_to_3d_object with plane_hits_3d, ray_plane_interact,
// get_wall_plane_equation, similarityTransformation, compute3D_BoxCorner, real_to_homo_coord / homo_to_real_coord)
// (detect_3d_cuboid/src/object_3d_util.cpp:14-50, 141-145, 175-252, 380-425, 566-648; matrix_utils.cpp) -- cut out of the reference at build time
// (oracle/_ref/extracted_geom.inc) and compiled against eigdyn.hpp.  tests/test_ref_pins.py compares the oracle's restatements with these.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include "cvshim.hpp"
#include "../oracle.h"
#include "eigdyn/eigdyn.hpp"
namespace Eigen = EigenDyn;

using namespace Eigen;
using namespace std;

namespace cv { enum { NORM_MINMAX = 32 }; inline void normalize(const Mat &, Mat &, double, double, int) {} } // behind the plot flags (off)

namespace { // (internal linkage: ref_extract_api.cpp cuts some of the same helpers against another stand-in)
class cuboid { // detect_3d_cuboid.h:15-36
  public:
    Eigen::Vector3d pos;
    Eigen::Vector3d scale;
    double rotY;
    Eigen::Vector2d box_config_type;
    Eigen::Matrix2Xi box_corners_2d;
    Eigen::Matrix3Xd box_corners_3d_world;
    Eigen::Vector4d rect_detect_2d;
    double edge_distance_error;
    double edge_angle_error;
    double normalized_error;
    double skew_ratio;
    double down_expand_height;
    double camera_roll_delta;
    double camera_pitch_delta;
};
typedef std::vector<cuboid *> ObjectSet;
struct cam_pose_infos { // detect_3d_cuboid.h:39-51
    Eigen::Matrix4d transToWolrd;
    Eigen::Matrix3d Kalib;
    Eigen::Matrix3d rotationToWorld;
    Eigen::Vector3d euler_angle;
    Eigen::Matrix3d invR;
    Eigen::Matrix3d invK;
    Eigen::Matrix<double, 3, 4> projectionMatrix;
    Eigen::Matrix3d KinvR;
    double camera_yaw;
};
class detect_3d_cuboid { // detect_3d_cuboid.h:53-80: the members detect_cuboid and the two setters use, with the header's defaults
  public:
    cam_pose_infos cam_pose;
    cam_pose_infos cam_pose_raw;
    void set_calibration(const Eigen::Matrix3d &Kalib);
    void set_cam_pose(const Eigen::Matrix4d &transToWolrd);
    void detect_cuboid(const cv::Mat &rgb_img, const Eigen::Matrix4d &transToWolrd, const Eigen::MatrixXd &obj_bbox_coors, Eigen::MatrixXd edges, std::vector<ObjectSet> &all_object_cuboids);
    bool whether_plot_detail_images = false;
    bool whether_plot_final_images = false;
    bool whether_save_final_images = false;
    cv::Mat cuboids_2d_img;
    bool print_details = false;
    bool consider_config_1 = true;
    bool consider_config_2 = true;
    bool whether_sample_cam_roll_pitch = false;
    bool whether_sample_bbox_height = false;
    int max_cuboid_num = 1;
    double nominal_skew_ratio = 1;
    double max_cut_skew = 3;
};
namespace ca { struct Profiler { static void tictoc(const char *) {} }; } // tictoc_profiler: timing only
// drawing helpers behind the plot flags (off): declared so that the text compiles, never called
void plot_image_with_edges(const cv::Mat &, cv::Mat &, MatrixXd &, const cv::Scalar &) {}
void plot_image_with_cuboid(cv::Mat &, const cuboid *) {}
// the defaults of the header's declarations (object_3d_util.h)
double box_edge_sum_dists(const cv::Mat &dist_map, const MatrixXd &box_corners_2d, const MatrixXi &edge_pt_ids, bool reweight_edge_distance = false);
template <class T> void quat_to_euler_zyx(const Eigen::Quaternion<T> &q, T &roll, T &pitch, T &yaw);

#include "extracted_geom.inc"
} // namespace

extern "C" {
static Vector2d v2(const double *p) { return Vector2d(p[0], p[1]); }
int ref_check_inside_box(const double *pt, const double *lt, const double *rb) { return check_inside_box(v2(pt), v2(lt), v2(rb)) ? 1 : 0; }
void ref_seg_hit_boundary(const double *ps, const double *pe, const double *seg4, double *out2) {
    const Vector2d h = seg_hit_boundary(v2(ps), v2(pe), Vector4d(seg4[0], seg4[1], seg4[2], seg4[3]));
    out2[0] = h(0); out2[1] = h(1);
}
void ref_line_segment_intersect(const double *p1s, const double *p1e, const double *p2s, const double *p2e, int infinite_line, double *out2) {
    const Vector2d h = lineSegmentIntersect(v2(p1s), v2(p1e), v2(p2s), v2(p2e), infinite_line != 0);
    out2[0] = h(0); out2[1] = h(1);
}
static Matrix3d m3(const double *p) { Matrix3d m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = p[i * 3 + j]; return m; }
static Matrix4d m4(const double *p) { Matrix4d m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = p[i * 4 + j]; return m; }
void ref_vanishing_points(const double *KinvR9, double yaw, double *out6) {
    Vector2d a, b, c;
    getVanishingPoints(m3(KinvR9), yaw, a, b, c);
    out6[0] = a(0); out6[1] = a(1); out6[2] = b(0); out6[3] = b(1); out6[4] = c(0); out6[5] = c(1);
}
// VPs 3 x 2, mids n x 2, angles n, thresholds (degrees) 2 -> 3 x 2 (NaN where no edge supports the vanishing point)
void ref_vp_support_edge_infos(const double *vps6, const double *mids, const double *angles, int n, const double *thre2, double *out6) {
    MatrixXd VPs(3, 2), mid(n, 2);
    VectorXd ang(n);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) VPs(i, j) = vps6[i * 2 + j];
    for (int i = 0; i < n; i++) { mid(i, 0) = mids[2 * i]; mid(i, 1) = mids[2 * i + 1]; ang(i) = angles[i]; }
    const MatrixXd r = VP_support_edge_infos(VPs, mid, ang, v2(thre2));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) out6[i * 2 + j] = r(i, j);
}
void ref_plane_hits_3d(const double *T16, const double *invK9, const double *plane4, const double *pixels /* 2 x n, row-major */, int n, double *out /* 3 x n */) {
    MatrixXd px(2, n);
    for (int i = 0; i < 2; i++) for (int j = 0; j < n; j++) px(i, j) = pixels[i * n + j];
    Matrix3Xd w;
    plane_hits_3d(m4(T16), m3(invK9), Vector4d(plane4[0], plane4[1], plane4[2], plane4[3]), px, w);
    for (int i = 0; i < 3; i++) for (int j = 0; j < n; j++) out[i * n + j] = w(i, j);
}
// corners 2 x 8 (row-major), configs (config id, vp_1_position, yaw) -> pos 3, rotY, scale 3, box_config_type 2, corners 2D (2 x 8 ints), corners 3D (3 x 8)
void ref_change_2d_corner_to_3d_object(const double *corners16, const double *configs3, const double *ground_plane4, const double *T16, const double *invK9,
                                       double *pos3, double *rotY, double *scale3, double *cfg2, int *corners2d16, double *corners3d24) {
    MatrixXd c(2, 8);
    for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) c(i, j) = corners16[i * 8 + j];
    Eigen::Matrix<double, 3, 4> proj(3, 4);
    cuboid o;
    change_2d_corner_to_3d_object(c, Vector3d(configs3[0], configs3[1], configs3[2]), Vector4d(ground_plane4[0], ground_plane4[1], ground_plane4[2], ground_plane4[3]), m4(T16), m3(invK9), proj, o);
    for (int i = 0; i < 3; i++) { pos3[i] = o.pos(i); scale3[i] = o.scale(i); }
    *rotY = o.rotY; cfg2[0] = o.box_config_type(0); cfg2[1] = o.box_config_type(1);
    for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) corners2d16[i * 8 + j] = o.box_corners_2d(i, j);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 8; j++) corners3d24[i * 8 + j] = o.box_corners_3d_world(i, j);
}
// detect_3d_cuboid::detect_cuboid with the arguments of orc_detect_cuboid (oracle.h): gray W x H, K 3 x 3, Twc 4 x 4, boxes nb x 5, lines nl x 4.
// out: room for nb * opts->max_cuboid_num records, counts[nb].
int ref_detect_cuboid(const uint8_t *gray, int W, int H, const double *K9, const double *Twc16, const double *boxes, int nb, const double *lines, int nl, const orc_cuboid_opts *opts,
                      orc_cuboid *out, int *counts) {
    detect_3d_cuboid det;
    det.consider_config_1 = opts->consider_config_1; det.consider_config_2 = opts->consider_config_2;
    det.whether_sample_cam_roll_pitch = opts->whether_sample_cam_roll_pitch; det.whether_sample_bbox_height = opts->whether_sample_bbox_height;
    det.max_cuboid_num = opts->max_cuboid_num; det.nominal_skew_ratio = opts->nominal_skew_ratio; det.max_cut_skew = opts->max_cut_skew;
    det.set_calibration(m3(K9));
    cv::Mat img(H, W, CV_8UC1, (void *)gray);
    MatrixXd bb(nb, 5), ln(nl, 4);
    for (int i = 0; i < nb; i++) for (int j = 0; j < 5; j++) bb(i, j) = boxes[i * 5 + j];
    for (int i = 0; i < nl; i++) for (int j = 0; j < 4; j++) ln(i, j) = lines[i * 4 + j];
    std::vector<ObjectSet> all;
    det.detect_cuboid(img, m4(Twc16), bb, ln, all);
    for (int b = 0; b < nb; b++) {
        counts[b] = (int)all[b].size();
        for (int k = 0; k < counts[b] && k < opts->max_cuboid_num; k++) {
            const cuboid &c = *all[b][k];
            orc_cuboid &o = out[(size_t)b * opts->max_cuboid_num + k];
            for (int i = 0; i < 3; i++) { o.pos[i] = c.pos(i); o.scale[i] = c.scale(i); }
            o.rotY = c.rotY; o.box_config_type[0] = c.box_config_type(0); o.box_config_type[1] = c.box_config_type(1);
            for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) o.box_corners_2d[i * 8 + j] = c.box_corners_2d(i, j);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 8; j++) o.box_corners_3d_world[i * 8 + j] = c.box_corners_3d_world(i, j);
            for (int i = 0; i < 4; i++) o.rect_detect_2d[i] = c.rect_detect_2d(i);
            o.edge_distance_error = c.edge_distance_error; o.edge_angle_error = c.edge_angle_error; o.normalized_error = c.normalized_error; o.skew_ratio = c.skew_ratio;
            o.down_expand_height = c.down_expand_height; o.camera_roll_delta = c.camera_roll_delta; o.camera_pitch_delta = c.camera_pitch_delta;
        }
    }
    return 0;
}
}
