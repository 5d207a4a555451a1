// ref_extract_api.cpp -- TEST INFRASTRUCTURE.  C entry points over reference functions whose files cannot be compiled whole here (Eigen,
// g2o and the SLAM classes are absent): their definitions are cut out of /root/reference at build time (extract_ref.py -> _ref/extracted.inc)
// and compiled against eigshim.hpp / cvshim.hpp.  merge_break_lines, box_edge_sum_dists, box_edge_alignment_angle_error,
// fuse_normalize_scores_v2 (detect_3d_cuboid/src/object_3d_util.cpp:300-565) with atan2_vector, fast_RemoveRow, sort_indexes,
// normalize_to_pi; ORBmatcher::DescriptorDistance / ComputeThreeMaxima (orb_object_slam/src/ORBmatcher.cc:1860-1921).
#include <algorithm>
#include <cmath>
#include <iostream>
#include <numeric>
#include <vector>

#include "cvshim.hpp"
#include "eigshim.hpp"

using namespace Eigen;
using std::vector;
template <class T> T normalize_to_pi(T angle);
void fast_RemoveRow(MatrixXd &matrix, int rowToRemove, int &total_line_number);
void sort_indexes(const Eigen::VectorXd &vec, std::vector<int> &idx, int top_k);
namespace ORB_SLAM2 { // the two members of ORBmatcher (include/ORBmatcher.h:36-95) that the extracted definitions belong to
class ORBmatcher {
public:
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);
};
} // namespace ORB_SLAM2
#include "extracted_cuboid.inc"
namespace ORB_SLAM2 {
#include "extracted_orb.inc"
}

extern "C" {
int ref_merge_break_lines(const double *lines, int n, double dist_thre, double angle_thre_deg, double len_thre, double *out) {
    MatrixXd in(n, 4), res;
    for (int i = 0; i < n * 4; i++) in.d[i] = lines[i];
    merge_break_lines(in, res, dist_thre, angle_thre_deg, len_thre);
    for (int i = 0; i < res.rows() * 4; i++) out[i] = res.d[i];
    return res.rows();
}
static void cfg_tables(int config_id, MatrixXi &vis, MatrixXi &vpe) { // box_proposal_detail.cpp:430-447 (0-based there as well)
    const int v1[9][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {3, 7}, {4, 7}, {4, 5}}, v2[7][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {4, 5}};
    const int e1[3][4] = {{0, 1, 7, 4}, {3, 0, 4, 5}, {3, 7, 1, 5}}, e2[3][4] = {{0, 1, 2, 3}, {3, 0, 4, 5}, {2, 4, 1, 5}};
    const int ne = config_id == 1 ? 9 : 7;
    vis.resize(ne, 2); vpe.resize(3, 4);
    for (int i = 0; i < ne; i++) for (int j = 0; j < 2; j++) vis(i, j) = config_id == 1 ? v1[i][j] : v2[i][j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) vpe(i, j) = config_id == 1 ? e1[i][j] : e2[i][j];
}
double ref_box_edge_sum_dists(const float *dist_map, int w, int h, const double *corners_shift /*2x8 row-major*/, int config_id) {
    cv::Mat dm(h, w, CV_32FC1, (void *)dist_map);
    MatrixXd c(2, 8); for (int i = 0; i < 16; i++) c.d[i] = corners_shift[i];
    MatrixXi vis, vpe; cfg_tables(config_id, vis, vpe);
    return box_edge_sum_dists(dm, c, vis, config_id == 2); // reweight_edge_distance for configuration 2, box_proposal_detail.cpp:448-452
}
double ref_box_edge_angle_error(const double *vp_bound_angles /*3x2*/, const double *corners /*2x8*/, int config_id) {
    MatrixXd a(3, 2), c(2, 8);
    for (int i = 0; i < 6; i++) a.d[i] = vp_bound_angles[i];
    for (int i = 0; i < 16; i++) c.d[i] = corners[i];
    MatrixXi vis, vpe; cfg_tables(config_id, vis, vpe);
    return box_edge_alignment_angle_error(a, vpe, c);
}
int ref_fuse_normalize_scores(const double *dist_err, const double *angle_err, int n, double weight_vp_angle, int whether_normalize, int *keep, double *scores) {
    VectorXd d(n), a(n), comb;
    for (int i = 0; i < n; i++) { d(i) = dist_err[i]; a(i) = angle_err[i]; }
    std::vector<int> k;
    fuse_normalize_scores_v2(d, a, comb, k, weight_vp_angle, whether_normalize != 0);
    for (size_t i = 0; i < k.size(); i++) { keep[i] = k[i]; scores[i] = comb(i); }
    return (int)k.size();
}
int ref_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    cv::Mat ma(1, 32, CV_8UC1, (void *)a), mb(1, 32, CV_8UC1, (void *)b);
    return ORB_SLAM2::ORBmatcher::DescriptorDistance(ma, mb);
}
void ref_three_maxima(const int *counts, int L, int *ind) {
    std::vector<std::vector<int>> histo(L);
    for (int i = 0; i < L; i++) histo[i].resize(counts[i]);
    ind[0] = ind[1] = ind[2] = -1;
    ORB_SLAM2::ORBmatcher m;
    m.ComputeThreeMaxima(histo.data(), L, ind[0], ind[1], ind[2]);
}
}
