/*
 * oracle/cuboid_oracle.cpp -- CPU oracle for the detect_3d_cuboid path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Restated from
 * /root/reference/detect_3d_cuboid/src/{box_proposal_detail,object_3d_util,matrix_utils}.cpp
 * plus the OpenCV / Eigen semantics those files reach (documented per function).  PINNED: tests/test_ref_pins.py runs
 * detect_3d_cuboid::detect_cuboid and every function it calls -- the reference's own text, cut out at build time (oracle/_ref) -- next to
 * orc_detect_cuboid in five settings: identical cuboid records.  What stays restated are the library primitives underneath (Eigen's
 * inverses and rotation -> quaternion, OpenCV's Canny / distanceTransform / cvtColor), which are not in the reference tree.
 * Build WITHOUT floating-point contraction (-ffp-contract=off): the reference is built
 * Release without -march=native (detect_3d_cuboid/CMakeLists.txt:2,56-57), i.e. no FMA, and
 * int() truncation of sample points that lie exactly on integer box edges depends on it.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// ----------------------------------------------------------------------------- tiny linear algebra
struct M3 { double m[3][3]; };
struct M4 { double m[4][4]; };
struct V2 { double x, y; };
struct V3 { double v[3]; };
struct V4 { double v[4]; };

static inline M3 mul(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = a.m[i][0] * b.m[0][j];
            s = s + a.m[i][1] * b.m[1][j];
            s = s + a.m[i][2] * b.m[2][j];
            r.m[i][j] = s;
        }
    return r;
}
static inline V3 mul(const M3 &a, const V3 &b) {
    V3 r;
    for (int i = 0; i < 3; i++) {
        double s = a.m[i][0] * b.v[0];
        s = s + a.m[i][1] * b.v[1];
        s = s + a.m[i][2] * b.v[2];
        r.v[i] = s;
    }
    return r;
}
// Eigen fixed-size 3x3 inverse: cofactors * (1/det)  (Eigen/src/LU/InverseImpl.h, size-3 path)
static inline double cof3(const M3 &a, int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
static inline M3 inv3(const M3 &a) {
    double c00 = cof3(a, 0, 0), c10 = cof3(a, 1, 0), c20 = cof3(a, 2, 0);
    double det = (c00 * a.m[0][0] + c10 * a.m[1][0]) + c20 * a.m[2][0];
    double invdet = 1.0 / det;
    M3 r;
    r.m[0][0] = c00 * invdet; r.m[0][1] = c10 * invdet; r.m[0][2] = c20 * invdet;
    r.m[1][0] = cof3(a, 0, 1) * invdet; r.m[1][1] = cof3(a, 1, 1) * invdet; r.m[1][2] = cof3(a, 2, 1) * invdet;
    r.m[2][0] = cof3(a, 0, 2) * invdet; r.m[2][1] = cof3(a, 1, 2) * invdet; r.m[2][2] = cof3(a, 2, 2) * invdet;
    return r;
}
// general 4x4 inverse (Gauss-Jordan with partial pivoting); only feeds cam_pose.projectionMatrix,
// which no output of detect_cuboid depends on (object_3d_util.cpp:610-648 never reads it).
static inline M4 inv4(const M4 &a) {
    double w[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { w[i][j] = a.m[i][j]; w[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c) for (int j = 0; j < 8; j++) std::swap(w[p][j], w[c][j]);
        double d = 1.0 / w[c][c];
        for (int j = 0; j < 8; j++) w[c][j] *= d;
        for (int r = 0; r < 4; r++) if (r != c) {
            double f = w[r][c];
            for (int j = 0; j < 8; j++) w[r][j] -= f * w[c][j];
        }
    }
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = w[i][4 + j];
    return r;
}

// Eigen::Quaterniond(Matrix3d)  (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>)
static inline void rot_to_quat(const M3 &m, double &qw, double &qx, double &qy, double &qz) {
    double t = m.m[0][0] + m.m[1][1] + m.m[2][2];
    double q[3];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        qw = 0.5 * t;
        t = 0.5 / t;
        qx = (m.m[2][1] - m.m[1][2]) * t;
        qy = (m.m[0][2] - m.m[2][0]) * t;
        qz = (m.m[1][0] - m.m[0][1]) * t;
    } else {
        int i = 0;
        if (m.m[1][1] > m.m[0][0]) i = 1;
        if (m.m[2][2] > m.m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m.m[i][i] - m.m[j][j] - m.m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        qw = (m.m[k][j] - m.m[j][k]) * t;
        q[j] = (m.m[j][i] + m.m[i][j]) * t;
        q[k] = (m.m[k][i] + m.m[i][k]) * t;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
}
// matrix_utils.cpp:35-46
static inline void quat_to_euler_zyx(double qw, double qx, double qy, double qz, double &roll, double &pitch, double &yaw) {
    roll = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
    pitch = std::asin(2 * (qw * qy - qz * qx));
    yaw = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}
// matrix_utils.cpp:74-89
static inline M3 euler_zyx_to_rot(double roll, double pitch, double yaw) {
    double cp = std::cos(pitch), sp = std::sin(pitch), sr = std::sin(roll), cr = std::cos(roll), sy = std::sin(yaw), cy = std::cos(yaw);
    M3 R;
    R.m[0][0] = cp * cy; R.m[0][1] = (sr * sp * cy) - (cr * sy); R.m[0][2] = (cr * sp * cy) + (sr * sy);
    R.m[1][0] = cp * sy; R.m[1][1] = (sr * sp * sy) + (cr * cy); R.m[1][2] = (cr * sp * sy) - (sr * cy);
    R.m[2][0] = -sp;     R.m[2][1] = sr * cp;                    R.m[2][2] = cr * cp;
    return R;
}
// matrix_utils.cpp:326-335
static inline double normalize_to_pi(double a) {
    if (a > M_PI / 2) return a - M_PI;
    else if (a < -M_PI / 2) return a + M_PI;
    else return a;
}
// matrix_utils.cpp:349-363
template <class T> static void linespace(T starting, T ending, T step, std::vector<T> &res) {
    while (starting <= ending) {
        res.push_back(starting);
        starting += step;
        if (res.size() > 1000) break;
    }
}

struct CamPose { // detect_3d_cuboid.h:39-51
    M4 transToWolrd; M3 Kalib, rotationToWorld, invR, invK, KinvR;
    double euler[3]; double proj[3][4]; double camera_yaw;
};
// box_proposal_detail.cpp:42-54
static void set_cam_pose(CamPose &c, const M4 &T) {
    c.transToWolrd = T;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.rotationToWorld.m[i][j] = T.m[i][j];
    double qw, qx, qy, qz;
    rot_to_quat(c.rotationToWorld, qw, qx, qy, qz);
    quat_to_euler_zyx(qw, qx, qy, qz, c.euler[0], c.euler[1], c.euler[2]);
    c.invR = inv3(c.rotationToWorld);
    M4 Ti = inv4(T);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = c.Kalib.m[i][0] * Ti.m[0][j];
            s = s + c.Kalib.m[i][1] * Ti.m[1][j];
            s = s + c.Kalib.m[i][2] * Ti.m[2][j];
            c.proj[i][j] = s;
        }
    c.KinvR = mul(c.Kalib, c.invR);
    c.camera_yaw = c.euler[2];
}
static inline V4 ground_plane_sensor_of(const M4 &T) { // box_proposal_detail.cpp:99-100: T^T * (0,0,1,0)
    V4 g;
    for (int i = 0; i < 4; i++) {
        double s = T.m[0][i] * 0.0;
        s = s + T.m[1][i] * 0.0;
        s = s + T.m[2][i] * 1.0;
        s = s + T.m[3][i] * 0.0;
        g.v[i] = s;
    }
    return g;
}

// ----------------------------------------------------------------------------- object_3d_util.cpp geometry
static inline bool check_inside_box(V2 pt, V2 lt, V2 rb) { // :141-144
    return lt.x <= pt.x && pt.x <= rb.x && lt.y <= pt.y && pt.y <= rb.y;
}
static inline V2 seg_hit_boundary(V2 ps, V2 pe, double bx0, double by0, double bx1, double by1) { // :194-230
    V2 direc{pe.x - ps.x, pe.y - ps.y};
    V2 hit{-1, -1};
    if (by0 == by1) {
        double lambd = (by0 - ps.y) / direc.y;
        if (lambd >= 0) {
            V2 t{ps.x + lambd * direc.x, ps.y + lambd * direc.y};
            if ((bx0 <= t.x) && (t.x <= bx1)) { hit = t; hit.y = by0; }
        }
    }
    if (bx0 == bx1) {
        double lambd = (bx0 - ps.x) / direc.x;
        if (lambd >= 0) {
            V2 t{ps.x + lambd * direc.x, ps.y + lambd * direc.y};
            if ((by0 <= t.y) && (t.y <= by1)) { hit = t; hit.x = bx0; }
        }
    }
    return hit;
}
static inline V2 line_intersect_inf(V2 p1s, V2 p1e, V2 p2s, V2 p2e) { // :233-252 with infinite_line=true
    double X2_X1 = p1e.x - p1s.x, Y2_Y1 = p1e.y - p1s.y;
    double X4_X3 = p2e.x - p2s.x, Y4_Y3 = p2e.y - p2s.y;
    double X1_X3 = p1s.x - p2s.x, Y1_Y3 = p1s.y - p2s.y;
    double u_a = (X4_X3 * Y1_Y3 - Y4_Y3 * X1_X3) / (Y4_Y3 * X2_X1 - X4_X3 * Y2_Y1);
    double INT_X = p1s.x + X2_X1 * u_a;
    double INT_Y = p1s.y + Y2_Y1 * u_a;
    return V2{INT_X * 1.0, INT_Y * 1.0};
}
static inline double dist2(V2 a, V2 b) { double dx = a.x - b.x, dy = a.y - b.y; return std::sqrt(dx * dx + dy * dy); }

// :602-607
static void getVanishingPoints(const M3 &KinvR, double yaw, V2 &vp1, V2 &vp2, V2 &vp3) {
    V3 a = mul(KinvR, V3{{std::cos(yaw), std::sin(yaw), 0}});
    V3 b = mul(KinvR, V3{{-std::sin(yaw), std::cos(yaw), 0}});
    V3 c = mul(KinvR, V3{{0, 0, 1}});
    vp1 = V2{a.v[0] / a.v[2], a.v[1] / a.v[2]};
    vp2 = V2{b.v[0] / b.v[2], b.v[1] / b.v[2]};
    vp3 = V2{c.v[0] / c.v[2], c.v[1] / c.v[2]};
}

// :380-425 (+ smooth_jump_angles :175-189).  out: 3x2, NaN where not found
static void VP_support_edge_infos(const V2 vps[3], const std::vector<V2> &mid, const std::vector<double> &ang,
                                  double thre12_deg, double thre3_deg, double out[6]) {
    for (int i = 0; i < 6; i++) out[i] = std::nan("");
    int n = (int)ang.size();
    if (n == 0) return;
    std::vector<double> raw(n);
    std::vector<int> ids;
    for (int vp_id = 0; vp_id < 3; vp_id++) {
        double thre = (vp_id != 2 ? thre12_deg : thre3_deg) / 180.0 * M_PI;
        ids.clear();
        for (int e = 0; e < n; e++) {
            double a_raw = std::atan2(mid[e].y - vps[vp_id].y, mid[e].x - vps[vp_id].x);
            double a_norm = normalize_to_pi(a_raw);
            double d = std::abs(ang[e] - a_norm);
            d = std::min(d, M_PI - d);
            if (d < thre) { raw[ids.size()] = a_raw; ids.push_back(e); }
        }
        if (!ids.empty()) {
            int m = (int)ids.size();
            std::vector<double> sh(raw.begin(), raw.begin() + m);
            double base = raw[0];
            for (int i = 0; i < m; i++) {
                if ((raw[i] - base) < -M_PI) sh[i] = raw[i] + 2 * M_PI;
                else if ((raw[i] - base) > M_PI) sh[i] = raw[i] - 2 * M_PI;
            }
            // Eigen maxCoeff/minCoeff(&idx): first occurrence of the extreme value
            int lo = 0, hi = 0;
            for (int i = 1; i < m; i++) { if (sh[i] > sh[lo]) lo = i; if (sh[i] < sh[hi]) hi = i; }
            int low_id = lo, top_id = hi;
            if (vp_id > 0) std::swap(low_id, top_id);
            out[vp_id * 2 + 0] = ang[ids[low_id]];
            out[vp_id * 2 + 1] = ang[ids[top_id]];
        }
    }
}

static const int VIS_EDGES_1[9][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {3, 7}, {4, 7}, {4, 5}}; // box_proposal_detail.cpp:432
static const int VIS_EDGES_2[7][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {4, 5}};                 // :442
static const int VP_EDGES_1[3][4] = {{0, 1, 7, 4}, {3, 0, 4, 5}, {3, 7, 1, 5}};                               // :434
static const int VP_EDGES_2[3][4] = {{0, 1, 2, 3}, {3, 0, 4, 5}, {2, 4, 1, 5}};                               // :444

// :427-453.  corners 2x8 row-major, shifted to ROI origin.  dist_map w x h, continuous (cv::Mat::at has no
// bounds check in Release; corners may sit on x==w or y==h): flat index y*w+x clamped to the buffer.
static double box_edge_sum_dists(const float *dist_map, int w, int h, const double *c, int config_id) {
    const int(*edges)[2] = config_id == 1 ? VIS_EDGES_1 : VIS_EDGES_2;
    int ne = config_id == 1 ? 9 : 7;
    bool reweight = (config_id != 1); // :437 default false, :447 passes reweight_edge_distance=true
    float sum_dist = 0;
    long last = (long)w * h - 1;
    for (int e = 0; e < ne; e++) {
        double x1 = c[edges[e][0]], y1 = c[8 + edges[e][0]];
        double x2 = c[edges[e][1]], y2 = c[8 + edges[e][1]];
        for (double s = 0; s < 11; s++) {
            double px = s / 10.0 * x1 + (1 - s / 10.0) * x2;
            double py = s / 10.0 * y1 + (1 - s / 10.0) * y2;
            long idx = (long)int(py) * w + int(px);
            if (idx < 0) idx = 0;
            if (idx > last) idx = last;
            float dist1 = dist_map[idx];
            if (reweight) {
                if ((4 <= e) && (e <= 5)) dist1 = dist1 * 3.0 / 2.0;
                if (6 == e) dist1 = dist1 * 2.0;
            }
            sum_dist = sum_dist + dist1;
        }
    }
    return double(sum_dist);
}

// :455-492
static double box_edge_alignment_angle_error(const double *vpa /*3x2*/, const double *c, int config_id) {
    const int(*ids)[4] = config_id == 1 ? VP_EDGES_1 : VP_EDGES_2;
    double total = 0;
    double not_found_penalty = 30.0 / 180.0 * M_PI * 2;
    for (int vp = 0; vp < 3; vp++) {
        double valid[2]; int nv = 0;
        for (int i = 0; i < 2; i++) if (!std::isnan(vpa[vp * 2 + i])) valid[nv++] = vpa[vp * 2 + i];
        if (nv > 0) {
            for (int ee = 0; ee < 2; ee++) {
                int a = ids[vp][2 * ee], b = ids[vp][2 * ee + 1];
                double ang = normalize_to_pi(std::atan2(c[8 + b] - c[8 + a], c[b] - c[a]));
                double best = 100;
                for (int i = 0; i < nv; i++) {
                    double t = std::abs(ang - valid[i]);
                    t = std::min(t, M_PI - t);
                    if (t < best) best = t;
                }
                total = total + best;
            }
        } else
            total = total + not_found_penalty;
    }
    return total;
}

// matrix_utils.cpp:316-319 pins std::partial_sort's unspecified tie order to (value, index) -- DESIGN.md D3.
static void sort_indexes_topk(const double *v, std::vector<int> &idx, int top_k) {
    std::stable_sort(idx.begin(), idx.end(), [v](int a, int b) { return v[a] < v[b]; });
    (void)top_k;
}

// object_3d_util.cpp:495-565
static void fuse_normalize_scores_v2(const double *dist_error, const double *angle_error, int n, std::vector<double> &combined,
                                     std::vector<int> &keep, double weight_vp_angle, bool whether_normalize) {
    keep.clear();
    if (n > 4) {
        int breaking_num = (int)std::round(float(n) / 3.0 * 2.0);
        std::vector<int> dist_sorted(n);
        std::iota(dist_sorted.begin(), dist_sorted.end(), 0);
        std::vector<int> angle_sorted = dist_sorted;
        sort_indexes_topk(dist_error, dist_sorted, breaking_num);
        sort_indexes_topk(angle_error, angle_sorted, breaking_num);
        std::vector<int> dist_keep(dist_sorted.begin(), dist_sorted.begin() + breaking_num - 1);
        if (angle_error[angle_sorted[breaking_num - 1]] > angle_error[angle_sorted[breaking_num - 2]]) {
            std::vector<int> angle_keep(angle_sorted.begin(), angle_sorted.begin() + breaking_num - 1);
            std::sort(dist_keep.begin(), dist_keep.end());
            std::sort(angle_keep.begin(), angle_keep.end());
            std::set_intersection(dist_keep.begin(), dist_keep.end(), angle_keep.begin(), angle_keep.end(), std::back_inserter(keep));
        } else
            keep = dist_keep;
    } else {
        keep.resize(n);
        std::iota(keep.begin(), keep.end(), 0);
    }
    int m = (int)keep.size();
    double min_d = 1e6, max_d = -1, min_a = 1e6, max_a = -1;
    std::vector<double> dk(m), ak(m);
    for (int i = 0; i < m; i++) {
        double td = dist_error[keep[i]], ta = angle_error[keep[i]];
        min_d = std::min(min_d, td); max_d = std::max(max_d, td);
        min_a = std::min(min_a, ta); max_a = std::max(max_a, ta);
        dk[i] = td; ak[i] = ta;
    }
    combined.resize(m);
    if (whether_normalize && (m > 1)) {
        for (int i = 0; i < m; i++) combined[i] = (dk[i] - min_d) / (max_d - min_d);
        if ((max_a - min_a) > 0)
            for (int i = 0; i < m; i++) ak[i] = (ak[i] - min_a) / (max_a - min_a);
        for (int i = 0; i < m; i++) combined[i] = (combined[i] + weight_vp_angle * ak[i]) / (1 + weight_vp_angle);
    } else
        for (int i = 0; i < m; i++) combined[i] = (dk[i] + weight_vp_angle * ak[i]) / (1 + weight_vp_angle);
}

// :574-585 plane_hits_3d for one pixel; T 4x4, plane in sensor frame
static inline V3 plane_hit_3d(const M4 &T, const M3 &invK, const V4 &plane, double px, double py) {
    V3 ray = mul(invK, V3{{px, py, 1.0}});
    double den = plane.v[0] * ray.v[0];
    den = den + plane.v[1] * ray.v[1];
    den = den + plane.v[2] * ray.v[2];
    double frac = -plane.v[3] / den;
    double s[4] = {frac * ray.v[0], frac * ray.v[1], frac * ray.v[2], 1.0};
    double w[4];
    for (int i = 0; i < 4; i++) {
        double a = T.m[i][0] * s[0];
        a = a + T.m[i][1] * s[1];
        a = a + T.m[i][2] * s[2];
        a = a + T.m[i][3] * s[3];
        w[i] = a;
    }
    return V3{{w[0] / w[3], w[1] / w[3], w[2] / w[3]}};
}

// :610-648
static void change_2d_corner_to_3d_object(const double *c /*2x8*/, double config_id, double vp_1_position, double yaw_esti,
                                          const V4 &ground_plane_sensor, const M4 &T, const M3 &invK, orc_cuboid &o) {
    V3 g[4];
    for (int i = 0; i < 4; i++) g[i] = plane_hit_3d(T, invK, ground_plane_sensor, c[4 + i], c[8 + 4 + i]);
    auto nrm = [](const V3 &a, const V3 &b) {
        double dx = a.v[0] - b.v[0], dy = a.v[1] - b.v[1], dz = a.v[2] - b.v[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    double length_half = nrm(g[0], g[3]) / 2;
    double width_half = nrm(g[0], g[1]) / 2;
    // get_wall_plane_equation :587-600
    double d[3] = {g[0].v[0] - g[1].v[0], g[0].v[1] - g[1].v[1], g[0].v[2] - g[1].v[2]};
    double nw[3] = {d[1] * 1.0 - d[2] * 0.0, d[2] * 0.0 - d[0] * 1.0, d[0] * 0.0 - d[1] * 0.0};
    double nn = std::sqrt(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
    nw[0] /= nn; nw[1] /= nn; nw[2] /= nn;
    double dist = -((nw[0] * g[0].v[0] + nw[1] * g[0].v[1]) + nw[2] * g[0].v[2]);
    V4 pw{{nw[0], nw[1], nw[2], dist}};
    if (dist < 0) for (int i = 0; i < 4; i++) pw.v[i] = -pw.v[i];
    V4 ps;
    for (int i = 0; i < 4; i++) {
        double s = T.m[0][i] * pw.v[0];
        s = s + T.m[1][i] * pw.v[1];
        s = s + T.m[2][i] * pw.v[2];
        s = s + T.m[3][i] * pw.v[3];
        ps.v[i] = s;
    }
    V3 top = plane_hit_3d(T, invK, ps, c[1], c[8 + 1]);
    double height_half = top.v[2] / 2;
    double mean_x = (((g[0].v[0] + g[1].v[0]) + g[2].v[0]) + g[3].v[0]) / 4.0;
    double mean_y = (((g[0].v[1] + g[1].v[1]) + g[2].v[1]) + g[3].v[1]) / 4.0;
    o.pos[0] = mean_x; o.pos[1] = mean_y; o.pos[2] = height_half;
    o.rotY = yaw_esti;
    o.scale[0] = length_half; o.scale[1] = width_half; o.scale[2] = height_half;
    o.box_config_type[0] = config_id; o.box_config_type[1] = vp_1_position;
    static const int map1[8] = {6, 5, 8, 7, 2, 3, 4, 1}, map2[8] = {5, 6, 7, 8, 3, 2, 1, 4};
    const int *mp = (vp_1_position == 1) ? map1 : map2;
    for (int i = 0; i < 8; i++) {
        o.box_corners_2d[i] = (int)c[mp[i] - 1];
        o.box_corners_2d[8 + i] = (int)c[8 + mp[i] - 1];
    }
    // compute3D_BoxCorner :41-50 / similarityTransformation :14-26
    static const double body[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
    double cy = std::cos(o.rotY), sy = std::sin(o.rotY);
    double rot[3][3] = {{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}};
    double rs[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = rot[i][0] * (j == 0 ? o.scale[0] : 0.0);
            s = s + rot[i][1] * (j == 1 ? o.scale[1] : 0.0);
            s = s + rot[i][2] * (j == 2 ? o.scale[2] : 0.0);
            rs[i][j] = s;
        }
    for (int k = 0; k < 8; k++) {
        double wv[4];
        for (int i = 0; i < 3; i++) {
            double s = rs[i][0] * body[0][k];
            s = s + rs[i][1] * body[1][k];
            s = s + rs[i][2] * body[2][k];
            s = s + o.pos[i] * 1.0;
            wv[i] = s;
        }
        wv[3] = ((0.0 * body[0][k] + 0.0 * body[1][k]) + 0.0 * body[2][k]) + 1.0;
        for (int i = 0; i < 3; i++) o.box_corners_3d_world[i * 8 + k] = wv[i] / wv[3];
    }
}

// The geometry helpers above one at a time (test hook: tests/test_ref_pins.py compares each with the reference's own function, oracle/_ref).
//   0 check_inside_box(pt, lt, rb)                  in: 6            out: 1
//   1 seg_hit_boundary(ps, pe, seg4)                in: 8            out: 2
//   2 lineSegmentIntersect(..., infinite_line=true) in: 8            out: 2
//   3 getVanishingPoints(KinvR 3x3, yaw)            in: 10           out: 6
//   4 plane_hits_3d(T 4x4, invK 3x3, plane4, px,py) in: 31           out: 3
//   5 change_2d_corner_to_3d_object(corners 2x8, config id, vp_1_position, yaw, ground plane 4, T 4x4, invK 3x3) in: 48
//                                                   out: pos 3, rotY, scale 3, box_config_type 2, corners 2D 16 (as doubles), corners 3D 24 = 49
//   6 VP_support_edge_infos(vps 3x2, thresholds 2, n, mids n x 2, angles n)                                      out: 6
extern "C" int orc_cuboid_geom(int op, const double *in, double *out) {
    auto V = [&](int i) { return V2{in[i], in[i + 1]}; };
    auto M3_ = [&](int o) { M3 m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.m[i][j] = in[o + i * 3 + j]; return m; };
    auto M4_ = [&](int o) { M4 m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.m[i][j] = in[o + i * 4 + j]; return m; };
    switch (op) {
    case 0: out[0] = check_inside_box(V(0), V(2), V(4)) ? 1.0 : 0.0; return 0;
    case 1: { const V2 h = seg_hit_boundary(V(0), V(2), in[4], in[5], in[6], in[7]); out[0] = h.x; out[1] = h.y; return 0; }
    case 2: { const V2 h = line_intersect_inf(V(0), V(2), V(4), V(6)); out[0] = h.x; out[1] = h.y; return 0; }
    case 3: { V2 a, b, c; getVanishingPoints(M3_(0), in[9], a, b, c); out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y; out[4] = c.x; out[5] = c.y; return 0; }
    case 4: { const V3 w = plane_hit_3d(M4_(0), M3_(16), V4{{in[25], in[26], in[27], in[28]}}, in[29], in[30]); out[0] = w.v[0]; out[1] = w.v[1]; out[2] = w.v[2]; return 0; }
    case 5: {
        orc_cuboid o;
        std::memset(&o, 0, sizeof o);
        change_2d_corner_to_3d_object(in, in[16], in[17], in[18], V4{{in[19], in[20], in[21], in[22]}}, M4_(23), M3_(39), o);
        for (int i = 0; i < 3; i++) { out[i] = o.pos[i]; out[4 + i] = o.scale[i]; }
        out[3] = o.rotY; out[7] = o.box_config_type[0]; out[8] = o.box_config_type[1];
        for (int i = 0; i < 16; i++) out[9 + i] = (double)o.box_corners_2d[i];
        for (int i = 0; i < 24; i++) out[25 + i] = o.box_corners_3d_world[i];
        return 0;
    }
    case 6: {
        const V2 vps[3] = {V(0), V(2), V(4)};
        const int n = (int)in[8];
        std::vector<V2> mid((size_t)n); std::vector<double> ang((size_t)n);
        for (int i = 0; i < n; i++) { mid[i] = V2{in[9 + 2 * i], in[10 + 2 * i]}; ang[i] = in[9 + 2 * n + i]; }
        VP_support_edge_infos(vps, mid, ang, in[6], in[7], out);
        return 0;
    }
    }
    return -1;
}

// ----------------------------------------------------------------------------- OpenCV imgproc restatements
// Sobel 3x3 on the parent image (cv::Canny calls cv::Sobel(..., BORDER_REPLICATE) on the ROI *view*, so the
// filter reads real pixels outside the ROI and replicates only at the image border).
static inline int px(const uint8_t *g, int W, int H, int x, int y) {
    x = x < 0 ? 0 : (x >= W ? W - 1 : x);
    y = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return g[(long)y * W + x];
}
static void canny_roi(const uint8_t *gray, int W, int H, int x0, int y0, int w, int h, int low, int high, uint8_t *dst) {
    // cv::Canny, aperture 3, L2gradient=false (OpenCV imgproc/src/canny.cpp, classic serial path)
    std::vector<short> dx((size_t)w * h), dy((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int X = x0 + x, Y = y0 + y;
            int p00 = px(gray, W, H, X - 1, Y - 1), p01 = px(gray, W, H, X, Y - 1), p02 = px(gray, W, H, X + 1, Y - 1);
            int p10 = px(gray, W, H, X - 1, Y), p12 = px(gray, W, H, X + 1, Y);
            int p20 = px(gray, W, H, X - 1, Y + 1), p21 = px(gray, W, H, X, Y + 1), p22 = px(gray, W, H, X + 1, Y + 1);
            dx[(size_t)y * w + x] = (short)((p02 + 2 * p12 + p22) - (p00 + 2 * p10 + p20));
            dy[(size_t)y * w + x] = (short)((p20 + 2 * p21 + p22) - (p00 + 2 * p01 + p02));
        }
    const int mapstep = w + 2;
    std::vector<int> magbuf((size_t)mapstep * 3, 0);
    std::vector<uint8_t> map((size_t)mapstep * (h + 2));
    int *mag_buf[3] = {magbuf.data(), magbuf.data() + mapstep, magbuf.data() + 2 * mapstep};
    std::memset(map.data(), 1, mapstep);
    std::memset(map.data() + (size_t)mapstep * (h + 1), 1, mapstep);
    std::vector<uint8_t *> stack;
    const int CANNY_SHIFT = 15;
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5);
    for (int i = 0; i <= h; i++) {
        int *_norm = mag_buf[(i > 0) + 1] + 1;
        if (i < h) {
            const short *_dx = &dx[(size_t)i * w], *_dy = &dy[(size_t)i * w];
            for (int j = 0; j < w; j++) _norm[j] = std::abs(int(_dx[j])) + std::abs(int(_dy[j]));
            _norm[-1] = _norm[w] = 0;
        } else
            std::memset(_norm - 1, 0, mapstep * sizeof(int));
        if (i == 0) continue;
        uint8_t *_map = map.data() + (size_t)mapstep * i + 1;
        _map[-1] = _map[w] = 1;
        int *_mag = mag_buf[1] + 1;
        const short *_x = &dx[(size_t)(i - 1) * w], *_y = &dy[(size_t)(i - 1) * w];
        long magstep1 = mag_buf[2] - mag_buf[1], magstep2 = mag_buf[0] - mag_buf[1];
        int prev_flag = 0;
        for (int j = 0; j < w; j++) {
            int m = _mag[j];
            bool push = false;
            if (m > low) {
                int xs = _x[j], ys = _y[j];
                int x = std::abs(xs), y = std::abs(ys) << CANNY_SHIFT;
                int tg22x = x * TG22;
                if (y < tg22x) {
                    if (m > _mag[j - 1] && m >= _mag[j + 1]) push = true;
                } else {
                    int tg67x = tg22x + (x << (CANNY_SHIFT + 1));
                    if (y > tg67x) {
                        if (m > _mag[j + magstep2] && m >= _mag[j + magstep1]) push = true;
                    } else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        if (m > _mag[j + magstep2 - s] && m > _mag[j + magstep1 + s]) push = true;
                    }
                }
            }
            if (!push) { prev_flag = 0; _map[j] = 1; continue; }
            if (!prev_flag && m > high && _map[j - mapstep] != 2) {
                _map[j] = 2; stack.push_back(_map + j); prev_flag = 1;
            } else
                _map[j] = 0;
        }
        _mag = mag_buf[0]; mag_buf[0] = mag_buf[1]; mag_buf[1] = mag_buf[2]; mag_buf[2] = _mag;
    }
    while (!stack.empty()) {
        uint8_t *m = stack.back(); stack.pop_back();
        const long offs[8] = {-1, 1, -mapstep - 1, -mapstep, -mapstep + 1, mapstep - 1, mapstep, mapstep + 1};
        for (int k = 0; k < 8; k++) if (!m[offs[k]]) { m[offs[k]] = 2; stack.push_back(m + offs[k]); }
    }
    for (int i = 0; i < h; i++) {
        const uint8_t *pm = map.data() + (size_t)mapstep * (i + 1) + 1;
        for (int j = 0; j < w; j++) dst[(size_t)i * w + j] = (uint8_t) - (pm[j] >> 1);
    }
}

// cv::distanceTransform(src, CV_DIST_L2, 3) -> distanceTransform_3x3 (OpenCV imgproc/src/distransform.cpp):
// two-pass 3x3 chamfer in 16.16 fixed point, a=0.955f, b=1.3693f.
static void dist_transform_3x3(const uint8_t *src, int w, int h, float *dist) {
    const int DIST_SHIFT = 16;
    const int INIT_DIST0 = (INT_MAX >> 2);
    const int HV_DIST = (int)std::lrint((double)(0.955f * (1 << DIST_SHIFT)));
    const int DIAG_DIST = (int)std::lrint((double)(1.3693f * (1 << DIST_SHIFT)));
    const float scale = 1.f / (1 << DIST_SHIFT);
    const int step = w + 2;
    std::vector<int> temp((size_t)step * (h + 2));
    for (int j = 0; j < step; j++) { temp[j] = INIT_DIST0; temp[(size_t)step * (h + 1) + j] = INIT_DIST0; }
    for (int i = 0; i < h; i++) {
        const uint8_t *s = src + (size_t)i * w;
        int *tmp = temp.data() + (size_t)(i + 1) * step + 1;
        tmp[-1] = tmp[w] = INIT_DIST0;
        for (int j = 0; j < w; j++) {
            if (!s[j]) tmp[j] = 0;
            else {
                int t0 = tmp[j - step - 1] + DIAG_DIST;
                int t = tmp[j - step] + HV_DIST; if (t0 > t) t0 = t;
                t = tmp[j - step + 1] + DIAG_DIST; if (t0 > t) t0 = t;
                t = tmp[j - 1] + HV_DIST; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
        }
    }
    for (int i = h - 1; i >= 0; i--) {
        float *d = dist + (size_t)i * w;
        int *tmp = temp.data() + (size_t)(i + 1) * step + 1;
        for (int j = w - 1; j >= 0; j--) {
            int t0 = tmp[j];
            if (t0 > HV_DIST) {
                int t = tmp[j + step + 1] + DIAG_DIST; if (t0 > t) t0 = t;
                t = tmp[j + step] + HV_DIST; if (t0 > t) t0 = t;
                t = tmp[j + step - 1] + DIAG_DIST; if (t0 > t) t0 = t;
                t = tmp[j + 1] + HV_DIST; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
            d[j] = (float)(t0 * scale);
        }
    }
}

// object_3d_util.cpp:300-376 (fast_RemoveRow matrix_utils.cpp:172-176)
static int merge_break_lines(const double *in, int n, double pre_merge_dist_thre, double angle_thre_deg, double edge_length_threshold,
                             std::vector<double> &out) {
    std::vector<double> L(in, in + (size_t)n * 4);
    bool can_force_merge = true;
    int total = n, counter = 0;
    double angle_thre = angle_thre_deg / 180.0 * M_PI;
    std::vector<double> ang(n > 0 ? n : 1);
    while (can_force_merge && (counter < 500)) {
        counter++;
        can_force_merge = false;
        for (int i = 0; i < total; i++) ang[i] = std::atan2(L[i * 4 + 3] - L[i * 4 + 1], L[i * 4 + 2] - L[i * 4 + 0]);
        for (int s1 = 0; s1 < total - 1; s1++) {
            for (int s2 = s1 + 1; s2 < total; s2++) {
                double diff = std::abs(ang[s1] - ang[s2]);
                double angle_diff = std::min(diff, M_PI - diff);
                if (angle_diff < angle_thre) {
                    double d12 = dist2(V2{L[s1 * 4 + 2], L[s1 * 4 + 3]}, V2{L[s2 * 4 + 0], L[s2 * 4 + 1]});
                    double d21 = dist2(V2{L[s2 * 4 + 2], L[s2 * 4 + 3]}, V2{L[s1 * 4 + 0], L[s1 * 4 + 1]});
                    if ((d12 < pre_merge_dist_thre) || (d21 < pre_merge_dist_thre)) {
                        V2 ms, me;
                        if (L[s1 * 4 + 0] < L[s2 * 4 + 0]) ms = V2{L[s1 * 4 + 0], L[s1 * 4 + 1]}; else ms = V2{L[s2 * 4 + 0], L[s2 * 4 + 1]};
                        if (L[s1 * 4 + 2] > L[s2 * 4 + 2]) me = V2{L[s1 * 4 + 2], L[s1 * 4 + 3]}; else me = V2{L[s2 * 4 + 2], L[s2 * 4 + 3]};
                        double merged_angle = std::atan2(me.y - ms.y, me.x - ms.x);
                        double temp = std::abs(ang[s1] - merged_angle);
                        double merge_angle_diff = std::min(temp, M_PI - temp);
                        if (merge_angle_diff < angle_thre) {
                            L[s1 * 4 + 0] = ms.x; L[s1 * 4 + 1] = ms.y; L[s1 * 4 + 2] = me.x; L[s1 * 4 + 3] = me.y;
                            for (int k = 0; k < 4; k++) L[s2 * 4 + k] = L[(total - 1) * 4 + k];
                            total--;
                            can_force_merge = true;
                            break;
                        }
                    }
                }
            }
            if (can_force_merge) break;
        }
    }
    out.clear();
    if (edge_length_threshold > 0) {
        for (int i = 0; i < total; i++) {
            double dx = L[i * 4 + 2] - L[i * 4 + 0], dy = L[i * 4 + 3] - L[i * 4 + 1];
            if (std::sqrt(dx * dx + dy * dy) > edge_length_threshold) out.insert(out.end(), &L[i * 4], &L[i * 4] + 4);
        }
    } else
        out.assign(L.begin(), L.begin() + (size_t)total * 4);
    return (int)(out.size() / 4);
}

} // namespace

// ============================================================================ C API
extern "C" {

void orc_cuboid_default_opts(orc_cuboid_opts *o) {
    o->consider_config_1 = 1; o->consider_config_2 = 1;
    o->whether_sample_cam_roll_pitch = 0; o->whether_sample_bbox_height = 0;
    o->max_cuboid_num = 1; o->nominal_skew_ratio = 1; o->max_cut_skew = 3;
    o->yaw_range_deg = 45; o->yaw_step_deg = 6; o->canny_low = 80; o->canny_high = 200;
    o->stateful_cam_pose = 0;
}

void orc_bgr2gray(const uint8_t *bgr, int w, int h, uint8_t *gray) {
    // cv::cvtColor(CV_BGR2GRAY) 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14
    for (long i = 0; i < (long)w * h; i++)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}

void orc_canny_roi(const uint8_t *gray, int W, int H, int x0, int y0, int w, int h, int low, int high, uint8_t *edges) {
    canny_roi(gray, W, H, x0, y0, w, h, low, high, edges);
}
void orc_dist_transform_3x3(const uint8_t *src, int w, int h, float *dist) { dist_transform_3x3(src, w, h, dist); }

void orc_canny_dt_roi(const uint8_t *gray, int W, int H, int x0, int y0, int w, int h, int low, int high, float *dist) {
    std::vector<uint8_t> e((size_t)w * h);
    canny_roi(gray, W, H, x0, y0, w, h, low, high, e.data());
    for (auto &v : e) v = (uint8_t)(255 - v); // box_proposal_detail.cpp:199: distanceTransform(255 - im_canny, ...)
    dist_transform_3x3(e.data(), w, h, dist);
}

int orc_merge_break_lines(const double *lines, int n, double dist_thre, double angle_thre_deg, double len_thre, double *out) {
    std::vector<double> o;
    int m = merge_break_lines(lines, n, dist_thre, angle_thre_deg, len_thre, o);
    std::memcpy(out, o.data(), o.size() * sizeof(double));
    return m;
}

double orc_box_edge_sum_dists(const float *dist_map, int w, int h, const double *corners_shift, int config_id) {
    return box_edge_sum_dists(dist_map, w, h, corners_shift, config_id);
}
double orc_box_edge_angle_error(const double *vp_bound_angles, const double *corners, int config_id) {
    return box_edge_alignment_angle_error(vp_bound_angles, corners, config_id);
}
int orc_fuse_normalize_scores(const double *dist_err, const double *angle_err, int n, double weight_vp_angle, int whether_normalize,
                              int *keep, double *scores) {
    std::vector<double> c; std::vector<int> k;
    fuse_normalize_scores_v2(dist_err, angle_err, n, c, k, weight_vp_angle, whether_normalize != 0);
    for (size_t i = 0; i < k.size(); i++) { keep[i] = k[i]; scores[i] = c[i]; }
    return (int)k.size();
}

int orc_detect_cuboid(const uint8_t *gray, int W, int H, const double *K, const double *Twc, const double *boxes, int nb,
                      const double *lines_in, int nl, const orc_cuboid_opts *opt, orc_cuboid *out, int *counts,
                      double *dbg_rows, long dbg_rows_cap, int *dbg_row_count) {
    CamPose cam, cam_raw;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cam.Kalib.m[i][j] = K[i * 3 + j];
    cam.invK = inv3(cam.Kalib); // box_proposal_detail.cpp:36-40
    M4 T;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T.m[i][j] = Twc[i * 4 + j];
    set_cam_pose(cam, T); // :59
    cam_raw = cam;        // :60
    const int img_width = W, img_height = H;
    const bool all_configs[2] = {opt->consider_config_1 != 0, opt->consider_config_2 != 0};
    const double vp12_edge_angle_thre = 15, vp3_edge_angle_thre = 10, shorted_edge_thre = 20; // :79-81
    const bool whether_normalize_two_errors = true;                                          // :85
    const double weight_vp_angle = 0.8, weight_skew_error = 1.5;                             // :86-87
    std::vector<double> lines(lines_in, lines_in + (size_t)nl * 4);
    for (int i = 0; i < nl; i++) // align_left_right_edges object_3d_util.cpp:147-158
        if (lines[i * 4 + 2] < lines[i * 4 + 0]) { std::swap(lines[i * 4 + 0], lines[i * 4 + 2]); std::swap(lines[i * 4 + 1], lines[i * 4 + 3]); }
    V4 ground_plane_sensor = ground_plane_sensor_of(cam.transToWolrd);
    long dbg_n = 0; int dbg_seg = 0;

    for (int object_id = 0; object_id < nb; object_id++) {
        const double *bb = boxes + object_id * 5;
        int left_x_raw = (int)bb[0], top_y_raw = (int)bb[1], obj_width_raw = (int)bb[2], obj_height_raw = (int)bb[3];
        int right_x_raw = (int)(left_x_raw + bb[2]);
        std::vector<int> down_expand_sample_all{0};
        if (opt->whether_sample_bbox_height) { // :116-123
            int r = std::max(std::min(20, obj_height_raw - 90), 20);
            r = std::min(r, img_height - top_y_raw - obj_height_raw - 1);
            if (r > 10) down_expand_sample_all.push_back((int)std::round(r / 2));
            down_expand_sample_all.push_back(r);
        }
        const CamPose &yaw_src = opt->stateful_cam_pose ? cam : cam_raw;
        double yaw_init = yaw_src.camera_yaw - 90.0 / 180.0 * M_PI; // :126
        std::vector<double> obj_yaw_samples;
        linespace<double>(yaw_init - opt->yaw_range_deg / 180.0 * M_PI, yaw_init + opt->yaw_range_deg / 180.0 * M_PI,
                          opt->yaw_step_deg / 180.0 * M_PI, obj_yaw_samples);
        std::vector<orc_cuboid> raw_obj_proposals;

        for (size_t hs = 0; hs < down_expand_sample_all.size(); hs++) {
            int down_expand_sample = down_expand_sample_all[hs];
            int obj_height_expan = obj_height_raw + down_expand_sample;
            int down_y_expan = top_y_raw + obj_height_expan;
            double obj_diaglength_expan = std::sqrt((double)(obj_width_raw * obj_width_raw + obj_height_expan * obj_height_expan));
            int top_sample_resolution = (int)std::round((double)std::min(20, obj_width_raw / 10));
            std::vector<int> top_x_samples;
            linespace<int>(left_x_raw + 5, right_x_raw - 5, top_sample_resolution, top_x_samples);
            int distmap_expand_wid = std::min(std::max(std::min(20, obj_width_raw - 100), 10), std::max(std::min(20, obj_height_expan - 100), 10));
            int left_x_e = std::max(0, left_x_raw - distmap_expand_wid);
            int right_x_e = std::min(img_width - 1, right_x_raw + distmap_expand_wid);
            int top_y_e = std::max(0, top_y_raw - distmap_expand_wid);
            int down_y_e = std::min(img_height - 1, down_y_expan + distmap_expand_wid);
            int height_e = down_y_e - top_y_e, width_e = right_x_e - left_x_e;
            V2 e_lt{(double)left_x_e, (double)top_y_e}, e_rb{(double)right_x_e, (double)down_y_e};
            if (width_e <= 0 || height_e <= 0 || left_x_e + width_e > W || top_y_e + height_e > H) return -2; // cv::Rect ROI assert
            std::vector<double> inside;
            for (int e = 0; e < nl; e++)
                if (check_inside_box(V2{lines[e * 4], lines[e * 4 + 1]}, e_lt, e_rb) && check_inside_box(V2{lines[e * 4 + 2], lines[e * 4 + 3]}, e_lt, e_rb))
                    inside.insert(inside.end(), &lines[e * 4], &lines[e * 4] + 4);
            std::vector<double> merged;
            int nm = merge_break_lines(inside.data(), (int)(inside.size() / 4), 20, 5, 30, merged); // :177-182
            std::vector<double> lines_inobj_angles(nm);
            std::vector<V2> edge_mid_pts(nm);
            for (int i = 0; i < nm; i++) {
                lines_inobj_angles[i] = std::atan2(merged[i * 4 + 3] - merged[i * 4 + 1], merged[i * 4 + 2] - merged[i * 4 + 0]);
                edge_mid_pts[i] = V2{(merged[i * 4 + 0] + merged[i * 4 + 2]) / 2, (merged[i * 4 + 1] + merged[i * 4 + 3]) / 2};
            }
            std::vector<float> dist_map((size_t)width_e * height_e);
            orc_canny_dt_roi(gray, W, H, left_x_e, top_y_e, width_e, height_e, opt->canny_low, opt->canny_high, dist_map.data()); // :195-199

            std::vector<double> rows; // 25 doubles per valid proposal
            std::vector<double> cam_roll_samples, cam_pitch_samples;
            if (opt->whether_sample_cam_roll_pitch) { // :217-221
                linespace<double>(cam_raw.euler[0] - 6.0 / 180.0 * M_PI, cam_raw.euler[0] + 6.0 / 180.0 * M_PI, 3.0 / 180.0 * M_PI, cam_roll_samples);
                linespace<double>(cam_raw.euler[1] - 6.0 / 180.0 * M_PI, cam_raw.euler[1] + 6.0 / 180.0 * M_PI, 3.0 / 180.0 * M_PI, cam_pitch_samples);
            } else {
                cam_roll_samples.push_back(cam_raw.euler[0]);
                cam_pitch_samples.push_back(cam_raw.euler[1]);
            }
            for (size_t ri = 0; ri < cam_roll_samples.size(); ri++)
                for (size_t pi = 0; pi < cam_pitch_samples.size(); pi++)
                    for (size_t yi = 0; yi < obj_yaw_samples.size(); yi++) {
                        if (opt->whether_sample_cam_roll_pitch) { // :233-239
                            M4 Tn = T;
                            M3 Rn = euler_zyx_to_rot(cam_roll_samples[ri], cam_pitch_samples[pi], cam_raw.euler[2]);
                            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tn.m[i][j] = Rn.m[i][j];
                            set_cam_pose(cam, Tn);
                            ground_plane_sensor = ground_plane_sensor_of(cam.transToWolrd);
                        }
                        double obj_yaw_esti = obj_yaw_samples[yi];
                        V2 vps[3];
                        getVanishingPoints(cam.KinvR, obj_yaw_esti, vps[0], vps[1], vps[2]);
                        double vpa[6];
                        VP_support_edge_infos(vps, edge_mid_pts, lines_inobj_angles, vp12_edge_angle_thre, vp3_edge_angle_thre, vpa);
                        const V2 vp_1 = vps[0], vp_2 = vps[1], vp_3 = vps[2];
                        for (size_t ti = 0; ti < top_x_samples.size(); ti++) {
                            V2 c1{(double)top_x_samples[ti], (double)top_y_raw};
                            int vp_1_position = 0;
                            V2 c2 = seg_hit_boundary(vp_1, c1, right_x_raw, top_y_raw, right_x_raw, down_y_expan);
                            if (c2.x == -1) {
                                c2 = seg_hit_boundary(vp_1, c1, left_x_raw, top_y_raw, left_x_raw, down_y_expan);
                                if (c2.x != -1) vp_1_position = 2;
                            } else
                                vp_1_position = 1;
                            if (!(vp_1_position > 0)) continue;
                            if (dist2(c1, c2) < shorted_edge_thre) continue;
                            for (int config_id = 1; config_id < 3; config_id++) {
                                if (!all_configs[config_id - 1]) continue;
                                V2 c3, c4;
                                if (config_id == 1) {
                                    if (vp_1_position == 1) c4 = seg_hit_boundary(vp_2, c1, left_x_raw, top_y_raw, left_x_raw, down_y_expan);
                                    else c4 = seg_hit_boundary(vp_2, c1, right_x_raw, top_y_raw, right_x_raw, down_y_expan);
                                    if (c4.y == -1) continue;
                                    if (dist2(c1, c4) < shorted_edge_thre) continue;
                                    c3 = line_intersect_inf(vp_2, c2, vp_1, c4);
                                    if (!check_inside_box(c3, V2{(double)left_x_raw, (double)top_y_raw}, V2{(double)right_x_raw, (double)down_y_expan})) continue;
                                    if ((dist2(c3, c4) < shorted_edge_thre) || (dist2(c3, c2) < shorted_edge_thre)) continue;
                                } else {
                                    if (vp_1_position == 1) c3 = seg_hit_boundary(vp_2, c2, left_x_raw, top_y_raw, left_x_raw, down_y_expan);
                                    else c3 = seg_hit_boundary(vp_2, c2, right_x_raw, top_y_raw, right_x_raw, down_y_expan);
                                    if (c3.y == -1) continue;
                                    if (dist2(c2, c3) < shorted_edge_thre) continue;
                                    c4 = line_intersect_inf(vp_1, c3, vp_2, c1);
                                    if (!check_inside_box(c4, V2{(double)left_x_raw, (double)top_y_e}, V2{(double)right_x_raw, (double)down_y_e})) continue; // :347
                                    if ((dist2(c3, c4) < shorted_edge_thre) || (dist2(c4, c1) < shorted_edge_thre)) continue;
                                }
                                V2 c5 = seg_hit_boundary(vp_3, c3, left_x_raw, down_y_expan, right_x_raw, down_y_expan);
                                if (c5.y == -1) continue;
                                if (dist2(c3, c5) < shorted_edge_thre) continue;
                                V2 c6 = line_intersect_inf(vp_2, c5, vp_3, c2);
                                if (!check_inside_box(c6, e_lt, e_rb)) continue;
                                if ((dist2(c6, c2) < shorted_edge_thre) || (dist2(c6, c5) < shorted_edge_thre)) continue;
                                V2 c7 = line_intersect_inf(vp_1, c6, vp_3, c1);
                                if (!check_inside_box(c7, e_lt, e_rb)) continue;
                                if ((dist2(c7, c1) < shorted_edge_thre) || (dist2(c7, c6) < shorted_edge_thre)) continue;
                                V2 c8 = line_intersect_inf(vp_1, c5, vp_2, c7);
                                if (!check_inside_box(c8, e_lt, e_rb)) continue;
                                if ((dist2(c8, c4) < shorted_edge_thre) || (dist2(c8, c5) < shorted_edge_thre) || (dist2(c8, c7) < shorted_edge_thre)) continue;

                                double cor[16] = {c1.x, c2.x, c3.x, c4.x, c5.x, c6.x, c7.x, c8.x, c1.y, c2.y, c3.y, c4.y, c5.y, c6.y, c7.y, c8.y};
                                double cs[16];
                                for (int i = 0; i < 8; i++) { cs[i] = cor[i] - left_x_e; cs[8 + i] = cor[8 + i] - top_y_e; }
                                double sum_dist = box_edge_sum_dists(dist_map.data(), width_e, height_e, cs, config_id);
                                double total_angle_diff = box_edge_alignment_angle_error(vpa, cor, config_id);
                                double row[25];
                                row[0] = config_id; row[1] = vp_1_position; row[2] = obj_yaw_esti; row[3] = (double)ti;
                                row[4] = sum_dist / obj_diaglength_expan; row[5] = total_angle_diff; row[6] = down_expand_sample;
                                if (opt->whether_sample_cam_roll_pitch) { row[7] = cam_roll_samples[ri]; row[8] = cam_pitch_samples[pi]; }
                                else { row[7] = cam_raw.euler[0]; row[8] = cam_raw.euler[1]; }
                                for (int i = 0; i < 16; i++) row[9 + i] = cor[i];
                                rows.insert(rows.end(), row, row + 25);
                            }
                        }
                    }
            int nvalid = (int)(rows.size() / 25);
            if (dbg_row_count) dbg_row_count[dbg_seg] = nvalid;
            dbg_seg++;
            if (dbg_rows)
                for (int i = 0; i < nvalid && dbg_n < dbg_rows_cap; i++, dbg_n++) std::memcpy(dbg_rows + dbg_n * 25, &rows[(size_t)i * 25], 25 * sizeof(double));

            std::vector<double> de(nvalid), ae(nvalid);
            for (int i = 0; i < nvalid; i++) { de[i] = rows[(size_t)i * 25 + 4]; ae[i] = rows[(size_t)i * 25 + 5]; }
            std::vector<double> normalized_score; std::vector<int> good;
            fuse_normalize_scores_v2(de.data(), ae.data(), nvalid, normalized_score, good, weight_vp_angle, whether_normalize_two_errors);
            for (size_t b = 0; b < good.size(); b++) {
                const double *r = &rows[(size_t)good[b] * 25];
                if (opt->whether_sample_cam_roll_pitch) { // :481-487
                    M4 Tn = T;
                    M3 Rn = euler_zyx_to_rot(r[7], r[8], cam_raw.euler[2]);
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tn.m[i][j] = Rn.m[i][j];
                    set_cam_pose(cam, Tn);
                    ground_plane_sensor = ground_plane_sensor_of(cam.transToWolrd);
                }
                orc_cuboid o;
                std::memset(&o, 0, sizeof(o));
                change_2d_corner_to_3d_object(r + 9, r[0], r[1], r[2], ground_plane_sensor, cam.transToWolrd, cam.invK, o);
                if (o.scale[0] < 0 || o.scale[1] < 0 || o.scale[2] < 0) continue; // :493
                o.rect_detect_2d[0] = left_x_raw; o.rect_detect_2d[1] = top_y_raw; o.rect_detect_2d[2] = obj_width_raw; o.rect_detect_2d[3] = obj_height_raw;
                o.edge_distance_error = r[4]; o.edge_angle_error = r[5];
                o.normalized_error = normalized_score[b];
                o.skew_ratio = std::max(o.scale[0], o.scale[1]) / std::min(o.scale[0], o.scale[1]);
                o.down_expand_height = r[6];
                if (opt->whether_sample_cam_roll_pitch) { o.camera_roll_delta = r[7] - cam_raw.euler[0]; o.camera_pitch_delta = r[8] - cam_raw.euler[1]; }
                else { o.camera_roll_delta = 0; o.camera_pitch_delta = 0; }
                raw_obj_proposals.push_back(o);
            }
        }
        // :517-536
        int n_raw = (int)raw_obj_proposals.size();
        int k = std::min(opt->max_cuboid_num, n_raw);
        std::vector<double> score(n_raw);
        for (int i = 0; i < n_raw; i++) {
            const orc_cuboid &o = raw_obj_proposals[i];
            double skew_error = weight_skew_error * std::max(o.skew_ratio - opt->nominal_skew_ratio, 0.0);
            if (o.skew_ratio > opt->max_cut_skew) skew_error = 100;
            score[i] = o.normalized_error + weight_skew_error * skew_error;
        }
        std::vector<int> idx(n_raw);
        std::iota(idx.begin(), idx.end(), 0);
        sort_indexes_topk(score.data(), idx, k);
        counts[object_id] = k;
        for (int i = 0; i < k; i++) out[(size_t)object_id * opt->max_cuboid_num + i] = raw_obj_proposals[idx[i]];
    }
    return 0;
}

} // extern "C"
