// local_ba_objects.hpp -- C++ host-side mirror of Optimizer::LocalBACameraPointObjects (reference orb_object_slam/include/Optimizer.h:48,
// src/Optimizer.cc:826-1534) on flat arrays over the C-ABI bundle adjuster (cs_ba_*): no OpenCV / Eigen / g2o.  The same flow as
// cube_slam_amd/ba_objects.py (which tests/test_local_ba_objects.py holds against the oracle); adapters/Optimizer_hip.cc gathers the
// window from KeyFrame* / MapPoint* / MapObject* and calls this.
//
// Steps (Optimizer.cc lines): object vertices with the KITTI half size and the height reset :983-1026; points with one observation skipped
// :1052; reprojection edges :1068-1137; point-object association -- count threshold max(int(0.4 largest), 2), 4 m / 3 m outlier filter,
// centroid reset above 5 points, unary edge above 10 :1141-1266; camera-object edges -- information (w [/2 above 5 objects])^2 q^2, 10 px margin,
// level 1 for an object seen once, left / right balancing :1268-1382; optimize(5), outliers to level 1 (chi2 5.991 / 7.815, depth, |bbox
// error| > 80), point kernels off, optimize(10) :1389-1438; erase list :1440-1475.  A level-1 edge is absent from the arrays a stage hands to
// the solver, and so is a vertex no active edge touches; a free camera left without edges is held fixed.  The outlier tests read the
// residuals at the accepted estimates (pin D4 of DESIGN.md).
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "orb_slam_mirrors.hpp"

namespace cubeslam {

struct LocalWindow { // in the reference's iteration order (see cube_slam_amd/ba_objects.py for the meaning of every array)
    int n_local = 0;                                   // the first n_local key frames are lLocalKeyFrames, the rest lFixedCameras
    std::vector<long> kf_id; std::vector<double> kf_pose; double cur_cam_center[3] = {0, 0, 0};
    std::vector<double> mp_pos; std::vector<int> mp_nobs;
    std::vector<int> obs_mp, obs_kf; std::vector<double> obs_uv, obs_ur, obs_inv_sigma2;
    std::vector<double> mo_pose, mo_scale, mo_meas_quality; std::vector<int> mo_largest_point_observations;
    std::vector<int> up_mo, up_count; std::vector<double> up_pos;
    std::vector<int> det_mo, det_kf, det_bbox_2d, det_left_right_to_car; std::vector<double> det_bbox_vec;
};
struct LocalBAParams {
    double K[9] = {0}; int img_width = 0, img_height = 0; double bf = 0, camera_object_BA_weight = 1.0;
    bool kitti = true, build_worldframe_on_ground = false, fixCamera = false;
};
struct LocalBAResult {
    std::vector<double> kf_pose;                       // n_local x 7
    std::vector<int> point_rows; std::vector<double> point_pos; // rows of mp_* that were vertices, and their positions
    std::vector<int> point_unwritten;                          // of those, the rows the erase list leaves with exactly one observation: the reference does not write them back (:1511-1512 after :1486-1496)
    std::vector<double> object_pose, object_scale;     // objects x 7, x 3
    std::vector<std::pair<int, int>> erase;            // (row of kf_*, row of mp_*)
    std::vector<uint8_t> erase_stereo;                 // per entry of `erase`: the observation has a right coordinate -- MapPoint::EraseObservation takes 2 off the point's count for it (MapPoint.cc:186-189)
    std::vector<uint8_t> obs_level, cobs_level, cobs_level2; std::vector<int> obs_rows, det_rows;
    std::vector<int> up_used, up_filtered;             // rows of up_*: MapObject::used_points_in_BA (count above the threshold) and used_points_in_BA_filtered (within 3 m)
    cs_ba_stats st1{}, st2{};
};

namespace local_ba_detail {
struct Graph { // owns what cs_ba_problem points into
    std::vector<double> cam_pose, points, cuboid_pose, cuboid_scale, obs_uv, obs_w, obs_ur, cobs_bbox, cobs_info, pc_points;
    std::vector<uint8_t> cam_fixed, cuboid_flags;
    std::vector<int> obs_cam, obs_point, cobs_cam, cobs_cuboid, pc_cuboid, pc_offsets;
    double fx = 0, fy = 0, cx = 0, cy = 0, huber_mono = 0, huber_stereo = 0, huber_obj = 0, bf = 0, ratio = 1, K[9] = {0};
    cs_ba_problem view() const {
        cs_ba_problem p{};
        p.n_cams = (int)cam_fixed.size(); p.cam_pose = cam_pose.data(); p.cam_fixed = cam_fixed.data();
        p.n_points = (int)points.size() / 3; p.points = points.data();
        p.n_cuboids = (int)cuboid_flags.size(); p.cuboid_pose = cuboid_pose.data(); p.cuboid_scale = cuboid_scale.data(); p.cuboid_flags = cuboid_flags.data();
        p.n_obs = (int)obs_cam.size(); p.obs_cam = obs_cam.data(); p.obs_point = obs_point.data(); p.obs_uv = obs_uv.data(); p.obs_inv_sigma2 = obs_w.data();
        p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.huber_mono = huber_mono;
        p.n_cobs = (int)cobs_cam.size(); p.cobs_cam = cobs_cam.data(); p.cobs_cuboid = cobs_cuboid.data(); p.cobs_bbox = cobs_bbox.data(); p.cobs_info = cobs_info.data();
        for (int i = 0; i < 9; i++) p.K[i] = K[i];
        p.huber_obj = huber_obj;
        p.n_pc = (int)pc_cuboid.size(); p.pc_cuboid = pc_cuboid.data(); p.pc_offsets = pc_offsets.data(); p.pc_points = pc_points.data(); p.max_outside_margin_ratio = ratio;
        p.obs_ur = obs_ur.data(); p.bf = bf; p.huber_stereo = huber_stereo;
        return p;
    }
};
inline double norm3(const double *a, const double *b) { const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return std::sqrt(x * x + y * y + z * z); }

// level-0 edges and the vertices they touch; pts_used / cub_used map the compact vertices back
inline Graph active_subgraph(const Graph &d, const std::vector<uint8_t> &keep_obs, const std::vector<uint8_t> &keep_cobs, std::vector<int> &pts_used, std::vector<int> &cub_used) {
    const int np = (int)d.points.size() / 3, nc = (int)d.cuboid_flags.size(), ncam = (int)d.cam_fixed.size();
    std::vector<int> pmap((size_t)np, -1), cmap((size_t)nc, -1);
    std::vector<uint8_t> used((size_t)ncam, 0);
    for (size_t o = 0; o < d.obs_cam.size(); o++) if (keep_obs[o]) { pmap[d.obs_point[o]] = 0; used[d.obs_cam[o]] = 1; }
    for (size_t o = 0; o < d.cobs_cam.size(); o++) if (keep_cobs[o]) { cmap[d.cobs_cuboid[o]] = 0; used[d.cobs_cam[o]] = 1; }
    for (int c : d.pc_cuboid) cmap[c] = 0;
    pts_used.clear(); cub_used.clear();
    for (int j = 0; j < np; j++) if (pmap[j] == 0) { pmap[j] = (int)pts_used.size(); pts_used.push_back(j); }
    for (int c = 0; c < nc; c++) if (cmap[c] == 0) { cmap[c] = (int)cub_used.size(); cub_used.push_back(c); }
    Graph s = d;
    for (int i = 0; i < ncam; i++) s.cam_fixed[i] = d.cam_fixed[i] || !used[i];
    s.points.clear(); for (int j : pts_used) s.points.insert(s.points.end(), &d.points[(size_t)j * 3], &d.points[(size_t)j * 3 + 3]);
    s.cuboid_pose.clear(); s.cuboid_scale.clear(); s.cuboid_flags.clear();
    for (int c : cub_used) {
        s.cuboid_pose.insert(s.cuboid_pose.end(), &d.cuboid_pose[(size_t)c * 7], &d.cuboid_pose[(size_t)c * 7 + 7]);
        s.cuboid_scale.insert(s.cuboid_scale.end(), &d.cuboid_scale[(size_t)c * 3], &d.cuboid_scale[(size_t)c * 3 + 3]);
        s.cuboid_flags.push_back(d.cuboid_flags[c]);
    }
    s.obs_cam.clear(); s.obs_point.clear(); s.obs_uv.clear(); s.obs_w.clear(); s.obs_ur.clear();
    for (size_t o = 0; o < d.obs_cam.size(); o++) if (keep_obs[o]) {
        s.obs_cam.push_back(d.obs_cam[o]); s.obs_point.push_back(pmap[d.obs_point[o]]); s.obs_uv.push_back(d.obs_uv[2 * o]); s.obs_uv.push_back(d.obs_uv[2 * o + 1]);
        s.obs_w.push_back(d.obs_w[o]); s.obs_ur.push_back(d.obs_ur[o]);
    }
    s.cobs_cam.clear(); s.cobs_cuboid.clear(); s.cobs_bbox.clear(); s.cobs_info.clear();
    for (size_t o = 0; o < d.cobs_cam.size(); o++) if (keep_cobs[o]) {
        s.cobs_cam.push_back(d.cobs_cam[o]); s.cobs_cuboid.push_back(cmap[d.cobs_cuboid[o]]);
        for (int k = 0; k < 4; k++) { s.cobs_bbox.push_back(d.cobs_bbox[4 * o + k]); s.cobs_info.push_back(d.cobs_info[4 * o + k]); }
    }
    for (size_t k = 0; k < d.pc_cuboid.size(); k++) s.pc_cuboid[k] = cmap[d.pc_cuboid[k]];
    // (never empty arrays behind the pointers)
    if (s.points.empty()) s.points.assign(3, 0.0);
    return s;
}
// one SparseOptimizer::optimize over the active part; estimates written back into d
inline void solve(Context &c, Graph &d, const std::vector<uint8_t> &keep_obs, const std::vector<uint8_t> &keep_cobs, int iterations, const volatile int *stop, cs_ba_stats *st,
                  const volatile bool *stop_bool = nullptr) {
    std::vector<int> pu, cu;
    Graph s = active_subgraph(d, keep_obs, keep_cobs, pu, cu);
    if (s.obs_cam.empty() && s.cobs_cam.empty() && s.pc_cuboid.empty()) return;
    cs_ba_problem p = s.view();
    p.n_points = (int)pu.size();
    cs_ba *ba = nullptr;
    check(c.ctx, cs_ba_create(c.ctx, &p, 0, 1, &ba), "cs_ba_create");
    std::vector<double> pts((size_t)pu.size() * 3 + 3), cub((size_t)cu.size() * 7 + 7);
    cs_ba_set_stop_flag_bool(ba, reinterpret_cast<const volatile unsigned char *>(stop_bool)); // optimizer.setForceStopFlag(pbStopFlag), Optimizer.cc:943-944: live during the solve
    int r = cs_ba_optimize(c.ctx, ba, iterations, stop, st);
    if (!r) r = cs_ba_read(c.ctx, ba, d.cam_pose.data(), pts.data(), cub.data());
    cs_ba_destroy(c.ctx, ba);
    check(c.ctx, r, "cs_ba stage");
    for (size_t k = 0; k < pu.size(); k++) for (int a = 0; a < 3; a++) d.points[(size_t)pu[k] * 3 + a] = pts[k * 3 + a];
    for (size_t k = 0; k < cu.size(); k++) for (int a = 0; a < 7; a++) d.cuboid_pose[(size_t)cu[k] * 7 + a] = cub[k * 7 + a];
}
// chi2 and depth of every reprojection edge, |error| of every camera-object edge, at the estimates in d
inline void residuals(Context &c, const Graph &d, std::vector<double> &chi2, std::vector<double> &depth, std::vector<double> &cnorm) {
    std::vector<uint8_t> all_o(d.obs_cam.size(), 1), all_c(d.cobs_cam.size(), 1);
    std::vector<int> pu, cu;
    Graph s = active_subgraph(d, all_o, all_c, pu, cu);
    cs_ba_problem p = s.view();
    p.n_points = (int)pu.size();
    cs_ba *ba = nullptr;
    check(c.ctx, cs_ba_create(c.ctx, &p, 0, 1, &ba), "cs_ba_create");
    std::vector<double> eo(d.obs_cam.size() * 3 + 3), ec(d.cobs_cam.size() * 4 + 4);
    double chi_total = 0;
    const int r = cs_ba_errors(c.ctx, ba, &chi_total, eo.data(), ec.data(), nullptr);
    cs_ba_destroy(c.ctx, ba);
    check(c.ctx, r, "cs_ba_errors");
    chi2.resize(d.obs_cam.size()); depth.resize(d.obs_cam.size()); cnorm.resize(d.cobs_cam.size());
    for (size_t o = 0; o < d.obs_cam.size(); o++) {
        const double *e = &eo[o * 3], w = d.obs_w[o];
        chi2[o] = d.obs_ur[o] >= 0 ? ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * w : (e[0] * e[0] + e[1] * e[1]) * w;
        const double *T = &d.cam_pose[(size_t)d.obs_cam[o] * 7], *X = &d.points[(size_t)d.obs_point[o] * 3];
        const double qx = T[3], qy = T[4], qz = T[5], qw = T[6]; // third row of R(q) times X plus t_z: EdgeSE3ProjectXYZ::isDepthPositive
        depth[o] = (2 * (qx * qz - qy * qw)) * X[0] + (2 * (qy * qz + qx * qw)) * X[1] + (1 - 2 * (qx * qx + qy * qy)) * X[2] + T[2];
    }
    for (size_t o = 0; o < d.cobs_cam.size(); o++) { const double *e = &ec[o * 4]; cnorm[o] = std::sqrt(((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) + e[3] * e[3]); }
}
} // namespace local_ba_detail

// pbStopFlag: an int flag (this library's convention); pbStopBool: the reference's `bool *pbStopFlag` as it is (another thread raises it): both are
// polled during the solves (between iterations and LM trials) and between the two stages (:1392-1396)
inline void LocalBACameraPointObjects(Context &c, const LocalWindow &w, const LocalBAParams &prm, LocalBAResult &out, const volatile int *pbStopFlag = nullptr,
                                      const volatile bool *pbStopBool = nullptr) {
    using namespace local_ba_detail;
    const int n_kf = (int)w.kf_id.size(), n_obj = (int)w.mo_meas_quality.size(), n_mp = (int)w.mp_nobs.size();
    Graph d;
    d.cam_pose = w.kf_pose; d.cam_fixed.assign((size_t)n_kf, 1);
    for (int i = 0; i < w.n_local; i++) d.cam_fixed[i] = (w.kf_id[i] == 0) || prm.fixCamera;                   // :958-960
    d.cuboid_pose = w.mo_pose; d.cuboid_scale = w.mo_scale; d.cuboid_flags.assign((size_t)n_obj, 1 | 8);           // whether_fixrollpitch, fixedscale
    if (prm.kitti)
        for (int i = 0; i < n_obj; i++) {                                                                           // :994-1014
            if (!prm.build_worldframe_on_ground) d.cuboid_pose[(size_t)i * 7 + 1] = (double)(float)w.cur_cam_center[1] + 1.0;
            else d.cuboid_pose[(size_t)i * 7 + 2] = (double)(float)w.cur_cam_center[2] - 1.0;
            d.cuboid_scale[(size_t)i * 3] = 1.9420; d.cuboid_scale[(size_t)i * 3 + 1] = 0.8143; d.cuboid_scale[(size_t)i * 3 + 2] = 0.7631;
        }
    std::vector<int> prow((size_t)n_mp, -1);
    out.point_rows.clear();
    for (int j = 0; j < n_mp; j++) if (w.mp_nobs[j] != 1) { prow[j] = (int)out.point_rows.size(); out.point_rows.push_back(j); }     // :1052
    for (int j : out.point_rows) d.points.insert(d.points.end(), &w.mp_pos[(size_t)j * 3], &w.mp_pos[(size_t)j * 3 + 3]);
    out.obs_rows.clear();
    for (size_t o = 0; o < w.obs_mp.size(); o++) if (prow[w.obs_mp[o]] >= 0) {
        out.obs_rows.push_back((int)o);
        d.obs_cam.push_back(w.obs_kf[o]); d.obs_point.push_back(prow[w.obs_mp[o]]); d.obs_uv.push_back(w.obs_uv[2 * o]); d.obs_uv.push_back(w.obs_uv[2 * o + 1]);
        d.obs_w.push_back(w.obs_inv_sigma2[o]); d.obs_ur.push_back(w.obs_ur[o]);
    }
    d.pc_offsets.push_back(0);
    out.up_used.clear(); out.up_filtered.clear();
    for (int i = 0; i < n_obj; i++) {                                                                               // :1141-1266
        int thr = (int)(w.mo_largest_point_observations[i] * 0.4); if (thr < 2) thr = 2;
        std::vector<const double *> P;
        std::vector<int> Prow;
        for (size_t u = 0; u < w.up_mo.size(); u++) if (w.up_mo[u] == i && w.up_count[u] > thr) { P.push_back(&w.up_pos[u * 3]); Prow.push_back((int)u); out.up_used.push_back((int)u); }
        if (P.empty()) continue;
        double mean[3] = {0, 0, 0};
        for (const double *p : P) for (int a = 0; a < 3; a++) mean[a] = mean[a] + p[a];
        for (int a = 0; a < 3; a++) mean[a] = mean[a] / (double)P.size();
        double mean2[3] = {0, 0, 0}; int n2 = 0;
        for (const double *p : P) if (norm3(mean, p) < 4.0) { for (int a = 0; a < 3; a++) mean2[a] = mean2[a] + p[a]; n2++; }
        if (n2 == 0) continue;
        for (int a = 0; a < 3; a++) mean2[a] = mean2[a] / (double)n2;
        std::vector<const double *> good;
        double acc[3] = {0, 0, 0};
        for (size_t k = 0; k < P.size(); k++) if (norm3(mean2, P[k]) < 3.0) { for (int a = 0; a < 3; a++) acc[a] = acc[a] + P[k][a]; good.push_back(P[k]); out.up_filtered.push_back(Prow[k]); }
        if (good.size() > 5) for (int a = 0; a < 3; a++) d.cuboid_pose[(size_t)i * 7 + a] = acc[a] / (double)good.size();
        if (good.size() > 10) {
            d.pc_cuboid.push_back(i);
            for (const double *p : good) d.pc_points.insert(d.pc_points.end(), p, p + 3);
            d.pc_offsets.push_back(d.pc_offsets.back() + (int)good.size());
        }
    }
    double inv_sigma = 1.0 * prm.camera_object_BA_weight;                                                           // :1283-1288
    if (n_obj > 5) inv_sigma = inv_sigma / 2;
    const int m = 10;
    out.det_rows.clear();
    std::vector<int> lr;
    for (size_t k = 0; k < w.det_mo.size(); k++) {
        const int *r = &w.det_bbox_2d[k * 4];
        if (!(r[0] > m && r[1] > m && r[0] + r[2] < prm.img_width - m && r[1] + r[3] < prm.img_height - m)) continue;   // :1322-1323
        out.det_rows.push_back((int)k);
        const double q = w.mo_meas_quality[w.det_mo[k]];
        d.cobs_cam.push_back(w.det_kf[k]); d.cobs_cuboid.push_back(w.det_mo[k]);
        for (int a = 0; a < 4; a++) { d.cobs_bbox.push_back(w.det_bbox_vec[k * 4 + a]); d.cobs_info.push_back(inv_sigma * inv_sigma * q * q); }
        lr.push_back(w.det_left_right_to_car[k]);
    }
    const size_t n_cobs = d.cobs_cam.size(), n_obs = d.obs_cam.size();
    out.cobs_level.assign(n_cobs + 1, 0); out.cobs_level.resize(n_cobs);
    for (int i = 0; i < n_obj; i++) {                                                                               // :1362-1365
        int cnt = 0, last = -1;
        for (size_t k = 0; k < n_cobs; k++) if (d.cobs_cuboid[k] == i) { cnt++; last = (int)k; }
        if (cnt == 1) out.cobs_level[last] = 1;
    }
    if (prm.kitti) {                                                                                                // :1368-1380
        int tl = 0, tr = 0, tm = 0;
        for (int v : lr) { tl += v == 1; tr += v == 2; tm += v == 0; }
        if (tl > 2 * (tr + tm)) for (size_t k = 0; k < n_cobs; k++) if (lr[k] == 1) for (int a = 0; a < 4; a++) d.cobs_info[k * 4 + a] = d.cobs_info[k * 4 + a] / 2.0;
        if (tr > 2 * (tl + tm)) for (size_t k = 0; k < n_cobs; k++) if (lr[k] == 2) for (int a = 0; a < 4; a++) d.cobs_info[k * 4 + a] = d.cobs_info[k * 4 + a] / 2.0;
    }
    d.fx = prm.K[0]; d.fy = prm.K[4]; d.cx = prm.K[2]; d.cy = prm.K[5];
    for (int i = 0; i < 9; i++) d.K[i] = prm.K[i];
    // `const float thHuberMono = sqrt(5.991)` :1043-1044, thHuberObject :1292: setDelta receives the float-rounded width
    d.huber_mono = (double)(float)std::sqrt(5.991); d.huber_stereo = (double)(float)std::sqrt(7.815); d.huber_obj = (double)(float)std::sqrt(900.0); d.bf = prm.bf; d.ratio = prm.kitti ? 2.0 : 1.0;
    out.object_scale = d.cuboid_scale;
    // two stages :1389-1438
    std::vector<uint8_t> keep_obs(n_obs + 1, 1), keep_cobs(n_cobs + 1, 1);
    keep_obs.resize(n_obs); keep_cobs.resize(n_cobs);
    for (size_t k = 0; k < n_cobs; k++) keep_cobs[k] = out.cobs_level[k] == 0;
    solve(c, d, keep_obs, keep_cobs, 5, pbStopFlag, &out.st1, pbStopBool);
    out.obs_level.assign(n_obs, 0); out.cobs_level2 = out.cobs_level;
    std::vector<double> chi1, z1, cn;
    residuals(c, d, chi1, z1, cn);
    const bool more = !(pbStopFlag && *pbStopFlag) && !(pbStopBool && *pbStopBool);
    if (more) {
        for (size_t o = 0; o < n_obs; o++) if (chi1[o] > (d.obs_ur[o] >= 0 ? 7.815 : 5.991) || !(z1[o] > 0)) out.obs_level[o] = 1;
        for (size_t k = 0; k < n_cobs; k++) if (out.cobs_level[k] == 0 && cn[k] > 80) out.cobs_level2[k] = 1;
        d.huber_mono = 0; d.huber_stereo = 0;
        for (size_t o = 0; o < n_obs; o++) keep_obs[o] = out.obs_level[o] == 0;
        for (size_t k = 0; k < n_cobs; k++) keep_cobs[k] = out.cobs_level2[k] == 0;
        solve(c, d, keep_obs, keep_cobs, 10, pbStopFlag, &out.st2, pbStopBool);
    }
    std::vector<double> chi2, z2;
    residuals(c, d, chi2, z2, cn);
    out.erase.clear(); out.erase_stereo.clear();
    for (int pass = 0; pass < 2; pass++)   // vpEdgesMono first, then vpEdgesStereo (:1445-1475)
        for (size_t o = 0; o < n_obs; o++) {
            const bool stereo = d.obs_ur[o] >= 0;
            if (stereo != (pass == 1)) continue;
            const double chi = out.obs_level[o] == 0 ? chi2[o] : chi1[o]; // a level-1 edge keeps the error of stage 1
            if (chi > (stereo ? 7.815 : 5.991) || !(z2[o] > 0)) { out.erase.emplace_back(w.obs_kf[out.obs_rows[o]], w.obs_mp[out.obs_rows[o]]); out.erase_stereo.push_back(stereo ? 1 : 0); }
        }
    out.kf_pose.assign(d.cam_pose.begin(), d.cam_pose.begin() + (size_t)w.n_local * 7);
    out.point_pos = d.points; out.object_pose = d.cuboid_pose;
    std::vector<int> left(w.mp_nobs);
    for (size_t k = 0; k < out.erase.size(); k++) left[out.erase[k].second] -= out.erase_stereo[k] ? 2 : 1; // mp_nobs is MapPoint::Observations() = nObs: a stereo observation counts twice
    out.point_unwritten.clear();
    for (int r : out.point_rows) if (left[r] == 1) out.point_unwritten.push_back(r);
}

} // namespace cubeslam
