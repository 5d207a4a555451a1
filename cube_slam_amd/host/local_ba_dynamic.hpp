// local_ba_dynamic.hpp -- C++ host-side mirror of Optimizer::LocalBACameraPointObjectsDynamic (reference orb_object_slam/include/Optimizer.h:57-58,
// src/Optimizer.cc:1537-2573) on flat arrays over the C-ABI dynamic-object bundle adjuster (cs_ba_dyn_*): no OpenCV / Eigen / g2o.  The same flow as
// cube_slam_amd/ba_dynamic.py (`build_graph`, `LocalBACameraPointObjectsDynamic`; tests/test_local_ba_dynamic.py holds both to the oracle's restatement, which is
// pinned to the reference's own function text); adapters/Optimizer_hip.cc gathers the window from KeyFrame* / MapPoint* / MapObject* and calls this.
//
// Steps (Optimizer.cc lines): pose vertices :1689-1713; one cuboid vertex per (object, observing key frame of the window) with the KITTI half size and the height
// reset from THAT key frame's camera :1727-1786; static points (one observation and dynamic points skipped) and their edges :1808-1906; dynamic points -- at least
// four observations, owned by a local object: PosToObj, UnaryLocalPoint, one three-vertex edge per observing key frame in which the owner has a vertex :1919-2001;
// point-object association as in the static function, over the STATIC function's vertex ids :2008-2115 (`mnId + maxKFid + 1` names the (mnId + 1)-th cuboid vertex
// created here, whichever object it belongs to); velocity vertices for objects with at least four vertices, motion edges between consecutive observing key frames of
// the last 5 s, a zero velocity initialised from the first and last stored pose :2137-2237; camera-object edges with the 10 px margin, key frames older than 5 s
// skipped when the motion edges are on, level 1 for an object left with one edge, left / right balancing :2243-2340; optimize(5), re-levelling (chi2 5.991 / 7.815 /
// depth; chi2 8 for the three-vertex edges; |bbox error| 80), point kernels off, optimize(10) :2353-2415; erase list :2417-2444; what is written back :2446-2572.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "orb_slam_mirrors.hpp"

namespace cubeslam {

struct DynamicWindow { // rows in the reference's iteration order (cube_slam_amd/ba_dynamic.py names every array)
    int n_local = 0;                                     // the first n_local key frames are lLocalKeyFrames (row 0: pKF), the rest lFixedCameras
    std::vector<long> kf_id; std::vector<double> kf_pose, kf_stamp, kf_cam_center;
    std::vector<double> mp_pos, mp_pos_to_obj; std::vector<int> mp_nobs, mp_best_mo; std::vector<uint8_t> mp_dynamic;
    std::vector<int> obs_mp, obs_kf; std::vector<double> obs_uv, obs_ur, obs_inv_sigma2;
    std::vector<long> mo_id; std::vector<double> mo_meas_quality, mo_velocity; std::vector<int> mo_largest_point_observations;
    std::vector<int> ov_mo, ov_kf, ov_bbox_2d, ov_left_right_to_car; std::vector<double> ov_pose, ov_bbox_vec;
    std::vector<int> seq_mo, seq_kf;
    std::vector<int> up_mo, up_count; std::vector<double> up_pos;
};
struct DynamicBAParams {
    double K[9] = {0}; int img_width = 0, img_height = 0; double bf = 0, camera_object_BA_weight = 1.0, object_velocity_BA_weight = 1.0;
    bool kitti = true, build_worldframe_on_ground = false, fixCamera = false, fixPoint = false, ba_dyna_pt_obj_cam = true, ba_dyna_obj_velo = true, ba_dyna_obj_cam = true;
};
struct DynamicGraphArrays { // owns what cs_ba_dyn_problem points into, plus the rows of the window behind the vertices and edges
    std::vector<double> cam_pose, obj_pose, obj_scale, vel, points, dpoints, obs_uv, obs_ur, obs_w, dobs_uv, dobs_w, mot_dt, cobs_bbox, cobs_info, pc_points;
    std::vector<uint8_t> cam_fixed, obj_flags, obs_level, dobs_level, cobs_level;
    std::vector<int> obs_cam, obs_point, dobs_cam, dobs_obj, dobs_point, mot_from, mot_to, mot_vel, cobs_cam, cobs_obj, pc_obj, pc_offsets;
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0, huber_mono = 0, huber_stereo = 0, huber_dyn = 0, huber_obj = 0, ulp_info = 10.0, ulp_ratio = 2.0, pc_ratio = 2.0;
    double ulp_scale[3] = {1.9420, 0.8143, 0.7631}, mot_info[3] = {1, 1, 25}, K[9] = {0};
    int fix_points = 0;
    std::vector<int> point_rows, obs_rows, dpoint_rows, dobs_rows, cobs_rows, vel_mo, up_used, up_filtered;
    std::vector<std::pair<int, std::pair<double, double>>> velocity_init; // (row of mo_*, the velocity :2223-2233 writes into the object before the solve)
    cs_ba_dyn_problem view() const {
        static const double zero_d[8] = {0}; static const int zero_i[2] = {0}; static const uint8_t zero_b[2] = {0};
        auto D = [&](const std::vector<double> &v) { return v.empty() ? zero_d : v.data(); };
        auto I = [&](const std::vector<int> &v) { return v.empty() ? zero_i : v.data(); };
        auto B = [&](const std::vector<uint8_t> &v) { return v.empty() ? zero_b : v.data(); };
        cs_ba_dyn_problem p{};
        p.n_cams = (int)cam_fixed.size(); p.cam_pose = D(cam_pose); p.cam_fixed = B(cam_fixed);
        p.n_objs = (int)obj_flags.size(); p.obj_pose = D(obj_pose); p.obj_scale = D(obj_scale); p.obj_flags = B(obj_flags);
        p.n_vels = (int)vel.size() / 2; p.vel = D(vel);
        p.n_points = (int)points.size() / 3; p.points = D(points); p.n_dpoints = (int)dpoints.size() / 3; p.dpoints = D(dpoints); p.fix_points = fix_points;
        p.n_obs = (int)obs_cam.size(); p.obs_cam = I(obs_cam); p.obs_point = I(obs_point); p.obs_uv = D(obs_uv); p.obs_ur = D(obs_ur); p.obs_inv_sigma2 = D(obs_w); p.obs_level = B(obs_level);
        p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.bf = bf; p.huber_mono = huber_mono; p.huber_stereo = huber_stereo;
        p.ulp_info = ulp_info; p.ulp_ratio = ulp_ratio;
        for (int i = 0; i < 3; i++) { p.ulp_scale[i] = ulp_scale[i]; p.mot_info[i] = mot_info[i]; }
        p.n_dobs = (int)dobs_cam.size(); p.dobs_cam = I(dobs_cam); p.dobs_obj = I(dobs_obj); p.dobs_point = I(dobs_point); p.dobs_uv = D(dobs_uv); p.dobs_inv_sigma2 = D(dobs_w); p.dobs_level = B(dobs_level);
        for (int i = 0; i < 9; i++) p.K[i] = K[i];
        p.huber_dyn = huber_dyn;
        p.n_mot = (int)mot_from.size(); p.mot_from = I(mot_from); p.mot_to = I(mot_to); p.mot_vel = I(mot_vel); p.mot_dt = D(mot_dt);
        p.n_cobs = (int)cobs_cam.size(); p.cobs_cam = I(cobs_cam); p.cobs_obj = I(cobs_obj); p.cobs_bbox = D(cobs_bbox); p.cobs_info = D(cobs_info); p.cobs_level = B(cobs_level); p.huber_obj = huber_obj;
        p.n_pc = (int)pc_obj.size(); p.pc_obj = I(pc_obj); p.pc_offsets = I(pc_offsets); p.pc_points = D(pc_points); p.pc_ratio = pc_ratio;
        return p;
    }
};
struct DynamicBAResult {
    std::vector<double> kf_pose;                                  // n_local x 7
    std::vector<int> point_rows; std::vector<double> point_pos;   // rows of mp_* that were static point vertices, and their positions
    std::vector<int> point_unwritten;                             // of those, the rows the erase list leaves with exactly one observation (:2478 after :2449-2459)
    std::vector<std::pair<int, int>> erase; std::vector<uint8_t> erase_stereo; // (row of kf_*, row of mp_*); whether the observation counts twice in MapPoint::Observations()
    std::vector<double> vertex_pose;                              // per row of ov_*: allDynamicPoses[key frame] (flag true)
    std::vector<int> object_latest;                               // per row of mo_*: the row of ov_* whose key frame has the largest mnId (pose_Twc_latestKF / SetWorldPos / pose_Twc_afterba), -1: no vertex
    std::vector<int> vel_mo; std::vector<double> velocity;        // rows of mo_* with a velocity vertex, and velocityPlanar / velocityhistory[pKF]
    std::vector<std::pair<int, std::pair<double, double>>> velocity_init; // written before the solve (also when the function is stopped)
    std::vector<int> dpoint_rows; std::vector<double> dpoint_local;       // rows of mp_* that were dynamic point vertices: PosToObj
    std::vector<int> dworld_rows; std::vector<double> dpoint_world;       // of those, the rows whose owner has a latest pose: mWorldPos_latestKF / SetWorldPos, is_optimized
    std::vector<int> up_used, up_filtered;                        // rows of up_*: used_points_in_BA / used_points_in_BA_filtered
    bool solved = false;                                          // false: stopped before the first optimize (:2344-2346) -- only velocity_init / up_* are valid
    cs_ba_stats st1{}, st2{};
};

namespace local_ba_dynamic_detail {
inline double dist3(const double *a, const double *b) { const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2]; return std::sqrt(x * x + y * y + z * z); }
} // namespace local_ba_dynamic_detail

// the g2o graph of :1667-2340 as the arrays of cs_ba_dyn_problem (no device needed)
inline DynamicGraphArrays build_dynamic_graph(const DynamicWindow &w, const DynamicBAParams &prm) {
    using namespace local_ba_dynamic_detail;
    if (!prm.kitti) throw std::runtime_error("LocalBACameraPointObjectsDynamic: the reference fixes the object size for scene_unique_id == kitti only (Optimizer.cc:1762-1765)");
    DynamicGraphArrays d;
    const int n_kf = (int)w.kf_id.size(), n_mp = (int)w.mp_nobs.size(), n_obj = (int)w.mo_id.size(), n_v = (int)w.ov_mo.size();
    const double now = w.kf_stamp.empty() ? 0.0 : w.kf_stamp[0];
    d.cam_pose = w.kf_pose; d.cam_fixed.assign((size_t)n_kf, 1);
    for (int i = 0; i < w.n_local; i++) d.cam_fixed[i] = (w.kf_id[i] == 0) || prm.fixCamera;                                   // :1693-1696
    // ---- cuboid vertices :1727-1786
    d.obj_pose = w.ov_pose; d.obj_flags.assign((size_t)n_v, 2 | 8);                                                            // whether_fixrotation, fixed scale
    std::map<std::pair<int, int>, int> vertex;
    std::vector<int> n_vert_of((size_t)n_obj, 0);
    for (int v = 0; v < n_v; v++) {
        if (!prm.build_worldframe_on_ground) d.obj_pose[(size_t)v * 7 + 1] = (double)(float)w.kf_cam_center[(size_t)w.ov_kf[v] * 3 + 1] + 1.0;   // :1771-1772
        for (int a = 0; a < 3; a++) d.obj_scale.push_back(d.ulp_scale[a]);
        vertex[{w.ov_mo[v], w.ov_kf[v]}] = v; n_vert_of[w.ov_mo[v]]++;
    }
    // ---- static points :1808-1906, dynamic points :1919-2001
    std::vector<int> prow((size_t)n_mp, -1), drow((size_t)n_mp, -1);
    for (int j = 0; j < n_mp; j++) {
        if (w.mp_nobs[j] != 1 && !w.mp_dynamic[j]) { prow[j] = (int)d.point_rows.size(); d.point_rows.push_back(j); for (int a = 0; a < 3; a++) d.points.push_back(w.mp_pos[(size_t)j * 3 + a]); }
        if (prm.ba_dyna_pt_obj_cam && w.mp_dynamic[j] && w.mp_nobs[j] >= 4 && w.mp_best_mo[j] >= 0) {
            drow[j] = (int)d.dpoint_rows.size(); d.dpoint_rows.push_back(j);
            for (int a = 0; a < 3; a++) d.dpoints.push_back(w.mp_pos_to_obj[(size_t)j * 3 + a]);
        }
    }
    for (size_t o = 0; o < w.obs_mp.size(); o++) {
        const int j = w.obs_mp[o];
        if (prow[j] >= 0) {
            d.obs_rows.push_back((int)o);
            d.obs_cam.push_back(w.obs_kf[o]); d.obs_point.push_back(prow[j]); d.obs_uv.push_back(w.obs_uv[2 * o]); d.obs_uv.push_back(w.obs_uv[2 * o + 1]);
            d.obs_ur.push_back(w.obs_ur[o]); d.obs_w.push_back(w.obs_inv_sigma2[o]);
        } else if (drow[j] >= 0) {
            const auto it = vertex.find({w.mp_best_mo[j], w.obs_kf[o]});                                                      // :1965-1966
            if (it == vertex.end()) continue;
            d.dobs_rows.push_back((int)o);
            d.dobs_cam.push_back(w.obs_kf[o]); d.dobs_obj.push_back(it->second); d.dobs_point.push_back(drow[j]);
            d.dobs_uv.push_back(w.obs_uv[2 * o]); d.dobs_uv.push_back(w.obs_uv[2 * o + 1]); d.dobs_w.push_back(w.obs_inv_sigma2[o]);
        }
    }
    d.obs_level.assign(d.obs_cam.size(), 0); d.dobs_level.assign(d.dobs_cam.size(), 0);
    // ---- point-object association :2008-2115
    d.pc_offsets.push_back(0);
    for (int i = 0; i < n_obj; i++) {
        int thr = (int)(w.mo_largest_point_observations[i] * 0.4); if (thr < 2) thr = 2;
        std::vector<const double *> P; std::vector<int> Prow;
        for (size_t u = 0; u < w.up_mo.size(); u++) if (w.up_mo[u] == i && w.up_count[u] > thr) { P.push_back(&w.up_pos[u * 3]); Prow.push_back((int)u); d.up_used.push_back((int)u); }
        std::vector<const double *> good;
        double acc[3] = {0, 0, 0};
        if (!P.empty()) {
            double mean[3] = {0, 0, 0}, mean2[3] = {0, 0, 0}; int n2 = 0;
            for (const double *p : P) for (int a = 0; a < 3; a++) mean[a] = mean[a] + p[a];
            for (int a = 0; a < 3; a++) mean[a] = mean[a] / (double)P.size();
            for (const double *p : P) if (dist3(mean, p) < 4.0) { for (int a = 0; a < 3; a++) mean2[a] = mean2[a] + p[a]; n2++; }
            for (int a = 0; a < 3; a++) mean2[a] = n2 ? mean2[a] / (double)n2 : NAN;
            for (size_t k = 0; k < P.size(); k++) if (dist3(mean2, P[k]) < 3.0) { for (int a = 0; a < 3; a++) acc[a] = acc[a] + P[k][a]; good.push_back(P[k]); d.up_filtered.push_back(Prow[k]); }
        }
        const long named = w.mo_id[i]; // `pMObj->mnId + maxKFid + 1` with maxKFid already incremented (:1730): the (mnId + 1)-th cuboid vertex created
        if (good.size() > 5) {
            if (named < 0 || named >= n_v)
                throw std::runtime_error("LocalBACameraPointObjectsDynamic: object " + std::to_string(i) + " names cuboid vertex " + std::to_string(named) + " of " + std::to_string(n_v) +
                                         " (the reference dereferences a null vertex here, Optimizer.cc:2075)");
            for (int a = 0; a < 3; a++) d.obj_pose[(size_t)named * 7 + a] = acc[a] / (double)good.size();
        }
        if (good.size() > 10) {
            d.pc_obj.push_back((int)named);
            for (const double *p : good) d.pc_points.insert(d.pc_points.end(), p, p + 3);
            d.pc_offsets.push_back(d.pc_offsets.back() + (int)good.size());
        }
    }
    // ---- velocity vertices, motion edges :2137-2237
    if (prm.ba_dyna_obj_velo)
        for (int i = 0; i < n_obj; i++) {
            if (n_vert_of[i] < 4) continue;
            const int vi = (int)d.vel_mo.size();
            d.vel_mo.push_back(i); d.vel.push_back(w.mo_velocity[(size_t)i * 2]); d.vel.push_back(w.mo_velocity[(size_t)i * 2 + 1]);
            int first = -1, last = -1, prev = -1;
            for (size_t s = 0; s < w.seq_mo.size(); s++) {
                if (w.seq_mo[s] != i) continue;
                const int k = w.seq_kf[s];
                if (!vertex.count({i, k}) || (now - w.kf_stamp[k]) > 5.0) continue;
                if (prev < 0) { prev = first = k; continue; }
                d.mot_from.push_back(vertex[{i, prev}]); d.mot_to.push_back(vertex[{i, k}]); d.mot_vel.push_back(vi); d.mot_dt.push_back(w.kf_stamp[k] - w.kf_stamp[prev]);
                prev = last = k;
            }
            if (d.vel[(size_t)vi * 2] == 0 && d.vel[(size_t)vi * 2 + 1] == 0 && first >= 0 && last >= 0) {
                const double lin = dist3(&w.ov_pose[(size_t)vertex[{i, last}] * 7], &w.ov_pose[(size_t)vertex[{i, first}] * 7]) / (w.kf_stamp[last] - w.kf_stamp[first]); // the stored poses, not the height-reset estimates
                d.vel[(size_t)vi * 2] = lin; d.vel[(size_t)vi * 2 + 1] = 0.0;
                d.velocity_init.push_back({i, {lin, 0.0}});
            }
        }
    // ---- camera-object edges :2243-2340
    if (prm.ba_dyna_obj_cam) {
        const double wc = 1.0 * prm.camera_object_BA_weight;
        const int m = 10;
        std::vector<int> lr;
        for (int i = 0; i < n_obj; i++) {
            int cnt = 0, only = -1;
            for (int v = 0; v < n_v; v++) {
                if (w.ov_mo[v] != i) continue;
                if (prm.ba_dyna_obj_velo && (now - w.kf_stamp[w.ov_kf[v]]) > 5.0) continue;
                const int *r = &w.ov_bbox_2d[(size_t)v * 4];
                if (!(r[0] > m && r[1] > m && r[0] + r[2] < prm.img_width - m && r[1] + r[3] < prm.img_height - m)) continue;
                const double q = w.mo_meas_quality[i];
                only = (int)d.cobs_rows.size(); cnt++;
                d.cobs_rows.push_back(v); d.cobs_cam.push_back(w.ov_kf[v]); d.cobs_obj.push_back(v);
                for (int a = 0; a < 4; a++) { d.cobs_bbox.push_back(w.ov_bbox_vec[(size_t)v * 4 + a]); d.cobs_info.push_back(wc * wc * q * q); }
                d.cobs_level.push_back(0); lr.push_back(w.ov_left_right_to_car[v]);
            }
            if (cnt == 1) d.cobs_level[only] = 1;                                                                              // :2316-2319
        }
        int tl = 0, tr = 0, tm = 0;
        for (int v : lr) { tl += v == 1; tr += v == 2; tm += v == 0; }
        if (tl > 2 * (tr + tm)) for (size_t k = 0; k < lr.size(); k++) if (lr[k] == 1) for (int a = 0; a < 4; a++) d.cobs_info[k * 4 + a] = d.cobs_info[k * 4 + a] / 2.0;
        if (tr > 2 * (tl + tm)) for (size_t k = 0; k < lr.size(); k++) if (lr[k] == 2) for (int a = 0; a < 4; a++) d.cobs_info[k * 4 + a] = d.cobs_info[k * 4 + a] / 2.0;
    }
    d.fx = prm.K[0]; d.fy = prm.K[4]; d.cx = prm.K[2]; d.cy = prm.K[5]; d.bf = prm.bf;
    for (int i = 0; i < 9; i++) d.K[i] = prm.K[i];
    // `const float thHuberMono = sqrt(5.991)` :1802-1803, thHuberObject :2262: setDelta receives the float-rounded width
    d.huber_mono = (double)(float)std::sqrt(5.991); d.huber_stereo = (double)(float)std::sqrt(7.815); d.huber_dyn = (double)(float)std::sqrt(5.991); d.huber_obj = (double)(float)std::sqrt(900.0);
    const double wv = prm.object_velocity_BA_weight;
    d.mot_info[0] = (1.0 * wv) * (1.0 * wv); d.mot_info[1] = (1.0 * wv) * (1.0 * wv); d.mot_info[2] = (5.0 * wv) * (5.0 * wv);
    d.fix_points = prm.fixPoint ? 1 : 0;
    return d;
}

// pbStopFlag: an int flag (this library's convention); pbStopBool: the reference's `bool *pbStopFlag` as it is (another thread raises it): polled during the solves and
// between the two stages (:2344-2351)
inline void LocalBACameraPointObjectsDynamic(Context &c, const DynamicWindow &w, const DynamicBAParams &prm, DynamicBAResult &out, const volatile int *pbStopFlag = nullptr,
                                             const volatile bool *pbStopBool = nullptr) {
    DynamicGraphArrays d = build_dynamic_graph(w, prm);
    out = DynamicBAResult();
    out.up_used = d.up_used; out.up_filtered = d.up_filtered; out.velocity_init = d.velocity_init; out.point_rows = d.point_rows; out.dpoint_rows = d.dpoint_rows; out.vel_mo = d.vel_mo;
    auto stopped = [&] { return (pbStopFlag && *pbStopFlag) || (pbStopBool && *pbStopBool); };
    if (stopped()) return;                                                                                                      // :2344-2346
    const size_t n_obs = d.obs_cam.size(), n_dobs = d.dobs_cam.size(), n_cobs = d.cobs_cam.size();
    std::vector<double> eo, ed, ec;
    auto stage = [&](int iterations, cs_ba_stats *st) {
        const cs_ba_dyn_problem P = d.view();
        cs_ba_dyn *ba = nullptr;
        check(c.ctx, cs_ba_dyn_create(c.ctx, &P, &ba), "cs_ba_dyn_create");
        cs_ba_dyn_set_stop_flag_bool(ba, reinterpret_cast<const volatile unsigned char *>(pbStopBool));
        std::vector<double> cam(d.cam_pose.size() + 7), obj(d.obj_pose.size() + 7), vel(d.vel.size() + 2), pts(d.points.size() + 3), dpts(d.dpoints.size() + 3);
        int r = cs_ba_dyn_optimize(c.ctx, ba, iterations, pbStopFlag, st);
        if (!r) r = cs_ba_dyn_read(c.ctx, ba, cam.data(), obj.data(), vel.data(), pts.data(), dpts.data());
        eo.assign(n_obs * 3 + 3, 0.0); ed.assign(n_dobs * 2 + 2, 0.0); ec.assign(n_cobs * 4 + 4, 0.0);
        if (!r) r = cs_ba_dyn_errors(c.ctx, ba, nullptr, eo.data(), ed.data(), nullptr, ec.data(), nullptr, nullptr);
        cs_ba_dyn_destroy(c.ctx, ba);
        check(c.ctx, r, "cs_ba_dyn stage");
        std::copy(cam.begin(), cam.begin() + d.cam_pose.size(), d.cam_pose.begin()); std::copy(obj.begin(), obj.begin() + d.obj_pose.size(), d.obj_pose.begin());
        std::copy(vel.begin(), vel.begin() + d.vel.size(), d.vel.begin()); std::copy(pts.begin(), pts.begin() + d.points.size(), d.points.begin());
        std::copy(dpts.begin(), dpts.begin() + d.dpoints.size(), d.dpoints.begin());
    };
    auto chi2_depth = [&](std::vector<double> &chi, std::vector<double> &z) {
        chi.resize(n_obs); z.resize(n_obs);
        for (size_t o = 0; o < n_obs; o++) {
            const double *e = &eo[o * 3], wgt = d.obs_w[o];
            chi[o] = d.obs_ur[o] >= 0 ? ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * wgt : (e[0] * e[0] + e[1] * e[1]) * wgt;
            const double *T = &d.cam_pose[(size_t)d.obs_cam[o] * 7], *X = &d.points[(size_t)d.obs_point[o] * 3];
            const double qx = T[3], qy = T[4], qz = T[5], qw = T[6]; // third row of R(q) times X plus t_z: isDepthPositive
            z[o] = (2 * (qx * qz - qy * qw)) * X[0] + (2 * (qy * qz + qx * qw)) * X[1] + (1 - 2 * (qx * qx + qy * qy)) * X[2] + T[2];
        }
    };
    stage(5, &out.st1);                                                                                                         // :2348-2349
    out.solved = true;
    std::vector<double> chi1, z1;
    chi2_depth(chi1, z1);
    if (!stopped()) {                                                                                                           // bDoMore :2351-2415
        for (size_t o = 0; o < n_obs; o++) if (chi1[o] > (d.obs_ur[o] >= 0 ? 7.815 : 5.991) || !(z1[o] > 0)) d.obs_level[o] = 1;
        for (size_t o = 0; o < n_dobs; o++) { const double *e = &ed[o * 2]; if ((e[0] * e[0] + e[1] * e[1]) * d.dobs_w[o] > 8) d.dobs_level[o] = 1; }
        for (size_t o = 0; o < n_cobs; o++) { const double *e = &ec[o * 4]; if (std::sqrt(((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) + e[3] * e[3]) > 80) d.cobs_level[o] = 1; }
        d.huber_mono = d.huber_stereo = d.huber_dyn = 0;                                                                        // setRobustKernel(0) on the three kinds of point edges
        stage(10, &out.st2);
    }
    std::vector<double> chi2, z2;
    chi2_depth(chi2, z2);
    for (int pass = 0; pass < 2; pass++)                                                                                        // vpEdgesMono, then vpEdgesStereo :2420-2444
        for (size_t o = 0; o < n_obs; o++) {
            const bool stereo = d.obs_ur[o] >= 0;
            if (stereo != (pass == 1)) continue;
            const double chi = d.obs_level[o] == 0 ? chi2[o] : chi1[o]; // a level-1 edge keeps the error of stage 1
            if (chi > (stereo ? 7.815 : 5.991) || !(z2[o] > 0)) { out.erase.emplace_back(w.obs_kf[d.obs_rows[o]], w.obs_mp[d.obs_rows[o]]); out.erase_stereo.push_back(stereo ? 1 : 0); }
        }
    out.kf_pose.assign(d.cam_pose.begin(), d.cam_pose.begin() + (size_t)w.n_local * 7);
    out.point_pos = d.points; out.vertex_pose = d.obj_pose; out.velocity = d.vel; out.dpoint_local = d.dpoints;
    std::vector<int> left(w.mp_nobs);
    for (size_t k = 0; k < out.erase.size(); k++) left[out.erase[k].second] -= out.erase_stereo[k] ? 2 : 1;
    for (int r : out.point_rows) if (left[r] == 1) out.point_unwritten.push_back(r);
    out.object_latest.assign(w.mo_id.size(), -1);                                                                               // :2503-2516
    for (size_t v = 0; v < w.ov_mo.size(); v++) {
        int &l = out.object_latest[w.ov_mo[v]];
        if (l < 0 || w.kf_id[w.ov_kf[v]] > w.kf_id[w.ov_kf[l]]) l = (int)v;
    }
    for (size_t k = 0; k < out.dpoint_rows.size(); k++) {                                                                       // :2536-2563
        const int r = out.dpoint_rows[k], l = out.object_latest[w.mp_best_mo[r]];
        if (l < 0) continue;
        const double *T = &out.vertex_pose[(size_t)l * 7], *p = &out.dpoint_local[k * 3];
        const double x = T[3], y = T[4], z = T[5], q = T[6];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * q), 2 * (x * z + y * q), 2 * (x * y + z * q), 1 - 2 * (x * x + z * z), 2 * (y * z - x * q),
                             2 * (x * z - y * q), 2 * (y * z + x * q), 1 - 2 * (x * x + y * y)};
        out.dworld_rows.push_back(r);
        for (int a = 0; a < 3; a++) out.dpoint_world.push_back((R[a * 3] * p[0] + R[a * 3 + 1] * p[1]) + R[a * 3 + 2] * p[2] + T[a]);
    }
}

} // namespace cubeslam
