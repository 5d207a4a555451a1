// TEST INFRASTRUCTURE (never linked into the product): the reference's graph-level optimisation functions, run as they are written.
//   Optimizer::BundleAdjustment (orb_object_slam/src/Optimizer.cc:64-251), Optimizer::PoseOptimization (:253-472) and
//   Optimizer::LocalBACameraPointObjects (:826-1534) are cut out of the reference at build time (oracle/_ref/extracted_graph.inc, extract_ref.py) together
//   with the Converter functions they call (src/Converter.cc:36-52, 62-70, 82-89, 109-115) and compiled here against
//     * the reference's vendored g2o, WHOLE, from where it lies (Thirdparty/g2o/g2o/{core,types,stuff}: SparseOptimizer, BlockSolver with its Schur complement
//       block_solver.hpp:354-486, OptimizationAlgorithmLevenberg, LinearSolverDense, the robust kernels) and the reference's g2o_Object.{h,cpp} and matrix_utils.cpp,
//     * a stand-in for Eigen (oracle/ref_shim/eigen_full: Eigen is not in this image) and for LinearSolverEigen (g2o_shadow: a wrapper of Eigen's sparse Cholesky),
//     * stand-ins for the map classes that hold the data those functions read (slam_graph_standins.hpp).
// tests/test_ref_graph_pins.py fills a window through the C entry points below, runs the reference's function and holds the oracle's graph-level restatements
// (oracle/local_ba_objects.py, orc_ba_optimize, orc_pose_optimization) to what it leaves in the map.
#include "slam_graph_standins.hpp"

#include <cstring>
#include <memory>

using namespace std;
using namespace Eigen;

namespace ORB_SLAM2 {
bool standin_verbose = false;
bool parallel_mapping = false, whether_dynamic_object = false, build_worldframe_on_ground = false, whether_detect_object = true, associate_point_with_object = true, bundle_object_opti = true;
bool remove_dynamic_features = false, use_dynamic_klt_features = false, mono_firstframe_truth_depth_init = false, mono_firstframe_Obj_depth_init = false, mono_allframe_Obj_depth_init = false;
bool enable_ground_height_scale = false, ba_dyna_pt_obj_cam = false, ba_dyna_obj_velo = false, ba_dyna_obj_cam = false, draw_map_truth_paths = false, draw_nonlocal_mappoint = false;
double camera_object_BA_weight = 1.0, object_velocity_BA_weight = 1.0, delta_t = 0.1;
Scene_Name scene_unique_id = kitti;
EraseLog *standin_log = nullptr;
std::mutex MapPoint::mGlobalMutex;
long int MapObject::nNextId = 0;

#include "extracted_graph_conv.inc"
#include "extracted_graph_map.inc"
#include "extracted_graph.inc"
} // namespace ORB_SLAM2

using namespace ORB_SLAM2;
#define API extern "C" __attribute__((visibility("default")))

#include "ref_graph_types.hpp"

namespace {
struct Quiet { // the functions narrate on std::cout
    std::streambuf *was;
    Quiet() : was(standin_verbose ? nullptr : std::cout.rdbuf(nullptr)) {}
    ~Quiet() { if (was) std::cout.rdbuf(was); }
};
} // namespace

static cv::Mat mat_f(int r, int c, const float *v) { cv::Mat m(r, c, CV_32F); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) m.at<float>(i, j) = v[i * c + j]; return m; }
static g2o::cuboid cuboid_of(const double *pose7, const double *scale3) {
    g2o::cuboid c;
    c.pose = g2o::SE3Quat(Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]), Eigen::Vector3d(pose7[0], pose7[1], pose7[2]));
    c.scale = Eigen::Vector3d(scale3[0], scale3[1], scale3[2]);
    return c;
}

API ref_graph *ref_graph_open() { return new ref_graph(); }
API void ref_graph_close(ref_graph *g) { delete g; }
API void ref_graph_set_params(ref_graph *g, int is_kitti, int worldframe_on_ground, double cam_obj_weight, int dynamic_objects, int img_w, int img_h, const double *K9, int verbose) {
    scene_unique_id = is_kitti ? kitti : voidtype;
    build_worldframe_on_ground = worldframe_on_ground != 0; camera_object_BA_weight = cam_obj_weight; whether_dynamic_object = dynamic_objects != 0; standin_verbose = verbose != 0;
    g->map.img_width = img_w; g->map.img_height = img_h;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { g->map.Kalib(i, j) = K9[i * 3 + j]; g->map.Kalib_f(i, j) = (float)K9[i * 3 + j]; }
    g->map.invKalib = g->map.Kalib.inverse();
}
// a key frame: pose as the 4 x 4 float matrix KeyFrame::GetPose returns, camera centre as GetCameraCenter returns it (3 floats), undistorted key points
API int ref_graph_add_kf(ref_graph *g, long id, int bad, const float *Tcw16, const float *Ow3, int n_keys, const float *keys_xy, const float *u_right, const int *octave, int n_levels,
                         const float *inv_level_sigma2, float fx, float fy, float cx, float cy, float bf) {
    std::unique_ptr<KeyFrame> k(new KeyFrame());
    k->mnId = (unsigned long)id; k->mnFrameId = (unsigned long)id; k->bad = bad != 0;
    k->Tcw = mat_f(4, 4, Tcw16); k->Ow = mat_f(3, 1, Ow3);
    k->mvKeysUn.resize(n_keys); k->mvuRight.assign(u_right, u_right + n_keys); k->mvpMapPoints.assign(n_keys, nullptr);
    for (int i = 0; i < n_keys; i++) { k->mvKeysUn[i].pt.x = keys_xy[2 * i]; k->mvKeysUn[i].pt.y = keys_xy[2 * i + 1]; k->mvKeysUn[i].octave = octave[i]; }
    k->mvInvLevelSigma2.assign(inv_level_sigma2, inv_level_sigma2 + n_levels);
    k->fx = fx; k->fy = fy; k->cx = cx; k->cy = cy; k->mbf = bf;
    g->map.all_kfs.push_back(k.get());
    g->kfs.push_back(std::move(k));
    return (int)g->kfs.size() - 1;
}
API int ref_graph_add_mp(ref_graph *g, long id, int bad, const float *pos3, int is_dynamic) {
    std::unique_ptr<MapPoint> p(new MapPoint());
    p->mnId = (unsigned long)id; p->bad = bad != 0; p->is_dynamic = is_dynamic != 0; p->mWorldPos = mat_f(3, 1, pos3);
    g->map.all_mps.push_back(p.get());
    g->mps.push_back(std::move(p));
    return (int)g->mps.size() - 1;
}
API int ref_graph_add_mo(ref_graph *g, long id, int bad, const double *pose7, const double *scale3, double meas_quality, int largest_point_observations) {
    std::unique_ptr<MapObject> o(new MapObject());
    o->mnId = id; o->bad = bad != 0; o->pose_Twc = cuboid_of(pose7, scale3); o->meas_quality = meas_quality; o->largest_point_observations = largest_point_observations;
    o->already_associated = true; o->associated_landmark = o.get(); // the state of every landmark: it was a candidate once (Tracking.cc:1945-1951)
    g->mos.push_back(std::move(o));
    return (int)g->mos.size() - 1;
}
API void ref_graph_kf_covisible(ref_graph *g, int kf, int other) { g->kfs[kf]->covisible.push_back(g->kfs[other].get()); }
API void ref_graph_kf_match(ref_graph *g, int kf, int key, int mp) { g->kfs[kf]->mvpMapPoints[key] = mp < 0 ? nullptr : g->mps[mp].get(); }
API void ref_graph_mp_observe(ref_graph *g, int mp, int kf, int key) { g->mps[mp]->mObservations[g->kfs[kf].get()] = (size_t)key; }
// a detection of a key frame (KeyFrame::local_cuboids entry) and the landmark it was associated with (KeyFrame::cuboids_landmark entry, -1: NULL, -2: no entry)
API int ref_graph_kf_detection(ref_graph *g, int kf, const double *bbox_vec4, const int *bbox_2d4, int left_right_to_car, double meas_quality, int landmark) {
    std::unique_ptr<MapObject> d(new MapObject());
    d->bbox_vec = Eigen::Vector4d(bbox_vec4[0], bbox_vec4[1], bbox_vec4[2], bbox_vec4[3]);
    d->bbox_2d = cv::Rect(bbox_2d4[0], bbox_2d4[1], bbox_2d4[2], bbox_2d4[3]);
    d->left_right_to_car = left_right_to_car; d->meas_quality = meas_quality;
    KeyFrame *k = g->kfs[kf].get();
    d->object_id_in_localKF = (int)k->local_cuboids.size();
    k->local_cuboids.push_back(d.get());
    if (landmark >= -1) k->cuboids_landmark.push_back(landmark < 0 ? nullptr : g->mos[landmark].get());
    g->dets.push_back(std::move(d));
    return (int)k->local_cuboids.size() - 1;
}
API void ref_graph_mo_observe(ref_graph *g, int mo, int kf, int det) { g->mos[mo]->addObservation(g->kfs[kf].get(), (size_t)det); } // MapObject::addObservation (MapObject.cc:117-131)
API void ref_graph_mo_unique_point(ref_graph *g, int mo, int mp, int count) {
    g->mos[mo]->mappoints_unique_own.insert(mp < 0 ? nullptr : g->mps[mp].get());
    if (mp >= 0) g->mps[mp]->MapObjObservations[g->mos[mo].get()] = count;
}
// ---- Tracking::AssociateCuboids: candidates are detections (KeyFrame::local_cuboids entries) that gathered potential points
API void ref_graph_det_candidate(ref_graph *g, int kf, int det, int become_candidate, int already_associated, const double *pose7, const double *scale3) {
    KeyFrame *k = g->kfs[kf].get(); MapObject *d = k->local_cuboids[det];
    d->become_candidate = become_candidate != 0; d->already_associated = already_associated != 0; d->moRefKF = k; d->pose_Twc = cuboid_of(pose7, scale3);
}
API void ref_graph_det_potential_point(ref_graph *g, int kf, int det, int mp) { g->mps[mp]->AddObjectObservation(g->kfs[kf]->local_cuboids[det]); } // a local object: LocalObjObservations + AddPotentialMapPoint
API void ref_graph_mp_vote(ref_graph *g, int mp, int mo, int count) { g->mps[mp]->MapObjObservations[g->mos[mo].get()] = count; }
API void ref_graph_mp_best(ref_graph *g, int mp, int mo, int max_vote) { g->mps[mp]->best_object = mo < 0 ? nullptr : g->mos[mo].get(); g->mps[mp]->max_object_vote = max_vote; }
API void ref_graph_associate_cuboids(ref_graph *g, int cur_kf, const int *local_kfs, int n_local, long next_id, int mono_allframe_depth_init) {
    Quiet q;
    Tracking t; t.mpMap = &g->map;
    for (auto &o : g->mos) g->map.AddMapObject(o.get());
    for (int i = 0; i < n_local; i++) t.mvpLocalKeyFrames.push_back(g->kfs[local_kfs[i]].get());
    MapObject::nNextId = next_id; mono_allframe_Obj_depth_init = mono_allframe_depth_init != 0;
    t.AssociateCuboids(g->kfs[cur_kf].get());
}
// an object pointer as (kind, index): 0 = landmark g->mos[index], 1 = detection (index = its creation order over all key frames), -1 = NULL
static void obj_ref(ref_graph *g, MapObject *o, int *kind, int *index) {
    *kind = -1; *index = -1;
    if (!o) return;
    for (size_t i = 0; i < g->mos.size(); i++) if (g->mos[i].get() == o) { *kind = 0; *index = (int)i; return; }
    for (size_t i = 0; i < g->dets.size(); i++) if (g->dets[i].get() == o) { *kind = 1; *index = (int)i; return; }
}
API int ref_graph_det_global_index(ref_graph *g, int kf, int det) { int k, i; obj_ref(g, g->kfs[kf]->local_cuboids[det], &k, &i); return i; }
API void ref_graph_det_state(ref_graph *g, int kf, int det, int *assoc_kind, int *assoc_index, long *mnId, int *already_associated, int *n_obs, double *scale3) {
    MapObject *d = g->kfs[kf]->local_cuboids[det];
    obj_ref(g, d->associated_landmark, assoc_kind, assoc_index);
    *mnId = d->mnId; *already_associated = d->already_associated ? 1 : 0; *n_obs = d->Observations();
    for (int i = 0; i < 3; i++) scale3[i] = d->pose_Twc.scale[i];
}
// the votes of a point: (kind, index, count) triples + best object and its vote; returns the number of triples
API int ref_graph_mp_votes(ref_graph *g, int mp, int *triples, int cap, int *best_kind, int *best_index, int *max_vote) {
    MapPoint *p = g->mps[mp].get();
    int n = 0;
    for (auto &v : p->MapObjObservations) { if (n < cap) { obj_ref(g, v.first, &triples[3 * n], &triples[3 * n + 1]); triples[3 * n + 2] = v.second; } n++; }
    obj_ref(g, p->best_object, best_kind, best_index); *max_vote = p->max_object_vote;
    return n;
}
API void ref_graph_mo_flags(ref_graph *g, int mo, int *n_obs, int *bad, int *is_good, int *n_unique, int *largest_point_observations, int *n_cuboids_landmark_refs) {
    MapObject *o = g->mos[mo].get();
    *n_obs = o->Observations(); *bad = o->bad ? 1 : 0; *is_good = o->isGood ? 1 : 0; *n_unique = o->NumUniqueMapPoints(); *largest_point_observations = o->largest_point_observations;
    int refs = 0;
    for (auto &k : g->kfs) for (MapObject *l : k->cuboids_landmark) if (l == o) refs++;
    *n_cuboids_landmark_refs = refs;
}

// ---- the dynamic-object BA (Optimizer.cc:1537-2573)
API void ref_graph_set_dyn_params(ref_graph *, int pt_obj_cam, int obj_velo, int obj_cam, double velocity_weight, int dynamic_objects) {
    ba_dyna_pt_obj_cam = pt_obj_cam != 0; ba_dyna_obj_velo = obj_velo != 0; ba_dyna_obj_cam = obj_cam != 0; object_velocity_BA_weight = velocity_weight;
    whether_dynamic_object = dynamic_objects != 0; use_dynamic_klt_features = false;
}
API void ref_graph_kf_stamp(ref_graph *g, int kf, double t) { g->kfs[kf]->mTimeStamp = t; }
API void ref_graph_mp_dynamic(ref_graph *g, int mp, const float *pos_to_obj3, int best_mo) {
    MapPoint *p = g->mps[mp].get();
    p->is_dynamic = true; p->PosToObj = mat_f(3, 1, pos_to_obj3); p->best_object = best_mo < 0 ? nullptr : g->mos[best_mo].get();
}
API void ref_graph_mo_dynamic_pose(ref_graph *g, int mo, int kf, const double *pose7, const double *scale3) {
    g->mos[mo]->allDynamicPoses[g->kfs[kf].get()] = std::make_pair(cuboid_of(pose7, scale3), false);
}
API void ref_graph_mo_velocity(ref_graph *g, int mo, const double *v2) { g->mos[mo]->velocityPlanar = Eigen::Vector2d(v2[0], v2[1]); g->mos[mo]->is_dynamic = true; }
API void ref_graph_local_ba_dynamic(ref_graph *g, int kf, int fix_camera, int fix_point, bool *stop) {
    Quiet q; standin_log = &g->log;
    Optimizer::LocalBACameraPointObjectsDynamic(g->kfs[kf].get(), stop, &g->map, fix_camera != 0, fix_point != 0);
    standin_log = nullptr;
}
// an object's pose in a key frame after the BA (allDynamicPoses[kf]): returns 1 if there is an entry; *baed = its flag
API int ref_graph_mo_dynamic_pose_out(ref_graph *g, int mo, int kf, double *pose7, int *baed) {
    MapObject *o = g->mos[mo].get();
    auto it = o->allDynamicPoses.find(g->kfs[kf].get());
    if (it == o->allDynamicPoses.end()) return 0;
    g2o::Vector7d v = it->second.first.pose.toVector();
    for (int i = 0; i < 7; i++) pose7[i] = v[i];
    *baed = it->second.second ? 1 : 0;
    return 1;
}
API void ref_graph_mo_dynamic_state(ref_graph *g, int mo, double *latest7, double *afterba7, double *velocity2, int *n_history, double *history2, long *local_for) {
    MapObject *o = g->mos[mo].get();
    g2o::Vector7d a = o->pose_Twc_latestKF.pose.toVector(), b = o->pose_Twc_afterba.pose.toVector();
    for (int i = 0; i < 7; i++) { latest7[i] = a[i]; afterba7[i] = b[i]; }
    velocity2[0] = o->velocityPlanar[0]; velocity2[1] = o->velocityPlanar[1];
    *n_history = (int)o->velocityhistory.size();
    if (!o->velocityhistory.empty()) { history2[0] = o->velocityhistory.rbegin()->second[0]; history2[1] = o->velocityhistory.rbegin()->second[1]; }
    *local_for = (long)o->mnBALocalForKF;
}
API void ref_graph_mp_dynamic_out(ref_graph *g, int mp, float *pos_to_obj3, float *latest3, int *is_optimized, int *bad, long *local_for) {
    MapPoint *p = g->mps[mp].get();
    for (int i = 0; i < 3; i++) { pos_to_obj3[i] = p->PosToObj.empty() ? 0.f : p->PosToObj.at<float>(i); latest3[i] = p->mWorldPos_latestKF.empty() ? 0.f : p->mWorldPos_latestKF.at<float>(i); }
    *is_optimized = p->is_optimized ? 1 : 0; *bad = p->bad ? 1 : 0; *local_for = (long)p->mnBALocalForKF;
}

// Converter::toSE3Quat of a 4 x 4 float pose: the estimate a pose vertex starts from (7 doubles [t, qx qy qz qw], SE3Quat::toVector)
API void ref_graph_pose_from_cvmat(const float *Tcw16, double *pose7) {
    g2o::SE3Quat s = Converter::toSE3Quat(mat_f(4, 4, Tcw16));
    g2o::Vector7d v = s.toVector();
    for (int i = 0; i < 7; i++) pose7[i] = v[i];
}
// Converter::toCvMat of a pose: the 4 x 4 float matrix SetPose receives
API void ref_graph_cvmat_from_pose(const double *pose7, float *Tcw16) {
    g2o::SE3Quat s(Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]), Eigen::Vector3d(pose7[0], pose7[1], pose7[2]));
    cv::Mat m = Converter::toCvMat(s);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tcw16[i * 4 + j] = m.at<float>(i, j);
}

// the functions themselves.  stop: NULL or the flag g2o polls
API void ref_graph_local_ba_objects(ref_graph *g, int kf, int fix_camera, int fix_point, bool *stop) {
    Quiet q; standin_log = &g->log;
    Optimizer::LocalBACameraPointObjects(g->kfs[kf].get(), stop, &g->map, fix_camera != 0, fix_point != 0);
    standin_log = nullptr;
}
API void ref_graph_bundle_adjustment(ref_graph *g, int iterations, unsigned long loop_kf, int robust, bool *stop) {
    Quiet q; standin_log = &g->log;
    std::vector<KeyFrame *> kfs; std::vector<MapPoint *> mps;
    for (auto &k : g->kfs) kfs.push_back(k.get());
    for (auto &p : g->mps) mps.push_back(p.get());
    Optimizer::BundleAdjustment(kfs, mps, iterations, stop, loop_kf, robust != 0);
    standin_log = nullptr;
}
API void ref_graph_local_ba(ref_graph *g, int kf, bool *stop) { // Optimizer::LocalBundleAdjustment (:474-825): the window without objects
    Quiet q; standin_log = &g->log;
    Optimizer::LocalBundleAdjustment(g->kfs[kf].get(), stop, &g->map);
    standin_log = nullptr;
}
API void ref_graph_global_ba(ref_graph *g, int iterations, unsigned long loop_kf, int robust, bool *stop) { // Optimizer::GlobalBundleAdjustemnt (:57-62)
    Quiet q; standin_log = &g->log;
    Optimizer::GlobalBundleAdjustemnt(&g->map, iterations, stop, loop_kf, robust != 0);
    standin_log = nullptr;
}
// PoseOptimization of a frame made of key frame `kf`'s pose, key points and matches; outliers out (n_keys bytes), returns the inlier count
API int ref_graph_pose_optimization(ref_graph *g, int kf, float *Tcw16_out, unsigned char *outlier) {
    Quiet q;
    KeyFrame *k = g->kfs[kf].get();
    Frame f;
    f.mTcw = k->Tcw.clone(); f.N = (int)k->mvKeysUn.size(); f.mvpMapPoints = k->mvpMapPoints; f.mvbOutlier.assign(f.N, false);
    f.mvKeysUn = k->mvKeysUn; f.mvuRight = k->mvuRight; f.mvInvLevelSigma2 = k->mvInvLevelSigma2;
    f.fx = k->fx; f.fy = k->fy; f.cx = k->cx; f.cy = k->cy; f.mbf = k->mbf;
    const int n = Optimizer::PoseOptimization(&f);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tcw16_out[i * 4 + j] = f.mTcw.at<float>(i, j);
    for (int i = 0; i < f.N; i++) outlier[i] = f.mvbOutlier[i] ? 1 : 0;
    return n;
}

// what the functions left in the map
API void ref_graph_kf_pose(ref_graph *g, int kf, float *Tcw16, int *n_writes, float *TcwGBA16) {
    KeyFrame *k = g->kfs[kf].get();
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tcw16[i * 4 + j] = k->Tcw.at<float>(i, j);
    if (n_writes) *n_writes = k->n_pose_writes;
    if (TcwGBA16 && !k->mTcwGBA.empty()) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) TcwGBA16[i * 4 + j] = k->mTcwGBA.at<float>(i, j);
}
API void ref_graph_kf_markers(ref_graph *g, int kf, long *local_for, long *fixed_for) { *local_for = (long)g->kfs[kf]->mnBALocalForKF; *fixed_for = (long)g->kfs[kf]->mnBAFixedForKF; }
API void ref_graph_mp_pos(ref_graph *g, int mp, float *pos3, int *n_writes, int *n_normal_updates) {
    MapPoint *p = g->mps[mp].get();
    for (int i = 0; i < 3; i++) pos3[i] = p->mWorldPos.at<float>(i);
    if (n_writes) *n_writes = p->n_pos_writes;
    if (n_normal_updates) *n_normal_updates = p->n_normal_updates;
}
API void ref_graph_mo_state(ref_graph *g, int mo, double *pose7, double *scale3, int *n_writes, int *been_optimized, int *point_threshold, int *n_used, int *n_filtered) {
    MapObject *o = g->mos[mo].get();
    g2o::Vector7d v = o->pose_Twc.pose.toVector();
    for (int i = 0; i < 7; i++) pose7[i] = v[i];
    for (int i = 0; i < 3; i++) scale3[i] = o->pose_Twc.scale[i];
    if (n_writes) *n_writes = o->n_pose_writes;
    if (been_optimized) *been_optimized = o->obj_been_optimized ? 1 : 0;
    if (point_threshold) *point_threshold = o->pointOwnedThreshold;
    if (n_used) *n_used = (int)o->used_points_in_BA.size();
    if (n_filtered) *n_filtered = (int)o->used_points_in_BA_filtered.size();
}
// the (key frame id, map point id) pairs of EraseMapPointMatch, in call order; returns their number
API int ref_graph_erased(ref_graph *g, long *pairs, int cap) {
    int n = 0;
    for (auto &e : g->log.match_erased) { if (n < cap) { pairs[2 * n] = (long)e.first->mnId; pairs[2 * n + 1] = (long)e.second->mnId; } n++; }
    return n;
}
