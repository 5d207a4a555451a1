// TEST INFRASTRUCTURE (never linked into the product): adapters/Optimizer_hip.cc -- the translation unit a maintainer compiles INSTEAD of the bodies of
// Optimizer::BundleAdjustment / Optimizer::LocalBACameraPointObjects -- compiled against the runnable stand-ins of the map classes (slam_graph_standins.hpp,
// under the header names the adapter includes: oracle/ref_shim/slam_graph/) and linked with libcubeslam_hip.so, so that on the GPU box the adapter's member
// functions run over the very windows on which oracle/_ref/libref_graph.so runs the reference's own function text (tests/test_adapters_gpu.py): the drop-in
// boundary b3 exercised end to end -- map in, map out -- against the reference itself.  The Converter / MapObject / MapPoint functions the adapter calls are the
// reference's text here too (cut out at build time).
#include "../../adapters/Optimizer_hip.cc"

#include "ref_graph_types.hpp"

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
} // namespace ORB_SLAM2

using namespace ORB_SLAM2;
#define API extern "C" __attribute__((visibility("default")))

API void adp_graph_set_params(int is_kitti, int worldframe_on_ground, double cam_obj_weight) {
    scene_unique_id = is_kitti ? kitti : voidtype; build_worldframe_on_ground = worldframe_on_ground != 0; camera_object_BA_weight = cam_obj_weight;
}
// 0 on success; the adapter throws when there is no device or the library reports an error: the message goes to `err`
API int adp_graph_local_ba_objects(ref_graph *g, int kf, int fix_camera, bool *stop, char *err, int err_cap) {
    standin_log = &g->log;
    int rc = 0;
    try { Optimizer::LocalBACameraPointObjects(g->kfs[kf].get(), stop, &g->map, fix_camera != 0, false); }
    catch (const std::exception &e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } }
    standin_log = nullptr;
    return rc;
}
API int adp_graph_bundle_adjustment(ref_graph *g, int iterations, unsigned long loop_kf, int robust, bool *stop, char *err, int err_cap) {
    standin_log = &g->log;
    int rc = 0;
    try {
        std::vector<KeyFrame *> kfs; std::vector<MapPoint *> mps;
        for (auto &k : g->kfs) kfs.push_back(k.get());
        for (auto &p : g->mps) mps.push_back(p.get());
        Optimizer::BundleAdjustment(kfs, mps, iterations, stop, loop_kf, robust != 0);
    } catch (const std::exception &e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } }
    standin_log = nullptr;
    return rc;
}
API int adp_graph_local_ba(ref_graph *g, int kf, bool *stop, char *err, int err_cap) { // Optimizer::LocalBundleAdjustment
    standin_log = &g->log;
    int rc = 0;
    try { Optimizer::LocalBundleAdjustment(g->kfs[kf].get(), stop, &g->map); }
    catch (const std::exception &e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } }
    standin_log = nullptr;
    return rc;
}
API int adp_graph_global_ba(ref_graph *g, int iterations, unsigned long loop_kf, int robust, bool *stop, char *err, int err_cap) { // Optimizer::GlobalBundleAdjustemnt
    standin_log = &g->log;
    int rc = 0;
    try { Optimizer::GlobalBundleAdjustemnt(&g->map, iterations, stop, loop_kf, robust != 0); }
    catch (const std::exception &e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } }
    standin_log = nullptr;
    return rc;
}
// Optimizer::PoseOptimization over a Frame made of key frame `kf`'s pose, key points and matches (like ref_graph_pose_optimization); -1 on an exception
API int adp_graph_pose_optimization(ref_graph *g, int kf, float *Tcw16_out, unsigned char *outlier, char *err, int err_cap) {
    KeyFrame *k = g->kfs[kf].get();
    Frame f;
    f.mTcw = k->Tcw.clone(); f.N = (int)k->mvKeysUn.size(); f.mvpMapPoints = k->mvpMapPoints; f.mvbOutlier.assign(f.N, false);
    f.mvKeysUn = k->mvKeysUn; f.mvuRight = k->mvuRight; f.mvInvLevelSigma2 = k->mvInvLevelSigma2;
    f.fx = k->fx; f.fy = k->fy; f.cx = k->cx; f.cy = k->cy; f.mbf = k->mbf;
    int n = -1;
    try { n = Optimizer::PoseOptimization(&f); }
    catch (const std::exception &e) { if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } return -1; }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Tcw16_out[i * 4 + j] = f.mTcw.at<float>(i, j);
    for (int i = 0; i < f.N; i++) outlier[i] = f.mvbOutlier[i] ? 1 : 0;
    return n;
}
// Optimizer::LocalBACameraPointObjectsDynamic; the switches of Parameters.h the function reads first
API void adp_graph_set_dyn_params(int pt_obj_cam, int obj_velo, int obj_cam, double velocity_weight, int dynamic_objects) {
    ba_dyna_pt_obj_cam = pt_obj_cam != 0; ba_dyna_obj_velo = obj_velo != 0; ba_dyna_obj_cam = obj_cam != 0; object_velocity_BA_weight = velocity_weight;
    whether_dynamic_object = dynamic_objects != 0; use_dynamic_klt_features = false;
}
API int adp_graph_local_ba_dynamic(ref_graph *g, int kf, int fix_camera, int fix_point, bool *stop, char *err, int err_cap) {
    standin_log = &g->log;
    int rc = 0;
    try { Optimizer::LocalBACameraPointObjectsDynamic(g->kfs[kf].get(), stop, &g->map, fix_camera != 0, fix_point != 0); }
    catch (const std::exception &e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } }
    standin_log = nullptr;
    return rc;
}
