// Optimizer_hip.cc -- replaces the body of ORB_SLAM2::Optimizer::BundleAdjustment (orb_object_slam/src/Optimizer.cc:64-251) in the
// reference's build: same signature (include/Optimizer.h:39-41), the g2o graph replaced by a flattening of the key frames / map points into
// cs_ba_problem arrays, cs_ba_optimize, and the write-back the reference does (:208-250).  Compile it INSTEAD of that function (the rest of
// Optimizer.cc -- PoseOptimization, the essential graph, Sim3 -- is untouched).  Needs the reference's SLAM headers, OpenCV and Eigen:
// it is not compiled in this repository's environment (adapters/README.md).
//
// Optimizer::LocalBACameraPointObjects (:826-1534, second half of this file): the window is gathered from the map exactly as :829-913 does,
// flattened into cubeslam::LocalWindow, and handed to cube_slam_amd/host/local_ba_objects.hpp, which holds the graph-level rules (which points
// and objects enter, information weights, the 5 + 10 two-stage scheme with re-levelling) -- the C++ twin of cube_slam_amd/ba_objects.py, both
// tested against the oracle's restatement (tests/test_local_ba_objects.py).  Write-back and erasure as :1477-1533.
//
// Also here, with the reference's signatures (include/Optimizer.h:42-51): Optimizer::GlobalBundleAdjustemnt (:57-62: all key frames and map points into
// BundleAdjustment), Optimizer::LocalBundleAdjustment (:474-825: the same window and two-stage scheme as LocalBACameraPointObjects without the objects -- the
// fallback LocalMapping.cc:74 takes when no object BA is wanted) and Optimizer::PoseOptimization(Frame *) (:253-472: the frame's matches flattened into
// cs_pose_optimization, which runs the 4 x 10 rounds with the re-classification in between; three calls per frame, Tracking.cc:1180,1321,1366).
//
// Optimizer::LocalBACameraPointObjectsDynamic (:1537-2573, LocalMapping.cc:66): the window of :1540-1665 gathered here (with its side effect: a dynamic point of another key
// frame with one observation is set bad), flattened into cubeslam::DynamicWindow and handed to cube_slam_amd/host/local_ba_dynamic.hpp (the graph-level rules over
// cs_ba_dyn_*: one cuboid vertex per (object, observing key frame), dynamic points in their object's frame, velocity vertices and motion edges); write-back as :2446-2572.
#include "Optimizer.h"

#include <map>
#include <stdexcept>
#include <vector>

#include "Converter.h"
#include "Frame.h"
#include "cubeslam_hip.h"
#include "cube_slam_amd/host/local_ba_dynamic.hpp"
#include "cube_slam_amd/host/local_ba_objects.hpp"
#include "MapObject.h"
#include "Parameters.h"
#include "g2o_Object.h"

namespace ORB_SLAM2 {
namespace {
// One context (device + stream) per CALLING THREAD: a cs_ctx is not thread-safe, and the SLAM system calls into this unit from several threads
// (tracking, local mapping, the detached global-BA thread).  Created at the thread's first call, destroyed when the thread exits.
struct ThreadCtx { cs_ctx *c = nullptr; ~ThreadCtx() { if (c) cs_destroy(c); } };
cs_ctx *shared_ctx() {
    thread_local ThreadCtx t;
    if (!t.c && cs_create(0, &t.c) != CS_OK) throw std::runtime_error("Optimizer (HIP): no device -- there is no CPU path");
    return t.c;
}
void pose_to_vec7(const cv::Mat &Tcw, double *v) { // SE3Quat::toVector of Converter::toSE3Quat(Tcw): t, then the unit quaternion (x y z w)
    const g2o::SE3Quat q = Converter::toSE3Quat(Tcw);
    const Eigen::Matrix<double, 7, 1> x = q.toVector();
    for (int i = 0; i < 7; i++) v[i] = x[i];
}
cubeslam::Context &window_ctx() { // the host mirrors' handle on a context: one per calling thread (LocalMapping); throws without a device: there is no CPU path
    thread_local cubeslam::Context c(0);
    return c;
}
cv::Mat vec7_to_pose(const double *v) {
    Eigen::Matrix<double, 7, 1> x;
    for (int i = 0; i < 7; i++) x[i] = v[i];
    g2o::SE3Quat q;
    q.fromVector(x);
    return Converter::toCvMat(q);
}
} // namespace

void Optimizer::BundleAdjustment(const std::vector<KeyFrame *> &vpKFs, const std::vector<MapPoint *> &vpMP, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF,
                                 const bool bRobust) {
    // ---- vertices: key frames that are not bad (:88-101); camera index = position in `cams`
    std::vector<KeyFrame *> cams;
    std::map<KeyFrame *, int> cam_index;
    long unsigned int maxKFid = 0;
    for (KeyFrame *pKF : vpKFs) {
        if (pKF->isBad()) continue;
        cam_index[pKF] = (int)cams.size();
        cams.push_back(pKF);
        if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
    }
    if (cams.empty()) return;
    std::vector<double> cam_pose(cams.size() * 7);
    std::vector<uint8_t> cam_fixed(cams.size());
    for (size_t i = 0; i < cams.size(); i++) { pose_to_vec7(cams[i]->GetPose(), &cam_pose[i * 7]); cam_fixed[i] = cams[i]->mnId == 0; }

    // ---- map points with at least one usable observation (:107-205), their reprojection edges in the reference's insertion order
    std::vector<bool> vbNotIncludedMP(vpMP.size(), true);
    std::vector<int> point_of_mp(vpMP.size(), -1);
    std::vector<double> points, obs_uv, obs_ur, obs_inv_sigma2;
    std::vector<int> obs_cam, obs_point;
    bool any_stereo = false;
    for (size_t i = 0; i < vpMP.size(); i++) {
        MapPoint *pMP = vpMP[i];
        if (pMP->isBad()) continue;
        const std::map<KeyFrame *, size_t> observations = pMP->GetObservations();
        const int pid = (int)(points.size() / 3);
        int nEdges = 0;
        for (const auto &ob : observations) {
            KeyFrame *pKF = ob.first;
            if (pKF->isBad() || pKF->mnId > maxKFid) continue;
            const auto ci = cam_index.find(pKF);
            if (ci == cam_index.end()) continue; // optimizer.vertex(pKF->mnId) would be NULL
            const cv::KeyPoint &kpUn = pKF->mvKeysUn[ob.second];
            const float ur = pKF->mvuRight[ob.second];
            obs_cam.push_back(ci->second); obs_point.push_back(pid);
            obs_uv.push_back(kpUn.pt.x); obs_uv.push_back(kpUn.pt.y);
            obs_ur.push_back(ur < 0 ? -1.0 : (double)ur);
            any_stereo = any_stereo || ur >= 0;
            obs_inv_sigma2.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
            nEdges++;
        }
        if (nEdges == 0) continue; // :197-201: the vertex is removed again
        const cv::Mat X = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) points.push_back(X.at<float>(k));
        point_of_mp[i] = pid;
        vbNotIncludedMP[i] = false;
    }

    cs_ba_problem p{};
    p.n_cams = (int)cams.size(); p.cam_pose = cam_pose.data(); p.cam_fixed = cam_fixed.data();
    p.n_points = (int)(points.size() / 3); p.points = points.data();
    p.n_obs = (int)obs_cam.size(); p.obs_cam = obs_cam.data(); p.obs_point = obs_point.data(); p.obs_uv = obs_uv.data(); p.obs_inv_sigma2 = obs_inv_sigma2.data();
    p.obs_ur = any_stereo ? obs_ur.data() : nullptr;
    KeyFrame *k0 = cams[0]; // one camera model per map (the reference copies pKF->fx ... into every edge)
    p.fx = k0->fx; p.fy = k0->fy; p.cx = k0->cx; p.cy = k0->cy; p.bf = k0->mbf;
    p.huber_mono = bRobust ? (double)(float)std::sqrt(5.99) : 0.0;    // `const float thHuber2D = sqrt(5.99)` :100: the double root, rounded to float
    p.huber_stereo = bRobust ? (double)(float)std::sqrt(7.815) : 0.0; // thHuber3D :101

    cs_ctx *ctx = shared_ctx();
    cs_ba *ba = nullptr;
    if (cs_ba_create(ctx, &p, 0, 1, &ba) != CS_OK) throw std::runtime_error(std::string("Optimizer (HIP): ") + cs_last_error(ctx));
    // optimizer.setForceStopFlag(pbStopFlag) :80-81: the caller's bool itself is polled between iterations and LM trials (GlobalBundleAdjustemnt hands in
    // mbStopGBA, which another thread raises)
    cs_ba_set_stop_flag_bool(ba, reinterpret_cast<const volatile unsigned char *>(pbStopFlag));
    static_assert(sizeof(bool) == 1, "the library polls the caller's bool as one byte");
    cs_ba_stats st;
    const int r = cs_ba_optimize(ctx, ba, nIterations, nullptr, &st);
    if (r == CS_OK) cs_ba_read(ctx, ba, cam_pose.data(), points.data(), nullptr);
    cs_ba_destroy(ctx, ba);
    if (r != CS_OK) throw std::runtime_error(std::string("Optimizer (HIP): ") + cs_last_error(ctx));

    // ---- write-back (:208-250)
    for (size_t i = 0; i < cams.size(); i++) {
        KeyFrame *pKF = cams[i];
        const cv::Mat Tcw = vec7_to_pose(&cam_pose[i * 7]);
        if (nLoopKF == 0) pKF->SetPose(Tcw);
        else { pKF->mTcwGBA.create(4, 4, CV_32F); Tcw.copyTo(pKF->mTcwGBA); pKF->mnBAGlobalForKF = nLoopKF; }
    }
    for (size_t i = 0; i < vpMP.size(); i++) {
        if (vbNotIncludedMP[i]) continue;
        MapPoint *pMP = vpMP[i];
        if (pMP->isBad()) continue;
        cv::Mat X(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) X.at<float>(k) = (float)points[(size_t)point_of_mp[i] * 3 + k];
        if (nLoopKF == 0) { pMP->SetWorldPos(X); pMP->UpdateNormalAndDepth(); }
        else { pMP->mPosGBA.create(3, 1, CV_32F); X.copyTo(pMP->mPosGBA); pMP->mnBAGlobalForKF = nLoopKF; }
    }
}


namespace {
// the window BA of LocalBACameraPointObjects (with_objects) and of LocalBundleAdjustment (!with_objects: :474-825 is the same function without the object
// vertices and edges; what differs is noted where it differs)
void local_window_ba(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera, bool with_objects) {
    // ---- the window :829-913 (pointer walking stays here; the marker fields are the reference's)
    std::vector<KeyFrame *> lLocalKeyFrames{pKF};
    pKF->mnBALocalForKF = pKF->mnId;
    for (KeyFrame *pKFi : pKF->GetVectorCovisibleKeyFrames()) { pKFi->mnBALocalForKF = pKF->mnId; if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi); }
    std::vector<MapPoint *> lLocalMapPoints;
    for (KeyFrame *k : lLocalKeyFrames)
        for (MapPoint *pMP : k->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                if (whether_dynamic_object && pMP->is_dynamic) continue;
                lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId;
            }
    std::vector<MapObject *> lLocalMapObjects;
    if (with_objects)
    for (KeyFrame *k : lLocalKeyFrames)
        for (MapObject *pMO : k->cuboids_landmark)
            if (pMO && !pMO->isBad() && pMO->mnBALocalForKF != pKF->mnId) { lLocalMapObjects.push_back(pMO); pMO->mnBALocalForKF = pKF->mnId; }
    std::vector<KeyFrame *> lFixedCameras;
    auto add_fixed = [&](KeyFrame *pKFi) {
        if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) { pKFi->mnBAFixedForKF = pKF->mnId; if (!pKFi->isBad()) lFixedCameras.push_back(pKFi); }
    };
    for (MapPoint *pMP : lLocalMapPoints) for (auto &ob : pMP->GetObservations()) add_fixed(ob.first);
    for (MapObject *pMO : lLocalMapObjects) for (auto &ob : pMO->GetObservations()) add_fixed(ob.first);

    // ---- flatten
    cubeslam::LocalWindow w;
    std::vector<KeyFrame *> kfs(lLocalKeyFrames);
    kfs.insert(kfs.end(), lFixedCameras.begin(), lFixedCameras.end());
    std::map<KeyFrame *, int> kf_row;
    w.n_local = (int)lLocalKeyFrames.size();
    w.kf_pose.resize(kfs.size() * 7);
    for (size_t i = 0; i < kfs.size(); i++) { kf_row[kfs[i]] = (int)i; w.kf_id.push_back((long)kfs[i]->mnId); pose_to_vec7(kfs[i]->GetPose(), &w.kf_pose[i * 7]); }
    const cv::Mat Ow = pKF->GetCameraCenter();
    for (int a = 0; a < 3; a++) w.cur_cam_center[a] = Ow.at<float>(a);
    for (size_t j = 0; j < lLocalMapPoints.size(); j++) {
        MapPoint *pMP = lLocalMapPoints[j];
        const cv::Mat X = pMP->GetWorldPos();
        for (int a = 0; a < 3; a++) w.mp_pos.push_back(X.at<float>(a));
        w.mp_nobs.push_back(pMP->Observations());
        for (auto &ob : pMP->GetObservations()) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->isBad()) continue;
            if (!with_objects && pKFi->KeysStatic.size() > 0 && !pKFi->KeysStatic[ob.second]) continue; // :603-607 (LocalBundleAdjustment only): a dynamic 2D feature
            const cv::KeyPoint &kpUn = pKFi->mvKeysUn[ob.second];
            w.obs_mp.push_back((int)j); w.obs_kf.push_back(kf_row.at(pKFi)); w.obs_uv.push_back(kpUn.pt.x); w.obs_uv.push_back(kpUn.pt.y);
            w.obs_ur.push_back(pKFi->mvuRight[ob.second] < 0 ? -1.0 : (double)pKFi->mvuRight[ob.second]);
            w.obs_inv_sigma2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    std::vector<MapPoint *> up_point; // the map point behind every row of w.up_*
    for (size_t i = 0; i < lLocalMapObjects.size(); i++) {
        MapObject *pMO = lLocalMapObjects[i];
        const g2o::cuboid cube = pMO->GetWorldPos();
        const Eigen::Matrix<double, 7, 1> pv = cube.pose.toVector();
        for (int a = 0; a < 7; a++) w.mo_pose.push_back(pv[a]);
        for (int a = 0; a < 3; a++) w.mo_scale.push_back(cube.scale[a]);
        w.mo_meas_quality.push_back(pMO->meas_quality);
        w.mo_largest_point_observations.push_back(pMO->largest_point_observations);
        pMO->point_object_BA_counter++; pMO->used_points_in_BA.clear(); pMO->used_points_in_BA_filtered.clear();                         // :1151-1153
        pMO->pointOwnedThreshold = std::max(int(pMO->largest_point_observations * 0.4), 2);                                              // :1157-1158
        for (MapPoint *pMP : pMO->GetUniqueMapPoints())
            if (pMP && !pMP->isBad()) {
                const cv::Mat X = pMP->GetWorldPos();
                up_point.push_back(pMP);
                w.up_mo.push_back((int)i); w.up_count.push_back(pMP->MapObjObservations[pMO]);
                for (int a = 0; a < 3; a++) w.up_pos.push_back(X.at<float>(a));
            }
        for (auto &ob : pMO->GetObservations()) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->isBad()) continue;
            const MapObject *local_object = pKFi->local_cuboids[ob.second];
            w.det_mo.push_back((int)i); w.det_kf.push_back(kf_row.at(pKFi));
            for (int a = 0; a < 4; a++) w.det_bbox_vec.push_back(local_object->bbox_vec[a]);
            const cv::Rect r = local_object->bbox_2d;
            w.det_bbox_2d.push_back(r.x); w.det_bbox_2d.push_back(r.y); w.det_bbox_2d.push_back(r.width); w.det_bbox_2d.push_back(r.height);
            w.det_left_right_to_car.push_back(local_object->left_right_to_car);
        }
    }
    cubeslam::LocalBAParams prm;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) prm.K[r * 3 + c] = pMap->Kalib(r, c);
    prm.img_width = pMap->img_width; prm.img_height = pMap->img_height; prm.bf = pKF->mbf; prm.camera_object_BA_weight = camera_object_BA_weight;
    prm.kitti = scene_unique_id == kitti; prm.build_worldframe_on_ground = build_worldframe_on_ground; prm.fixCamera = fixCamera;
    if (pbStopFlag && *pbStopFlag) return;                                                                                               // :1386-1388

    cubeslam::Context &ctx = window_ctx();
    cubeslam::LocalBAResult res;
    // LocalMapping::InterruptBA raises *pbStopFlag from another thread: the flag itself goes down (setForceStopFlag :943-944; re-checked between the stages :1392-1396)
    cubeslam::LocalBACameraPointObjects(ctx, w, prm, res, nullptr, pbStopFlag);

    for (int u : res.up_used) lLocalMapObjects[w.up_mo[u]]->used_points_in_BA.push_back(up_point[u]);                     // :1164, read by Tracking.cc:2012 / MapDrawer.cc:146
    for (int u : res.up_filtered) lLocalMapObjects[w.up_mo[u]]->used_points_in_BA_filtered.push_back(up_point[u]);       // :1209
    // ---- erase, write back :1477-1533
    // (the reference writes `if (parallel_mapping) unique_lock<mutex> lock(...)`, Optimizer.cc:1478 / :2447: the lock is the body of the if and is gone before the
    //  write-back starts -- a slip that races Tracking when parallel_mapping is on.  The adapter holds the map mutex over the write-back, which is what the line means.)
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate, std::defer_lock);
    if (parallel_mapping) lock.lock();
    for (auto &e : res.erase) { KeyFrame *pKFi = kfs[e.first]; MapPoint *pMPi = lLocalMapPoints[e.second]; pKFi->EraseMapPointMatch(pMPi); pMPi->EraseObservation(pKFi); }
    for (size_t i = 0; i < lLocalKeyFrames.size(); i++) { lLocalKeyFrames[i]->mnBALocalForKF = 0; lLocalKeyFrames[i]->SetPose(vec7_to_pose(&res.kf_pose[i * 7])); }
    for (MapPoint *pMP : lLocalMapPoints) pMP->mnBALocalForKF = 0;
    for (size_t k = 0; k < res.point_rows.size(); k++) {
        MapPoint *pMP = lLocalMapPoints[res.point_rows[k]];
        if (pMP->Observations() == 1) continue; // :1511-1512, read AFTER the erasures above: a point they left with one observation keeps its position (res.point_unwritten)
        cv::Mat X(3, 1, CV_32F);
        for (int a = 0; a < 3; a++) X.at<float>(a) = (float)res.point_pos[k * 3 + a];
        pMP->SetWorldPos(X); pMP->UpdateNormalAndDepth();
    }
    for (KeyFrame *k : lFixedCameras) { k->mnBAFixedForKF = 0; if (with_objects) k->mnBALocalForKF = 0; } // (:1521-1523 resets both marks, :820-824 the first)
    for (size_t i = 0; i < lLocalMapObjects.size(); i++) {
        MapObject *pMO = lLocalMapObjects[i];
        pMO->mnBALocalForKF = 0; pMO->obj_been_optimized = true;
        g2o::cuboid cube;
        Eigen::Matrix<double, 7, 1> pv;
        for (int a = 0; a < 7; a++) pv[a] = res.object_pose[i * 7 + a];
        cube.pose.fromVector(pv);
        for (int a = 0; a < 3; a++) cube.scale[a] = res.object_scale[i * 3 + a];
        pMO->SetWorldPos(cube);
    }
}
} // namespace

void Optimizer::LocalBACameraPointObjects(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera, bool fixPoint) {
    if (fixPoint) throw std::runtime_error("LocalBACameraPointObjects (HIP): fixPoint is not supported (LocalMapping.cc:68 passes false)");
    local_window_ba(pKF, pbStopFlag, pMap, fixCamera, true);
}

void Optimizer::LocalBACameraPointObjectsDynamic(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera, bool fixPoint) {
    // ---- the window :1540-1665
    std::vector<KeyFrame *> lLocalKeyFrames{pKF};
    pKF->mnBALocalForKF = pKF->mnId;
    for (KeyFrame *pKFi : pKF->GetVectorCovisibleKeyFrames()) { pKFi->mnBALocalForKF = pKF->mnId; if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi); }
    std::vector<MapPoint *> lLocalMapPoints;
    auto take_points = [&](KeyFrame *k, const std::vector<MapPoint *> &vpMPs) {
        for (MapPoint *pMP : vpMPs)
            if (pMP && !pMP->isBad()) {
                if (k != pKF && pMP->is_dynamic && pMP->Observations() == 1) pMP->SetBadFlag();       // :1566-1570 "delete old mappoint which is only observed once" -- and it stays in the list
                if (pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
            }
    };
    for (KeyFrame *k : lLocalKeyFrames) take_points(k, k->GetMapPointMatches());
    if (use_dynamic_klt_features) for (KeyFrame *k : lLocalKeyFrames) take_points(k, k->GetHarrisMapPointMatches());     // :1581-1606
    std::vector<MapObject *> lLocalMapObjects;
    for (KeyFrame *k : lLocalKeyFrames)
        for (MapObject *pMO : k->cuboids_landmark)
            if (pMO && !pMO->isBad() && pMO->mnBALocalForKF != pKF->mnId) { lLocalMapObjects.push_back(pMO); pMO->mnBALocalForKF = pKF->mnId; }
    std::vector<KeyFrame *> lFixedCameras;
    auto add_fixed = [&](KeyFrame *pKFi) {
        if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) { pKFi->mnBAFixedForKF = pKF->mnId; if (!pKFi->isBad()) lFixedCameras.push_back(pKFi); }
    };
    for (MapPoint *pMP : lLocalMapPoints) for (auto &ob : pMP->GetObservations()) add_fixed(ob.first);
    for (MapObject *pMO : lLocalMapObjects)
        for (auto &ob : pMO->GetObservations())
            if ((ob.first->mTimeStamp - pKF->mTimeStamp) > 8.0) add_fixed(ob.first);                                    // :1655 (as written: key frames 8 s NEWER than the current one)

    // ---- flatten
    cubeslam::DynamicWindow w;
    std::vector<KeyFrame *> kfs(lLocalKeyFrames);
    kfs.insert(kfs.end(), lFixedCameras.begin(), lFixedCameras.end());
    std::map<KeyFrame *, int> kf_row;
    std::map<MapObject *, int> mo_row;
    w.n_local = (int)lLocalKeyFrames.size();
    w.kf_pose.resize(kfs.size() * 7);
    unsigned long maxKFid = 0;
    for (size_t i = 0; i < kfs.size(); i++) {
        kf_row[kfs[i]] = (int)i; w.kf_id.push_back((long)kfs[i]->mnId); pose_to_vec7(kfs[i]->GetPose(), &w.kf_pose[i * 7]); w.kf_stamp.push_back(kfs[i]->mTimeStamp);
        const cv::Mat Ow = kfs[i]->GetCameraCenter();
        for (int a = 0; a < 3; a++) w.kf_cam_center.push_back(Ow.at<float>(a));
        if (kfs[i]->mnId > maxKFid) maxKFid = kfs[i]->mnId;
    }
    for (size_t i = 0; i < lLocalMapObjects.size(); i++) mo_row[lLocalMapObjects[i]] = (int)i;
    for (size_t j = 0; j < lLocalMapPoints.size(); j++) {
        MapPoint *pMP = lLocalMapPoints[j];
        const cv::Mat X = pMP->GetWorldPos();
        for (int a = 0; a < 3; a++) w.mp_pos.push_back(X.at<float>(a));
        w.mp_nobs.push_back(pMP->Observations()); w.mp_dynamic.push_back(pMP->is_dynamic ? 1 : 0);
        for (int a = 0; a < 3; a++) w.mp_pos_to_obj.push_back(pMP->is_dynamic && !pMP->PosToObj.empty() ? pMP->PosToObj.at<float>(a) : 0.f);
        MapObject *owner = pMP->is_dynamic ? pMP->GetBelongedObject() : nullptr;
        w.mp_best_mo.push_back(owner && owner->mnBALocalForKF == pKF->mnId && mo_row.count(owner) ? mo_row[owner] : -1);  // :1933-1935
        for (auto &ob : pMP->GetObservations()) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->isBad()) continue;
            const cv::KeyPoint &kpUn = (pMP->is_dynamic && use_dynamic_klt_features) ? pKFi->mvKeysHarris[ob.second] : pKFi->mvKeysUn[ob.second];   // :1972-1975
            w.obs_mp.push_back((int)j); w.obs_kf.push_back(kf_row.at(pKFi)); w.obs_uv.push_back(kpUn.pt.x); w.obs_uv.push_back(kpUn.pt.y);
            w.obs_ur.push_back(pMP->is_dynamic || pKFi->mvuRight[ob.second] < 0 ? -1.0 : (double)pKFi->mvuRight[ob.second]);
            w.obs_inv_sigma2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    std::vector<MapPoint *> up_point;                          // the map point behind every row of w.up_*
    std::vector<std::pair<MapObject *, KeyFrame *>> ov_key;    // the (object, key frame) behind every row of w.ov_*
    int next_vertex_id = (int)maxKFid + 1;                     // maxIdTillObject :1730-1731
    for (size_t i = 0; i < lLocalMapObjects.size(); i++) {
        MapObject *pMO = lLocalMapObjects[i];
        w.mo_id.push_back(pMO->mnId); w.mo_meas_quality.push_back(pMO->meas_quality); w.mo_largest_point_observations.push_back(pMO->largest_point_observations);
        w.mo_velocity.push_back(pMO->velocityPlanar[0]); w.mo_velocity.push_back(pMO->velocityPlanar[1]);
        pMO->bundle_vertex_ids.clear();
        for (auto &ob : pMO->GetObservations()) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->isBad() || !kf_row.count(pKFi)) continue;                                                       // :1742-1746: bad, or not used in this BA
            const auto dp = pMO->allDynamicPoses.find(pKFi);
            if (dp == pMO->allDynamicPoses.end()) throw std::runtime_error("LocalBACameraPointObjectsDynamic (HIP): BA not found frame object pose (the reference exits here, Optimizer.cc:1752-1757)");
            const Eigen::Matrix<double, 7, 1> pv = dp->second.first.pose.toVector();
            const MapObject *local_object = pKFi->local_cuboids[ob.second];
            w.ov_mo.push_back((int)i); w.ov_kf.push_back(kf_row[pKFi]);
            for (int a = 0; a < 7; a++) w.ov_pose.push_back(pv[a]);
            for (int a = 0; a < 4; a++) w.ov_bbox_vec.push_back(local_object->bbox_vec[a]);
            const cv::Rect r = local_object->bbox_2d;
            w.ov_bbox_2d.push_back(r.x); w.ov_bbox_2d.push_back(r.y); w.ov_bbox_2d.push_back(r.width); w.ov_bbox_2d.push_back(r.height);
            w.ov_left_right_to_car.push_back(local_object->left_right_to_car);
            ov_key.emplace_back(pMO, pKFi);
            pMO->bundle_vertex_ids[pKFi] = ++next_vertex_id;                                                           // :1781-1784
        }
        for (KeyFrame *k : pMO->GetObserveFramesSequential())
            if (!k->isBad() && kf_row.count(k)) { w.seq_mo.push_back((int)i); w.seq_kf.push_back(kf_row[k]); }
        pMO->point_object_BA_counter++; pMO->used_points_in_BA.clear(); pMO->used_points_in_BA_filtered.clear();       // :2018-2020
        pMO->pointOwnedThreshold = std::max(int(pMO->largest_point_observations * 0.4), 2);                            // :2024-2025
        for (MapPoint *pMP : pMO->GetUniqueMapPoints())
            if (pMP && !pMP->isBad()) {
                const cv::Mat X = pMP->GetWorldPos();
                up_point.push_back(pMP);
                w.up_mo.push_back((int)i); w.up_count.push_back(pMP->MapObjObservations[pMO]);
                for (int a = 0; a < 3; a++) w.up_pos.push_back(X.at<float>(a));
            }
    }
    cubeslam::DynamicBAParams prm;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) prm.K[r * 3 + c] = pMap->Kalib(r, c);
    prm.img_width = pMap->img_width; prm.img_height = pMap->img_height; prm.bf = pKF->mbf; prm.camera_object_BA_weight = camera_object_BA_weight;
    prm.object_velocity_BA_weight = object_velocity_BA_weight; prm.kitti = scene_unique_id == kitti; prm.build_worldframe_on_ground = build_worldframe_on_ground;
    prm.fixCamera = fixCamera; prm.fixPoint = fixPoint; prm.ba_dyna_pt_obj_cam = ba_dyna_pt_obj_cam; prm.ba_dyna_obj_velo = ba_dyna_obj_velo; prm.ba_dyna_obj_cam = ba_dyna_obj_cam;

    cubeslam::Context &ctx = window_ctx();
    cubeslam::DynamicBAResult res;
    cubeslam::LocalBACameraPointObjectsDynamic(ctx, w, prm, res, nullptr, pbStopFlag);
    for (int u : res.up_used) lLocalMapObjects[w.up_mo[u]]->used_points_in_BA.push_back(up_point[u]);                  // :2032
    for (int u : res.up_filtered) lLocalMapObjects[w.up_mo[u]]->used_points_in_BA_filtered.push_back(up_point[u]);     // :2080
    for (auto &v : res.velocity_init) lLocalMapObjects[v.first]->velocityPlanar = Eigen::Vector2d(v.second.first, v.second.second);   // :2231, written while the graph is built
    if (!res.solved) return;                                                                                           // :2344-2346: stopped before the first optimize -- the marks stay, like there

    // ---- erase, write back :2446-2572
    // (the reference writes `if (parallel_mapping) unique_lock<mutex> lock(...)`, Optimizer.cc:1478 / :2447: the lock is the body of the if and is gone before the
    //  write-back starts -- a slip that races Tracking when parallel_mapping is on.  The adapter holds the map mutex over the write-back, which is what the line means.)
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate, std::defer_lock);
    if (parallel_mapping) lock.lock();
    for (auto &e : res.erase) { KeyFrame *pKFi = kfs[e.first]; MapPoint *pMPi = lLocalMapPoints[e.second]; pKFi->EraseMapPointMatch(pMPi); pMPi->EraseObservation(pKFi); }
    for (size_t i = 0; i < lLocalKeyFrames.size(); i++) { lLocalKeyFrames[i]->mnBALocalForKF = 0; lLocalKeyFrames[i]->SetPose(vec7_to_pose(&res.kf_pose[i * 7])); }
    for (MapPoint *pMP : lLocalMapPoints) pMP->mnBALocalForKF = 0;
    for (size_t k = 0; k < res.point_rows.size(); k++) {
        MapPoint *pMP = lLocalMapPoints[res.point_rows[k]];
        if (pMP->Observations() == 1) continue;            // :2478, read AFTER the erasures (res.point_unwritten)
        cv::Mat X(3, 1, CV_32F);
        for (int a = 0; a < 3; a++) X.at<float>(a) = (float)res.point_pos[k * 3 + a];
        pMP->SetWorldPos(X); pMP->UpdateNormalAndDepth();
    }
    for (KeyFrame *k : lFixedCameras) { k->mnBAFixedForKF = 0; k->mnBALocalForKF = 0; }
    auto cuboid_of = [&](const double *p7) {
        g2o::cuboid cube;
        Eigen::Matrix<double, 7, 1> pv;
        for (int a = 0; a < 7; a++) pv[a] = p7[a];
        cube.pose.fromVector(pv);
        cube.scale = Eigen::Vector3d(1.9420, 0.8143, 0.7631); // the estimate carries the fixed KITTI size (:1779)
        return cube;
    };
    for (size_t v = 0; v < ov_key.size(); v++) ov_key[v].first->allDynamicPoses[ov_key[v].second] = std::make_pair(cuboid_of(&res.vertex_pose[v * 7]), true);   // :2510
    for (size_t i = 0; i < lLocalMapObjects.size(); i++) {
        MapObject *pMO = lLocalMapObjects[i];
        if (!whether_dynamic_object) pMO->mnBALocalForKF = 0;
        pMO->obj_been_optimized = true;
        if (res.object_latest[i] >= 0) {   // (an object without a vertex: the reference reads allDynamicPoses[nullptr] here)
            pMO->pose_Twc_latestKF = cuboid_of(&res.vertex_pose[(size_t)res.object_latest[i] * 7]);
            pMO->SetWorldPos(pMO->pose_Twc_latestKF);
            pMO->pose_Twc_afterba = pMO->pose_Twc_latestKF;
        }
    }
    for (size_t k = 0; k < res.vel_mo.size(); k++) {
        MapObject *pMO = lLocalMapObjects[res.vel_mo[k]];
        pMO->velocityPlanar = Eigen::Vector2d(res.velocity[k * 2], res.velocity[k * 2 + 1]);
        pMO->velocityhistory[pKF] = pMO->velocityPlanar;
    }
    if (ba_dyna_pt_obj_cam) {
        size_t wk = 0;
        for (size_t k = 0; k < res.dpoint_rows.size(); k++) {
            MapPoint *pMP = lLocalMapPoints[res.dpoint_rows[k]];
            const bool has_world = wk < res.dworld_rows.size() && res.dworld_rows[wk] == res.dpoint_rows[k];
            const size_t wrow = wk;
            if (has_world) wk++;
            if (pMP->Observations() == 1) continue;
            cv::Mat L(3, 1, CV_32F);
            for (int a = 0; a < 3; a++) L.at<float>(a) = (float)res.dpoint_local[k * 3 + a];
            pMP->PosToObj = L;
            MapObject *belongedobj = pMP->GetBelongedObject();
            if (!belongedobj || belongedobj->mnBALocalForKF != pKF->mnId || !has_world) continue;   // :2554-2556 (the objects' marks were reset above unless whether_dynamic_object)
            cv::Mat Xw(3, 1, CV_32F);
            for (int a = 0; a < 3; a++) Xw.at<float>(a) = (float)res.dpoint_world[wrow * 3 + a];
            pMP->mWorldPos_latestKF = Xw;
            pMP->SetWorldPos(pMP->mWorldPos_latestKF);
            pMP->is_optimized = true;
        }
    }
    if (whether_dynamic_object) for (MapObject *pMO : lLocalMapObjects) pMO->mnBALocalForKF = 0;
}

void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap) { local_window_ba(pKF, pbStopFlag, pMap, false, false); }

void Optimizer::GlobalBundleAdjustemnt(Map *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust) { // :57-62
    const std::vector<KeyFrame *> vpKFs = pMap->GetAllKeyFrames();
    const std::vector<MapPoint *> vpMP = pMap->GetAllMapPoints();
    BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

int Optimizer::PoseOptimization(Frame *pFrame) {
    // ---- the unary edges :283-379, in key point order (the engine keeps mono and stereo apart like vpEdgesMono / vpEdgesStereo)
    const int N = pFrame->N;
    std::vector<int> idx;
    std::vector<double> Xw, obs, w;
    int nInitialCorrespondences = 0, n_no_edge = 0;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
        for (int i = 0; i < N; i++) {
            MapPoint *pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            if (whether_dynamic_object && pMP->is_dynamic) continue;
            const bool mono = pFrame->mvuRight[i] < 0;
            nInitialCorrespondences++;
            pFrame->mvbOutlier[i] = false;
            if (mono && pFrame->KeysStatic.size() > 0 && !pFrame->KeysStatic[i]) { n_no_edge++; continue; } // :305-309: counted, never an edge (so never an outlier)
            const cv::KeyPoint &kpUn = pFrame->mvKeysUn[i];
            const cv::Mat X = pMP->GetWorldPos();
            idx.push_back(i);
            for (int a = 0; a < 3; a++) Xw.push_back(X.at<float>(a));
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(mono ? -1.0 : (double)pFrame->mvuRight[i]);
            w.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    if (nInitialCorrespondences < 3) return 0; // :381-382
    const int edge_off[2] = {0, (int)idx.size()};
    const double intr[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};
    double pose_in[7], pose_out[7];
    pose_to_vec7(pFrame->mTcw, pose_in);
    std::vector<uint8_t> outlier(idx.size() + 1, 0);
    int n_inliers = 0;
    if (Xw.empty()) { Xw.assign(3, 0.0); obs.assign(3, 0.0); w.assign(1, 0.0); }
    cs_ctx *ctx = shared_ctx();
    if (cs_pose_optimization(ctx, 1, edge_off, Xw.data(), obs.data(), w.data(), intr, pose_in, pose_out, outlier.data(), &n_inliers) != CS_OK)
        throw std::runtime_error(std::string("Optimizer (HIP): ") + cs_last_error(ctx));
    for (size_t k = 0; k < idx.size(); k++) pFrame->mvbOutlier[idx[k]] = outlier[k] != 0; // :400-445
    pFrame->SetPose(vec7_to_pose(pose_out));                                                // :466-469
    return n_inliers + n_no_edge;                                                           // nInitialCorrespondences - nBad
}

} // namespace ORB_SLAM2
