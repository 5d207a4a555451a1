// Optimizer_hip.cc -- replaces the body of ORB_SLAM2::Optimizer::BundleAdjustment (orb_object_slam/src/Optimizer.cc:64-251) in the
// reference's build: same signature (include/Optimizer.h:39-41), the g2o graph replaced by a flattening of the key frames / map points into
// cs_ba_problem arrays, cs_ba_optimize, and the write-back the reference does (:208-250).  Compile it INSTEAD of that function (the rest of
// Optimizer.cc -- PoseOptimization, the essential graph, Sim3 -- is untouched).  Needs the reference's SLAM headers, OpenCV and Eigen:
// it is not compiled in this repository's environment (adapters/README.md).
//
// Optimizer::LocalBACameraPointObjects (:826-1534) uses the same flattening for its local window plus the cuboid vertices / edges; its
// graph-level rules (which points and objects enter, information weights, the 5 + 10 two-stage scheme with re-levelling) are restated and
// tested on flat arrays in cube_slam_amd/ba_objects.py (LocalBACameraPointObjects there), which is the specification for this file's second
// entry point.
#include "Optimizer.h"

#include <map>
#include <stdexcept>
#include <vector>

#include "Converter.h"
#include "cubeslam_hip.h"

namespace ORB_SLAM2 {
namespace {
cs_ctx *shared_ctx() {
    static cs_ctx *ctx = nullptr;
    if (!ctx && cs_create(0, &ctx) != CS_OK) throw std::runtime_error("Optimizer (HIP): no device -- there is no CPU path");
    return ctx;
}
void pose_to_vec7(const cv::Mat &Tcw, double *v) { // SE3Quat::toVector of Converter::toSE3Quat(Tcw): t, then the unit quaternion (x y z w)
    const g2o::SE3Quat q = Converter::toSE3Quat(Tcw);
    const Eigen::Matrix<double, 7, 1> x = q.toVector();
    for (int i = 0; i < 7; i++) v[i] = x[i];
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
    p.huber_mono = bRobust ? std::sqrt(5.99f) : 0.0;    // thHuber2D, float like :103
    p.huber_stereo = bRobust ? std::sqrt(7.815f) : 0.0; // thHuber3D

    cs_ctx *ctx = shared_ctx();
    cs_ba *ba = nullptr;
    if (cs_ba_create(ctx, &p, 0, 1, &ba) != CS_OK) throw std::runtime_error(std::string("Optimizer (HIP): ") + cs_last_error(ctx));
    volatile int stop = 0; // g2o polls *pbStopFlag (a bool); the library polls an int: mirror it before the call (a flag raised during the solve ends the NEXT call early)
    if (pbStopFlag && *pbStopFlag) stop = 1;
    cs_ba_stats st;
    const int r = cs_ba_optimize(ctx, ba, nIterations, pbStopFlag ? &stop : nullptr, &st);
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

} // namespace ORB_SLAM2
