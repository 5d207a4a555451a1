// slam_graph_standins.hpp -- TEST INFRASTRUCTURE (never linked into the product): stand-ins for the SLAM map classes (orb_object_slam/include/{KeyFrame,MapPoint,
// MapObject,Map,Frame,Converter,Parameters}.h) with exactly the members that the graph-building functions of orb_object_slam/src/Optimizer.cc and
// Tracking::AssociateCuboids read and write, so that the TEXT of those functions -- cut out of the reference at build time (oracle/ref_shim/extract_ref.py) --
// compiles and runs on the reference's own vendored g2o (Thirdparty/g2o, compiled whole from where it lies, oracle/Makefile.ref) over a small pointer graph
// that tests/test_ref_graph_pins.py fills through oracle/ref_shim/ref_graph_api.cpp.  The real classes need DBoW2, Pangolin, ROS and OpenCV; what is kept here
// is their data: ids, marker fields, poses as 4 x 4 float cv::Mat, key points, observation maps, detections.  Writes that the functions make (SetPose,
// SetWorldPos, EraseMapPointMatch, EraseObservation) are stored and logged so that the test can read the outcome.
#pragma once
#include <iostream>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "cvshim.hpp"

#include <Eigen/Core>
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <Eigen/StdVector>

#include "Thirdparty/g2o/g2o/core/block_solver.h"
#include "Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_dense.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h" // (the stand-in of oracle/ref_shim/g2o_shadow: Eigen's sparse Cholesky is not in this image)
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
#include "g2o_Object.h" // the reference's own vertex / edge classes of the object BA

#define ROS_ERROR_STREAM(x) do { if (ORB_SLAM2::standin_verbose) std::cerr << x << std::endl; } while (0)
#define ROS_WARN_STREAM(x) do { if (ORB_SLAM2::standin_verbose) std::cerr << x << std::endl; } while (0)
#define ROS_INFO_STREAM(x) do { if (ORB_SLAM2::standin_verbose) std::cerr << x << std::endl; } while (0)
#define ROS_ERROR(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)

namespace ORB_SLAM2 {
extern bool standin_verbose;
// Parameters.h: the globals the cut functions read
extern bool parallel_mapping, whether_dynamic_object, build_worldframe_on_ground, whether_detect_object, associate_point_with_object, bundle_object_opti;
extern bool remove_dynamic_features, use_dynamic_klt_features, mono_firstframe_truth_depth_init, mono_firstframe_Obj_depth_init, mono_allframe_Obj_depth_init;
extern bool enable_ground_height_scale, ba_dyna_pt_obj_cam, ba_dyna_obj_velo, ba_dyna_obj_cam, draw_map_truth_paths, draw_nonlocal_mappoint;
extern double camera_object_BA_weight, object_velocity_BA_weight, delta_t;
enum Scene_Name { voidtype = 0, kitti };
extern Scene_Name scene_unique_id;

class KeyFrame; class MapPoint; class MapObject; class Map;
struct EraseLog { std::vector<std::pair<KeyFrame *, MapPoint *>> match_erased, observation_erased; };
extern EraseLog *standin_log;

struct cmpKeyframe { bool operator()(const KeyFrame *a, const KeyFrame *b) const; }; // MapObject.h: by mnId

class MapPoint {
  public:
    long unsigned int mnId = 0, mnBALocalForKF = 0, mnBAGlobalForKF = 0;
    bool is_dynamic = false, bad = false;
    cv::Mat mWorldPos, mPosGBA; // 3 x 1 float
    std::map<KeyFrame *, size_t> mObservations;
    std::map<MapObject *, int> MapObjObservations;
    int n_pos_writes = 0, n_normal_updates = 0;
    static std::mutex mGlobalMutex;
    // object votes (MapPoint.h:133-137; AddObjectObservation is the reference's text, MapPoint.cc:219-247)
    MapObject *best_object = nullptr; // one point can only belong to at most one object
    int max_object_vote = 0;
    std::set<MapObject *> LocalObjObservations;
    std::mutex mMutexObject;
    void AddObjectObservation(MapObject *obj);
    // dynamic-object BA (:1537-2573)
    cv::Mat mWorldPos_latestKF, PosToObj; // world position at the latest key frame; position in the object's frame
    bool is_optimized = false;
    int n_bad_flags = 0;
    void SetBadFlag() { bad = true; n_bad_flags++; } // (the real one also erases the point's observations: map bookkeeping)
    std::map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    int Observations(); // nObs: a stereo observation counts twice (MapPoint.cc:78-81, 186-189); defined below KeyFrame
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat &p) { p.copyTo(mWorldPos); n_pos_writes++; }
    void UpdateNormalAndDepth() { n_normal_updates++; }
    void EraseObservation(KeyFrame *kf) { if (standin_log) standin_log->observation_erased.push_back(std::make_pair(kf, this)); mObservations.erase(kf); }
    MapObject *GetBelongedObject() { return best_object; }
};

class KeyFrame {
  public:
    long unsigned int mnId = 0, mnFrameId = 0, mnBALocalForKF = 0, mnBAFixedForKF = 0, mnBAGlobalForKF = 0;
    bool bad = false;
    cv::Mat Tcw, Ow, mTcwGBA; // 4 x 4 and 3 x 1 float
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<bool> KeysStatic; // KeyFrame.h: per key point, empty unless dynamic features are tracked (LocalBundleAdjustment skips the others, Optimizer.cc:603-607)
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    double mTimeStamp = 0;
    std::vector<cv::KeyPoint> mvKeysHarris;
    std::vector<MapPoint *> GetHarrisMapPointMatches() { return std::vector<MapPoint *>(); }
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<KeyFrame *> covisible;
    std::vector<MapObject *> local_cuboids, cuboids_landmark;
    int n_pose_writes = 0;
    bool isBad() { return bad; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    void SetPose(const cv::Mat &T) { T.copyTo(Tcw); n_pose_writes++; }
    void EraseMapPointMatch(MapPoint *mp) {
        if (standin_log) standin_log->match_erased.push_back(std::make_pair(this, mp));
        for (auto &m : mvpMapPoints) if (m == mp) m = nullptr; // KeyFrame.cc: the slot the point's observation names is set to NULL
    }
};
inline int MapPoint::Observations() { int n = 0; for (auto &o : mObservations) n += o.first->mvuRight[o.second] >= 0 ? 2 : 1; return n; }
inline bool cmpKeyframe::operator()(const KeyFrame *a, const KeyFrame *b) const { return a->mnId < b->mnId; }

class MapObject {
  public:
    long int mnId = 0;
    long unsigned int mnBALocalForKF = 0;
    bool bad = false, obj_been_optimized = false, is_dynamic = false, isGood = false, already_associated = false, become_candidate = false;
    int point_object_BA_counter = -1, largest_point_observations = 0, pointOwnedThreshold = 0, left_right_to_car = -1, object_id_in_localKF = 0, truth_tracklet_id = -1;
    long unsigned int association_refid_in_tracking = 0;
    std::vector<MapPoint *> used_points_in_BA, used_points_in_BA_filtered;
    std::set<MapPoint *> mappoints_unique_own, mappoints_potential_own; // MapObject.h:171-172 (pointer order, like the reference)
    std::mutex mMutexFeatures, mMutexPos, mMutexParam;
    int nObs = 0, n_bad_flags = 0;
    static long int nNextId;
    std::unordered_map<KeyFrame *, size_t> mObservations;
    std::vector<KeyFrame *> observed_frames;
    cv::Rect bbox_2d, bbox_2d_tight;
    Eigen::Vector4d bbox_vec;
    double meas_quality = 1.0;
    g2o::cuboid pose_Twc, cube_meas, pose_Twc_latestKF, pose_Twc_afterba, pose_noopti;
    Eigen::Vector2d velocityPlanar;
    Vector6d velocityTwist;
    std::map<KeyFrame *, std::pair<g2o::cuboid, bool>, cmpKeyframe> allDynamicPoses;
    std::map<KeyFrame *, Eigen::Vector2d, cmpKeyframe> velocityhistory;
    std::unordered_map<KeyFrame *, int> bundle_vertex_ids;
    MapObject *associated_landmark = nullptr;
    KeyFrame *moRefKF = nullptr, *mLatestKF = nullptr;
    int n_pose_writes = 0;
    bool isBad() { return bad; }
    g2o::cuboid GetWorldPos() { return pose_Twc; }
    void SetWorldPos(const g2o::cuboid &c) { pose_Twc = c; n_pose_writes++; }
    std::unordered_map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    int Observations() { return (int)mObservations.size(); }
    // the reference's text (MapObject.cc:44-131), cut out at build time
    static long int getIncrementedIndex();
    std::vector<MapPoint *> GetUniqueMapPoints();
    int NumUniqueMapPoints();
    void AddUniqueMapPoint(MapPoint *pMP, int obs_num);
    void EraseUniqueMapPoint(MapPoint *pMP, int obs_num);
    std::vector<MapPoint *> GetPotentialMapPoints();
    void AddPotentialMapPoint(MapPoint *pMP);
    bool check_whether_valid_object(int own_point_thre = 30);
    void SetAsLandmark();
    void MergeIntoLandmark(MapObject *otherLocalObject);
    void addObservation(KeyFrame *pKF, size_t idx);
    void SetBadFlag() { bad = true; n_bad_flags++; } // (the real one also unhooks the object from its key frames and points: map bookkeeping)
    std::vector<KeyFrame *> GetObserveFrames() { std::vector<KeyFrame *> v; for (auto &o : mObservations) v.push_back(o.first); return v; }
    std::vector<KeyFrame *> GetObserveFramesSequential() { return observed_frames; }
    KeyFrame *GetReferenceKeyFrame() { return moRefKF; }
    KeyFrame *GetLatestKeyFrame() { return mLatestKF; }
    bool IsInKeyFrame(KeyFrame *kf) { return mObservations.count(kf) > 0; }
    int GetIndexInKeyFrame(KeyFrame *kf) { auto it = mObservations.find(kf); return it == mObservations.end() ? -1 : (int)it->second; }
};

class Map {
  public:
    std::set<MapObject *> mspMapObjects;
    std::vector<KeyFrame *> all_kfs; std::vector<MapPoint *> all_mps; // what GetAllKeyFrames / GetAllMapPoints return (Map.cc: copies of the sets)
    std::vector<KeyFrame *> GetAllKeyFrames() { return all_kfs; }
    std::vector<MapPoint *> GetAllMapPoints() { return all_mps; }
    void AddMapObject(MapObject *pMO) { mspMapObjects.insert(pMO); }
    std::vector<MapObject *> GetAllMapObjects() { return std::vector<MapObject *>(mspMapObjects.begin(), mspMapObjects.end()); }
    int img_width = 0, img_height = 0;
    Eigen::Matrix3d Kalib, invKalib;
    Eigen::Matrix3f Kalib_f, invKalib_f;
    std::mutex mMutexMapUpdate;
};

class Frame { // what PoseOptimization (:253-472) touches
  public:
    cv::Mat mTcw;
    int N = 0;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier, KeysStatic;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int n_pose_writes = 0;
    void SetPose(const cv::Mat &T) { T.copyTo(mTcw); n_pose_writes++; }
};

class Converter { // Converter.h: the conversions the cut functions call (bodies cut out of src/Converter.cc)
  public:
    static g2o::SE3Quat toSE3Quat(const cv::Mat &cvT);
    static cv::Mat toCvMat(const g2o::SE3Quat &SE3);
    static cv::Mat toCvMat(const Eigen::Matrix<double, 4, 4> &m);
    static cv::Mat toCvMat(const Eigen::Matrix<double, 3, 1> &m);
    static Eigen::Matrix<double, 3, 1> toVector3d(const cv::Mat &cvVector);
};

class Tracking { // what Tracking::AssociateCuboids (Tracking.cc:1848-2043) touches
  public:
    std::vector<KeyFrame *> mvpLocalKeyFrames;
    Map *mpMap = nullptr;
    bool use_truth_trackid = false;
    std::unordered_map<int, MapObject *> trackletid_to_landmark;
    void AssociateCuboids(KeyFrame *pKF);
};

class Optimizer { // Optimizer.h:36-49
  public:
    void static BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                 const bool bRobust = true);
    void static LocalBACameraPointObjects(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera = false, bool fixPoint = false);
    void static LocalBACameraPointObjectsDynamic(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera = false, bool fixPoint = false);
    int static PoseOptimization(Frame *pFrame);
    void static GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
};
} // namespace ORB_SLAM2
