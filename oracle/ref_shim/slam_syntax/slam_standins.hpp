// slam_standins.hpp -- TEST INFRASTRUCTURE, never executed.  Stand-in DECLARATIONS of the ORB-SLAM / CubeSLAM classes that
// adapters/Optimizer_hip.cc walks, so that the adapter can be type-checked (`g++ -fsyntax-only`, tests/test_adapters.py) where the
// reference's own headers cannot be included: KeyFrame.h / MapPoint.h / MapObject.h / Converter.h pull in DBoW2, g2o's core and the full
// Eigen.  Every member below has the name and type the reference declares (file:line under /root/reference/orb_object_slam/include); only the
// members the adapter touches are listed.  `class Optimizer` itself is NOT restated: the test cuts its declaration out of the reference's
// Optimizer.h and the adapter is checked against that text (Optimizer_decl.inc, written next to this file at test time).
#pragma once
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include <Eigen/Dense>
#include <opencv2/core/core.hpp>

namespace g2o {
class SE3Quat { // Thirdparty/g2o/g2o/types/se3quat.h:43-220
public:
    SE3Quat() {}
    Eigen::Matrix<double, 7, 1> toVector() const { return Eigen::Matrix<double, 7, 1>(); } // :124
    void fromVector(const Eigen::Matrix<double, 7, 1> &) {}                                // :133
};
class Sim3; // types/sim3.h
class cuboid { // include/g2o_Object.h:27-37
public:
    SE3Quat pose;
    Eigen::Vector3d scale;
};
} // namespace g2o

namespace ORB_SLAM2 {
class MapPoint;
class MapObject;
class Map;
class Frame;
class KeyFrame { // KeyFrame.h
public:
    void SetPose(const cv::Mat &Tcw);                          // :49
    cv::Mat GetPose();                                         // :50
    cv::Mat GetCameraCenter();                                 // :52
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames();     // :66
    void EraseMapPointMatch(MapPoint *pMP);                    // :90
    std::vector<MapPoint *> GetMapPointMatches();              // :93
    bool isBad();                                              // :111
    std::vector<MapObject *> local_cuboids;                    // :135
    std::vector<MapObject *> cuboids_landmark;                 // :136
    long unsigned int mnId;                                    // :166
    long unsigned int mnBALocalForKF;                          // :184
    long unsigned int mnBAFixedForKF;                          // :185
    cv::Mat mTcwGBA;                                           // :196
    long unsigned int mnBAGlobalForKF;                         // :198
    const float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0; // :201
    const std::vector<cv::KeyPoint> mvKeysUn;                  // :208
    const std::vector<float> mvuRight;                         // :209
    const std::vector<float> mvInvLevelSigma2;                 // :226
    std::vector<bool> KeysStatic;                              // :146
    std::vector<cv::KeyPoint> mvKeysHarris;                    // :141
    std::vector<MapPoint *> GetHarrisMapPointMatches();        // :144
    const double mTimeStamp = 0;                               // :169
};
class Frame { // Frame.h (what PoseOptimization touches)
public:
    void SetPose(cv::Mat Tcw);                                 // :66
    std::vector<bool> KeysStatic;                              // :109
    static float fx, fy, cx, cy;                               // :136-139
    float mbf;                                                 // :145
    int N;                                                     // :155
    std::vector<cv::KeyPoint> mvKeysUn;                        // :162
    std::vector<float> mvuRight;                               // :166
    std::vector<MapPoint *> mvpMapPoints;                      // :181
    std::vector<bool> mvbOutlier;                              // :184
    cv::Mat mTcw;                                              // :192
    std::vector<float> mvInvLevelSigma2;                       // :209
};
class MapPoint { // MapPoint.h
public:
    void SetWorldPos(const cv::Mat &Pos);                      // :44
    cv::Mat GetWorldPos();                                     // :45
    std::map<KeyFrame *, size_t> GetObservations();            // :52
    int Observations();                                        // :53
    void EraseObservation(KeyFrame *pKF);                      // :57
    bool isBad();                                              // :63
    void UpdateNormalAndDepth();                               // :80
    long unsigned int mnBALocalForKF;                          // :112
    static std::mutex mGlobalMutex;                            // :122
    cv::Mat mPosGBA;                                           // :119
    long unsigned int mnBAGlobalForKF;                         // :120
    bool is_dynamic = false;                                   // :125
    std::map<MapObject *, int> MapObjObservations;             // :136
    void SetBadFlag();                                         // :62
    MapObject *GetBelongedObject();                            // :92
    cv::Mat mWorldPos_latestKF;                                // :126
    cv::Mat PosToObj;                                          // :128
    bool is_optimized = false;                                 // :131
};
struct cmpKeyframe { bool operator()(const KeyFrame *a, const KeyFrame *b) const { return a->mnId < b->mnId; } }; // MapObject.h:20-26
class MapObject { // MapObject.h
public:
    void SetWorldPos(const g2o::cuboid &Pos);                  // :34
    g2o::cuboid GetWorldPos();                                 // :35
    std::unordered_map<KeyFrame *, size_t> GetObservations();  // :42
    bool isBad();                                              // :59
    std::vector<MapPoint *> GetUniqueMapPoints();              // :62
    int largest_point_observations;                            // :69
    int pointOwnedThreshold;                                   // :70
    long unsigned int mnBALocalForKF;                          // :78
    bool obj_been_optimized = false;                           // :79
    int point_object_BA_counter = -1;                          // :80
    std::vector<MapPoint *> used_points_in_BA;                 // :84
    std::vector<MapPoint *> used_points_in_BA_filtered;        // :85
    cv::Rect bbox_2d;                                          // :107
    Eigen::Vector4d bbox_vec;                                  // :108
    double meas_quality;                                       // :110
    int left_right_to_car;                                     // :115
    std::vector<KeyFrame *> GetObserveFramesSequential();      // :45
    long int mnId;                                             // :48
    Eigen::Vector2d velocityPlanar;                            // :91
    g2o::cuboid pose_Twc_latestKF;                             // :92
    std::map<KeyFrame *, Eigen::Vector2d, cmpKeyframe> velocityhistory; // :93
    g2o::cuboid pose_Twc_afterba;                              // :94
    std::map<KeyFrame *, std::pair<g2o::cuboid, bool>, cmpKeyframe> allDynamicPoses; // :96
    std::unordered_map<KeyFrame *, int> bundle_vertex_ids;     // :97
};
class Map { // Map.h
public:
    std::vector<KeyFrame *> GetAllKeyFrames();                 // :52
    std::vector<MapPoint *> GetAllMapPoints();                 // :54
    std::mutex mMutexMapUpdate;                                // :70
    Eigen::Matrix3d Kalib, invKalib;                           // :80
    int img_width, img_height;                                 // :82
};
class Converter { // Converter.h
public:
    static g2o::SE3Quat toSE3Quat(const cv::Mat &cvT);         // :39
    static cv::Mat toCvMat(const g2o::SE3Quat &SE3);           // :42
};
class LoopClosing { // LoopClosing.h:48-50 (only the typedef Optimizer.h names)
public:
    typedef std::map<KeyFrame *, g2o::Sim3 *> KeyFrameAndPose; // (the reference's mapped type is g2o::Sim3 with an aligned allocator)
};
// Parameters.h:32-68
extern bool parallel_mapping;
extern bool whether_dynamic_object;
extern bool build_worldframe_on_ground;
extern bool use_dynamic_klt_features;
extern bool ba_dyna_pt_obj_cam, ba_dyna_obj_velo, ba_dyna_obj_cam;
extern double camera_object_BA_weight;
extern double object_velocity_BA_weight;
enum Scene_Name { voidtype = 0, kitti };
extern Scene_Name scene_unique_id;
} // namespace ORB_SLAM2
