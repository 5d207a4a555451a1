// TEST INFRASTRUCTURE (never linked into the product): the reference's ORB matcher searches -- ORBmatcher::SearchByProjection(Frame&, const Frame&)
// (orb_object_slam/src/ORBmatcher.cc:1373-1522), SearchByProjection(Frame&, vector<MapPoint*>) (:50-142), SearchForInitialization (:429-542) with
// RadiusByViewingCos, ComputeThreeMaxima, DescriptorDistance and the constants they use, and Frame::GetFeaturesInArea / PosInGrid /
// AssignFeaturesToGrid (src/Frame.cc:303-318, 404-459, 525-535) -- cut out of the reference at build time (oracle/_ref/extracted_match.inc) and
// compiled against stand-ins for the SLAM classes they touch.  The reference's headers pull in DBoW2, g2o, Eigen and the whole map, so Frame,
// MapPoint and the ORBmatcher declaration below carry just the members those functions read, under the reference's names; every statement of the
// searches themselves is the reference's.  tests/test_ref_pins.py runs them next to the oracle's restatement on the same inputs.
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "cvshim.hpp"
#include "../oracle.h"

// ---- float matrices the way cv::MatExpr evaluates them: A * B (+ C) is one gemm with double accumulation and a single rounding to float
namespace cv {
struct MulExpr {
    Mat a, b; double alpha;
    Mat eval(const Mat *c) const {
        Mat r(a.rows, b.cols, CV_32F);
        for (int i = 0; i < a.rows; i++) for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(i, k) * (double)b.at<float>(k, j);
            r.at<float>(i, j) = (float)(s * alpha + (c ? (double)c->at<float>(i, j) * 1.0 : 0.0));
        }
        return r;
    }
    operator Mat() const { return eval(nullptr); }
};
struct NegExpr { Mat m; };
inline MulExpr operator*(const Mat &a, const Mat &b) { return MulExpr{a, b, 1.0}; }
inline MulExpr operator*(const NegExpr &a, const Mat &b) { return MulExpr{a.m, b, -1.0}; }
inline Mat operator+(const MulExpr &e, const Mat &c) { return e.eval(&c); }
inline NegExpr operator-(const Mat &m) { return NegExpr{m}; }
inline Mat operator-(const Mat &a, const Mat &b) { Mat r(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows * a.cols; i++) r.at<float>(i) = a.at<float>(i) - b.at<float>(i); return r; } // CV_32F vectors
inline double norm(const Mat &m) { double s = 0; for (int i = 0; i < m.rows * m.cols; i++) s += (double)m.at<float>(i) * (double)m.at<float>(i); return std::sqrt(s); } // NORM_L2 of CV_32F: double accumulation
} // namespace cv

namespace DBoW2 { typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector; } // DBoW2/FeatureVector.h: a std::map<NodeId, std::vector<unsigned int>>

namespace ORB_SLAM2_m { // (its own namespace: ref_extract_api.cpp holds another stand-in ORBmatcher for the two primitives)
#define FRAME_GRID_ROWS 48 // Frame.h:32-33
#define FRAME_GRID_COLS 64

class KeyFrame;
class MapPoint {
  public:
    // ... and what Fuse asks of a map point: the scale prediction and the distance / normal data are the map's, given here; Replace / AddObservation are recorded
    int pred_level = 0; cv::Mat normal; MapPoint *replaced_by = nullptr; long added_idx = -1;
    bool IsInKeyFrame(KeyFrame *) { return false; }
    float GetMaxDistanceInvariance() { return 1e9f; }
    float GetMinDistanceInvariance() { return 0.f; }
    cv::Mat GetNormal() { return normal.clone(); }
    int PredictScale(const float &, const float &) { return pred_level; }
    void Replace(MapPoint *p) { replaced_by = p; }
    void AddObservation(KeyFrame *, size_t idx) { added_idx = (long)idx; }
    bool is_dynamic = false, mbTrackInView = false, bad = false;
    int mnTrackScaleLevel = 0, nobs = 0;
    float mTrackViewCos = 0, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    cv::Mat pos, desc;
    cv::Mat GetWorldPos() { return pos.clone(); }
    static MapPoint *&current() { static MapPoint *p = nullptr; return p; } // the map point whose descriptor was fetched last (Fuse: the one being searched for)
    cv::Mat GetDescriptor() { current() = this; return desc.clone(); }
    int Observations() { return nobs; }
    bool isBad() { return bad; }
};
class Frame {
  public:
    cv::Mat mTcw;
    float mb = 0, mbf = 0, fx = 0, fy = 0, cx = 0, cy = 0, mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    int N = 0;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier, KeysStatic;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvScaleFactors, mvuRight;
    cv::Mat mDescriptors;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    DBoW2::FeatureVector mFeatVec;
    void AssignFeaturesToGrid();
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1, const int maxLevel = -1) const;
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);
};
class KeyFrame { // the members the two SearchByBoW read
  public:
    std::vector<cv::KeyPoint> mvKeysUn;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors;
    std::vector<bool> KeysStatic;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    // ... and SearchForTriangulation
    int N = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    cv::Mat Ow, Rcw, tcw;
    std::vector<float> mvuRight, mvLevelSigma2, mvScaleFactors;
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }
    std::vector<size_t> queried; std::vector<MapPoint *> queried_for; // (the indices GetMapPoint was asked for, in order: Fuse asks once per fused map point, with its bestIdx)
    MapPoint *GetMapPoint(const size_t &idx) { queried.push_back(idx); queried_for.push_back(MapPoint::current()); return mvpMapPoints[idx]; }
    // ... and Fuse
    float mbf = 0, mfLogScaleFactor = 0, mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
    std::vector<float> mvInvLevelSigma2;
    std::vector<std::vector<std::vector<size_t>>> mGrid;
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r) const;
    bool IsInImage(const float &x, const float &y) const;
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
};
class ORBmatcher {
  public:
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo);
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    bool CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2);
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
  protected:
    float RadiusByViewingCos(const float &viewCos);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);
    float mfNNratio;
    bool mbCheckOrientation;
};
using namespace std;
} // namespace ORB_SLAM2_m


namespace ORB_SLAM2_m { // (its own namespace: ref_extract_api.cpp holds another stand-in ORBmatcher for the two primitives)
#include "extracted_match.inc"
} // namespace ORB_SLAM2_m

namespace {
using ORB_SLAM2_m::Frame; using ORB_SLAM2_m::MapPoint; using ORB_SLAM2_m::ORBmatcher;
void fill_frame(Frame &F, const orc_frame *f, const float *scale_factors, int n_levels) {
    F.N = f->N;
    F.mnMinX = f->minX; F.mnMaxX = f->maxX; F.mnMinY = f->minY; F.mnMaxY = f->maxY;
    F.mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (F.mnMaxX - F.mnMinX);   // Frame.cc:128-129
    F.mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (F.mnMaxY - F.mnMinY);
    F.mvKeysUn.resize(f->N);
    for (int i = 0; i < f->N; i++) { const orc_keypoint &k = f->keysUn[i]; F.mvKeysUn[i] = cv::KeyPoint(k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id); }
    F.mvKeys = F.mvKeysUn;
    F.mDescriptors = cv::Mat(f->N, 32, CV_8UC1);
    for (int i = 0; i < f->N; i++) std::memcpy(F.mDescriptors.ptr<uchar>(i), f->desc + (size_t)i * 32, 32);
    F.mvpMapPoints.assign(f->N, nullptr);
    F.mvbOutlier.assign(f->N, false);
    F.mvuRight.assign(f->N, -1.0f); // monocular
    if (scale_factors) F.mvScaleFactors.assign(scale_factors, scale_factors + n_levels);
    F.AssignFeaturesToGrid();
}
cv::Mat desc_row(const uint8_t *d) { cv::Mat m(1, 32, CV_8UC1); std::memcpy(m.ptr<uchar>(0), d, 32); return m; }
// keypoints that may not be matched: the odd ones are not static (KeysStatic, which GetFeaturesInArea drops too), the even ones hold a map point with observations
void block_train(Frame &F, const uint8_t *train_blocked, std::vector<MapPoint> &holders) {
    if (!train_blocked) return;
    bool any_static = false;
    for (int i = 0; i < F.N; i++) if (train_blocked[i] && (i & 1)) any_static = true;
    if (any_static) F.KeysStatic.assign(F.N, true);
    holders.resize(F.N);
    for (int i = 0; i < F.N; i++) if (train_blocked[i]) {
        if (i & 1) F.KeysStatic[i] = false;
        else { holders[i].nobs = 1; F.mvpMapPoints[i] = &holders[i]; }
    }
}
} // namespace

extern "C" {
int ref_get_features_in_area(const orc_frame *f, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap) {
    Frame F; fill_frame(F, f, nullptr, 0);
    const std::vector<size_t> v = F.GetFeaturesInArea(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int)v[i];
    return (int)v.size();
}
// the arguments of orc_search_by_projection_frame (oracle.h); bMono = true
int ref_search_by_projection_frame(const orc_frame *cur, int n_last, const float *world_pos, const uint8_t *valid, const uint8_t *blocks, const uint8_t *mp_desc,
                                   const int *last_octave, const float *last_angle, const float *Tcw12, float fx, float fy, float cx, float cy, const float *scale_factors,
                                   int n_levels, float th, int check_orientation, const uint8_t *train_blocked, int *train_match) {
    Frame C; fill_frame(C, cur, scale_factors, n_levels);
    C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy; C.mb = 0.1f; C.mbf = 40.f;
    C.mTcw = cv::Mat(4, 4, CV_32F);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) C.mTcw.at<float>(i, j) = Tcw12[i * 4 + j];
    C.mTcw.at<float>(3, 0) = 0; C.mTcw.at<float>(3, 1) = 0; C.mTcw.at<float>(3, 2) = 0; C.mTcw.at<float>(3, 3) = 1;
    std::vector<MapPoint> holders;
    block_train(C, train_blocked, holders);
    Frame L;
    L.N = n_last; L.mTcw = C.mTcw.clone();
    L.mvKeys.resize(n_last); L.mvKeysUn.resize(n_last); L.mvpMapPoints.assign(n_last, nullptr); L.mvbOutlier.assign(n_last, false);
    std::vector<MapPoint> pts(n_last);
    for (int i = 0; i < n_last; i++) {
        L.mvKeys[i].octave = last_octave[i]; L.mvKeysUn[i].octave = last_octave[i]; L.mvKeys[i].angle = last_angle[i]; L.mvKeysUn[i].angle = last_angle[i];
        MapPoint &p = pts[i];
        p.pos = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; k++) p.pos.at<float>(k) = world_pos[i * 3 + k];
        p.desc = desc_row(mp_desc + (size_t)i * 32); p.nobs = blocks[i] ? 1 : 0;
        L.mvpMapPoints[i] = &p;
        if (!valid[i]) { // the three ways a last-frame feature drops out (:1399-1405)
            if (i % 3 == 0) L.mvpMapPoints[i] = nullptr; else if (i % 3 == 1) L.mvbOutlier[i] = true; else p.is_dynamic = true;
        }
    }
    ORBmatcher m(0.9f, check_orientation != 0);
    const int n = m.SearchByProjection(C, L, th, true);
    for (int i = 0; i < C.N; i++) {
        MapPoint *p = C.mvpMapPoints[i];
        train_match[i] = (p && p >= pts.data() && p < pts.data() + n_last) ? (int)(p - pts.data()) : -1;
    }
    return n;
}
// the arguments of orc_search_local_map
int ref_search_local_map(const orc_frame *f, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level, const uint8_t *in_view, const uint8_t *blocks,
                         const uint8_t *mp_desc, const float *scale_factors, int n_levels, float th, float nnratio, const uint8_t *train_blocked, int *train_match) {
    Frame F; fill_frame(F, f, scale_factors, n_levels);
    std::vector<MapPoint> holders;
    block_train(F, train_blocked, holders);
    std::vector<MapPoint> pts(n_mp);
    std::vector<MapPoint *> vp(n_mp);
    for (int i = 0; i < n_mp; i++) {
        MapPoint &p = pts[i];
        p.mbTrackInView = in_view[i] != 0; p.mnTrackScaleLevel = pred_level[i]; p.mTrackViewCos = view_cos[i]; p.mTrackProjX = proj_xy[2 * i]; p.mTrackProjY = proj_xy[2 * i + 1];
        p.desc = desc_row(mp_desc + (size_t)i * 32); p.nobs = blocks[i] ? 1 : 0;
        vp[i] = &p;
    }
    ORBmatcher m(nnratio, true);
    const int n = m.SearchByProjection(F, vp, th);
    for (int i = 0; i < F.N; i++) {
        MapPoint *p = F.mvpMapPoints[i];
        train_match[i] = (p && p >= pts.data() && p < pts.data() + n_mp) ? (int)(p - pts.data()) : -1;
    }
    return n;
}
int ref_search_for_initialization(const orc_frame *f1, const orc_frame *f2, float *prev_matched, int window_size, float nnratio, int check_orientation, int *matches12) {
    Frame F1, F2; fill_frame(F1, f1, nullptr, 0); fill_frame(F2, f2, nullptr, 0);
    std::vector<cv::Point2f> prev(f1->N);
    for (int i = 0; i < f1->N; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
    std::vector<int> m12;
    ORBmatcher m(nnratio, check_orientation != 0);
    const int n = m.SearchForInitialization(F1, F2, prev, m12, window_size);
    for (int i = 0; i < f1->N; i++) { matches12[i] = m12[i]; prev_matched[2 * i] = prev[i].x; prev_matched[2 * i + 1] = prev[i].y; }
    return n;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:171-307) with the arguments of orc_search_by_bow: a skipped key-frame feature has, by turns, no map
// point / a bad one / a dynamic one / a non-static key point; a skipped frame feature is a non-static key point.
static void fill_kf(ORB_SLAM2_m::KeyFrame &K, const orc_frame *f, const int *node, const uint8_t *skip, std::vector<MapPoint> &pts, bool all_ways) {
    K.mvKeysUn.resize(f->N);
    for (int i = 0; i < f->N; i++) { const orc_keypoint &k = f->keysUn[i]; K.mvKeysUn[i] = cv::KeyPoint(k.x, k.y, k.size, k.angle, k.response, k.octave, k.class_id); }
    K.mDescriptors = cv::Mat(f->N, 32, CV_8UC1);
    for (int i = 0; i < f->N; i++) std::memcpy(K.mDescriptors.ptr<uchar>(i), f->desc + (size_t)i * 32, 32);
    pts.assign(f->N, MapPoint());
    K.mvpMapPoints.assign(f->N, nullptr);
    if (all_ways) K.KeysStatic.assign(f->N, true);
    for (int i = 0; i < f->N; i++) {
        K.mvpMapPoints[i] = &pts[i];
        if (skip && skip[i]) {
            const int way = all_ways ? i % 4 : i % 2;
            if (way == 0) K.mvpMapPoints[i] = nullptr; else if (way == 1) pts[i].bad = true; else if (way == 2) pts[i].is_dynamic = true; else K.KeysStatic[i] = false;
        }
        if (node[i] >= 0) K.mFeatVec[(unsigned)node[i]].push_back((unsigned)i);
    }
}
int ref_search_by_bow(const orc_frame *KF, const int *nodeKF, const uint8_t *skipKF, const orc_frame *Ff, const int *nodeF, const uint8_t *skipF, float nnratio, int check_orientation,
                      int *matchesF) {
    ORB_SLAM2_m::KeyFrame K; std::vector<MapPoint> pts;
    fill_kf(K, KF, nodeKF, skipKF, pts, true);
    Frame F; fill_frame(F, Ff, nullptr, 0);
    for (int i = 0; i < Ff->N; i++) if (nodeF[i] >= 0) F.mFeatVec[(unsigned)nodeF[i]].push_back((unsigned)i);
    if (skipF) { F.KeysStatic.assign(Ff->N, true); for (int i = 0; i < Ff->N; i++) if (skipF[i]) F.KeysStatic[i] = false; }
    std::vector<MapPoint *> m;
    ORBmatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByBoW(&K, F, m);
    for (int i = 0; i < Ff->N; i++) matchesF[i] = m[i] ? (int)(m[i] - pts.data()) : -1;
    return n;
}
// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (:544-677) with the arguments of orc_search_by_bow_kf: a skipped feature has no map point or a bad one
int ref_search_by_bow_kf(const orc_frame *K1f, const int *node1, const uint8_t *skip1, const orc_frame *K2f, const int *node2, const uint8_t *skip2, float nnratio, int check_orientation,
                         int *matches12) {
    ORB_SLAM2_m::KeyFrame K1, K2; std::vector<MapPoint> p1, p2;
    fill_kf(K1, K1f, node1, skip1, p1, false); fill_kf(K2, K2f, node2, skip2, p2, false);
    std::vector<MapPoint *> m;
    ORBmatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByBoW(&K1, &K2, m);
    for (int i = 0; i < K1f->N; i++) matches12[i] = m[i] ? (int)(m[i] - p2.data()) : -1;
    return n;
}

// ORBmatcher::SearchForTriangulation (:679-850) with CheckDistEpipolarLine (:152-169), the arguments of orc_search_for_triangulation except the epipole, which the
// reference computes itself from the first key frame's centre Ow (given in the second camera's frame: R2w = I, t2w = 0).  skip = the feature has a map point.
int ref_search_for_triangulation(const orc_frame *F1, const int *node1, const uint8_t *skip1, const float *u_right1, const uint8_t *static1, const orc_frame *F2, const int *node2,
                                 const uint8_t *skip2, const float *u_right2, const uint8_t *static2, const float *F12, const float *Ow3, float fx, float fy, float cx, float cy,
                                 const float *scale_factors2, const float *level_sigma2_2, int n_levels, int only_stereo, int check_orientation, int *matches12) {
    ORB_SLAM2_m::KeyFrame K1, K2; std::vector<MapPoint> p1, p2;
    auto fill = [&](ORB_SLAM2_m::KeyFrame &K, const orc_frame *f, const int *node, const uint8_t *skip, const float *ur, const uint8_t *stat, std::vector<MapPoint> &pts) {
        fill_kf(K, f, node, nullptr, pts, false);
        K.N = f->N;
        for (int i = 0; i < f->N; i++) K.mvpMapPoints[i] = skip[i] ? &pts[i] : nullptr;
        K.mvuRight.assign(ur, ur + f->N);
        if (stat) { K.KeysStatic.assign(f->N, true); for (int i = 0; i < f->N; i++) K.KeysStatic[i] = stat[i] != 0; }
    };
    fill(K1, F1, node1, skip1, u_right1, static1, p1); fill(K2, F2, node2, skip2, u_right2, static2, p2);
    K2.fx = fx; K2.fy = fy; K2.cx = cx; K2.cy = cy;
    K2.mvScaleFactors.assign(scale_factors2, scale_factors2 + n_levels); K2.mvLevelSigma2.assign(level_sigma2_2, level_sigma2_2 + n_levels);
    K1.Ow = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; k++) K1.Ow.at<float>(k) = Ow3[k];
    K2.Rcw = cv::Mat(3, 3, CV_32F); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K2.Rcw.at<float>(i, j) = i == j ? 1.f : 0.f;
    K2.tcw = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; k++) K2.tcw.at<float>(k) = 0.f;
    cv::Mat Fm(3, 3, CV_32F); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Fm.at<float>(i, j) = F12[i * 3 + j];
    std::vector<std::pair<size_t, size_t>> pairs;
    ORBmatcher matcher(0.6f, check_orientation != 0);
    const int n = matcher.SearchForTriangulation(&K1, &K2, Fm, pairs, only_stereo != 0);
    for (int i = 0; i < F1->N; i++) matches12[i] = -1;
    for (auto &pr : pairs) matches12[pr.first] = (int)pr.second;
    return n;
}

// ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (:852-1003) whole, over a key frame at the origin (Rcw = I, tcw = 0, so that the camera coordinates are the
// world coordinates and the caller can state the projection the oracle is given in the same float arithmetic).  KeyFrame::GetFeaturesInArea / IsInImage are the
// reference's (KeyFrame.cc:627-673); the grid is filled like KeyFrame's constructor copies it from the Frame (Frame::AssignFeaturesToGrid).  A map point drops out, by
// turns, as NULL / bad / dynamic.  Out: for every fused map point in order, its index and the key point it went to; returns nFused.
int ref_fuse(const orc_frame *f, const float *u_right, const float *inv_level_sigma2, const uint8_t *keys_static, int n_mp, const float *world_pos, const int *pred_level,
             const uint8_t *drop, const uint8_t *mp_desc, const float *scale_factors, int n_levels, float fx, float fy, float cx, float cy, float bf, float th, int *fused_mp, int *fused_idx) {
    ORB_SLAM2_m::KeyFrame K; std::vector<MapPoint> holders;
    std::vector<int> node(f->N, -1);
    fill_kf(K, f, node.data(), nullptr, holders, false);
    K.mvpMapPoints.assign(f->N, nullptr);
    K.N = f->N; K.fx = fx; K.fy = fy; K.cx = cx; K.cy = cy; K.mbf = bf;
    K.mnMinX = f->minX; K.mnMaxX = f->maxX; K.mnMinY = f->minY; K.mnMaxY = f->maxY;
    K.mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (K.mnMaxX - K.mnMinX);
    K.mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (K.mnMaxY - K.mnMinY);
    K.mvuRight.assign(u_right, u_right + f->N);
    K.mvScaleFactors.assign(scale_factors, scale_factors + n_levels); K.mvInvLevelSigma2.assign(inv_level_sigma2, inv_level_sigma2 + n_levels);
    if (keys_static) { K.KeysStatic.assign(f->N, true); for (int i = 0; i < f->N; i++) K.KeysStatic[i] = keys_static[i] != 0; }
    { Frame F; fill_frame(F, f, nullptr, 0); // the grid of the frame the key frame was made from (KeyFrame.cc:51-57)
      K.mGrid.resize(FRAME_GRID_COLS);
      for (int i = 0; i < FRAME_GRID_COLS; i++) { K.mGrid[i].resize(FRAME_GRID_ROWS); for (int j = 0; j < FRAME_GRID_ROWS; j++) K.mGrid[i][j] = F.mGrid[i][j]; } }
    K.Rcw = cv::Mat(3, 3, CV_32F); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K.Rcw.at<float>(i, j) = i == j ? 1.f : 0.f;
    K.tcw = cv::Mat(3, 1, CV_32F); K.Ow = cv::Mat(3, 1, CV_32F); for (int k = 0; k < 3; k++) { K.tcw.at<float>(k) = 0.f; K.Ow.at<float>(k) = 0.f; }
    std::vector<MapPoint> pts(n_mp); std::vector<MapPoint *> vp(n_mp);
    for (int i = 0; i < n_mp; i++) {
        MapPoint &p = pts[i];
        p.pos = cv::Mat(3, 1, CV_32F); p.normal = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) { p.pos.at<float>(k) = world_pos[i * 3 + k]; p.normal.at<float>(k) = world_pos[i * 3 + k]; } // seen head-on: PO . Pn = |PO|^2 >= 0.5 |PO|
        p.desc = desc_row(mp_desc + (size_t)i * 32); p.pred_level = pred_level[i];
        vp[i] = &p;
        if (drop[i]) { if (i % 3 == 0) vp[i] = nullptr; else if (i % 3 == 1) p.bad = true; else p.is_dynamic = true; }
    }
    ORBmatcher matcher(0.6f, true);
    const int n = matcher.Fuse(&K, vp, th);
    int k = 0;
    for (size_t q = 0; q < K.queried.size(); q++, k++) { fused_idx[k] = (int)K.queried[q]; fused_mp[k] = (int)(K.queried_for[q] - pts.data()); }
    return n;
}
}
