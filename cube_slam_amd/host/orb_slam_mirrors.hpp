// orb_slam_mirrors.hpp -- dependency-free C++ host-side mirrors of the orb_object_slam / line_lbd entry points on the hot path, over
// the C-ABI (include/cubeslam_hip.h): same class names, constructor arguments, method names and member defaults as the reference;
// plain arrays where the reference passes cv::Mat / KeyFrame*.  The adapters with the reference's exact signatures
// are thin wrappers around these and compile only in a tree that provides OpenCV / Eigen / the ORB-SLAM2 map classes.
//   cubeslam::ORBextractor      ORB_SLAM2::ORBextractor      (orb_object_slam/include/ORBextractor.h:44-112)
//   cubeslam::line_lbd_detect   line_lbd_detect              (line_lbd/include/line_lbd/line_lbd_allclass.h:22-70)
//   cubeslam::Optimizer         ORB_SLAM2::Optimizer         (orb_object_slam/include/Optimizer.h:39-62): BundleAdjustment over the
//                               flattened graph (cs_ba_problem), PoseOptimization over flattened matches,
//                               LocalBACameraPointObjectsDynamic over the flattened dynamic graph (cs_ba_dyn_problem)
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cubeslam_hip.h"
#include "detect_3d_cuboid.hpp" // cubeslam::Context

namespace cubeslam {

inline void check(cs_ctx *ctx, int r, const char *what) {
    if (r != CS_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(r) + "): " + cs_last_error(ctx));
}

class ORBextractor {
  public:
    // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) + the image size the device buffers are for
    ORBextractor(Context &c, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width, int height)
        : ctx_(c), nfeatures_(nfeatures), nlevels_(nlevels), W_(width), H_(height) {
        check(ctx_.ctx, cs_orb_create(ctx_.ctx, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 1, &e_), "cs_orb_create");
    }
    ~ORBextractor() { cs_orb_destroy(ctx_.ctx, e_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;
    // operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors): mask is ignored like in the
    // reference (:1036-1040); descriptors = keypoints.size() x 32 bytes
    void operator()(const uint8_t *gray, int stride, std::vector<cs_keypoint> &keypoints, std::vector<uint8_t> &descriptors) {
        const int cap = nfeatures_ + 4 * nlevels_ + 64;
        keypoints.resize((size_t)cap); descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(ctx_.ctx, cs_orb_extract(ctx_.ctx, e_, gray, 1, stride, keypoints.data(), descriptors.data(), cap, &n), "cs_orb_extract");
        keypoints.resize((size_t)n); descriptors.resize((size_t)n * 32);
    }
    int GetLevels() const { return nlevels_; }
    std::vector<float> GetScaleFactors() const { return table(0); }
    std::vector<float> GetInverseScaleFactors() const { return table(1); }
    std::vector<float> GetScaleSigmaSquares() const { return table(2); }
    std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
    cs_orb *handle() { return e_; }

  private:
    std::vector<float> table(int which) const { std::vector<float> t((size_t)nlevels_); cs_orb_get_table(e_, which, t.data()); return t; }
    Context &ctx_;
    cs_orb *e_ = nullptr;
    int nfeatures_, nlevels_, W_, H_;
};

class line_lbd_detect {
  public:
    line_lbd_detect(Context &c, int width, int height) : ctx_(c), W_(width), H_(height) { check(ctx_.ctx, cs_lsd_create(ctx_.ctx, width, height, 1, &l_), "cs_lsd_create"); }
    ~line_lbd_detect() { cs_lsd_destroy(ctx_.ctx, l_); }
    line_lbd_detect(const line_lbd_detect &) = delete;
    line_lbd_detect &operator=(const line_lbd_detect &) = delete;
    // detect_raw_lines(const cv::Mat& gray_img, std::vector<KeyLine>& keylines_out)
    void detect_raw_lines(const uint8_t *gray, int stride, std::vector<cs_keyline> &keylines_out) {
        keylines_out.resize(cap_);
        int n = 0;
        check(ctx_.ctx, cs_lsd_detect(ctx_.ctx, l_, gray, 1, stride, keylines_out.data(), (int)cap_, &n), "cs_lsd_detect");
        keylines_out.resize((size_t)n);
    }
    // detect_filter_lines(const cv::Mat& gray_img, cv::Mat& linesmat_out): rows of x1 y1 x2 y2 (CV_32F), lineLength > line_length_thres
    void detect_filter_lines(const uint8_t *gray, int stride, std::vector<float> &linesmat_out) {
        linesmat_out.resize(cap_ * 4);
        int n = 0;
        check(ctx_.ctx, cs_lsd_detect_filter_lines(ctx_.ctx, l_, gray, 1, stride, (float)line_length_thres, linesmat_out.data(), (int)cap_, &n), "cs_lsd_detect_filter_lines");
        linesmat_out.resize((size_t)n * 4);
    }
    // get_line_descriptors(gray, keylines, descriptors): 32-byte LBD per KeyLine
    void get_line_descriptors(const uint8_t *gray, int stride, const std::vector<cs_keyline> &keylines, std::vector<uint8_t> &line_descrips) {
        line_descrips.assign(keylines.size() * 32, 0);
        if (keylines.empty()) return;
        check(ctx_.ctx, cs_lbd_compute(ctx_.ctx, gray, W_, H_, stride, keylines.data(), (int)keylines.size(), line_descrips.data(), nullptr), "cs_lbd_compute");
    }
    // public members of the reference class, same names and defaults (line_lbd_allclass.h:27-31, constructor line_lbd/class/line_lbd_allclass.cpp:117-123)
    bool use_LSD = false; // the EDLine detector is not on the hot path: every method here runs the LSD detector
    float line_length_thres = 50;
    int numoctaves_ = 1;
    float octaveratio_ = 2.0f;

  private:
    Context &ctx_;
    cs_lsd *l_ = nullptr;
    int W_, H_;
    size_t cap_ = 20000;
};

struct Optimizer {
    // Optimizer::BundleAdjustment / LocalBACameraPointObjects: the caller flattens key frames, map points, objects and their edges into
    // cs_ba_problem; estimates come back in the same order.  pbStopFlag is polled where g2o polls forceStopFlag.
    static cs_ba_stats BundleAdjustment(Context &c, const cs_ba_problem &problem, int nIterations, const volatile int *pbStopFlag, std::vector<double> &cam_pose,
                                        std::vector<double> &points, std::vector<double> &cuboid_pose) {
        cs_ba *b = nullptr;
        check(c.ctx, cs_ba_create(c.ctx, &problem, 0, 1, &b), "cs_ba_create");
        cs_ba_stats st{};
        int r = cs_ba_optimize(c.ctx, b, nIterations, pbStopFlag, &st);
        cam_pose.assign((size_t)problem.n_cams * 7, 0.0); points.assign((size_t)problem.n_points * 3, 0.0); cuboid_pose.assign((size_t)(problem.n_cuboids > 0 ? problem.n_cuboids : 1) * 7, 0.0);
        if (r == CS_OK) r = cs_ba_read(c.ctx, b, cam_pose.data(), points.data(), cuboid_pose.data());
        cs_ba_destroy(c.ctx, b);
        check(c.ctx, r, "cs_ba_optimize");
        cuboid_pose.resize((size_t)problem.n_cuboids * 7);
        return st;
    }
    // Optimizer::PoseOptimization(Frame*): matched map points Xw (n x 3), observations (u, v, u_right or < 0) (n x 3), invSigma2 (n),
    // intrinsics (fx fy cx cy bf), pose [t, q] in / out, mvbOutlier out; returns nInitialCorrespondences - nBad
    static int PoseOptimization(Context &c, int n, const double *Xw, const double *obs, const double *inv_sigma2, const double intr[5], const double pose_in[7],
                                double pose_out[7], std::vector<uint8_t> &mvbOutlier) {
        const int off[2] = {0, n};
        int n_inl = 0;
        mvbOutlier.assign((size_t)n + 1, 0);
        check(c.ctx, cs_pose_optimization(c.ctx, 1, off, Xw, obs, inv_sigma2, intr, pose_in, pose_out, mvbOutlier.data(), &n_inl), "cs_pose_optimization");
        mvbOutlier.resize((size_t)n);
        return n_inl;
    }
    // Optimizer::LocalBACameraPointObjectsDynamic (Optimizer.cc:2353-2415) on a graph the caller flattened from the map (:1684-2344):
    // optimize(5); reprojection edges with chi2 > 5.991 / 7.815 or a non-positive depth and dynamic-point edges with chi2 > 8 go to level 1
    // and all of them lose their kernel, camera-object edges with |error| > 80 go to level 1; optimize(10).  The estimates in `g` are
    // updated in place; the returned levels are what :2418-2447 turns into vToErase.
    struct DynamicGraph { // owns the arrays cs_ba_dyn_problem points into
        cs_ba_dyn_problem P{};
        std::vector<double> cam_pose, obj_pose, vel, points, dpoints;
        std::vector<uint8_t> obs_level, dobs_level, cobs_level;
        void bind() {
            P.cam_pose = cam_pose.data(); P.obj_pose = obj_pose.data(); P.vel = vel.data(); P.points = points.data(); P.dpoints = dpoints.data();
            P.obs_level = obs_level.data(); P.dobs_level = dobs_level.data(); P.cobs_level = cobs_level.data();
        }
    };
    static void LocalBACameraPointObjectsDynamic(Context &c, DynamicGraph &g, const volatile int *pbStopFlag = nullptr, cs_ba_stats *st1 = nullptr, cs_ba_stats *st2 = nullptr) {
        cs_ba_dyn_problem &P = g.P;
        g.obs_level.resize((size_t)P.n_obs + 1, 0); g.dobs_level.resize((size_t)P.n_dobs + 1, 0); g.cobs_level.resize((size_t)P.n_cobs + 1, 0);
        g.bind();
        auto stage = [&](int iterations, cs_ba_stats *st, std::vector<double> &eo, std::vector<double> &ed, std::vector<double> &ec) {
            cs_ba_dyn *ba = nullptr;
            check(c.ctx, cs_ba_dyn_create(c.ctx, &P, &ba), "cs_ba_dyn_create");
            int r = cs_ba_dyn_optimize(c.ctx, ba, iterations, pbStopFlag, st);
            if (!r) r = cs_ba_dyn_read(c.ctx, ba, g.cam_pose.data(), g.obj_pose.data(), g.vel.data(), g.points.data(), g.dpoints.data());
            eo.assign((size_t)P.n_obs * 3 + 1, 0.0); ed.assign((size_t)P.n_dobs * 2 + 1, 0.0); ec.assign((size_t)P.n_cobs * 4 + 1, 0.0);
            if (!r) r = cs_ba_dyn_errors(c.ctx, ba, nullptr, eo.data(), ed.data(), nullptr, ec.data(), nullptr, nullptr);
            cs_ba_dyn_destroy(c.ctx, ba);
            check(c.ctx, r, "cs_ba_dyn stage");
        };
        std::vector<double> eo, ed, ec;
        stage(5, st1, eo, ed, ec);
        if (pbStopFlag && *pbStopFlag) return; // bDoMore = false
        for (int o = 0; o < P.n_obs; o++) { // :2366-2394
            const double *e = &eo[(size_t)o * 3], w = P.obs_inv_sigma2[o];
            const bool stereo = P.obs_ur && P.obs_ur[o] >= 0;
            const double chi2 = stereo ? ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * w : (e[0] * e[0] + e[1] * e[1]) * w;
            const double *T = &g.cam_pose[(size_t)P.obs_cam[o] * 7], *X = &g.points[(size_t)P.obs_point[o] * 3];
            const double qx = T[3], qy = T[4], qz = T[5], qw = T[6]; // third row of R(q) times X plus t_z: isDepthPositive
            const double z = (2 * (qx * qz - qy * qw)) * X[0] + (2 * (qy * qz + qx * qw)) * X[1] + (1 - 2 * (qx * qx + qy * qy)) * X[2] + T[2];
            if (chi2 > (stereo ? 7.815 : 5.991) || !(z > 0)) g.obs_level[o] = 1;
        }
        for (int o = 0; o < P.n_dobs; o++) { const double *e = &ed[(size_t)o * 2]; if ((e[0] * e[0] + e[1] * e[1]) * P.dobs_inv_sigma2[o] > 8) g.dobs_level[o] = 1; } // :2396-2404
        for (int o = 0; o < P.n_cobs; o++) { const double *e = &ec[(size_t)o * 4]; if (((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) + e[3] * e[3] > 80.0 * 80.0) g.cobs_level[o] = 1; } // :2406-2411
        P.huber_mono = P.huber_stereo = P.huber_dyn = 0; // setRobustKernel(0)
        stage(10, st2, eo, ed, ec);
    }
};

} // namespace cubeslam
