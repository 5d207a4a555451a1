// detect_3d_cuboid.hpp -- C++ host-side mirror of the reference class detect_3d_cuboid
// (reference detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:53-79) over the C-ABI.
//
// Two layers:
//   1. cubeslam::detect_3d_cuboid  -- dependency-free (plain arrays); same member names and defaults as the reference.
//   2. ::detect_3d_cuboid adapter  -- the reference's exact signature (cv::Mat / Eigen), compiled only where the
//      caller's tree provides OpenCV and Eigen (they are not in this image).  See INTEGRATION.md.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cubeslam_hip.h"

namespace cubeslam {

struct Context { // one per thread, like the reference objects (not re-entrant)
    cs_ctx *ctx = nullptr;
    explicit Context(int device = 0) {
        int r = cs_create(device, &ctx);
        if (r != CS_OK) throw std::runtime_error("cs_create failed (" + std::to_string(r) + "): no HIP device; there is no CPU path");
    }
    ~Context() { cs_destroy(ctx); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
};

class detect_3d_cuboid {
  public:
    explicit detect_3d_cuboid(Context &c) : ctx_(c) {}
    void set_calibration(const double Kalib[9]) { for (int i = 0; i < 9; i++) K_[i] = Kalib[i]; }

    // img: height x width x channels u8 (1 = gray, 3 = BGR); transToWolrd 4x4 row-major; obj_bbox_coors n x 5
    // [x y w h prob]; edges m x 4.  Returns per box up to max_cuboid_num cuboids, best first (ObjectSet order).
    std::vector<std::vector<cs_cuboid>> detect_cuboid(const unsigned char *img, int width, int height, int channels, int stride,
                                                       const double transToWolrd[16], const std::vector<double> &obj_bbox_coors,
                                                       const std::vector<double> &edges) {
        const int nb = (int)(obj_bbox_coors.size() / 5), nl = (int)(edges.size() / 4);
        cs_cuboid_opts o;
        cs_cuboid_default_opts(&o);
        o.consider_config_1 = consider_config_1; o.consider_config_2 = consider_config_2;
        o.whether_sample_cam_roll_pitch = whether_sample_cam_roll_pitch; o.whether_sample_bbox_height = whether_sample_bbox_height;
        o.max_cuboid_num = max_cuboid_num; o.nominal_skew_ratio = nominal_skew_ratio; o.max_cut_skew = max_cut_skew;
        std::vector<cs_cuboid> out((size_t)(nb > 0 ? nb : 1) * max_cuboid_num);
        std::vector<int> counts(nb > 0 ? nb : 1);
        int r = cs_cuboid_detect(ctx_.ctx, img, width, height, channels, stride, K_, transToWolrd, obj_bbox_coors.data(), nb,
                                 edges.data(), nl, &o, out.data(), counts.data());
        if (r != CS_OK) throw std::runtime_error(std::string("cs_cuboid_detect failed: ") + cs_last_error(ctx_.ctx));
        std::vector<std::vector<cs_cuboid>> res(nb);
        for (int b = 0; b < nb; b++) res[b].assign(out.begin() + (size_t)b * max_cuboid_num, out.begin() + (size_t)b * max_cuboid_num + counts[b]);
        return res;
    }

    // public members of the reference class, same names and defaults (detect_3d_cuboid.h:65-79)
    bool print_details = false;
    bool consider_config_1 = true;
    bool consider_config_2 = true;
    bool whether_sample_cam_roll_pitch = false;
    bool whether_sample_bbox_height = false;
    int max_cuboid_num = 1;
    double nominal_skew_ratio = 1;
    double max_cut_skew = 3;

  private:
    Context &ctx_;
    double K_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
};

} // namespace cubeslam

// ---------------------------------------------------------------------------------------------------------------------
// Adapter with the reference's exact signature.  Compiled in the caller's environment only.
#if defined(CUBESLAM_WITH_OPENCV_EIGEN) && __has_include(<opencv2/core/core.hpp>) && __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#include <opencv2/core/core.hpp>

class cuboid { // detect_3d_cuboid.h:15-36
  public:
    Eigen::Vector3d pos, scale;
    double rotY;
    Eigen::Vector2d box_config_type;
    Eigen::Matrix2Xi box_corners_2d;
    Eigen::Matrix3Xd box_corners_3d_world;
    Eigen::Vector4d rect_detect_2d;
    double edge_distance_error, edge_angle_error, normalized_error, skew_ratio, down_expand_height, camera_roll_delta, camera_pitch_delta;
};
typedef std::vector<cuboid *> ObjectSet;

class detect_3d_cuboid {
  public:
    detect_3d_cuboid() : impl_(ctx_) {}
    void set_calibration(const Eigen::Matrix3d &Kalib) {
        double K[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K[i * 3 + j] = Kalib(i, j);
        impl_.set_calibration(K);
    }
    // detect_3d_cuboid.h:62-63.  all_object_cuboids is resized (not cleared) and filled with new cuboid* owned by the caller,
    // exactly like box_proposal_detail.cpp:72,489.
    void detect_cuboid(const cv::Mat &rgb_img, const Eigen::Matrix4d &transToWolrd, const Eigen::MatrixXd &obj_bbox_coors,
                       Eigen::MatrixXd edges, std::vector<ObjectSet> &all_object_cuboids) {
        impl_.consider_config_1 = consider_config_1; impl_.consider_config_2 = consider_config_2;
        impl_.whether_sample_cam_roll_pitch = whether_sample_cam_roll_pitch; impl_.whether_sample_bbox_height = whether_sample_bbox_height;
        impl_.max_cuboid_num = max_cuboid_num; impl_.nominal_skew_ratio = nominal_skew_ratio; impl_.max_cut_skew = max_cut_skew;
        double T[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = transToWolrd(i, j);
        std::vector<double> boxes((size_t)obj_bbox_coors.rows() * 5), lines((size_t)edges.rows() * 4);
        for (int i = 0; i < obj_bbox_coors.rows(); i++) for (int j = 0; j < 5; j++) boxes[(size_t)i * 5 + j] = obj_bbox_coors(i, j);
        for (int i = 0; i < edges.rows(); i++) for (int j = 0; j < 4; j++) lines[(size_t)i * 4 + j] = edges(i, j);
        auto res = impl_.detect_cuboid(rgb_img.data, rgb_img.cols, rgb_img.rows, rgb_img.channels(), (int)rgb_img.step, T, boxes, lines);
        all_object_cuboids.resize(res.size());
        for (size_t b = 0; b < res.size(); b++)
            for (const cs_cuboid &c : res[b]) {
                cuboid *o = new cuboid();
                o->pos = Eigen::Vector3d(c.pos[0], c.pos[1], c.pos[2]);
                o->scale = Eigen::Vector3d(c.scale[0], c.scale[1], c.scale[2]);
                o->rotY = c.rotY;
                o->box_config_type = Eigen::Vector2d(c.box_config_type[0], c.box_config_type[1]);
                o->box_corners_2d.resize(2, 8);
                o->box_corners_3d_world.resize(3, 8);
                for (int k = 0; k < 8; k++) {
                    o->box_corners_2d(0, k) = c.box_corners_2d[k]; o->box_corners_2d(1, k) = c.box_corners_2d[8 + k];
                    for (int i = 0; i < 3; i++) o->box_corners_3d_world(i, k) = c.box_corners_3d_world[i * 8 + k];
                }
                o->rect_detect_2d = Eigen::Vector4d(c.rect_detect_2d[0], c.rect_detect_2d[1], c.rect_detect_2d[2], c.rect_detect_2d[3]);
                o->edge_distance_error = c.edge_distance_error; o->edge_angle_error = c.edge_angle_error;
                o->normalized_error = c.normalized_error; o->skew_ratio = c.skew_ratio; o->down_expand_height = c.down_expand_height;
                o->camera_roll_delta = c.camera_roll_delta; o->camera_pitch_delta = c.camera_pitch_delta;
                all_object_cuboids[b].push_back(o);
            }
    }
    bool whether_plot_detail_images = false, whether_plot_final_images = false, whether_save_final_images = false, print_details = false;
    bool consider_config_1 = true, consider_config_2 = true, whether_sample_cam_roll_pitch = false, whether_sample_bbox_height = false;
    int max_cuboid_num = 1;
    double nominal_skew_ratio = 1, max_cut_skew = 3;

  private:
    cubeslam::Context ctx_;
    cubeslam::detect_3d_cuboid impl_;
};
#endif
