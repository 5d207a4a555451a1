// TEST INFRASTRUCTURE (never linked into the product): adapters/detect_3d_cuboid_hip.cpp -- the translation unit a maintainer compiles INSTEAD of
// detect_3d_cuboid/src/box_proposal_detail.cpp -- compiled against the reference's OWN class definition (detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h,
// object_3d_util.h, matrix_utils.h, included from where they lie) with the stand-ins for Eigen (ref_shim/eigen_full) and OpenCV (ref_shim/cvshim.hpp), linked with
// libcubeslam_hip.so, and driven through the reference's class interface exactly as oracle/ref_shim/ref_geom_api.cpp::ref_detect_cuboid drives the reference's own
// detect_cuboid text: on the GPU box tests/test_adapters_gpu.py holds the two side by side (boundary b1, run).
#include "../../adapters/detect_3d_cuboid_hip.cpp"

#include "../oracle.h"

// object_3d_util.cpp stays in the reference's library in a real build; the adapter calls this one function of it, and only to draw (never in the tests)
void plot_image_with_cuboid(cv::Mat &, const cuboid *) {}
// matrix_utils.cpp's quat_to_euler_zyx is compiled from the reference (oracle/Makefile.ref: gg_matrix_utils.o)

extern "C" __attribute__((visibility("default")))
int adp_detect_cuboid(const uint8_t *gray, int W, int H, const double *K9, const double *Twc16, const double *boxes, int nb, const double *lines, int nl, const orc_cuboid_opts *opts,
                      orc_cuboid *out, int *counts, double *cam_pose_raw_euler3, char *err, int err_cap) {
    try {
        detect_3d_cuboid det;
        det.consider_config_1 = opts->consider_config_1; det.consider_config_2 = opts->consider_config_2;
        det.whether_sample_cam_roll_pitch = opts->whether_sample_cam_roll_pitch; det.whether_sample_bbox_height = opts->whether_sample_bbox_height;
        det.max_cuboid_num = opts->max_cuboid_num; det.nominal_skew_ratio = opts->nominal_skew_ratio; det.max_cut_skew = opts->max_cut_skew;
        det.whether_plot_detail_images = false; det.whether_plot_final_images = false; det.whether_save_final_images = false; det.print_details = false;
        Eigen::Matrix3d K; Eigen::Matrix4d T;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K(i, j) = K9[i * 3 + j];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T(i, j) = Twc16[i * 4 + j];
        det.set_calibration(K);
        cv::Mat img(H, W, CV_8UC1, (void *)gray);
        Eigen::MatrixXd bb(nb, 5), ln(nl, 4);
        for (int i = 0; i < nb; i++) for (int j = 0; j < 5; j++) bb(i, j) = boxes[i * 5 + j];
        for (int i = 0; i < nl; i++) for (int j = 0; j < 4; j++) ln(i, j) = lines[i * 4 + j];
        std::vector<ObjectSet> all;
        det.detect_cuboid(img, T, bb, ln, all);
        for (int b = 0; b < nb; b++) {
            counts[b] = (int)all[b].size();
            for (int k = 0; k < counts[b] && k < opts->max_cuboid_num; k++) {
                const cuboid &c = *all[b][k];
                orc_cuboid &o = out[(size_t)b * opts->max_cuboid_num + k];
                for (int i = 0; i < 3; i++) { o.pos[i] = c.pos(i); o.scale[i] = c.scale(i); }
                o.rotY = c.rotY; o.box_config_type[0] = c.box_config_type(0); o.box_config_type[1] = c.box_config_type(1);
                for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) o.box_corners_2d[i * 8 + j] = c.box_corners_2d(i, j);
                for (int i = 0; i < 3; i++) for (int j = 0; j < 8; j++) o.box_corners_3d_world[i * 8 + j] = c.box_corners_3d_world(i, j);
                for (int i = 0; i < 4; i++) o.rect_detect_2d[i] = c.rect_detect_2d(i);
                o.edge_distance_error = c.edge_distance_error; o.edge_angle_error = c.edge_angle_error; o.normalized_error = c.normalized_error; o.skew_ratio = c.skew_ratio;
                o.down_expand_height = c.down_expand_height; o.camera_roll_delta = c.camera_roll_delta; o.camera_pitch_delta = c.camera_pitch_delta;
                delete all[b][k];
            }
        }
        for (int i = 0; i < 3; i++) cam_pose_raw_euler3[i] = det.cam_pose_raw.euler_angle(i); // what main_obj.cpp reads after the call (:450, :465)
        return 0;
    } catch (const std::exception &e) {
        if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
        return 1;
    }
}
