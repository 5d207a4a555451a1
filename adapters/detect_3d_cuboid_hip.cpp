// detect_3d_cuboid_hip.cpp -- replaces detect_3d_cuboid/src/box_proposal_detail.cpp in the reference's build: the three member functions of
// class detect_3d_cuboid, with the reference's own class definition (include/detect_3d_cuboid/detect_3d_cuboid.h:53-79), on top of
// libcubeslam_hip.so.  Everything callers read is filled: all_object_cuboids (new cuboid*, owned by the caller like :489 does), cam_pose,
// cam_pose_raw (object_slam/src/main_obj.cpp:450,465 read cam_pose_raw / cam_pose after the call) and cuboids_2d_img.
#include "detect_3d_cuboid/detect_3d_cuboid.h"

#include <stdexcept>
#include <vector>

#include <opencv2/highgui/highgui.hpp>

#include "cubeslam_hip.h"
#include "detect_3d_cuboid/object_3d_util.h" // plot_image_with_cuboid (object_3d_util.cpp stays in the reference's library)

namespace {
// One context (device + stream) per CALLING THREAD: a cs_ctx is not thread-safe, and the SLAM system calls into this unit from several threads
// (tracking, local mapping, the detached global-BA thread).  Created at the thread's first call, destroyed when the thread exits.
struct ThreadCtx { cs_ctx *c = nullptr; ~ThreadCtx() { if (c) cs_destroy(c); } };
cs_ctx *shared_ctx() {
    thread_local ThreadCtx t;
    if (!t.c && cs_create(0, &t.c) != CS_OK) throw std::runtime_error("detect_3d_cuboid (HIP): no device -- there is no CPU path");
    return t.c;
}
} // namespace

void detect_3d_cuboid::set_calibration(const Eigen::Matrix3d &Kalib) {
    cam_pose.Kalib = Kalib;
    cam_pose.invK = Kalib.inverse();
}

// the camera tables of one pose (box_proposal_detail.cpp:42-54); the library computes its own copy on the device, these are for the callers
void detect_3d_cuboid::set_cam_pose(const Eigen::Matrix4d &transToWolrd) {
    cam_pose.transToWolrd = transToWolrd;
    cam_pose.rotationToWorld = transToWolrd.block<3, 3>(0, 0);
    double roll, pitch, yaw;
    quat_to_euler_zyx(Eigen::Quaterniond(cam_pose.rotationToWorld), roll, pitch, yaw); // matrix_utils.cpp:35-46
    cam_pose.euler_angle = Eigen::Vector3d(roll, pitch, yaw);
    cam_pose.invR = cam_pose.rotationToWorld.inverse();
    const Eigen::Matrix4d Tcw = transToWolrd.inverse();
    cam_pose.projectionMatrix = cam_pose.Kalib * Tcw.block<3, 4>(0, 0);
    cam_pose.KinvR = cam_pose.Kalib * cam_pose.invR;
    cam_pose.camera_yaw = yaw;
}

void detect_3d_cuboid::detect_cuboid(const cv::Mat &rgb_img, const Eigen::Matrix4d &transToWolrd, const Eigen::MatrixXd &obj_bbox_coors, Eigen::MatrixXd edges,
                                     std::vector<ObjectSet> &all_object_cuboids) {
    set_cam_pose(transToWolrd);
    cam_pose_raw = cam_pose; // :59
    const int nb = (int)obj_bbox_coors.rows(), nl = (int)edges.rows();
    all_object_cuboids.resize(nb); // :72: resized, not cleared -- what a caller left in the first nb entries is appended to, like the reference's push_back (:536)
    cs_cuboid_opts o;
    cs_cuboid_default_opts(&o);
    o.consider_config_1 = consider_config_1; o.consider_config_2 = consider_config_2;
    o.whether_sample_cam_roll_pitch = whether_sample_cam_roll_pitch; o.whether_sample_bbox_height = whether_sample_bbox_height;
    o.max_cuboid_num = max_cuboid_num; o.nominal_skew_ratio = nominal_skew_ratio; o.max_cut_skew = max_cut_skew;
    double K[9], T[16];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K[i * 3 + j] = cam_pose.Kalib(i, j);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = transToWolrd(i, j);
    std::vector<double> boxes((size_t)nb * 5, 0.0), lines((size_t)nl * 4);
    for (int i = 0; i < nb; i++) for (int j = 0; j < 5 && j < (int)obj_bbox_coors.cols(); j++) boxes[(size_t)i * 5 + j] = obj_bbox_coors(i, j); // only columns 0..3 are read (:107-110)
    for (int i = 0; i < nl; i++) for (int j = 0; j < 4; j++) lines[(size_t)i * 4 + j] = edges(i, j);
    std::vector<cs_cuboid> out((size_t)(nb > 0 ? nb : 1) * max_cuboid_num);
    std::vector<int> counts(nb > 0 ? nb : 1, 0);
    cs_ctx *ctx = shared_ctx();
    if (nb > 0 && cs_cuboid_detect(ctx, rgb_img.data, rgb_img.cols, rgb_img.rows, rgb_img.channels(), (int)rgb_img.step, K, T, boxes.data(), nb, lines.data(), nl, &o, out.data(),
                                   counts.data()) != CS_OK)
        throw std::runtime_error(std::string("detect_3d_cuboid (HIP): ") + cs_last_error(ctx));
    const bool draw = whether_plot_final_images || whether_save_final_images;
    cv::Mat frame_all_cubes_img;
    if (draw) frame_all_cubes_img = rgb_img.clone(); // :74-75
    for (int b = 0; b < nb; b++)
        for (int k = 0; k < counts[b]; k++) {
            const cs_cuboid &c = out[(size_t)b * max_cuboid_num + k];
            cuboid *obj = new cuboid();
            obj->pos = Eigen::Vector3d(c.pos[0], c.pos[1], c.pos[2]);
            obj->scale = Eigen::Vector3d(c.scale[0], c.scale[1], c.scale[2]);
            obj->rotY = c.rotY;
            obj->box_config_type = Eigen::Vector2d(c.box_config_type[0], c.box_config_type[1]);
            obj->box_corners_2d.resize(2, 8);
            obj->box_corners_3d_world.resize(3, 8);
            for (int q = 0; q < 8; q++) {
                obj->box_corners_2d(0, q) = c.box_corners_2d[q]; obj->box_corners_2d(1, q) = c.box_corners_2d[8 + q];
                for (int i = 0; i < 3; i++) obj->box_corners_3d_world(i, q) = c.box_corners_3d_world[i * 8 + q];
            }
            obj->rect_detect_2d = Eigen::Vector4d(c.rect_detect_2d[0], c.rect_detect_2d[1], c.rect_detect_2d[2], c.rect_detect_2d[3]);
            obj->edge_distance_error = c.edge_distance_error; obj->edge_angle_error = c.edge_angle_error; obj->normalized_error = c.normalized_error;
            obj->skew_ratio = c.skew_ratio; obj->down_expand_height = c.down_expand_height;
            obj->camera_roll_delta = c.camera_roll_delta; obj->camera_pitch_delta = c.camera_pitch_delta;
            all_object_cuboids[b].push_back(obj);
            if (draw && k == 0) plot_image_with_cuboid(frame_all_cubes_img, obj); // the best cuboid of every box (:540-543)
        }
    if (whether_save_final_images) cuboids_2d_img = frame_all_cubes_img; // :548-549
    if (whether_plot_final_images) { cv::imshow("frame_all_cubes_img", frame_all_cubes_img); cv::waitKey(0); } // :550-554
    // roll / pitch sampling leaves the sampled pose of the LAST box in cam_pose in the reference (:237, :485; pin D1 of DESIGN.md).  Inside the call the library
    // chains the boxes through that pose like the reference does (cs_cuboid_detect); the caller-visible cam_pose here is the raw pose, which is what main_obj.cpp
    // uses (it reads cam_pose_raw)
}
