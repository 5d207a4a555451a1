/*
 * oracle/oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the
 * reference (shichaoy/cube_slam) hot path, used as the parity checker by
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  The
 * product (cube_slam_amd/, include/) never includes, links or calls anything
 * in this directory.
 *
 * PINNED TO THE REFERENCE'S OWN CODE: the reference has no
 * tests or golden vectors for this path and its build needs OpenCV / Eigen /
 * ROS, all absent here; oracle/Makefile.ref therefore compiles the reference's
 * translation units where they lie under /root/reference -- whole where they
 * need nothing but OpenCV (against a stand-in for its headers), cut out
 * function by function at build time where they need Eigen / g2o / the SLAM
 * classes (against stand-ins for those) -- into oracle/_ref/libref.so, and
 * tests/test_ref_pins.py demands bit equality of every oracle of the path with
 * it.  The graph-level functions (Optimizer::LocalBACameraPointObjects and
 * its Dynamic variant, BundleAdjustment, PoseOptimization,
 * Tracking::AssociateCuboids) run as text
 * on the reference's vendored g2o, compiled WHOLE against a stand-in for
 * Eigen's interface (oracle/_ref/libref_graph.so): tests/test_ref_graph_pins.py
 * holds the graph-level oracles to what they leave in the map.  What stays
 * restated and unpinned: the third-party primitives that are not in the
 * reference tree (OpenCV imgproc / features2d, Eigen's sparse Cholesky); each
 * file's header says which.
 * Every function cites the reference file:line it follows.  The expected
 * outputs the reference ships (object_slam/data/detect_cuboids_saved.txt, the
 * author's offline MATLAB detections) pin the line + cuboid chain loosely:
 * tests/test_cuboid_oracle.py.
 */
#ifndef CUBESLAM_ORACLE_H
#define CUBESLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ cuboid */

typedef struct orc_cuboid_opts {
    int consider_config_1;             /* detect_3d_cuboid.h:71 */
    int consider_config_2;             /* :72 */
    int whether_sample_cam_roll_pitch; /* :73 */
    int whether_sample_bbox_height;    /* :74 */
    int max_cuboid_num;                /* :76 */
    double nominal_skew_ratio;         /* :77 */
    double max_cut_skew;               /* :78 */
    /* extensions (defaults reproduce box_proposal_detail.cpp:126-128,197) */
    double yaw_range_deg;              /* 45 */
    double yaw_step_deg;               /* 6  */
    int canny_low;                     /* 80 */
    int canny_high;                    /* 200 */
    /* 0: yaw_init from the raw camera pose for every box (pinned, what the
     *    product implements); 1: the reference's stateful cam_pose carry-over
     *    between boxes (box_proposal_detail.cpp:126 reads cam_pose, which
     *    :237/:485 overwrite while sampling roll/pitch). */
    int stateful_cam_pose;
} orc_cuboid_opts;

typedef struct orc_cuboid {
    double pos[3];
    double scale[3];
    double rotY;
    double box_config_type[2];
    int32_t box_corners_2d[16];       /* 2x8 row-major (x0..x7, y0..y7) */
    double box_corners_3d_world[24];  /* 3x8 row-major */
    double rect_detect_2d[4];
    double edge_distance_error;
    double edge_angle_error;
    double normalized_error;
    double skew_ratio;
    double down_expand_height;
    double camera_roll_delta;
    double camera_pitch_delta;
} orc_cuboid;

void orc_cuboid_default_opts(orc_cuboid_opts *o);

/* cv::cvtColor(BGR2GRAY) on u8 */
void orc_bgr2gray(const uint8_t *bgr, int w, int h, uint8_t *gray);

/* cv::Canny(gray(roi), low, high) with aperture 3, L1 gradient.  edges: 0/255, w*h */
void orc_canny_roi(const uint8_t *gray, int W, int H, int x0, int y0, int w, int h,
                   int low, int high, uint8_t *edges);

/* cv::distanceTransform(src, CV_DIST_L2, 3); zero pixels of src are sources */
void orc_dist_transform_3x3(const uint8_t *src, int w, int h, float *dist);

/* merge_break_lines (object_3d_util.cpp:300-376); out has room for n*4; returns rows */
/* the geometry helpers of the proposal construction one at a time (test hook, see cuboid_oracle.cpp) */
int orc_cuboid_geom(int op, const double *in, double *out);
int orc_merge_break_lines(const double *lines, int n, double dist_thre, double angle_thre_deg,
                          double len_thre, double *out);

/* box_edge_sum_dists / box_edge_alignment_angle_error on one proposal
 * corners: 2x8 row-major (already shifted for sum_dists); config 1 or 2 */
double orc_box_edge_sum_dists(const float *dist_map, int w, int h, const double *corners_shift, int config_id);
double orc_box_edge_angle_error(const double *vp_bound_angles /*3x2*/, const double *corners, int config_id);

/* fuse_normalize_scores_v2; keep has room n; scores has room n; returns kept count */
int orc_fuse_normalize_scores(const double *dist_err, const double *angle_err, int n, double weight_vp_angle,
                              int whether_normalize, int *keep, double *scores);

/*
 * detect_3d_cuboid::detect_cuboid (box_proposal_detail.cpp:56-557).
 *   gray: H x W u8.  K: 3x3 row-major.  Twc: 4x4 row-major.  boxes: nb x 5.  lines: nl x 4.
 *   out: nb * max_cuboid_num records, counts[nb].
 * Optional debug outputs (may be NULL):
 *   dbg_rows     : room for dbg_rows_cap rows of 25 doubles
 *                  [cfg, vp1pos, yaw, top_id, dist/diag, angle, hExp, roll, pitch, x0..x7, y0..y7]
 *                  for every valid proposal, in reference order, all boxes / height samples concatenated
 *   dbg_row_count: per (box,height-sample) count, room for nb*3
 * returns 0 or <0 on bad input.
 */
int orc_detect_cuboid(const uint8_t *gray, int W, int H, const double *K, const double *Twc,
                      const double *boxes, int nb, const double *lines, int nl,
                      const orc_cuboid_opts *opts, orc_cuboid *out, int *counts,
                      double *dbg_rows, long dbg_rows_cap, int *dbg_row_count);

/* full-frame helper used by tests: Canny+DT of one ROI as detect_cuboid does it */
void orc_canny_dt_roi(const uint8_t *gray, int W, int H, int x0, int y0, int w, int h, int low, int high,
                      float *dist);

/* ------------------------------------------------------------------ ORB extractor
 * ORB_SLAM2::ORBextractor (orb_object_slam/src/ORBextractor.cc, include/ORBextractor.h). */
typedef struct orc_keypoint { /* cv::KeyPoint layout (28 bytes) */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;
typedef struct orc_orb orc_orb;

orc_orb *orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void orc_orb_destroy(orc_orb *e);
/* ORBextractor::operator() (:1036-1099); returns the number of keypoints (<= cap), desc: n x 32 */
int orc_orb_extract(orc_orb *e, const uint8_t *gray, int W, int H, orc_keypoint *kps, uint8_t *desc, int cap);
/* introspection of the last extract() */
int orc_orb_features_per_level(orc_orb *e, int *out);
int orc_orb_level_dims(orc_orb *e, int level, int *w, int *h);
int orc_orb_get_level(orc_orb *e, int level, int blurred, uint8_t *out);
int orc_orb_get_candidates(orc_orb *e, int level, float *xyr, int cap); /* pre-quadtree FAST keypoints: x,y,response */
/* cv::FAST(img, kps, threshold, true) on a w x h view; xyr gets x,y,score triplets; returns count */
int orc_fast(const uint8_t *img, int stride, int w, int h, int threshold, float *xyr, int cap);
float orc_fast_atan2(float y, float x);
/* the float cos/sin the descriptor rotation uses (correctly rounded to float via a double evaluation) */
void orc_sincos_f(float angle_rad, float *s, float *c);

/* ------------------------------------------------------------------ ORB matcher
 * ORB_SLAM2::ORBmatcher (orb_object_slam/src/ORBmatcher.cc) + Frame grid (src/Frame.cc:303-318,404-459,525-535). */
typedef struct orc_frame {      /* the parts of ORB_SLAM2::Frame the matchers read */
    int N;
    const orc_keypoint *keysUn; /* mvKeysUn */
    const uint8_t *desc;        /* mDescriptors, N x 32 */
    float minX, maxX, minY, maxY; /* mnMinX .. mnMaxY */
} orc_frame;

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);                 /* ORBmatcher.cc:1905-1921 */
void orc_three_maxima(const int *counts, int L, int *ind /*3*/);                  /* ORBmatcher::ComputeThreeMaxima :1860-1901 on bin sizes */
int orc_get_features_in_area(const orc_frame *F, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap);
/* SearchByProjection(Frame &Cur, const Frame &Last, th, bMono=true) (:1373-1522).  Per last-frame keypoint: valid (has a
 * non-outlier, non-dynamic map point), world_pos (float xyz), the map point's descriptor, its octave and angle; blocks[i]
 * = map point has Observations() > 0.  train_match[N_cur] receives the last-frame index or -1.  Returns nmatches. */
int orc_search_by_projection_frame(const orc_frame *cur, int n_last, const float *world_pos, const uint8_t *valid,
                                   const uint8_t *blocks, const uint8_t *mp_desc, const int *last_octave, const float *last_angle,
                                   const float *Tcw12, float fx, float fy, float cx, float cy, const float *scale_factors,
                                   float th, int check_orientation, const uint8_t *train_blocked /* may be NULL */, int *train_match);
/* SearchByProjection(Frame &F, vector<MapPoint*>, th) (:50-142): proj_xy (mTrackProjX/Y), view_cos, pred_level, in_view,
 * blocks, mp_desc per map point; train_blocked[N] marks keypoints that already hold a map point with observations. */
int orc_search_local_map(const orc_frame *F, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level,
                         const uint8_t *in_view, const uint8_t *blocks, const uint8_t *mp_desc, const float *scale_factors,
                         float th, float nnratio, const uint8_t *train_blocked, int *train_match);
/* SearchForInitialization (:429-542): prev_matched is N1 x 2 in/out; matches12[N1] out.  Returns nmatches. */
int orc_search_for_initialization(const orc_frame *F1, const orc_frame *F2, float *prev_matched, int window_size, float nnratio,
                                  int check_orientation, int *matches12);
/* ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (:852-1003), search part: per valid map point (u, v, ur projections and
 * predicted level from the caller) the keypoint of the key frame with the smallest descriptor distance among GetFeaturesInArea(u, v,
 * th * scale[level]) that passes the level and chi-square tests (first one wins ties).  best_dist 256 / best_idx -1: none.  Returns nFused. */
int orc_fuse(const orc_frame *F, const float *u_right, const float *inv_level_sigma2, const uint8_t *keys_static, int n_mp, const float *uv, const float *ur,
             const int *pred_level, const uint8_t *valid, const uint8_t *mp_desc, const float *scale_factors, float th, int *best_idx, int *best_dist);
/* ORBmatcher::SearchForTriangulation (:679-850); node = DBoW2 FeatureVector node id of every feature (-1 none), F12 row-major 3x3. */
int orc_search_for_triangulation(const orc_frame *F1, const int *node1, const uint8_t *skip1, const float *u_right1, const uint8_t *static1, const orc_frame *F2,
                                 const int *node2, const uint8_t *skip2, const float *u_right2, const uint8_t *static2, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation, int *matches12);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:171-310) with the FeatureVector given as a node id per feature. */
int orc_search_by_bow(const orc_frame *KF, const int *nodeKF, const uint8_t *skipKF, const orc_frame *F, const int *nodeF, const uint8_t *skipF, float nnratio,
                      int check_orientation, int *matchesF);
int orc_search_by_bow_kf(const orc_frame *K1, const int *node1, const uint8_t *skip1, const orc_frame *K2, const int *node2, const uint8_t *skip2, float nnratio,
                         int check_orientation, int *matches12);
/* exact 2-NN in Hamming space over all pairs (first index wins ties) */
void orc_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx, int *best_dist, int *second_dist);

/* ------------------------------------------------------------------ object bundle adjustment
 * g2o machinery restated for BlockSolver_6_3 + Levenberg (vendored g2o core/{block_solver.hpp,
 * optimization_algorithm_levenberg.cpp,sparse_optimizer.cpp,base_*_edge.hpp,robust_kernel_impl.cpp}, types/{se3quat.h,
 * types_six_dof_expmap.*,types_sba.h}) with the CubeSLAM vertices/edges of orb_object_slam/{include/g2o_Object.h,
 * src/g2o_Object.cpp} as Optimizer::BundleAdjustment / LocalBACameraPointObjects (src/Optimizer.cc:64-251,826-1534) use them.
 * Poses are 7-vectors [tx ty tz qx qy qz qw] (SE3Quat::toVector). */
typedef struct orc_ba_problem {
    /* VertexSE3Expmap: world-to-camera poses */
    int n_cams; const double *cam_pose; const uint8_t *cam_fixed;
    /* VertexSBAPointXYZ, marginalised */
    int n_points; const double *points;
    /* VertexCuboidFixScale: object-to-world pose + half scale; per-vertex flags */
    int n_cuboids; const double *cuboid_pose; const double *cuboid_scale; /* n x 3, also used as fixedscale when fixedscale[0] > 0 */
    const uint8_t *cuboid_flags; /* bit0 whether_fixrollpitch, bit1 whether_fixrotation, bit2 whether_fixheight, bit3 fixedscale set */
    /* EdgeSE3ProjectXYZ (mono): information = inv_sigma2 * I, Huber(delta_mono) if > 0 */
    int n_obs; const int *obs_cam; const int *obs_point; const double *obs_uv; const double *obs_inv_sigma2;
    double fx, fy, cx, cy, huber_mono;
    /* EdgeSE3CuboidFixScaleProj: measurement = bbox [cx cy w h], information = diag(info4), Huber(delta_obj) if > 0 */
    int n_cobs; const int *cobs_cam; const int *cobs_cuboid; const double *cobs_bbox; const double *cobs_info;
    double K[9], huber_obj;
    /* EdgePointCuboidOnlyObjectFixScale (unary, information = I): fixed world points per edge */
    int n_pc; const int *pc_cuboid; const int *pc_offsets; const double *pc_points; double max_outside_margin_ratio;
    /* EdgeStereoSE3ProjectXYZ: obs_ur[o] >= 0 makes observation o a stereo edge (u, v, u_right), information = inv_sigma2 * I3, Huber(huber_stereo); NULL = all monocular */
    const double *obs_ur; double bf, huber_stereo;
} orc_ba_problem;

typedef struct orc_ba_stats {
    int iterations;            /* LM iterations executed (calls of OptimizationAlgorithmLevenberg::solve) */
    int lm_trials;             /* total solve() trials inside them */
    double chi2_init, chi2_final, lambda_final;
    double chi2_trace[64];     /* robust chi2 after each iteration */
} orc_ba_stats;

/* SparseOptimizer::optimize(iterations); outputs may alias nothing; stop_flag may be NULL */
int orc_ba_optimize(const orc_ba_problem *p, int iterations, double *cam_pose_out, double *points_out, double *cuboid_pose_out,
                    orc_ba_stats *stats);
/* computeActiveErrors + activeRobustChi2 at the given estimates; err_* may be NULL */
double orc_ba_errors(const orc_ba_problem *p, double *err_obs, double *err_cobs, double *err_pc);
/* Dense reduced camera system (Schur complement) restricted to the landmarks [lm_begin, lm_end) and, when with_pose_edges,
 * the camera-cuboid / point-cuboid edges; lambda is added to the landmark blocks always and to the pose diagonal when
 * with_pose_edges.  H is (6P)x(6P) row-major, b is 6P, P = number of non-fixed cameras + cuboids.  Returns P. */
int orc_ba_reduced_dense(const orc_ba_problem *p, int lm_begin, int lm_end, int with_pose_edges, double lambda, double *H, double *b);

/* ------------------------------------------------------------------ LSD line detection
 * line_lbd_detect::detect_filter_lines (line_lbd/class/line_lbd_allclass.cpp:125-148,200-221) ->
 * LSDDetector::detectImpl (libs/LSDDetector.cpp:153-287, one octave) -> LineSegmentDetectorImpl (libs/lsd.cpp, LSD_REFINE_ADV,
 * scale 0.8, sigma_scale 0.6, quant 2, ang_th 22.5, log_eps 0, density_th 0.7, 1024 bins). */
typedef struct orc_keyline { /* the KeyLine fields detectImpl fills (descriptor.hpp:105-150) */
    float angle; int32_t class_id; int32_t octave; float pt_x, pt_y; float response; float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int32_t numOfPixels;
} orc_keyline;
/* detect_raw_lines: all KeyLines (cap entries); returns count */
int orc_lsd_detect(const uint8_t *gray, int W, int H, orc_keyline *out, int cap);
/* detect_filter_lines(gray, linesmat): octave 0 and lineLength > length_thres; lines: n x 4 floats; returns n */
int orc_lsd_detect_filter_lines(const uint8_t *gray, int W, int H, float length_thres, float *lines, int cap);
/* introspection: the scaled image (double), gradient norm, level-line angle (NOTDEF = -1024) and the pseudo-ordered
 * coordinate list (x + y*width) of ll_angle; sw/sh out.  Any pointer may be NULL. */
int orc_lsd_maps(const uint8_t *gray, int W, int H, int *sw, int *sh, double *scaled, double *modgrad, double *angles, int *order, int *n_order);

/* ------------------------------------------------------------------ LBD line descriptor
 * BinaryDescriptor::compute -> computeImpl (line_lbd/libs/binary_descriptor.cpp:603-790, useDetectionData = false, one octave):
 * computeGaussianPyramid (:352-370, GaussianBlur 5x5 sigma 1), computeSobel (:373-402), computeLBD (:1146-1509),
 * binaryConversion (:405-416) with the 32 band pairs of `combinations` (:74-107). */
int orc_lbd_compute(const uint8_t *gray, int W, int H, const orc_keyline *kl, int n, uint8_t *desc, float *fdesc);
int orc_lbd_maps(const uint8_t *gray, int W, int H, uint8_t *blur, int16_t *dx, int16_t *dy);

/* ------------------------------------------------------------------ Optimizer::PoseOptimization (orb_object_slam/src/Optimizer.cc:253-472)
 * One frame: n map-point matches (Xw n x 3; obs n x 3 = u, v, u_right with u_right < 0 for monocular observations; inv_sigma2 n),
 * pose = [t, qx qy qz qw] of Tcw.  outlier: n flags (mvbOutlier).  Returns nInitialCorrespondences - nBad. */
int orc_pose_optimization(int n, const double *Xw, const double *obs, const double *inv_sigma2, double fx, double fy, double cx, double cy, double bf,
                          const double *pose_in, double *pose_out, uint8_t *outlier);

/* ------------------------------------------------------------------ 9-dof g2o::cuboid of object_slam (g2o_Object.h:23-252)
 * cuboid = 10 doubles [t, qx qy qz qw, half scale].  oplus: VertexCuboid::oplusImpl (pose * exp(update[0:6]), scale + update[6:9]);
 * edge_error: EdgeSE3Cuboid::computeError (min_log_error over the four 90-degree rotations of the measured cuboid moved to the
 * world by the camera); edge_linearize: its numeric Jacobians as g2o computes them (central differences, 1e-9). */
int orc_cuboid9_oplus(int n, const double *cub, const double *upd, double *out);
int orc_cuboid9_edge_error(int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err);
int orc_cuboid9_edge_linearize(int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err, double *Jcam, double *Jcub);
/* test hook: one pose helper at a time (SE3Quat exp / log / product / inverse, exptwist_norollpitch, the cuboid's log errors, rotations and
 * transforms, point_boundary_error) -- see ba_oracle.cpp; tests/test_ref_pins.py holds them against the reference's own code */
/* the LM loop's pieces one at a time (test hook, see ba_oracle.cpp) */
void orc_huber(double e, double delta, double *rho3); /* RobustKernelHuber::robustify as the BA uses it */
typedef struct orc_ba_handle orc_ba_handle;
orc_ba_handle *orc_ba_open(const orc_ba_problem *p);
struct orc_badyn_problem;
orc_ba_handle *orc_badyn_open(const struct orc_badyn_problem *p); /* the same pieces of the dynamic-object BA oracle (read with orc_badyn_read) */
int orc_ba_block_dim(orc_ba_handle *h, int block);
void orc_badyn_read(orc_ba_handle *h, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints);
void orc_ba_close(orc_ba_handle *h);
void orc_ba_compute_errors(orc_ba_handle *h);
double orc_ba_robust_chi2(orc_ba_handle *h);
void orc_ba_build_system(orc_ba_handle *h);
void orc_ba_sizes(orc_ba_handle *h, int *P, int *L);
double orc_ba_hessian_diag(orc_ba_handle *h, int block, int j);
int orc_ba_solve(orc_ba_handle *h, double lambda);
void orc_ba_update(orc_ba_handle *h);
void orc_ba_push(orc_ba_handle *h);
void orc_ba_pop(orc_ba_handle *h);
void orc_ba_discard_top(orc_ba_handle *h);
const double *orc_ba_x(orc_ba_handle *h, long *n);
const double *orc_ba_b(orc_ba_handle *h);
int orc_ba_block(orc_ba_handle *h, int kind, int i, int j, double *out); /* blocks of the system orc_ba_build_system left: 0 Hpp(i,i) 1 Hpp(i,j) i<j 2 Hll(i) 3 Hpl of observation i (6 x 3) */
int orc_ba_pose_index(orc_ba_handle *h, int is_cuboid, int i);
double orc_ba_edge_chi2(orc_ba_handle *h, int kind, int o); /* BaseEdge::chi2 of an edge: 0 point observation 1 camera-cuboid 2 point-cuboid */
void orc_ba_read(orc_ba_handle *h, double *cam_pose_out, double *points_out, double *cuboid_pose_out);
int orc_se3_op(int op, const double *a, const double *b, double s, double *out);
void orc_pose_linearize(int n, const double *Xw, const double *obs, const double *inv_sigma2, double fx, double fy, double cx, double cy, double bf, const double *pose_in,
                        int robust, double *H, double *b, double *chi2);

/* --------------------------------------------------------------------------------------------------------------------
 * Optimizer::LocalBACameraPointObjectsDynamic (orb_object_slam/src/Optimizer.cc:1537-2573): the graph it hands to g2o
 * (BlockSolverX + LinearSolverDense + Levenberg) -- key-frame poses, one VertexCuboidFixScale per (object, key frame),
 * one VelocityPlanarVelocity per object, static points (world frame) and dynamic points (object frame), both marginalised.
 * Indices refer to the arrays of this struct; *_level[o] != 0 takes edge o out of the optimisation (setLevel(1));
 * NULL level arrays = all zero; obs_ur NULL or < 0 = monocular.  All information matrices are diagonal in the reference. */
typedef struct orc_badyn_problem {
    int n_cams; const double *cam_pose; const uint8_t *cam_fixed;                               /* VertexSE3Expmap (:1689-1711) */
    int n_objs; const double *obj_pose; const double *obj_scale; const uint8_t *obj_flags;      /* VertexCuboidFixScale (:1729-1786), flags as in orc_ba_problem */
    int n_vels; const double *vel;                                                              /* VelocityPlanarVelocity [v, steer] (:2160-2165) */
    int n_points; const double *points;                                                         /* static VertexSBAPointXYZ (:1819-1830) */
    int n_dpoints; const double *dpoints;                                                       /* dynamic points, PosToObj (:1934-1945) */
    int fix_points;                                                                             /* fixPoint: points fixed instead of marginalised */
    /* EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (:1843-1906) */
    int n_obs; const int *obs_cam; const int *obs_point; const double *obs_uv; const double *obs_ur; const double *obs_inv_sigma2; const uint8_t *obs_level;
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
    /* UnaryLocalPoint, one per dynamic point (:1947-1955): information = ulp_info * I */
    double ulp_info, ulp_scale[3], ulp_ratio;
    /* EdgeDynamicPointCuboidCamera (:1977-1993): vertices (camera, object pose, dynamic point), information = inv_sigma2 * I2 */
    int n_dobs; const int *dobs_cam; const int *dobs_obj; const int *dobs_point; const double *dobs_uv; const double *dobs_inv_sigma2; const uint8_t *dobs_level;
    double K[9], huber_dyn;
    /* EdgeObjectMotion (:2192-2202): vertices (object pose from, object pose to, velocity), information = diag(mot_info), no kernel */
    int n_mot; const int *mot_from; const int *mot_to; const int *mot_vel; const double *mot_dt; double mot_info[3];
    /* EdgeSE3CuboidFixScaleProj (:2279-2298): information = diag(cobs_info[o*4..]), Huber(huber_obj) */
    int n_cobs; const int *cobs_cam; const int *cobs_obj; const double *cobs_bbox; const double *cobs_info; const uint8_t *cobs_level; double huber_obj;
    /* EdgePointCuboidOnlyObjectFixScale (:2096-2115): information = I, no kernel */
    int n_pc; const int *pc_obj; const int *pc_offsets; const double *pc_points; double pc_ratio;
} orc_badyn_problem;

/* computeError of every edge (inactive ones too) + activeRobustChi2 over the level-0 edges; any output may be NULL.
 * e_obs n x 3 (third 0 for mono), e_dobs n x 2, e_mot n x 3, e_cobs n x 4, e_pc n x 3, e_ulp n_dpoints x 3 */
double orc_badyn_errors(const orc_badyn_problem *p, double *e_obs, double *e_dobs, double *e_mot, double *e_cobs, double *e_pc, double *e_ulp);
/* reduced pose system (Hpp + lambda I - Hpl (Hll + lambda I)^-1 Hlp, bp - Hpl (Hll + lambda I)^-1 bl) at the given estimates, dense row-major;
 * pose scalars are ordered cameras (6 each, non-fixed), object poses (6), velocities (2).  Returns the dimension; H may be NULL to query it. */
int orc_badyn_reduced_dense(const orc_badyn_problem *p, double lambda, double *H, double *bvec);
void orc_badyn_edge_jacobians(const orc_badyn_problem *p, double *J_dobs, double *J_mot); /* test hook: per-edge Jacobians of the three-vertex edges */
/* one linear step with a given damping (computeActiveErrors, buildSystem, solve, update); returns 1 if the factorisation failed */
int orc_badyn_step(const orc_badyn_problem *p, double lambda, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints);
/* SparseOptimizer::optimize(iterations) */
int orc_badyn_optimize(const orc_badyn_problem *p, int iterations, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints, orc_ba_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
