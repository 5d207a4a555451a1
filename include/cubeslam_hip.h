/*
 * cubeslam_hip.h -- C-ABI of libcubeslam_hip.so: MI355X (gfx950) kernels for CubeSLAM's per-frame hot path.
 *
 * Plain C, POD pointers and sizes only.  Every entry point cites the reference (shichaoy/cube_slam) interface
 * it replaces.  All functions return 0 (CS_OK) or a negative cs_status; they never throw and never fall back to
 * a CPU path: without a usable HIP device cs_create() fails with CS_ERR_NO_DEVICE.
 *
 * Threading: one cs_ctx per host thread / HIP stream (the reference objects are not re-entrant either:
 * detect_3d_cuboid.h:55-57 keeps cam_pose state, ORBextractor.h:85 keeps the pyramid).
 */
#ifndef CUBESLAM_HIP_H
#define CUBESLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_VERSION 107 /* 107: cs_match_by_projection_stream takes n_queries / n_train, cs_cuboid_batch_set_boxes, the matchers' claim passes run on the device; 106: cs_cuboid_batch_n_frames, cs_frontend_queues, cs_match_by_projection_stream, cs_frontend_stream_*, cs_*_set_frames_device, cs_orb_read_packed; 105: cs_frontend_set_backlog; 104: cs_frontend_set_cuboid_ctx; 103: cs_lsd_read_filter_lines takes the caller's frame count, cs_frontend_set_chain; 102: cs_cuboid_batch_set_lines, cs_cuboid_batch_set_shared_gpu, cs_lsd_read_filter_lines; 101: cs_ba_set_stop_flag_bool / cs_ba_dyn_set_stop_flag_bool; cs_match_by_projection_frame takes train_blocked */

typedef enum cs_status {
    CS_OK = 0,
    CS_ERR_NO_DEVICE = -1,   /* no HIP device / runtime error at create */
    CS_ERR_BAD_ARG = -2,     /* null pointer, negative size, ROI outside image (cv::Rect assert in the reference) */
    CS_ERR_HIP = -3,         /* a HIP runtime call failed; see cs_last_error() */
    CS_ERR_CAPACITY = -4,    /* an internal fixed capacity was exceeded (e.g. > CS_MAX_ROI_LINES lines in one box) */
    CS_ERR_NOMEM = -5
} cs_status;

typedef struct cs_ctx cs_ctx;

/* Creates a context bound to HIP device `device_id` with its own stream. */
int cs_create(int device_id, cs_ctx **out);
/* Same with a stream priority: > 0 highest, < 0 lowest, 0 default.  The batch front-end runs ORB + cuboid (what the tracking thread waits
 * for) on a high-priority stream and the line detectors' device phases, which overlap their own host stage, on low-priority ones. */
int cs_create_with_priority(int device_id, int priority, cs_ctx **out);
void cs_destroy(cs_ctx *ctx);
const char *cs_last_error(const cs_ctx *ctx);

/* RCCL over xGMI inside the library (one process per GPU).  Rank 0 calls cs_comm_unique_id and hands the 128 bytes to the other ranks by
 * any out-of-band means (MPI, a socket, torch.distributed, a file); every rank then calls cs_comm_init on its context.  A sharded cs_ba
 * (world > 1) without a cs_ba_set_allreduce callback all-reduces the reduced camera system with ncclAllReduce(ncclDouble, ncclSum) on the
 * context's own stream: one fused buffer [36 * slots | 6 * P] per LM trial, no host synchronisation around it.  librccl.so is opened at
 * cs_comm_init (not a link-time dependency). */
#define CS_COMM_ID_BYTES 128
int cs_comm_unique_id(void *id128);
int cs_comm_init(cs_ctx *ctx, int rank, int world, const void *id128);
void cs_comm_destroy(cs_ctx *ctx);
/* ncclAllReduce(ncclDouble, ncclSum) in place over n doubles of device memory, enqueued on the context's stream (no synchronisation). */
int cs_comm_allreduce_f64(cs_ctx *ctx, double *device_buf, long n);
int cs_version(void);
/* Blocks until all work queued on the context's stream is done. */
int cs_sync(cs_ctx *ctx);
/* CPU threads the host stages use (affinity mask capped by the cgroup CPU quota; env CUBESLAM_HOST_THREADS overrides) */
int cs_host_thread_count(void);

/* Per-kernel hipEvent timing on the context's stream (replaces ca::Profiler::tictoc,
 * dependency/tictoc_profiler/src/profiler.cpp:40-67).  Disabled by default. */
int cs_timing_enable(cs_ctx *ctx, int on);
int cs_timing_reset(cs_ctx *ctx);
/* total milliseconds and launch count recorded for kernel `name` since the last reset; unknown name -> 0,0 */
int cs_timing_get(cs_ctx *ctx, const char *name, double *total_ms, long *count);

/* ===================================================================== detect_3d_cuboid
 * Replaces detect_3d_cuboid::detect_cuboid (detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:62-63,
 * detect_3d_cuboid/src/box_proposal_detail.cpp:56-557) and everything it calls in object_3d_util.cpp. */

#define CS_MAX_ROI_LINES 1024

typedef struct cs_cuboid_opts {          /* public members of class detect_3d_cuboid, detect_3d_cuboid.h:65-79 */
    int consider_config_1;
    int consider_config_2;
    int whether_sample_cam_roll_pitch;
    int whether_sample_bbox_height;
    int max_cuboid_num;
    double nominal_skew_ratio;
    double max_cut_skew;
    /* extensions; the defaults reproduce the constants hard-coded at box_proposal_detail.cpp:126-128,197 */
    double yaw_range_deg;                /* 45 */
    double yaw_step_deg;                 /* 6  */
    int canny_low;                       /* 80 */
    int canny_high;                      /* 200 */
} cs_cuboid_opts;

typedef struct cs_cuboid {               /* class cuboid, detect_3d_cuboid.h:15-36 */
    double pos[3];
    double scale[3];
    double rotY;
    double box_config_type[2];
    int32_t box_corners_2d[16];          /* 2x8 row-major */
    double box_corners_3d_world[24];     /* 3x8 row-major */
    double rect_detect_2d[4];
    double edge_distance_error;
    double edge_angle_error;
    double normalized_error;
    double skew_ratio;
    double down_expand_height;
    double camera_roll_delta;
    double camera_pitch_delta;
} cs_cuboid;

void cs_cuboid_default_opts(cs_cuboid_opts *opts);

/*
 * One frame, host buffers in / host buffers out (the drop-in call; H2D + kernels + D2H).
 *   img      : height x width x channels u8, row stride `stride` bytes; channels 1 (gray) or 3 (BGR, converted like
 *              cv::cvtColor(CV_BGR2GRAY), box_proposal_detail.cpp:62-66)
 *   K        : 3x3 row-major calibration (set_calibration, box_proposal_detail.cpp:36-40)
 *   Twc      : 4x4 row-major camera-to-world (transToWolrd)
 *   boxes    : n_boxes x 5  [x y w h prob], 0-based (obj_bbox_coors)
 *   lines    : n_lines x 4  [x1 y1 x2 y2] (all_lines_raw; copied, like the by-value MatrixXd of the reference)
 *   out      : n_boxes * max_cuboid_num records; box b's cuboids are out[b*max_cuboid_num + i], i < counts[b],
 *              sorted by combined score (ObjectSet order, box_proposal_detail.cpp:517-536)
 * With whether_sample_cam_roll_pitch and more than one box the boxes are taken one after the other, like the reference's loop
 * over objects: box b + 1 reads the camera yaw that box b left in cam_pose (box_proposal_detail.cpp:126 after :233-239 /
 * :481-487) -- the configuration of object_slam/src/main_obj.cpp:442.  A cs_cuboid_batch starts every box from the raw pose.
 */
int cs_cuboid_detect(cs_ctx *ctx, const uint8_t *img, int width, int height, int channels, int stride,
                     const double *K, const double *Twc, const double *boxes, int n_boxes,
                     const double *lines, int n_lines, const cs_cuboid_opts *opts,
                     cs_cuboid *out, int *counts);

/*
 * Batched, device-resident form (BASELINE config 4: many frames x boxes).  create() uploads everything to HBM and
 * plans the arenas; run() only launches kernels on the context's stream (asynchronous); read() synchronises and
 * copies results back.  All frames share width/height/K/opts.
 *   gray        : n_frames x height x width u8, densely packed
 *   Twc         : n_frames x 16
 *   box_offsets : n_frames+1 prefix offsets into `boxes` (rows)
 *   line_offsets: n_frames+1 prefix offsets into `lines` (rows)
 */
typedef struct cs_cuboid_batch cs_cuboid_batch;
int cs_cuboid_batch_create(cs_ctx *ctx, int n_frames, int width, int height, const uint8_t *gray,
                           const double *K, const double *Twc, const int *box_offsets, const double *boxes,
                           const int *line_offsets, const double *lines, const cs_cuboid_opts *opts,
                           cs_cuboid_batch **out);
int cs_cuboid_batch_run(cs_ctx *ctx, cs_cuboid_batch *b);
/* New edge lists (line_offsets[n_frames + 1], lines m x 4 as in cs_cuboid_batch_create) for the frames of an existing batch: the hand-over of the
 * chain detect_filter_lines -> detect_cuboid (main_obj.cpp:428-449) when frames and boxes stay resident. */
int cs_cuboid_batch_set_lines(cs_ctx *ctx, cs_cuboid_batch *b, const int *line_offsets, const double *lines);
/* frames the batch was created with (line_offsets of cs_cuboid_batch_set_lines holds one more entry); -1 for NULL */
int cs_cuboid_batch_n_frames(const cs_cuboid_batch *b);
/* Other 2-D boxes, camera poses and (line_offsets non-NULL) edge lists for the frames of an existing batch -- the arguments every call of detect_cuboid brings with its
 * pixels (detect_3d_cuboid.h:62-63).  Same frame count, image size and options; layouts as in cs_cuboid_batch_create.  The plan of the sweep (ROIs, top samples, arena
 * slices: box_proposal_detail.cpp:107-161) is rebuilt for them on the host (a function of boxes and poses alone) and uploaded from pinned staging on the context's stream
 * behind the run that still reads the old one; the call waits only when an arena has to grow.  A box whose ROI leaves the image: CS_ERR_BAD_ARG, the batch stays as it was.
 * Results (cs_cuboid_batch_read) then hold box_offsets[n_frames] boxes. */
int cs_cuboid_batch_set_scene(cs_ctx *ctx, cs_cuboid_batch *b, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines);
int cs_cuboid_batch_n_boxes(const cs_cuboid_batch *b);
/* shared != 0: long-running kernels of other streams hold most CUs while this batch runs (cs_frontend's alternating runner does this itself): the edge-scoring
 * kernel takes the launch shape that fits beside them.  A speed hint only. */
int cs_cuboid_batch_set_shared_gpu(cs_cuboid_batch *b, int shared);
/* out: total_boxes * max_cuboid_num, counts: total_boxes */
int cs_cuboid_batch_read(cs_ctx *ctx, cs_cuboid_batch *b, cs_cuboid *out, int *counts);
void cs_cuboid_batch_destroy(cs_ctx *ctx, cs_cuboid_batch *b);

/* Workload facts for roofline accounting (valid after run + sync): number of (box, height-sample) units, total ROI
 * pixels A, total enumerated hypotheses, total valid proposals. */
int cs_cuboid_batch_stats(cs_ctx *ctx, cs_cuboid_batch *b, long *n_units, long *roi_pixels, long *n_hypotheses,
                          long *n_valid);
/* The same facts split by scoring kernel (valid after run + sync).  out[0..2] = units, ROI pixels, valid proposals scored by
 * cuboid_sweep_score (the unit's 16-bit chamfer code map fits one CU's LDS); out[3..5] = the same for cuboid_sweep_score_big (larger
 * ROIs, or a pixel farther than 244 px from every edge), which gathers from the float map. */
int cs_cuboid_batch_score_stats(cs_ctx *ctx, cs_cuboid_batch *b, long out[6]);

/* Introspection for parity tests (after run).  unit = index over (frame, box, height-sample) in that order.
 *   dims    : [roi_x, roi_y, roi_w, roi_h, n_hyp_capacity, n_valid, n_merged_lines, n_yaw, frame, box, height_sample, n_height_samples]
 *   edges   : roi_w*roi_h u8 Canny output (0/255), may be NULL
 *   dist    : roi_w*roi_h f32 distance map, may be NULL
 *   rows    : n_valid x 25 doubles in the reference's row layout
 *             [cfg, vp1pos, yaw, top_id, dist/diag, angle, hExp, roll, pitch, x0..x7, y0..y7], may be NULL (cap rows_cap)
 *   merged  : n_merged_lines x 4 doubles, may be NULL (cap merged_cap rows)
 */
int cs_cuboid_batch_unit(cs_ctx *ctx, cs_cuboid_batch *b, int unit, int dims[12], uint8_t *edges, float *dist,
                         double *rows, long rows_cap, double *merged, long merged_cap);

/* ===================================================================== ORBextractor
 * Replaces ORB_SLAM2::ORBextractor (orb_object_slam/include/ORBextractor.h:46-117, src/ORBextractor.cc):
 * constructor tables (:412-471), ComputePyramid (:1101-1125), ComputeKeyPointsOctTree (:766-853), IC_Angle (:74-101),
 * 7x7 Gaussian blur + steered rBRIEF (:104-150,:1069-1098).  FAST scores / cell NMS / orientation / blur / descriptors
 * run on the GPU; the order-dependent quadtree selection (DistributeOctTree :540-763) runs on the host between the two
 * GPU phases. */
typedef struct cs_keypoint {             /* cv::KeyPoint layout, 28 bytes */
    float x, y;                          /* pt */
    float size;
    float angle;                         /* degrees, cv::fastAtan2 */
    float response;                      /* FAST corner score */
    int32_t octave;
    int32_t class_id;                    /* -1 */
} cs_keypoint;

typedef struct cs_orb cs_orb;
/* ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) for images of width x height; up to max_frames
 * frames are processed per call (device buffers are sized once). */
int cs_orb_create(cs_ctx *ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                  int width, int height, int max_frames, cs_orb **out);
void cs_orb_destroy(cs_ctx *ctx, cs_orb *e);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares (ORBextractor.h:63-83);
 * which = 0..3, out has nlevels floats.  which = 4: mnFeaturesPerLevel as ints (out reinterpreted as int32). */
int cs_orb_get_table(const cs_orb *e, int which, void *out);
/* ORBextractor::operator() for n_frames gray images (host, row stride `stride`).  Per frame f: counts[f] keypoints at
 * kps[f*cap_per_frame ...] and descriptors at desc[(f*cap_per_frame + i)*32], level-major order (:1065-1098).
 * Returns CS_ERR_CAPACITY if a frame yields more than cap_per_frame keypoints. */
int cs_orb_extract(cs_ctx *ctx, cs_orb *e, const uint8_t *gray, int n_frames, int stride,
                   cs_keypoint *kps, uint8_t *desc, int cap_per_frame, int *counts);
/* Device-resident form: upload once, run() = both GPU phases + host quadtree, read() = D2H of the results. */
int cs_orb_upload(cs_ctx *ctx, cs_orb *e, const uint8_t *gray, int n_frames, int stride);
int cs_orb_run(cs_ctx *ctx, cs_orb *e);
int cs_orb_read(cs_ctx *ctx, cs_orb *e, cs_keypoint *kps, uint8_t *desc, int cap_per_frame, int *counts);
/* The same results packed: every frame's key points / descriptors one frame behind the other, frame f = first[f] .. first[f + 1] (n_frames + 1 entries), two copies
 * for the whole batch.  kps / desc NULL: only *total and first are filled (size query); CS_ERR_CAPACITY when cap_total < *total. */
int cs_orb_read_packed(cs_ctx *ctx, cs_orb *e, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total);
/* Introspection for parity tests (after run): pyramid level (mvImagePyramid, ORBextractor.h:85) or its blurred copy;
 * the FAST keypoints handed to DistributeOctTree (x, y, response; cell-major order). */
int cs_orb_get_level(cs_ctx *ctx, cs_orb *e, int frame, int level, int blurred, uint8_t *out, int *w, int *h);
int cs_orb_get_candidates(cs_ctx *ctx, cs_orb *e, int frame, int level, float *xyr, int cap, int *n);

/* ===================================================================== ORBmatcher
 * Replaces the Hamming searches of ORB_SLAM2::ORBmatcher (orb_object_slam/include/ORBmatcher.h:43-89, src/ORBmatcher.cc)
 * and the Frame grid they use (src/Frame.cc:303-318 AssignFeaturesToGrid, :404-459 GetFeaturesInArea, :525-535 PosInGrid).
 * GPU: 64x48 grid build, projection, per-query candidate lists (grid order) with 256-bit Hamming distances.
 * Host: the order-dependent greedy claim / ratio / rotation-histogram pass over those lists (sequential in the reference).
 * Monocular paths (mvuRight < 0). */
typedef struct cs_matcher cs_matcher;
int cs_matcher_create(cs_ctx *ctx, int max_keypoints, int max_queries, long max_candidates, cs_matcher **out);
void cs_matcher_destroy(cs_ctx *ctx, cs_matcher *m);
/* The frame searched in (CurrentFrame / F / F2): mvKeysUn, mDescriptors, mnMinX..mnMaxY. */
int cs_matcher_set_frame(cs_ctx *ctx, cs_matcher *m, const cs_keypoint *keysUn, const uint8_t *desc, int N,
                         float minX, float maxX, float minY, float maxY);
/* Frame post-processing without a host round trip: the keypoints and descriptors of frame `frame` of the last
 * cs_orb_run stay in HBM; Frame::UndistortKeyPoints (Frame.cc:546-576: copy when dist[0] == 0, else cv::undistortPoints(K, dist, P = K),
 * classic five-iteration form) and Frame::AssignFeaturesToGrid (:303-318) run on the device and the matcher is ready for the searches.
 * K4 = fx fy cx cy, dist5 = k1 k2 p1 p2 k3 (NULL = none); bounds from cs_frame_image_bounds.  keysUn_out (nullable, room for the
 * frame's keypoints) receives mvKeysUn; with NULL nothing is copied back and the call does not wait for the device (every search reads what it needs of a train key point
 * from the device copy; only cs_match_for_initialization, which updates vbPrevMatched on the host, needs a frame set WITH the copy). */
int cs_matcher_set_frame_from_orb(cs_ctx *ctx, cs_matcher *m, const cs_orb *orb, int frame, const float *K4, const float *dist5, float minX, float maxX, float minY,
                                  float maxY, cs_keypoint *keysUn_out, int *n_out);
/* Frame::ComputeImageBounds (Frame.cc:578-609): bounds = mnMinX, mnMaxX, mnMinY, mnMaxY of the undistorted image corners. */
int cs_frame_image_bounds(int cols, int rows, const float *K4, const float *dist5, float *bounds);
/* Workload facts of the last search on this matcher (queries, candidates the search windows enumerated), for roofline accounting. */
int cs_matcher_last_counts(const cs_matcher *m, int *queries, long *candidates);
/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) on the frame set above; *n = count (may exceed cap). */
int cs_matcher_features_in_area(cs_ctx *ctx, cs_matcher *m, float x, float y, float r, int minLevel, int maxLevel,
                                int *out, int cap, int *n);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono=true) (ORBmatcher.cc:1373-1522).
 * Per last-frame keypoint i: valid[i] (map point present, not outlier/dynamic), world_pos (float xyz), blocks[i] (map point
 * has Observations() > 0), the map point descriptor, LastFrame.mvKeys[i].octave and mvKeysUn[i].angle.  Tcw: 3x4 float
 * row-major.  train_blocked (N bytes, may be NULL): current-frame keypoints a candidate loop must skip -- the ones that already carry a
 * map point with Observations() > 0 before the call, and, in dynamic-object mode, the ones with KeysStatic[i2] == false
 * (ORBmatcher.cc:1451-1457).  train_match[N] receives the last-frame index matched to each current keypoint or -1. */
int cs_match_by_projection_frame(cs_ctx *ctx, cs_matcher *m, int n_last, const float *world_pos, const uint8_t *valid,
                                 const uint8_t *blocks, const uint8_t *mp_desc, const int *last_octave, const float *last_angle,
                                 const float *Tcw, float fx, float fy, float cx, float cy, const float *scale_factors, int n_levels,
                                 float th, int check_orientation, const uint8_t *train_blocked, int *train_match, int *nmatches);
/* The same search for a whole WINDOW of a stream whose frames an extractor holds in HBM: pair p = (last frame f0 + p, current frame f0 + p + 1), p < n_pairs.  Frame
 * post-processing of every current frame (UndistortKeyPoints + AssignFeaturesToGrid, Frame.cc:303-318, 546-576, from the extractor's device buffers) and the searches of all
 * pairs run as a handful of launches over the window (the per-frame calls take five launches and three host round trips per frame); results equal the per-frame calls'.
 * Queries: the key points of the last frames, concatenated over the pairs in order (n_q = key points of frames f0 .. f0 + n_pairs - 1): world_pos (3 floats each), valid,
 * blocks; their level and angle are the last frame's key points' (ORBmatcher.cc:1424, 1483); mp_desc: the map points' descriptors (32 B per query) or NULL = the last
 * frame's own descriptors.  Tcw: 12 floats per pair.  train_match: concatenated over the current frames f0 + 1 .. f0 + n_pairs (key point counts as the extractor reports
 * them), the matched query's index WITHIN its last frame or -1; nmatches[n_pairs].  n_queries / n_train: what the caller sized the per-query arrays and train_match for -- the
 * call fails with CS_ERR_BAD_ARG when they are not the extractor's counts for the window (an extractor that ran again in between).  There is no train_blocked here: a current frame
 * of the window carries no map points before its own search.  The claims (a key point claimed by a map point with observations is skipped by later queries), the rotation
 * histogram and its three-maxima cut run on the device, one wave per pair; candidates never leave HBM. */
typedef struct cs_match_stream cs_match_stream;
int cs_match_stream_create(cs_match_stream **out);
void cs_match_stream_destroy(cs_ctx *ctx, cs_match_stream *m);
int cs_match_by_projection_stream(cs_ctx *ctx, cs_match_stream *m, const cs_orb *orb, int f0, int n_pairs, const float *K4, const float *dist5 /* nullable */,
                                  float minX, float maxX, float minY, float maxY, int n_queries, int n_train, const float *world_pos, const uint8_t *valid, const uint8_t *blocks,
                                  const uint8_t *mp_desc /* nullable */, const float *Tcw, float fx, float fy, float cx, float cy, const float *scale_factors, int n_levels,
                                  float th, int check_orientation, int *train_match, int *nmatches);
/* queries and window candidates of the last call (roofline accounting) */
int cs_match_stream_last_counts(const cs_match_stream *m, long *queries, long *candidates);
/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th) (:50-142). */
int cs_match_local_map(cs_ctx *ctx, cs_matcher *m, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level,
                       const uint8_t *in_view, const uint8_t *blocks, const uint8_t *mp_desc, const float *scale_factors, int n_levels,
                       float th, float nnratio, const uint8_t *train_blocked, int *train_match, int *nmatches);
/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:429-542); F2 = the frame set above. */
int cs_match_for_initialization(cs_ctx *ctx, cs_matcher *m, const cs_keypoint *keys1Un, const uint8_t *desc1, int N1,
                                float *prev_matched, int window_size, float nnratio, int check_orientation,
                                int *matches12, int *nmatches);
/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (ORBmatcher.cc:852-1003), the search: for every valid map point (the
 * caller did the preamble :871-919 -- projection u, v, ur = u - bf/z, image bounds, distance and viewing-angle tests, PredictScale) the
 * keypoint of the key frame set with cs_matcher_set_frame that has the smallest descriptor distance among GetFeaturesInArea(u, v,
 * th * scale[level]), after the level (:941), KeysStatic (:944), and chi-square tests (stereo 7.8 / mono 5.99, :947-971); first one
 * wins ties.  u_right = mvuRight, inv_level_sigma2 = mvInvLevelSigma2.  best_idx -1 / best_dist 256: none; *n_fused = number with
 * best_dist <= TH_LOW (what the reference returns).  Replace / AddObservation (:985-1000) stay with the caller's map. */
int cs_match_fuse(cs_ctx *ctx, cs_matcher *m, const float *u_right, const float *inv_level_sigma2, int n_levels, const uint8_t *keys_static, int n_mp, const float *uv,
                  const float *ur, const int *pred_level, const uint8_t *valid, const uint8_t *mp_desc, const float *scale_factors, float th, int *best_idx,
                  int *best_dist, int *n_fused);
/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:679-850).  node1 / node2: vocabulary node of every feature (the DBoW2 FeatureVector
 * built by KeyFrame::ComputeBoW, -1 = none; DBoW2 itself is out of scope); skip = the feature already has a map point (or is
 * not static); u_right < 0 = monocular; F12 row-major 3x3 (float), (ex, ey) the epipole of KF1's centre in KF2 (:686-692).
 * matches12[N1] = index in KF2 or -1 (vMatchedPairs = the pairs with matches12 >= 0). */
int cs_match_for_triangulation(cs_ctx *ctx, const cs_keypoint *keys1Un, const uint8_t *desc1, int N1, const int *node1, const uint8_t *skip1, const float *u_right1,
                               const cs_keypoint *keys2Un, const uint8_t *desc2, int N2, const int *node2, const uint8_t *skip2, const float *u_right2,
                               const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2_2, int n_levels, int only_stereo,
                               int check_orientation, int *matches12, int *nmatches);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (ORBmatcher.cc:171-310): node = vocabulary node of every feature
 * (-1 none; DBoW2 is out of scope); skipKF = no usable map point (NULL / bad / dynamic / not static); skipF (nullable) = !KeysStatic.
 * matchesF[NF] = index of the key-frame feature whose map point the frame feature receives, -1 none. */
int cs_match_by_bow(cs_ctx *ctx, const cs_keypoint *keysKF, const uint8_t *descKF, int NK, const int *nodeKF, const uint8_t *skipKF, const cs_keypoint *keysF,
                    const uint8_t *descF, int NF, const int *nodeF, const uint8_t *skipF, float nnratio, int check_orientation, int *matchesF, int *nmatches);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORBmatcher.cc:544-677), the loop-closing overload: skip1 / skip2 =
 * the feature has no usable map point (NULL or isBad, :580-584, :598-603); a KF2 feature is claimed once (vbMatched2); the acceptance
 * test is bestDist1 < TH_LOW (strict here, :625) and the ratio test; the rotation histogram holds KF1 indices.  matches12[N1] = index
 * of the KF2 feature whose map point vpMatches12[idx1] receives, -1 none. */
int cs_match_by_bow_kf(cs_ctx *ctx, const cs_keypoint *keys1, const uint8_t *desc1, int N1, const int *node1, const uint8_t *skip1, const cs_keypoint *keys2,
                       const uint8_t *desc2, int N2, const int *node2, const uint8_t *skip2, float nnratio, int check_orientation, int *matches12, int *nmatches);
/* ORBmatcher::DescriptorDistance over all pairs: exact best / second-best per query (first index wins ties). */
int cs_hamming_knn2(cs_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx, int *best_dist, int *second_dist);

/* ===================================================================== object bundle adjustment
 * Replaces the g2o machinery behind ORB_SLAM2::Optimizer::BundleAdjustment (orb_object_slam/include/Optimizer.h:39-41,
 * src/Optimizer.cc:64-251) and Optimizer::LocalBACameraPointObjects (:826-1534): SparseOptimizer::optimize with
 * OptimizationAlgorithmLevenberg + BlockSolver_6_3 (Schur) (vendored g2o core/optimization_algorithm_levenberg.cpp:61-189,
 * core/block_solver.hpp:143-604) over VertexSE3Expmap / VertexSBAPointXYZ / VertexCuboidFixScale vertices and
 * EdgeSE3ProjectXYZ / EdgeSE3CuboidFixScaleProj / EdgePointCuboidOnlyObjectFixScale edges (types_six_dof_expmap.*,
 * orb_object_slam/include/g2o_Object.h, src/g2o_Object.cpp).  Residuals, Jacobians (analytic for reprojection, central
 * differences with delta 1e-9 for the cuboid edges, as g2o does), Hessian blocks, the block-Schur reduction, the reduced
 * solve (block-band Cholesky) and the landmark back-substitution run on the GPU; the LM control loop runs on the host.
 * Poses are 7-vectors [tx ty tz qx qy qz qw] (SE3Quat::toVector). */
typedef struct cs_ba_problem {
    int n_cams; const double *cam_pose; const uint8_t *cam_fixed;                /* VertexSE3Expmap, world-to-camera */
    int n_points; const double *points;                                         /* VertexSBAPointXYZ, marginalised */
    int n_cuboids; const double *cuboid_pose; const double *cuboid_scale;       /* VertexCuboidFixScale (pose + fixedscale / scale) */
    const uint8_t *cuboid_flags;  /* bit0 whether_fixrollpitch, bit1 whether_fixrotation, bit2 whether_fixheight, bit3 fixedscale set */
    int n_obs; const int *obs_cam; const int *obs_point; const double *obs_uv; const double *obs_inv_sigma2; /* EdgeSE3ProjectXYZ */
    double fx, fy, cx, cy, huber_mono;                                          /* Huber delta (sqrt(5.991)), 0 = no kernel */
    int n_cobs; const int *cobs_cam; const int *cobs_cuboid; const double *cobs_bbox; const double *cobs_info; /* EdgeSE3CuboidFixScaleProj */
    double K[9], huber_obj;
    int n_pc; const int *pc_cuboid; const int *pc_offsets; const double *pc_points; double max_outside_margin_ratio; /* EdgePointCuboidOnlyObjectFixScale */
    /* EdgeStereoSE3ProjectXYZ (types_six_dof_expmap.h:195-232, Optimizer.cc:158-184): obs_ur[o] >= 0 makes observation o a stereo edge
     * (u, v, u_right) with information inv_sigma2 * I3 and Huber(huber_stereo = sqrt(7.815)); NULL = all monocular */
    const double *obs_ur; double bf, huber_stereo;
} cs_ba_problem;

typedef struct cs_ba_stats {
    int iterations;            /* calls of OptimizationAlgorithmLevenberg::solve executed */
    int lm_trials;             /* linear solves (accepted + rejected trials) */
    double chi2_init, chi2_final, lambda_final;
    double chi2_trace[64];
} cs_ba_stats;

typedef struct cs_ba cs_ba;
/* all-reduce(sum) of n doubles in device memory across the ranks of a multi-GPU run; NULL for a single GPU */
typedef int (*cs_allreduce_fn)(void *user, double *device_buf, long n);

/* Builds the solver structure (index mapping, Schur pattern, ordering; BlockSolver::buildStructure) and uploads the graph.
 * rank/world: landmarks (and their observations) are sharded across `world` ranks, cameras and cuboids are replicated;
 * with world > 1 an all-reduce callback must be set. */
int cs_ba_create(cs_ctx *ctx, const cs_ba_problem *p, int rank, int world, cs_ba **out);
void cs_ba_destroy(cs_ctx *ctx, cs_ba *b);
int cs_ba_set_allreduce(cs_ba *b, cs_allreduce_fn fn, void *user);
/* SparseOptimizer::optimize(iterations).  stop_flag (may be NULL) is polled between iterations and LM trials like g2o's forceStopFlag
 * (sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149). */
int cs_ba_optimize(cs_ctx *ctx, cs_ba *b, int iterations, const volatile int *stop_flag, cs_ba_stats *stats);
/* SparseOptimizer::setForceStopFlag(bool*) (Optimizer.cc:80-81, 943-944): a second flag, ONE BYTE wide -- the C++ `bool` another thread of the
 * caller raises (LocalMapping::InterruptBA, mbStopGBA) -- polled at the same points of every later cs_ba_optimize on this graph; NULL removes it. */
int cs_ba_set_stop_flag_bool(cs_ba *b, const volatile unsigned char *flag);
/* current estimates (any pointer may be NULL) */
int cs_ba_read(cs_ctx *ctx, cs_ba *b, double *cam_pose, double *points, double *cuboid_pose);
/* computeActiveErrors + activeRobustChi2 at the current estimates (this rank's share when sharded); err_* may be NULL */
int cs_ba_errors(cs_ctx *ctx, cs_ba *b, double *chi2, double *err_obs, double *err_cobs, double *err_pc);
/* Dense copy of this rank's reduced camera system (before the all-reduce) for lambda: H (6P x 6P row-major), b (6P); *P out */
int cs_ba_reduced_dense(cs_ctx *ctx, cs_ba *b, double lambda, double *H, double *bvec, int *P);

/* ===================================================================== dynamic-object bundle adjustment
 * Replaces the g2o machinery behind ORB_SLAM2::Optimizer::LocalBACameraPointObjectsDynamic (orb_object_slam/include/Optimizer.h:57-58,
 * src/Optimizer.cc:1537-2573): SparseOptimizer::optimize with OptimizationAlgorithmLevenberg + BlockSolverX + LinearSolverDense (:1670-1678)
 * over VertexSE3Expmap (key frames), one VertexCuboidFixScale per (object, key frame) (:1729-1786), VelocityPlanarVelocity (:2160-2165),
 * static VertexSBAPointXYZ (world frame, :1819-1830) and dynamic ones (object frame, :1934-1945), both marginalised, and the edges
 * EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (:1843-1906), UnaryLocalPoint (:1947-1955), EdgeDynamicPointCuboidCamera (:1977-1993),
 * EdgePointCuboidOnlyObjectFixScale (:2096-2115), EdgeObjectMotion (:2192-2202), EdgeSE3CuboidFixScaleProj (:2279-2298)
 * (orb_object_slam/include/g2o_Object.h, src/g2o_Object.cpp).  The caller builds the graph (the map bookkeeping of :1540-1665 stays in the
 * reference) and flattens it into the arrays below; indices refer to these arrays.  *_level[o] != 0 = setLevel(1) (NULL: all zero);
 * obs_ur NULL or < 0 = monocular; every information matrix of the reference is diagonal.  The two stages of :2353-2415 are two handles:
 * optimize(5), read + errors, then a second problem with the levels / removed kernels, optimize(10). */
typedef struct cs_ba_dyn_problem {
    int n_cams; const double *cam_pose; const uint8_t *cam_fixed;                               /* world-to-camera 7-vectors */
    int n_objs; const double *obj_pose; const double *obj_scale; const uint8_t *obj_flags;      /* object-to-world; flags as in cs_ba_problem */
    int n_vels; const double *vel;                                                              /* [linear velocity, steering angle] */
    int n_points; const double *points;
    int n_dpoints; const double *dpoints;                                                       /* MapPoint::PosToObj */
    int fix_points;                                                                             /* fixPoint: setFixed(true) instead of setMarginalized(true) */
    int n_obs; const int *obs_cam; const int *obs_point; const double *obs_uv; const double *obs_ur; const double *obs_inv_sigma2; const uint8_t *obs_level;
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
    double ulp_info, ulp_scale[3], ulp_ratio;                                                   /* UnaryLocalPoint: info * I, objectscale, max_outside_margin_ratio */
    int n_dobs; const int *dobs_cam; const int *dobs_obj; const int *dobs_point; const double *dobs_uv; const double *dobs_inv_sigma2; const uint8_t *dobs_level;
    double K[9], huber_dyn;
    int n_mot; const int *mot_from; const int *mot_to; const int *mot_vel; const double *mot_dt; double mot_info[3];
    int n_cobs; const int *cobs_cam; const int *cobs_obj; const double *cobs_bbox; const double *cobs_info; const uint8_t *cobs_level; double huber_obj;
    int n_pc; const int *pc_obj; const int *pc_offsets; const double *pc_points; double pc_ratio;
} cs_ba_dyn_problem;

typedef struct cs_ba_dyn cs_ba_dyn;
/* initializeOptimization: index mapping (non-fixed cameras, object poses, velocities; landmarks = static then dynamic points), the
 * pose-landmark slot lists, upload.  CS_ERR_BAD_ARG for an index out of range or a pose system above 4000 scalars. */
int cs_ba_dyn_create(cs_ctx *ctx, const cs_ba_dyn_problem *p, cs_ba_dyn **out);
void cs_ba_dyn_destroy(cs_ctx *ctx, cs_ba_dyn *b);
/* SparseOptimizer::optimize(iterations); stop_flag as in cs_ba_optimize */
int cs_ba_dyn_optimize(cs_ctx *ctx, cs_ba_dyn *b, int iterations, const volatile int *stop_flag, cs_ba_stats *stats);
int cs_ba_dyn_set_stop_flag_bool(cs_ba_dyn *b, const volatile unsigned char *flag); /* as cs_ba_set_stop_flag_bool */
/* current estimates (any pointer may be NULL) */
int cs_ba_dyn_read(cs_ctx *ctx, cs_ba_dyn *b, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints);
/* computeError of every edge + activeRobustChi2 over the active ones: e_obs n x 3 (third 0 for mono), e_dobs n x 2, e_mot n x 3,
 * e_cobs n x 4, e_pc n x 3, e_ulp n_dpoints x 3; any output may be NULL */
int cs_ba_dyn_errors(cs_ctx *ctx, cs_ba_dyn *b, double *chi2, double *e_obs, double *e_dobs, double *e_mot, double *e_cobs, double *e_pc, double *e_ulp);
/* dense reduced pose system for lambda at the current estimates: H (n x n row-major), bvec (n); *n out (H NULL: query) */
int cs_ba_dyn_reduced_dense(cs_ctx *ctx, cs_ba_dyn *b, double lambda, double *H, double *bvec, int *n);

/* ===================================================================== LSD line detector
 * Replaces line_lbd_detect::detect_raw_lines / detect_filter_lines (line_lbd/include/line_lbd/line_lbd_allclass.h:37-52,
 * class/line_lbd_allclass.cpp:125-148,200-221) with use_LSD = true, one octave: LSDDetector::detectImpl
 * (libs/LSDDetector.cpp:153-287) over LineSegmentDetectorImpl (libs/lsd.cpp; LSD_REFINE_ADV, default parameters).
 * GPU: Gaussian blur + 0.8x bilinear resize in double, level-line angles and gradient norms (ll_angle).  Host (one thread
 * per frame): the pseudo-ordering and the inherently sequential region growing / rectangle / NFA stages. */
typedef struct cs_keyline {              /* the KeyLine fields (line_descriptor/descriptor.hpp:105-150) */
    float angle; int32_t class_id; int32_t octave; float pt_x, pt_y; float response; float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int32_t numOfPixels;
} cs_keyline;
typedef struct cs_lsd cs_lsd;
int cs_lsd_create(cs_ctx *ctx, int width, int height, int max_frames, cs_lsd **out);
void cs_lsd_destroy(cs_ctx *ctx, cs_lsd *l);
/* detect_raw_lines on n_frames gray images: per frame counts[f] KeyLines at out[f*cap_per_frame ...] */
int cs_lsd_detect(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride, cs_keyline *out, int cap_per_frame, int *counts);
/* detect_filter_lines(gray, linesmat): octave 0, lineLength > length_thres; lines: [f][cap_per_frame][4] floats (CV_32F N x 4) */
int cs_lsd_detect_filter_lines(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride, float length_thres,
                               float *lines, int cap_per_frame, int *counts);
/* the same rows from the KeyLines of the last cs_lsd_run over the resident frames (counts[f] rows of frame f at lines + f * cap * 4); n_frames: the frames
 * `lines` and `counts` have room for -- CS_ERR_BAD_ARG when the last run processed more */
int cs_lsd_read_filter_lines(cs_ctx *ctx, cs_lsd *l, float length_thres, float *lines, int cap, int *counts, int n_frames);
/* introspection after a detect call: scaled image, gradient norm and level-line angle maps (sw x sh doubles each) */
int cs_lsd_get_maps(cs_ctx *ctx, cs_lsd *l, int frame, double *scaled, double *modgrad, double *angles, int *sw, int *sh);

/* Resident-batch form of the line front-end (what bench.py times): frames stay in HBM; cs_lsd_run detects the KeyLines of every
 * uploaded frame and, with_lbd != 0, their LBD descriptors (= line_lbd_detect::detect_descrip_lines before its length filter,
 * class/line_lbd_allclass.cpp:222-256); cs_lsd_read returns one frame's lines (out == NULL: count only) and descriptors. */
int cs_lsd_upload(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride);
int cs_lsd_run(cs_ctx *ctx, cs_lsd *l, int with_lbd);
/* The region stage (lsd.cpp:464-535: region_grow ... rect_improve) of the last batch.  Batches of 512 frames and more run it on the device, one
 * wave per frame walking the reference's sequence (cube_slam_amd/csrc/lsd_regions.hip); smaller ones on the host's cores, one frame per thread;
 * cs_lsd_set_region_stage picks the stage for a detector's later runs.  Every stage gives the same KeyLines byte for byte.
 * out[0] 1 when the device stage was chosen, out[1] region_grow calls, out[2] rectangles that reached rect_improve,
 * out[3] 1 when the batch fell back to the host stage (a region larger than the device list), out[4] pixel-window fetches.  All 0 for the host stage. */
int cs_lsd_region_stats(cs_ctx *ctx, cs_lsd *l, long out[5]);
/* Which formulation of the region stage the detector's next runs take:
 *   CS_LSD_REGIONS_AUTO            the host stage below 512 frames per run, one wave per frame from 512 on (the default);
 *   CS_LSD_REGIONS_HOST            one frame per host thread (a frame: ~3.5 ms);
 *   CS_LSD_REGIONS_WAVE_PER_FRAME  one wave walks a frame's sequence (~110 ms per frame whatever the batch; 36 k frames/s with the chip full of frames);
 *   CS_LSD_REGIONS_BACKLOG         a walker lane per frame + rectangle waves (64 frames per wave slot, ~450 ms per frame, 134 k frames/s with the chip full): for an offline
 *                                  backlog of tens of thousands of frames whose lines nobody waits for. */
enum { CS_LSD_REGIONS_AUTO = 0, CS_LSD_REGIONS_HOST = 1, CS_LSD_REGIONS_WAVE_PER_FRAME = 2, CS_LSD_REGIONS_BACKLOG = 3 };
int cs_lsd_set_region_stage(cs_lsd *l, int stage);
int cs_lsd_read(cs_ctx *ctx, cs_lsd *l, int frame, cs_keyline *out, int cap, int *count, uint8_t *desc /* cap x 32 or NULL */);

/* ===================================================================== LBD line descriptor + matcher
 * Replaces BinaryDescriptor::compute (line_lbd/libs/binary_descriptor.cpp:588-790,1146-1509; what
 * line_lbd_detect::get_line_descriptors / detect_descrip_lines call, class/line_lbd_allclass.cpp:192-269) for one octave, and
 * line_lbd_detect::match_line_descrip (:339-356) = BinaryDescriptorMatcher::match (exact 1-NN in 256-bit Hamming space,
 * libs/binary_descriptor_matcher.cpp:196-262) followed by the distance threshold. */
int cs_lbd_compute(cs_ctx *ctx, const uint8_t *gray, int width, int height, int stride, const cs_keyline *keylines, int n,
                   uint8_t *desc /* n x 32 */, float *float_desc /* n x 72 or NULL */);
/* good matches: for every query whose nearest train descriptor is closer than dist_thres: (query, train, distance) */
int cs_lbd_match(cs_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, float dist_thres,
                 int *query_idx, int *train_idx, int *distance, int *n_matches);
/* introspection: the blurred image and the Sobel derivative maps of the last cs_lbd_compute-like pass over `gray` */
int cs_lbd_maps(cs_ctx *ctx, const uint8_t *gray, int width, int height, int stride, uint8_t *blur, int16_t *dx, int16_t *dy);

/* ===================================================================== Optimizer::PoseOptimization  
 * Replaces ORB_SLAM2::Optimizer::PoseOptimization(Frame*) (orb_object_slam/include/Optimizer.h:51, src/Optimizer.cc:253-472) for a
 * batch of frames: frame f owns the matches [edge_off[f], edge_off[f+1]) -- Xw (world point), obs (u, v, u_right; u_right < 0 =
 * monocular, Optimizer.cc:303), inv_sigma2 (mvInvLevelSigma2[octave]); intrinsics n_frames x 5 = fx fy cx cy bf; poses = Tcw as
 * [t, qx qy qz qw].  Out: optimised pose, per-match mvbOutlier flags, n_inliers[f] = nInitialCorrespondences - nBad (:471).
 * One workgroup per frame runs the 4 x 10 Levenberg-Marquardt rounds with the re-classification in between. */
int cs_pose_optimization(cs_ctx *ctx, int n_frames, const int *edge_off, const double *Xw, const double *obs, const double *inv_sigma2, const double *intrinsics,
                         const double *pose_in, double *pose_out, uint8_t *outlier, int *n_inliers);

/* ===================================================================== batch front-end runner
 * One pass of the per-frame path (ORBextractor::operator(), line_lbd_detect::detect_descrip_lines, detect_3d_cuboid::detect_cuboid --
 * what Tracking / main_obj.cpp call per frame, object_slam/src/main_obj.cpp:395-470) over a batch that is resident in HBM.  Handles
 * are borrowed: orb / batch run on `ctx` in the calling thread, every line detector on its own context in a worker thread; with two
 * detectors (both holding the same frames) consecutive passes alternate between them, so the host stage of LSD overlaps GPU work.
 * cs_frontend_step returns when ORB and the cuboid batch of this pass are done; cs_frontend_drain waits for the line passes.
 * Streams: the caller's, one per line detector and, for batches whose region stage runs on the device, one lowest-priority background stream per detector for
 * that stage's one long kernel.  The HIP runtime serialises streams that share a hardware queue and creates four by default: export GPU_MAX_HW_QUEUES=16 (or
 * more) before the runtime starts in a process that uses the runner. */
typedef struct cs_frontend cs_frontend;
int cs_frontend_create(cs_ctx *ctx, cs_orb *orb /* nullable */, cs_cuboid_batch *batch /* nullable */, int n_line_workers, cs_ctx *const *line_ctx,
                       cs_lsd *const *lsd, cs_frontend **out);
/* 1: the process offers at least as many hardware queues (GPU_MAX_HW_QUEUES as the process saw it at cs_frontend_create, 4 when unset) as the runner keeps streams
 * busy (1 + 2 per line worker); 0: it does not -- cs_frontend_create has said so on stderr -- and streams that share a queue serialise; < 0: error. */
int cs_frontend_queues(const cs_frontend *fe, int *streams, int *hw_queues);
int cs_frontend_step(cs_frontend *fe);
int cs_frontend_drain(cs_frontend *fe);
/* Streaming source: the frames of every step arrive from the HOST (main_obj.cpp:420-449 / Frame.cc:320-326 take a new image per call) instead of staying resident.
 * cs_frontend_stream_begin gives the runner a ring of n_slots (2..8) device slots of n_frames x height x width bytes and a copy stream of its own;
 * cs_frontend_stream_push(gray) enqueues the H2D copy of the frames of the next step that has none yet (pinned host memory: asynchronous; it waits on the device for the
 * slot's last readers, on the host only for the line pass that last used the slot to have taken its frames) -- push step k + 1 before calling step k and the upload
 * overlaps step k; cs_frontend_step then feeds ORB, the cuboid batch (same boxes / poses / plan, new pixels) and the step's line pass from the slot by device copies.
 * A step without pushed frames is CS_ERR_BAD_ARG, a push with every slot waiting for its step CS_ERR_CAPACITY.  Not with phased passes.  cs_frontend_stream_end drains
 * and returns to resident frames. */
int cs_frontend_stream_begin(cs_frontend *fe, int n_frames, int width, int height, int n_slots);
int cs_frontend_stream_push(cs_frontend *fe, const uint8_t *gray);
/* ... with what detect_cuboid takes beside the pixels (detect_3d_cuboid/include/detect_3d_cuboid/detect_3d_cuboid.h:62-63, object_slam/src/main_obj.cpp:420-449): the
 * frames' camera poses (n_frames x 16), their 2-D boxes (box_offsets[n_frames + 1], rows of 5) and their edge lists (line_offsets NULL: the batch's lists stay).  The step that takes the slot hands them to the cuboid batch (cs_cuboid_batch_set_scene: the plan is rebuilt for them). */
int cs_frontend_stream_push_scene(cs_frontend *fe, const uint8_t *gray, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines);
int cs_frontend_stream_end(cs_frontend *fe);
/* The results of the step that has just been enqueued, copied to the caller's (pinned) buffers on a copy stream of the ring behind the step's kernels: ORB key points /
 * descriptors packed like cs_orb_read_packed (first / total are filled at once), the cuboids like cs_cuboid_batch_read; either pair may be NULL.  Returns at once; the next
 * step's kernels wait on the device for the copies before they overwrite what is read; cs_frontend_stream_read_wait blocks until the copies of the last call have arrived
 * (and reports the batch's status).  One call per step at most. */
int cs_frontend_stream_read_async(cs_frontend *fe, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total, cs_cuboid *cuboids, int *counts);
int cs_frontend_stream_read_wait(cs_frontend *fe);
/* the device-side hand-overs the streaming source uses (a copy on the context's stream, nothing waits): frames one behind the other, rows of `width` bytes */
int cs_orb_set_frames_device(cs_ctx *ctx, cs_orb *e, const uint8_t *d_gray, int n_frames);
int cs_lsd_set_frames_device(cs_ctx *ctx, cs_lsd *l, const uint8_t *d_gray, int n_frames);
int cs_cuboid_batch_set_gray_device(cs_ctx *ctx, cs_cuboid_batch *b, const uint8_t *d_gray);
/* Phased passes (off by default; results are the same either way).  With the device region stage of LSD (batches of >= 512 frames) a
 * super-step = one pass per line detector: the detectors stop in front of the region stage, and after the last pass of the super-step
 * (or at cs_frontend_drain) the region stages of all of them run together while `ctx`'s stream is idle -- that cs_frontend_step
 * returns when they have left the GPU.  The cuboid score kernel and the one-wave-per-frame region kernel then never share a CU. */
int cs_frontend_set_phased(cs_frontend *fe, int on);
/* The reference's chain (object_slam/src/main_obj.cpp:428-449: detect_cuboid consumes this frame's detect_filter_lines) as a pipeline: on != 0 makes
 * step k hand the lines of line pass k - W (octave 0, lineLength > length_thres; line pass k belongs to step k, W = number of line workers) to the cuboid
 * batch before that batch runs: a pass that was started at least W steps earlier, so in the steady state no step waits for a line pass.  The first W steps
 * after the switch run on the lists the batch holds.  Needs a cuboid batch and >= 1 line worker. */
int cs_frontend_set_chain(cs_frontend *fe, int on, float length_thres);
/* A backlog: `n_steps` more cs_frontend_step calls will follow on the frames the detectors hold.  The line passes of those steps are then started as soon as a
 * worker is free -- at most 2 W passes ahead of the caller's step -- instead of one per step, so the line pipeline is full from the first step and does not
 * drain behind the last one.  Nothing is added or dropped: step k still needs line pass k to have been started, cs_frontend_drain waits for every pass in
 * flight and cuts what is left of the backlog (passes already done are found done by the steps that follow).  No effect on phased passes. */
int cs_frontend_set_backlog(cs_frontend *fe, int n_steps);
/* The cuboid batch on a stream of its own: its launches (no host round trip among them) are enqueued on `cuboid_ctx` and run beside the ORB pass of the same step
 * instead of in front of the next one.  NULL: back onto the caller's stream.  cs_frontend_drain waits for that stream too. */
int cs_frontend_set_cuboid_ctx(cs_frontend *fe, cs_ctx *cuboid_ctx);
void cs_frontend_destroy(cs_frontend *fe);

/* ===================================================================== 9-dof g2o::cuboid of object_slam
 * Replaces, for batches, g2o::VertexCuboid::oplusImpl (object_slam/include/object_slam/g2o_Object.h:193-204: pose * exp(update[0:6]),
 * scale + update[6:9]) and g2o::EdgeSE3Cuboid::computeError (:227-252: cuboid::min_log_error of the measured cuboid moved to the
 * world by the camera, :77-101) with the numeric Jacobians BaseBinaryEdge::linearizeOplus builds for it (central differences, 1e-9;
 * Jcam n x 9 x 6, Jcub n x 9 x 9, row-major; pass both NULL for residuals only).  cuboid = [t, qx qy qz qw, half scale] (10 doubles). */
int cs_cuboid9_oplus(cs_ctx *ctx, int n, const double *cub, const double *upd, double *out);
int cs_cuboid9_edge_linearize(cs_ctx *ctx, int n, const double *cam_Tcw, const double *cub_global, const double *cub_meas_local, double *err, double *Jcam, double *Jcub);

/* ===================================================================== object association after detect_cuboid
 * cs_associate_keypoints replaces the keypoint -> local cuboid association of Tracking::DetectCuboid
 * (orb_object_slam/src/Tracking.cc:1717-1775, with bboxOverlapratio detect_3d_cuboid/src/object_3d_util.cpp:650-654) for a batch of
 * keyframes: boxes are cv::Rect (x, y, w, h) of pKF->local_cuboids in order, keypoints mvKeys[i].pt.  Out: keypoint_associate_objectID
 * (index of the single non-overlapped box that contains the rounded keypoint, else -1), keypoint_inany_object (only meaningful with
 * enable_ground_height_scale, may be NULL) and the per-box `overlapped` flags (IoU > 0.15 with an earlier unflagged box, may be NULL).
 * At most 64 boxes per frame. */
int cs_associate_keypoints(cs_ctx *ctx, int n_frames, const int *kp_off, const float *kp_xy, const int *box_off, const int *boxes, int enable_ground_height_scale,
                           int *assoc, uint8_t *inany, uint8_t *overlapped);
/* Tracking::AssociateCuboids (Tracking.cc:1848-1990, use_truth_trackid = false) on index arrays instead of MapObject* / MapPoint*:
 * candidates (cand_id = the id a candidate gets as a landmark, its GetPotentialMapPoints() as CSR cand_off / cand_pts) are taken
 * in order against LocalObjectsLandmarks (landmark_id, landmark_bad = isBad()), counting for every landmark the candidate's points
 * whose MapObjObservations (CSR pobs_off / pobs_obj / pobs_cnt, pobs_cnt NULL = all 1) contain it; the first landmark in list
 * order with the strictly largest count above the threshold absorbs the candidate (MergeIntoLandmark), otherwise the candidate
 * becomes a landmark (SetAsLandmark) and joins the list before the NEXT candidate; both add one vote per point
 * (MapPoint::AddObjectObservation, MapPoint.cc:219-242), which later candidates see.  Out: assoc[i] = landmark id candidate i ends
 * up in, created[i]; best_object / max_vote (n_points, in/out, may be NULL) follow the `best_object` bookkeeping of the points;
 * the changed votes come back as (point, object, new count) triples (upd_*, may be NULL; *n_upd = their number).  Host code: the
 * loop is serial by construction. */
int cs_associate_cuboids(int n_cand, const int *cand_id, const int *cand_off, const int *cand_pts, int n_landmarks, const int *landmark_id, const uint8_t *landmark_bad,
                         int n_points, const int *pobs_off, const int *pobs_obj, const int *pobs_cnt, int *best_object, int *max_vote, int largest_shared_num_points_thres,
                         int *assoc, uint8_t *created, int upd_cap, int *upd_point, int *upd_obj, int *upd_cnt, int *n_upd);

#ifdef __cplusplus
}
#endif
#endif
