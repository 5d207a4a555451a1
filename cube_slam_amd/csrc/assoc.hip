// Object association step that follows detect_cuboid in the tracking thread.
//   assoc_keypoints    per-keyframe keypoint -> local cuboid association from the 2-D boxes (Tracking::DetectCuboid,
//                      orb_object_slam/src/Tracking.cc:1717-1775; bboxOverlapratio, detect_3d_cuboid/src/object_3d_util.cpp:650-654):
//                      one workgroup per frame, the few boxes' overlap flags by thread 0 in the reference's order, then thread per keypoint.
//   cs_associate_cuboids  Tracking::AssociateCuboids (Tracking.cc:1848-1990): candidates are taken one at a time, every decision changes
//                      the point -> object votes the next candidate sees (SetAsLandmark / MergeIntoLandmark, MapObject.cc:100-115,
//                      MapPoint::AddObjectObservation MapPoint.cc:219-250) -- a short serial loop over tens of objects with no data
//                      parallelism, so it stays host code (the reference's is too); it sits behind the same C-ABI so that the step
//                      between detect_cuboid and the object BA needs nothing from the reference's map classes.
#include "common.h"

#include <algorithm>
#include <map>
#include <vector>

namespace {
constexpr int MAX_BOXES = 64;

// cv::Rect & cv::Rect, areas in int, ratio in float (object_3d_util.cpp:650-654)
__host__ __device__ inline float bbox_overlap_ratio(const int *a, const int *b) {
    int x1 = max(a[0], b[0]), y1 = max(a[1], b[1]);
    int w = min(a[0] + a[2], b[0] + b[2]) - x1, h = min(a[1] + a[3], b[1] + b[3]) - y1;
    if (w <= 0 || h <= 0) { w = 0; h = 0; } // operator& returns an empty Rect
    const int ov = w * h;
    return (float)ov / ((float)(a[2] * a[3] + b[2] * b[3] - ov));
}

// frame f: keypoints kp_off[f]..kp_off[f+1]-1 (x, y), boxes box_off[f]..box_off[f+1]-1 (x, y, w, h)
__global__ void __launch_bounds__(256) assoc_keypoints(const int *kp_off, const float *kp_xy, const int *box_off, const int *boxes, int ground_height_mode,
                                                       int *assoc, uint8_t *inany, uint8_t *overlapped) {
    __shared__ int s_box[MAX_BOXES * 4];
    __shared__ uint8_t s_ov[MAX_BOXES];
    const int f = blockIdx.x, b0 = box_off[f], nb = min(box_off[f + 1] - b0, MAX_BOXES);
    for (int i = threadIdx.x; i < nb * 4; i += 256) s_box[i] = boxes[(long)b0 * 4 + i];
    for (int i = threadIdx.x; i < nb; i += 256) s_ov[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { // :1722-1736: a box already marked is skipped as i, and never tested again as j
        for (int i = 0; i < nb; i++)
            if (!s_ov[i])
                for (int j = i + 1; j < nb; j++)
                    if (!s_ov[j] && (double)bbox_overlap_ratio(s_box + i * 4, s_box + j * 4) > 0.15) { s_ov[i] = 1; s_ov[j] = 1; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256) overlapped[b0 + i] = s_ov[i];
    for (int k = kp_off[f] + threadIdx.x; k < kp_off[f + 1]; k += 256) {
        // Rect::contains(Point2f -> Point2i): saturate_cast<int>(float) is cvRound = round to nearest even
        const int px = __float2int_rn(kp_xy[(long)k * 2]), py = __float2int_rn(kp_xy[(long)k * 2 + 1]);
        int id = -1, times = 0; bool any = false;
        for (int j = 0; j < nb; j++) {
            const int *r = s_box + j * 4;
            const bool in = r[0] <= px && px < r[0] + r[2] && r[1] <= py && py < r[1] + r[3];
            if (ground_height_mode) { // :1755-1772: inany counts overlapped boxes too
                if (in) { any = true; if (!s_ov[j]) { times++; id = times == 1 ? j : -1; } }
            } else if (!s_ov[j] && in) { times++; id = times == 1 ? j : -1; } // :1739-1753
        }
        assoc[k] = id;
        if (inany) inany[k] = any ? 1 : 0;
    }
}
} // namespace

extern "C" {

int cs_associate_keypoints(cs_ctx *ctx, int n_frames, const int *kp_off, const float *kp_xy, const int *box_off, const int *boxes, int enable_ground_height_scale,
                           int *assoc, uint8_t *inany, uint8_t *overlapped) {
    if (!ctx || n_frames < 0 || !kp_off || !box_off || !assoc) return CS_ERR_BAD_ARG;
    if (n_frames == 0) return CS_OK;
    const int nk = kp_off[n_frames], nb = box_off[n_frames];
    if (nk < 0 || nb < 0 || (nk && !kp_xy) || (nb && !boxes)) return CS_ERR_BAD_ARG;
    for (int f = 0; f < n_frames; f++) if (box_off[f + 1] - box_off[f] > MAX_BOXES || kp_off[f + 1] < kp_off[f] || box_off[f + 1] < box_off[f]) return CS_ERR_CAPACITY;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    int *d_ko = nullptr, *d_bo = nullptr, *d_bx = nullptr, *d_as = nullptr; float *d_xy = nullptr; uint8_t *d_in = nullptr, *d_ov = nullptr;
    int r = cs_dalloc(ctx, &d_ko, (size_t)n_frames + 1);
    if (!r) r = cs_dalloc(ctx, &d_bo, (size_t)n_frames + 1);
    if (!r) r = cs_dalloc(ctx, &d_bx, (size_t)std::max(nb, 1) * 4);
    if (!r) r = cs_dalloc(ctx, &d_as, (size_t)std::max(nk, 1));
    if (!r) r = cs_dalloc(ctx, &d_xy, (size_t)std::max(nk, 1) * 2);
    if (!r) r = cs_dalloc(ctx, &d_in, (size_t)std::max(nk, 1));
    if (!r) r = cs_dalloc(ctx, &d_ov, (size_t)std::max(nb, 1));
    if (!r) r = cs_h2d(ctx, d_ko, kp_off, (size_t)n_frames + 1);
    if (!r) r = cs_h2d(ctx, d_bo, box_off, (size_t)n_frames + 1);
    if (!r && nb) r = cs_h2d(ctx, d_bx, boxes, (size_t)nb * 4);
    if (!r && nk) r = cs_h2d(ctx, d_xy, kp_xy, (size_t)nk * 2);
    if (!r) {
        CS_LAUNCH(ctx, "assoc_keypoints", assoc_keypoints, dim3(n_frames), dim3(256), 0, d_ko, d_xy, d_bo, d_bx, enable_ground_height_scale, d_as, d_in, d_ov);
        if (nk) r = cs_d2h(ctx, assoc, d_as, (size_t)nk);
        if (!r && nk && inany) r = cs_d2h(ctx, inany, d_in, (size_t)nk);
        if (!r && nb && overlapped) r = cs_d2h(ctx, overlapped, d_ov, (size_t)nb);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_ko, d_bo, d_bx, d_as, d_xy, d_in, d_ov};
    for (void *p : ptrs) if (p) hipFree(p);
    return r;
}

int cs_associate_cuboids(int n_cand, const int *cand_id, const int *cand_off, const int *cand_pts, int n_landmarks, const int *landmark_id, const uint8_t *landmark_bad,
                         int n_points, const int *pobs_off, const int *pobs_obj, const int *pobs_cnt, int *best_object, int *max_vote, int largest_shared_num_points_thres,
                         int *assoc, uint8_t *created, int upd_cap, int *upd_point, int *upd_obj, int *upd_cnt, int *n_upd) {
    if (n_cand < 0 || n_landmarks < 0 || n_points < 0 || (n_cand && (!cand_id || !cand_off || !assoc || !created)) || (n_landmarks && !landmark_id) || (n_points && !pobs_off))
        return CS_ERR_BAD_ARG;
    std::vector<std::map<int, int>> votes((size_t)n_points); // MapPoint::MapObjObservations
    for (int p = 0; p < n_points; p++)
        for (int e = pobs_off[p]; e < pobs_off[p + 1]; e++) votes[p][pobs_obj[e]] = pobs_cnt ? pobs_cnt[e] : 1;
    std::vector<int> L(landmark_id, landmark_id + n_landmarks); // LocalObjectsLandmarks, in order
    std::map<int, bool> bad;
    for (int j = 0; j < n_landmarks; j++) bad[landmark_id[j]] = landmark_bad && landmark_bad[j];
    std::map<std::pair<int, int>, int> touched; // (point, object) -> count after this call
    auto add_observation = [&](int p, int obj) { // MapPoint::AddObjectObservation, already_associated branch (MapPoint.cc:223-242)
        int &c = votes[p][obj];
        c++;
        touched[std::make_pair(p, obj)] = c;
        if (best_object && max_vote && c > max_vote[p]) { best_object[p] = obj; max_vote[p] = c; }
    };
    int last_new = -1;
    for (int i = 0; i < n_cand; i++) {
        if (last_new >= 0) { L.push_back(last_new); bad[last_new] = false; } // Tracking.cc:1891-1893
        last_new = -1;
        for (int e = cand_off[i]; e < cand_off[i + 1]; e++) if (cand_pts[e] < 0 || cand_pts[e] >= n_points) return CS_ERR_BAD_ARG;
        int best = -1;
        if (!L.empty()) { // :1900-1923
            std::map<int, int> counter;
            for (int e = cand_off[i]; e < cand_off[i + 1]; e++)
                for (const auto &kv : votes[cand_pts[e]]) counter[kv.first]++;
            int largest = largest_shared_num_points_thres;
            for (int obj : L)
                if (!bad[obj]) {
                    auto it = counter.find(obj);
                    if (it != counter.end() && it->second > largest) { largest = it->second; best = obj; }
                }
        }
        if (best < 0) { // :1934-1959 new landmark: SetAsLandmark()
            for (int e = cand_off[i]; e < cand_off[i + 1]; e++) add_observation(cand_pts[e], cand_id[i]);
            assoc[i] = cand_id[i]; created[i] = 1;
            last_new = cand_id[i];
        } else { // :1960-1982 MergeIntoLandmark()
            for (int e = cand_off[i]; e < cand_off[i + 1]; e++) add_observation(cand_pts[e], best);
            assoc[i] = best; created[i] = 0;
        }
    }
    int k = 0;
    for (const auto &kv : touched) {
        if (k < upd_cap && upd_point && upd_obj && upd_cnt) { upd_point[k] = kv.first.first; upd_obj[k] = kv.first.second; upd_cnt[k] = kv.second; }
        k++;
    }
    if (n_upd) *n_upd = k;
    return (upd_point && k > upd_cap) ? CS_ERR_CAPACITY : CS_OK;
}

} // extern "C"
