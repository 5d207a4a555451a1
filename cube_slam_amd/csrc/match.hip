// match.hip -- ORB_SLAM2::ORBmatcher Hamming searches on MI355X (gfx950).
//
// Replaces ORBmatcher::SearchByProjection (both overloads), SearchForInitialization and DescriptorDistance
// (reference orb_object_slam/src/ORBmatcher.cc:50-142, :429-542, :1373-1522, :1905-1921) and the Frame grid
// (src/Frame.cc:303-318, :404-459, :525-535).
//
//   match_grid        counting sort of the keypoints into the 64x48 grid, per-cell lists in keypoint order
//   match_project     per last-frame map point: Rcw*x+tcw (cv::gemm: double accumulate, one rounding), pinhole projection,
//                     window radius th*scale[octave]
//   match_candidates  one wave per query: lane = grid cell of the window (ix outer, iy inner = the reference's candidate
//                     order), level + |dx|<r,|dy|<r filters, ordered compaction by a wave prefix sum, 256-bit Hamming
//                     distance as 4 x popcount(u64).  pass 0 counts, pass 1 fills a CSR list (idx, dist).
//   match_knn2        all-pairs best / second best, train descriptors staged through LDS
// The greedy claim / ratio / rotation-histogram logic of the reference is sequential over the queries (a candidate claimed by
// an earlier query is skipped by later ones); it runs on the host over the CSR lists.
#include "common.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
constexpr int GRID_ROWS = 48, GRID_COLS = 64, NCELL = GRID_ROWS * GRID_COLS; // Frame.h:32-33
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;                // ORBmatcher.cc:42-44

struct FrameP { int N; float minX, maxX, minY, maxY, wInv, hInv; };

__global__ void __launch_bounds__(1024) match_grid(FrameP F, const cs_keypoint *keys, int *cell_start /*NCELL+1*/, int *cell_items, int *kp_cell) {
    __shared__ int cnt[NCELL];
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c < NCELL; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < F.N; i += 1024) { // PosInGrid, Frame.cc:525-535
        int px = (int)roundf((keys[i].x - F.minX) * F.wInv), py = (int)roundf((keys[i].y - F.minY) * F.hInv);
        int c = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
        kp_cell[i] = c;
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan over the cells: 3 cells per thread
    int c0 = cnt[tid * 3], c1 = cnt[tid * 3 + 1], c2 = cnt[tid * 3 + 2];
    s_part[tid] = c0 + c1 + c2;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int base = s_part[tid] - (c0 + c1 + c2);
    cell_start[tid * 3] = base; cell_start[tid * 3 + 1] = base + c0; cell_start[tid * 3 + 2] = base + c0 + c1;
    if (tid == 1023) cell_start[NCELL] = s_part[1023];
    __syncthreads();
    cnt[tid * 3] = base; cnt[tid * 3 + 1] = base + c0; cnt[tid * 3 + 2] = base + c0 + c1; // running fill positions
    __syncthreads();
    for (int i = tid; i < F.N; i += 1024) { int c = kp_cell[i]; if (c >= 0) cell_items[atomicAdd(&cnt[c], 1)] = i; }
    __syncthreads();
    // mGrid[x][y] holds the indices in push_back order = ascending keypoint index: sort each (tiny) cell list
    for (int c = tid; c < NCELL; c += 1024) {
        int b = cell_start[c], e = cnt[c];
        for (int i = b + 1; i < e; i++) {
            int v = cell_items[i], j = i - 1;
            while (j >= b && cell_items[j] > v) { cell_items[j + 1] = cell_items[j]; j--; }
            cell_items[j + 1] = v;
        }
    }
}

// cv::undistortPoints(src, dst, K, D, Mat(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call it (Frame.cc:546-609): the
// classic cvUndistortPoints of OpenCV 2.4 - 3.2 -- normalise with the double copies of the float intrinsics, five fixed-point
// iterations of the Brown model (k1 k2 p1 p2 k3), re-project with P = K, round to float.  (Later OpenCV versions stop the iteration on
// a reprojection-error criterion; the reference's target version is not pinned)  -ffp-contract=off: plain double ops.
struct UndP { double fx, fy, cx, cy, k[5]; int identity; };
__host__ __device__ inline void undistort_point(const UndP &U, float xin, float yin, float &xo, float &yo) {
    if (U.identity) { xo = xin; yo = yin; return; }
    const double ifx = 1. / U.fx, ify = 1. / U.fy;
    double x = ((double)xin - U.cx) * ifx, y = ((double)yin - U.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((U.k[4] * r2 + U.k[1]) * r2 + U.k[0]) * r2);
        const double deltaX = 2 * U.k[2] * x * y + U.k[3] * (r2 + 2 * x * x);
        const double deltaY = U.k[2] * (r2 + 2 * y * y) + 2 * U.k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = U.fx * x + 0.0 * y + U.cx, yy = 0.0 * x + U.fy * y + U.cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    xo = (float)(xx * ww); yo = (float)(yy * ww);
}
__global__ void match_undistort(int n, const cs_keypoint *in, UndP U, cs_keypoint *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cs_keypoint k = in[i];
    float x, y;
    undistort_point(U, k.x, k.y, x, y);
    k.x = x; k.y = y;
    out[i] = k;
}

struct Query { float x, y, r; int minLevel, maxLevel, valid; };

__global__ void match_project(int n, const float *world_pos, const uint8_t *valid, const int *octave, const float *T, float fx, float fy, float cx,
                              float cy, const float *scale_factors, float th, FrameP F, Query *q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Query Q{0, 0, 0, 0, 0, 0};
    if (valid[i]) {
        float x3Dc[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)T[r * 4 + k] * (double)world_pos[i * 3 + k];
            x3Dc[r] = (float)(s * 1.0 + (double)T[r * 4 + 3] * 1.0);
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (!(invzc < 0)) {
            float u = fx * xc * invzc + cx;
            float v = fy * yc * invzc + cy;
            if (!(u < F.minX || u > F.maxX) && !(v < F.minY || v > F.maxY)) {
                int o = octave[i];
                Q.x = u; Q.y = v; Q.r = th * scale_factors[o]; Q.minLevel = o - 1; Q.maxLevel = o + 1; Q.valid = 1;
            }
        }
    }
    q[i] = Q;
}

__device__ __forceinline__ int hamming256(const unsigned long long *a, unsigned long long b0, unsigned long long b1, unsigned long long b2,
                                          unsigned long long b3) {
    return __popcll(a[0] ^ b0) + __popcll(a[1] ^ b1) + __popcll(a[2] ^ b2) + __popcll(a[3] ^ b3);
}

// GetFeaturesInArea (Frame.cc:404-459) for every query + distances.  pass 0: counts[q]; pass 1: CSR fill.
__global__ void __launch_bounds__(256) match_candidates(FrameP F, const cs_keypoint *keys, const unsigned long long *desc, const int *cell_start,
                                                        const int *cell_items, int nq, const Query *q, const unsigned long long *qdesc, int pass,
                                                        int *counts, const int *offsets, int2 *cands) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const Query Q = q[qi];
    int total = 0;
    if (Q.valid) {
        const float x = Q.x, y = Q.y, r = Q.r;
        const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x - F.minX + r) * F.wInv));
        const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y - F.minY + r) * F.hInv));
        if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0)) {
            const bool bCheckLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
            const int ny = nMaxCellY - nMinCellY + 1, ncell = (nMaxCellX - nMinCellX + 1) * ny;
            unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            if (pass == 1 && qdesc) { d0 = qdesc[(long)qi * 4]; d1 = qdesc[(long)qi * 4 + 1]; d2 = qdesc[(long)qi * 4 + 2]; d3 = qdesc[(long)qi * 4 + 3]; }
            const long base = pass == 1 ? offsets[qi] : 0;
            for (int k0 = 0; k0 < ncell; k0 += 64) {
                const int k = k0 + lane;
                int b = 0, e = 0;
                if (k < ncell) { int c = (nMinCellX + k / ny) * GRID_ROWS + nMinCellY + k % ny; b = cell_start[c]; e = cell_start[c + 1]; }
                int mine = 0;
                for (int p = b; p < e; p++) {
                    const cs_keypoint kp = keys[cell_items[p]];
                    bool ok = true;
                    if (bCheckLevels) { if (kp.octave < Q.minLevel) ok = false; if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) ok = false; }
                    const float distx = kp.x - x, disty = kp.y - y;
                    ok = ok && fabsf(distx) < r && fabsf(disty) < r;
                    mine += ok;
                }
                int inc = mine;
                for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
                if (pass == 1 && mine) {
                    long o = base + total + inc - mine;
                    for (int p = b; p < e; p++) {
                        const int id = cell_items[p];
                        const cs_keypoint kp = keys[id];
                        bool ok = true;
                        if (bCheckLevels) { if (kp.octave < Q.minLevel) ok = false; if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) ok = false; }
                        const float distx = kp.x - x, disty = kp.y - y;
                        ok = ok && fabsf(distx) < r && fabsf(disty) < r;
                        if (ok) { cands[o] = make_int2(id, hamming256(desc + (long)id * 4, d0, d1, d2, d3)); o++; }
                    }
                }
                total += __shfl(inc, 63);
            }
        }
    }
    if (pass == 0 && lane == 0) counts[qi] = total;
}

__global__ void __launch_bounds__(1024) match_scan(int n, const int *counts, int *offsets) { // exclusive scan, single block
    __shared__ int s[1024];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        int i = b + threadIdx.x, v = i < n ? counts[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) offsets[i] = s_run + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = s_run;
}

__global__ void __launch_bounds__(256) match_knn2(const unsigned long long *q, int nq, const unsigned long long *t, int nt, int *best_idx, int *best_dist,
                                                  int *second_dist) {
    __shared__ unsigned long long tile[256 * 4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long a[4] = {0, 0, 0, 0};
    if (i < nq) for (int k = 0; k < 4; k++) a[k] = q[(long)i * 4 + k];
    int b = INT_MAX, b2 = INT_MAX, bi = -1;
    for (int j0 = 0; j0 < nt; j0 += 256) {
        const int nj = min(256, nt - j0);
        __syncthreads();
        for (int k = threadIdx.x; k < nj * 4; k += 256) tile[k] = t[(long)j0 * 4 + k];
        __syncthreads();
        if (i < nq)
            for (int j = 0; j < nj; j++) {
                int d = hamming256(a, tile[j * 4], tile[j * 4 + 1], tile[j * 4 + 2], tile[j * 4 + 3]);
                if (d < b) { b2 = b; b = d; bi = j0 + j; }
                else if (d < b2) b2 = d;
            }
    }
    if (i < nq) { best_idx[i] = bi; best_dist[i] = b; second_dist[i] = b2; }
}

// SearchForTriangulation (:679-850): one wave per keypoint of KF1; lanes stride over the KF2 features of the same vocabulary node
// (ascending index).  The reference keeps the LAST candidate that reaches the running minimum (`dist > bestDist -> continue` lets
// equal distances through and the winner only changes on a candidate that passes the geometric tests), i.e. among the candidates
// that pass every test the smallest distance and, for equal distances, the largest index: min over the key dist << 20 | (2^20-1 - idx2).
struct TriP { float F12[9]; float ex, ey; int only_stereo; };
__global__ void __launch_bounds__(256) match_triangulation(int N1, const cs_keypoint *keys1, const unsigned long long *desc1, const int *node1, const uint8_t *skip1,
                                                           const float *ur1, const cs_keypoint *keys2, const unsigned long long *desc2, const uint8_t *skip2,
                                                           const float *ur2, const int *node_start2, const int *node_items2, int n_nodes, TriP P,
                                                           const float *scale2, const float *sigma2_2, int *matches12) {
    const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i1 >= N1) return;
    unsigned best = 0xffffffffu;
    const int nd = node1[i1];
    const bool stereo1 = ur1[i1] >= 0;
    if (nd >= 0 && nd < n_nodes && !skip1[i1] && !(P.only_stereo && !stereo1)) {
        const cs_keypoint kp1 = keys1[i1];
        const unsigned long long a0 = desc1[(long)i1 * 4], a1 = desc1[(long)i1 * 4 + 1], a2 = desc1[(long)i1 * 4 + 2], a3 = desc1[(long)i1 * 4 + 3];
        // epipolar line l = x1' F12 (CheckDistEpipolarLine :152-169), float, no contraction
        const float a = kp1.x * P.F12[0] + kp1.y * P.F12[3] + P.F12[6];
        const float b = kp1.x * P.F12[1] + kp1.y * P.F12[4] + P.F12[7];
        const float c = kp1.x * P.F12[2] + kp1.y * P.F12[5] + P.F12[8];
        const float den = a * a + b * b;
        for (int p = node_start2[nd] + lane; p < node_start2[nd + 1]; p += 64) {
            const int i2 = node_items2[p];
            if (skip2[i2]) continue;
            const bool stereo2 = ur2[i2] >= 0;
            if (P.only_stereo && !stereo2) continue;
            const int dist = hamming256(desc2 + (long)i2 * 4, a0, a1, a2, a3);
            if (dist > TH_LOW) continue;
            const cs_keypoint kp2 = keys2[i2];
            if (!stereo1 && !stereo2) {
                const float dx = P.ex - kp2.x, dy = P.ey - kp2.y;
                if (dx * dx + dy * dy < 100 * scale2[kp2.octave]) continue;
            }
            const float num = a * kp2.x + b * kp2.y + c;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (!((double)dsqr < 3.84 * (double)sigma2_2[kp2.octave])) continue;
            const unsigned key = ((unsigned)dist << 20) | (0xfffffu - (unsigned)i2);
            best = min(best, key);
        }
    }
    for (int off = 32; off > 0; off >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, off));
    if (lane == 0) matches12[i1] = best == 0xffffffffu ? -1 : (int)(0xfffffu - (best & 0xfffffu));
}

// SearchByBoW (:171-310): distances of every key-frame feature to the frame features of its vocabulary node, one wave per key-frame
// feature, written to that feature's slice of a CSR buffer (the slice length is the node's size); the greedy claim order is the host's.
__global__ void __launch_bounds__(256) match_bow_dists(int NK, const unsigned long long *descK, const int *nodeK, const unsigned long long *descF, const int *node_start,
                                                       const int *node_items, const int *out_off, int *dists) {
    const int ik = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ik >= NK) return;
    const int nd = nodeK[ik];
    if (nd < 0) return;
    const unsigned long long a0 = descK[(long)ik * 4], a1 = descK[(long)ik * 4 + 1], a2 = descK[(long)ik * 4 + 2], a3 = descK[(long)ik * 4 + 3];
    const int b = node_start[nd], n = node_start[nd + 1] - b;
    for (int p = lane; p < n; p += 64) dists[out_off[ik] + p] = hamming256(descF + (long)node_items[b + p] * 4, a0, a1, a2, a3);
}


// ---- a whole window of a stream at once (cs_match_by_projection_stream): pair p = (last frame f0 + p, current frame f0 + p + 1) of the frames an extractor holds in HBM.
// Frame post-processing of every current frame and the searches of all pairs in a handful of launches -- the per-frame calls take five launches, six copies and three
// host round trips per frame, and at 2 000 queries a launch is all latency.  Same arithmetic, same candidate order, same claims as the per-frame calls.
struct QueryS { float x, y, r; int minLevel, maxLevel, valid, pair; };
// AssignFeaturesToGrid of the current frames: one workgroup per frame, key points kfirst[p] .. kfirst[p + 1] of `keys`; cell lists relative to the frame
__global__ void __launch_bounds__(1024) match_grid_batch(FrameP F, const cs_keypoint *keys, const int *kfirst, int *cell_start_all /*P x (NCELL+1)*/, int *cell_items_all, int *kp_cell_all) {
    __shared__ int cnt[NCELL];
    __shared__ int s_part[1024];
    const int tid = threadIdx.x, p = blockIdx.x, kb = kfirst[p], N = kfirst[p + 1] - kb;
    const cs_keypoint *k = keys + kb;
    int *cell_start = cell_start_all + (size_t)p * (NCELL + 1), *cell_items = cell_items_all + kb, *kp_cell = kp_cell_all + kb;
    for (int c = tid; c < NCELL; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) { // PosInGrid, Frame.cc:525-535
        int px = (int)roundf((k[i].x - F.minX) * F.wInv), py = (int)roundf((k[i].y - F.minY) * F.hInv);
        int c = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
        kp_cell[i] = c;
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    int c0 = cnt[tid * 3], c1 = cnt[tid * 3 + 1], c2 = cnt[tid * 3 + 2];
    s_part[tid] = c0 + c1 + c2;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int base = s_part[tid] - (c0 + c1 + c2);
    cell_start[tid * 3] = base; cell_start[tid * 3 + 1] = base + c0; cell_start[tid * 3 + 2] = base + c0 + c1;
    if (tid == 1023) cell_start[NCELL] = s_part[1023];
    __syncthreads();
    cnt[tid * 3] = base; cnt[tid * 3 + 1] = base + c0; cnt[tid * 3 + 2] = base + c0 + c1;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) { int c = kp_cell[i]; if (c >= 0) cell_items[atomicAdd(&cnt[c], 1)] = i; }
    __syncthreads();
    for (int c = tid; c < NCELL; c += 1024) { // mGrid[x][y] holds the indices in ascending key point order
        int b = cell_start[c], e = cnt[c];
        for (int i = b + 1; i < e; i++) {
            int v = cell_items[i], j = i - 1;
            while (j >= b && cell_items[j] > v) { cell_items[j + 1] = cell_items[j]; j--; }
            cell_items[j + 1] = v;
        }
    }
}
// the queries of all pairs: query j belongs to pair p with qfirst[p] <= j < qfirst[p + 1]; its level is the last frame's key point's (ORBmatcher.cc:1424)
__global__ void __launch_bounds__(256) match_project_stream(int nq, int n_pairs, const int *qfirst, const float *world_pos, const uint8_t *valid, const cs_keypoint *last_keys /* raw key points of frame f0 on: query j = last_keys[j] */,
                                                           const float *T_all, float fx, float fy, float cx, float cy, const float *scale_factors, float th, FrameP F, QueryS *q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    int lo = 0, hi = n_pairs; // qfirst[lo] <= i < qfirst[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qfirst[mid] <= i) lo = mid; else hi = mid; }
    const float *T = T_all + 12 * lo;
    QueryS Q{0, 0, 0, 0, 0, 0, lo};
    if (valid[i]) {
        float x3Dc[3];
        for (int r = 0; r < 3; r++) {
            double sacc = 0;
            for (int k = 0; k < 3; k++) sacc += (double)T[r * 4 + k] * (double)world_pos[(size_t)i * 3 + k];
            x3Dc[r] = (float)(sacc * 1.0 + (double)T[r * 4 + 3] * 1.0);
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (!(invzc < 0)) {
            float u = fx * xc * invzc + cx;
            float v = fy * yc * invzc + cy;
            if (!(u < F.minX || u > F.maxX) && !(v < F.minY || v > F.maxY)) {
                int o = last_keys[i].octave;
                Q.x = u; Q.y = v; Q.r = th * scale_factors[o]; Q.minLevel = o - 1; Q.maxLevel = o + 1; Q.valid = 1;
            }
        }
    }
    q[i] = Q;
}
// match_candidates over the queries of every pair: the train frame of pair p is key points kfirst[p] .. of `keys` (undistorted) / `desc`, its cell lists at p x (NCELL + 1) / kfirst[p]
__global__ void __launch_bounds__(256) match_candidates_stream(FrameP F, const cs_keypoint *keys, const unsigned long long *desc, const int *kfirst, const int *cell_start_all, const int *cell_items_all, int nq,
                                                               const QueryS *q, const unsigned long long *qdesc, int pass, int *counts, const long *offsets, int2 *cands) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const QueryS Q = q[qi];
    int total = 0;
    if (Q.valid) {
        const int kb = kfirst[Q.pair];
        const cs_keypoint *tk = keys + kb; const unsigned long long *td = desc + (size_t)kb * 4;
        const int *cell_start = cell_start_all + (size_t)Q.pair * (NCELL + 1), *cell_items = cell_items_all + kb;
        const float x = Q.x, y = Q.y, r = Q.r;
        const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x - F.minX + r) * F.wInv));
        const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y - F.minY + r) * F.hInv));
        if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0)) {
            const bool bCheckLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
            const int ny = nMaxCellY - nMinCellY + 1, ncell = (nMaxCellX - nMinCellX + 1) * ny;
            unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            if (pass == 1) { d0 = qdesc[(size_t)qi * 4]; d1 = qdesc[(size_t)qi * 4 + 1]; d2 = qdesc[(size_t)qi * 4 + 2]; d3 = qdesc[(size_t)qi * 4 + 3]; }
            const long base = pass == 1 ? offsets[qi] : 0;
            for (int k0 = 0; k0 < ncell; k0 += 64) {
                const int k = k0 + lane;
                int b = 0, e = 0;
                if (k < ncell) { int c = (nMinCellX + k / ny) * GRID_ROWS + nMinCellY + k % ny; b = cell_start[c]; e = cell_start[c + 1]; }
                int mine = 0;
                for (int p = b; p < e; p++) {
                    const cs_keypoint kp = tk[cell_items[p]];
                    bool ok = true;
                    if (bCheckLevels) { if (kp.octave < Q.minLevel) ok = false; if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) ok = false; }
                    const float distx = kp.x - x, disty = kp.y - y;
                    ok = ok && fabsf(distx) < r && fabsf(disty) < r;
                    mine += ok;
                }
                int inc = mine;
                for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
                if (pass == 1 && mine) {
                    long o = base + total + inc - mine;
                    for (int p = b; p < e; p++) {
                        const int id = cell_items[p];
                        const cs_keypoint kp = tk[id];
                        bool ok = true;
                        if (bCheckLevels) { if (kp.octave < Q.minLevel) ok = false; if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) ok = false; }
                        const float distx = kp.x - x, disty = kp.y - y;
                        ok = ok && fabsf(distx) < r && fabsf(disty) < r;
                        if (ok) { cands[o] = make_int2(id, hamming256(td + (size_t)id * 4, d0, d1, d2, d3)); o++; }
                    }
                }
                total += __shfl(inc, 63);
            }
        }
    }
    if (pass == 0 && lane == 0) counts[qi] = total;
}
// exclusive scan of n counts into 64-bit offsets (offsets[n] = total): per-block sums, then the blocks' bases, then the fill
__global__ void __launch_bounds__(1024) match_scan_blocks(int n, const int *counts, long *block_sum) {
    __shared__ long s[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    long v = i < n ? counts[i] : 0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { long t = 0; for (int k = 0; k < 16; k++) t += s[k]; block_sum[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(1024) match_scan_fill(int n, int n_blocks, const int *counts, const long *block_sum, long *offsets) {
    __shared__ long s[1024];
    __shared__ long s_base;
    const int tid = threadIdx.x, i = blockIdx.x * 1024 + tid;
    if (tid == 0) { long t = 0; for (int b = 0; b < (int)blockIdx.x; b++) t += block_sum[b]; s_base = t; }
    const long v = i < n ? counts[i] : 0;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { const long t = tid >= off ? s[tid - off] : 0; __syncthreads(); s[tid] += t; __syncthreads(); }
    if (i < n) offsets[i] = s_base + s[tid] - v;
    if (blockIdx.x == (unsigned)n_blocks - 1 && tid == 1023) offsets[n] = s_base + s[1023];
}

static void three_maxima(const int *sizes, int L, int &ind1, int &ind2, int &ind3) { // ORBmatcher.cc:1860-1901
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}
static inline int rot_bin(float a1, float a2) { // :1483-1491
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
} // namespace

struct cs_matcher {
    int max_kp = 0, max_q = 0; long max_cand = 0;
    FrameP F{};
    std::vector<cs_keypoint> keys; // host copy of mvKeysUn (angles / octaves / points for the resolve pass)
    cs_keypoint *d_keys = nullptr; unsigned long long *d_desc = nullptr, *d_qdesc = nullptr;
    int *d_cell_start = nullptr, *d_cell_items = nullptr, *d_kp_cell = nullptr, *d_counts = nullptr, *d_offsets = nullptr;
    Query *d_q = nullptr; int2 *d_cands = nullptr;
    float *d_f = nullptr; uint8_t *d_u8 = nullptr; int *d_i = nullptr; // staging for projection inputs
    std::vector<int> offsets; std::vector<int2> cands;
};

static int run_candidates(cs_ctx *ctx, cs_matcher *m, int nq, bool with_desc) {
    if (nq == 0) { m->offsets.assign(1, 0); m->cands.clear(); return CS_OK; }
    CS_LAUNCH(ctx, "match_candidates", match_candidates, dim3((nq + 3) / 4), dim3(256), 0, m->F, m->d_keys, m->d_desc, m->d_cell_start, m->d_cell_items, nq,
              m->d_q, with_desc ? m->d_qdesc : nullptr, 0, m->d_counts, m->d_offsets, m->d_cands);
    CS_LAUNCH(ctx, "match_scan", match_scan, dim3(1), dim3(1024), 0, nq, m->d_counts, m->d_offsets);
    m->offsets.resize((size_t)nq + 1);
    int r = cs_d2h(ctx, m->offsets.data(), m->d_offsets, (size_t)nq + 1); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long total = m->offsets[nq];
    if (total > m->max_cand) { ctx->err = "matcher candidate capacity exceeded"; return CS_ERR_CAPACITY; }
    CS_LAUNCH(ctx, "match_candidates", match_candidates, dim3((nq + 3) / 4), dim3(256), 0, m->F, m->d_keys, m->d_desc, m->d_cell_start, m->d_cell_items, nq,
              m->d_q, with_desc ? m->d_qdesc : nullptr, 1, m->d_counts, m->d_offsets, m->d_cands);
    m->cands.resize((size_t)std::max<long>(total, 1));
    r = cs_d2h(ctx, m->cands.data(), m->d_cands, (size_t)total); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

struct cs_match_stream {
    size_t cap_k = 0, cap_q = 0, cap_pairs = 0; long cap_c = 0;
    cs_keypoint *d_keys = nullptr; int *d_cell_start = nullptr, *d_cell_items = nullptr, *d_kp_cell = nullptr, *d_kfirst = nullptr, *d_qfirst = nullptr, *d_counts = nullptr;
    long *d_offsets = nullptr, *d_bsum = nullptr; QueryS *d_q = nullptr; int2 *d_cands = nullptr; float *d_wp = nullptr, *d_T = nullptr, *d_sf = nullptr; uint8_t *d_valid = nullptr; unsigned long long *d_qdesc = nullptr;
    std::vector<cs_keypoint> keys; std::vector<long> offsets; std::vector<int2> cands;
    long last_q = 0, last_c = 0;
};
template <class T> static int ms_grow(cs_ctx *ctx, T **p, size_t *cap, size_t need, size_t unit = 1) { // (cap in elements of `unit` T's; a shared cap is raised by its first array: call in groups)
    if (*p && need <= *cap) return CS_OK;
    if (*p) { hipFree(*p); *p = nullptr; }
    const size_t c = need + need / 4 + 64;
    const int r = cs_dalloc(ctx, p, c * unit);
    if (r == CS_OK && cap) *cap = c;
    return r;
}
extern "C" {

void cs_matcher_destroy(cs_ctx *ctx, cs_matcher *m) {
    if (!m) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {m->d_keys, m->d_desc, m->d_qdesc, m->d_cell_start, m->d_cell_items, m->d_kp_cell, m->d_counts, m->d_offsets, m->d_q, m->d_cands,
                    m->d_f, m->d_u8, m->d_i};
    for (void *p : ptrs) if (p) hipFree(p);
    delete m;
}

int cs_matcher_last_counts(const cs_matcher *m, int *queries, long *candidates) { // workload facts of the last search, for roofline accounting
    if (!m || !queries || !candidates) return CS_ERR_BAD_ARG;
    *queries = m->offsets.empty() ? 0 : (int)m->offsets.size() - 1;
    *candidates = m->offsets.empty() ? 0 : (long)m->offsets.back();
    return CS_OK;
}

int cs_matcher_create(cs_ctx *ctx, int max_keypoints, int max_queries, long max_candidates, cs_matcher **out) {
    if (!ctx || !out || max_keypoints < 1 || max_queries < 1 || max_candidates < 1) return CS_ERR_BAD_ARG;
    *out = nullptr;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_matcher *m = new (std::nothrow) cs_matcher();
    if (!m) return CS_ERR_NOMEM;
    m->max_kp = max_keypoints; m->max_q = max_queries; m->max_cand = max_candidates;
#define A_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_matcher_destroy(ctx, m); return r__; } } while (0)
    A_(cs_dalloc(ctx, &m->d_keys, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_desc, (size_t)max_keypoints * 4));
    A_(cs_dalloc(ctx, &m->d_qdesc, (size_t)max_queries * 4));
    A_(cs_dalloc(ctx, &m->d_cell_start, (size_t)NCELL + 1));
    A_(cs_dalloc(ctx, &m->d_cell_items, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_kp_cell, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_counts, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_offsets, (size_t)max_queries + 1));
    A_(cs_dalloc(ctx, &m->d_q, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_cands, (size_t)max_candidates));
    A_(cs_dalloc(ctx, &m->d_f, (size_t)max_queries * 3 + 64));
    A_(cs_dalloc(ctx, &m->d_u8, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_i, (size_t)max_queries));
#undef A_
    *out = m;
    return CS_OK;
}

int cs_matcher_set_frame(cs_ctx *ctx, cs_matcher *m, const cs_keypoint *keysUn, const uint8_t *desc, int N, float minX, float maxX, float minY,
                         float maxY) {
    if (!ctx || !m || !keysUn || !desc || N < 0 || N > m->max_kp || !(maxX > minX) || !(maxY > minY)) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    m->F.N = N; m->F.minX = minX; m->F.maxX = maxX; m->F.minY = minY; m->F.maxY = maxY;
    m->F.wInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX); // Frame.cc:285-286
    m->F.hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);
    m->keys.assign(keysUn, keysUn + N);
    int r = cs_h2d(ctx, m->d_keys, keysUn, (size_t)N); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_desc, desc, (size_t)N * 32); if (r) return r;
    CS_LAUNCH(ctx, "match_grid", match_grid, dim3(1), dim3(1024), 0, m->F, m->d_keys, m->d_cell_start, m->d_cell_items, m->d_kp_cell);
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

static UndP make_undp(const float *K4, const float *dist5) {
    UndP U; U.fx = K4[0]; U.fy = K4[1]; U.cx = K4[2]; U.cy = K4[3];
    for (int i = 0; i < 5; i++) U.k[i] = dist5 ? (double)dist5[i] : 0.0;
    U.identity = !dist5 || dist5[0] == 0.0f; // Frame.cc:548: only the first coefficient is looked at
    return U;
}
int cs_frame_image_bounds(int cols, int rows, const float *K4, const float *dist5, float *bounds) { // Frame::ComputeImageBounds (Frame.cc:578-609)
    if (!K4 || !bounds || cols < 1 || rows < 1) return CS_ERR_BAD_ARG;
    const UndP U = make_undp(K4, dist5);
    if (U.identity) { bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows; return CS_OK; }
    const float cx[4] = {0.0f, (float)cols, 0.0f, (float)cols}, cy[4] = {0.0f, 0.0f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) undistort_point(U, cx[i], cy[i], ux[i], uy[i]);
    bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]); bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    return CS_OK;
}
int cs_matcher_set_frame_from_orb(cs_ctx *ctx, cs_matcher *m, const cs_orb *orb, int frame, const float *K4, const float *dist5, float minX, float maxX, float minY,
                                  float maxY, cs_keypoint *keysUn_out, int *n_out) {
    if (!ctx || !m || !orb || !K4 || !(maxX > minX) || !(maxY > minY)) return CS_ERR_BAD_ARG;
    const cs_keypoint *d_k = nullptr; const unsigned long long *d_d = nullptr; int N = 0;
    int r = cs_orb_device_frame(orb, frame, &d_k, &d_d, &N); if (r) return r;
    if (N > m->max_kp) return CS_ERR_CAPACITY;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    m->F.N = N; m->F.minX = minX; m->F.maxX = maxX; m->F.minY = minY; m->F.maxY = maxY;
    m->F.wInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX); // Frame.cc:285-286
    m->F.hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);
    // UndistortKeyPoints + the descriptors, device to device; AssignFeaturesToGrid on the undistorted keypoints
    if (N > 0) {
        CS_LAUNCH(ctx, "match_undistort", match_undistort, dim3((N + 255) / 256), dim3(256), 0, N, d_k, make_undp(K4, dist5), m->d_keys);
        CS_HIP(ctx, hipMemcpyAsync(m->d_desc, d_d, (size_t)N * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    CS_LAUNCH(ctx, "match_grid", match_grid, dim3(1), dim3(1024), 0, m->F, m->d_keys, m->d_cell_start, m->d_cell_items, m->d_kp_cell);
    m->keys.resize((size_t)N); // the host resolve passes read angle / octave / point of the candidates
    r = cs_d2h(ctx, m->keys.data(), m->d_keys, (size_t)N); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keysUn_out) memcpy(keysUn_out, m->keys.data(), sizeof(cs_keypoint) * (size_t)N);
    if (n_out) *n_out = N;
    return CS_OK;
}

int cs_matcher_features_in_area(cs_ctx *ctx, cs_matcher *m, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap, int *n) {
    if (!ctx || !m || !n) return CS_ERR_BAD_ARG;
    Query Q{x, y, r, minLevel, maxLevel, 1};
    int rr = cs_h2d(ctx, m->d_q, &Q, 1); if (rr) return rr;
    rr = run_candidates(ctx, m, 1, false); if (rr) return rr;
    *n = m->offsets[1];
    if (out) for (int i = 0; i < *n && i < cap; i++) out[i] = m->cands[i].x;
    return CS_OK;
}

int cs_match_by_projection_frame(cs_ctx *ctx, cs_matcher *m, int n_last, const float *world_pos, const uint8_t *valid, const uint8_t *blocks,
                                 const uint8_t *mp_desc, const int *last_octave, const float *last_angle, const float *Tcw, float fx, float fy, float cx,
                                 float cy, const float *scale_factors, int n_levels, float th, int check_orientation, const uint8_t *train_blocked, int *train_match, int *nmatches) {
    if (!ctx || !m || n_last < 0 || n_last > m->max_q || !world_pos || !valid || !blocks || !mp_desc || !last_octave || !last_angle || !Tcw ||
        !scale_factors || n_levels < 1 || n_levels > 32 || !train_match || !nmatches)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n_last; i++) if (valid[i] && (last_octave[i] < 0 || last_octave[i] >= n_levels)) return CS_ERR_BAD_ARG;
    for (int i = 0; i < m->F.N; i++) train_match[i] = -1;
    *nmatches = 0;
    if (n_last == 0) return CS_OK;
    float *d_T = m->d_f + (size_t)m->max_q * 3, *d_sf = d_T + 12;
    int r = cs_h2d(ctx, m->d_f, world_pos, (size_t)n_last * 3); if (r) return r;
    r = cs_h2d(ctx, d_T, Tcw, 12); if (r) return r;
    r = cs_h2d(ctx, d_sf, scale_factors, (size_t)n_levels); if (r) return r;
    r = cs_h2d(ctx, m->d_u8, valid, (size_t)n_last); if (r) return r;
    r = cs_h2d(ctx, m->d_i, last_octave, (size_t)n_last); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_last * 32); if (r) return r;
    CS_LAUNCH(ctx, "match_project", match_project, dim3((n_last + 255) / 256), dim3(256), 0, n_last, m->d_f, m->d_u8, m->d_i, d_T, fx, fy, cx, cy, d_sf, th,
              m->F, m->d_q);
    r = run_candidates(ctx, m, n_last, true); if (r) return r;
    // sequential greedy pass (:1397-1494)
    int nm = 0;
    std::vector<int> rot_items[HISTO_LENGTH];
    for (int i = 0; i < n_last; i++) {
        const int b = m->offsets[i], e = m->offsets[i + 1];
        if (e == b) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int p = b; p < e; p++) {
            const int i2 = m->cands[p].x;
            if (train_match[i2] >= 0 && blocks[train_match[i2]]) continue;
            if (train_blocked && train_blocked[i2]) continue; // map point from before the call / KeysStatic[i2] == false, :1451-1457
            const int dist = m->cands[p].y;
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            train_match[bestIdx2] = i;
            nm++;
            if (check_orientation) rot_items[rot_bin(last_angle[i], m->keys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rot_items[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int id : rot_items[i]) { train_match[id] = -1; nm--; }
    }
    *nmatches = nm;
    return CS_OK;
}


// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1373-1522) for the pairs (f0 + p, f0 + p + 1), p < n_pairs, of the frames `orb` holds in HBM, with
// Frame::UndistortKeyPoints + AssignFeaturesToGrid (Frame.cc:303-318, 546-576) of every current frame: see include/cubeslam_hip.h.
void cs_match_stream_destroy(cs_ctx *ctx, cs_match_stream *m) {
    if (!m) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {m->d_keys, m->d_cell_start, m->d_cell_items, m->d_kp_cell, m->d_kfirst, m->d_qfirst, m->d_counts, m->d_offsets, m->d_bsum, m->d_q, m->d_cands, m->d_wp, m->d_T, m->d_sf, m->d_valid, m->d_qdesc};
    for (void *p : ptrs) if (p) hipFree(p);
    delete m;
}
int cs_match_stream_create(cs_match_stream **out) {
    if (!out) return CS_ERR_BAD_ARG;
    *out = new (std::nothrow) cs_match_stream();
    return *out ? CS_OK : CS_ERR_NOMEM;
}
int cs_match_stream_last_counts(const cs_match_stream *m, long *queries, long *candidates) {
    if (!m || !queries || !candidates) return CS_ERR_BAD_ARG;
    *queries = m->last_q; *candidates = m->last_c;
    return CS_OK;
}
int cs_match_by_projection_stream(cs_ctx *ctx, cs_match_stream *m, const cs_orb *orb, int f0, int n_pairs, const float *K4, const float *dist5, float minX, float maxX, float minY, float maxY,
                                  const float *world_pos, const uint8_t *valid, const uint8_t *blocks, const uint8_t *mp_desc, const float *Tcw, float fx, float fy, float cx, float cy,
                                  const float *scale_factors, int n_levels, float th, int check_orientation, int *train_match, int *nmatches) {
    if (!ctx || !m || !orb || n_pairs < 1 || f0 < 0 || !K4 || !(maxX > minX) || !(maxY > minY) || !world_pos || !valid || !blocks || !Tcw || !scale_factors || n_levels < 1 || n_levels > 32 || !train_match || !nmatches)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    // the frames f0 .. f0 + n_pairs lie one behind the other in the extractor's buffers
    std::vector<int> first((size_t)n_pairs + 2);
    const cs_keypoint *d_k0 = nullptr; const unsigned long long *d_d0 = nullptr;
    { int n0 = 0; int r = cs_orb_device_frame(orb, f0, &d_k0, &d_d0, &n0); if (r) return r; first[0] = 0; first[1] = n0; }
    for (int p = 1; p <= n_pairs; p++) {
        const cs_keypoint *dk; const unsigned long long *dd; int n = 0;
        int r = cs_orb_device_frame(orb, f0 + p, &dk, &dd, &n); if (r) return r;
        if (dk != d_k0 + first[p]) { ctx->err = "cs_match_by_projection_stream: the extractor's frames are not contiguous"; return CS_ERR_BAD_ARG; }
        first[p + 1] = first[p] + n;
    }
    const int nq = first[n_pairs], nk_all = first[n_pairs + 1]; // queries: key points of frames f0 .. f0 + n_pairs - 1; train key points: frames f0 + 1 .. f0 + n_pairs (indices first[1] ..)
    for (int p = 0; p < n_pairs; p++) nmatches[p] = 0;
    const int nk = nk_all - first[1];
    for (int i = 0; i < nk; i++) train_match[i] = -1;
    m->last_q = nq; m->last_c = 0;
    if (nq == 0 || nk == 0) return CS_OK;
    FrameP F; F.N = 0; F.minX = minX; F.maxX = maxX; F.minY = minY; F.maxY = maxY;
    F.wInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX); F.hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);
    int r;
#define G_(call) do { r = (call); if (r != CS_OK) return r; } while (0)
    { size_t c = m->cap_k; G_(ms_grow(ctx, &m->d_keys, &c, (size_t)nk_all)); c = m->cap_k; G_(ms_grow(ctx, &m->d_cell_items, &c, (size_t)nk_all)); G_(ms_grow(ctx, &m->d_kp_cell, &m->cap_k, (size_t)nk_all)); }
    { size_t c = m->cap_pairs; G_(ms_grow(ctx, &m->d_cell_start, &c, (size_t)n_pairs, (size_t)NCELL + 1)); c = m->cap_pairs; G_(ms_grow(ctx, &m->d_kfirst, &c, (size_t)n_pairs + 2)); c = m->cap_pairs; G_(ms_grow(ctx, &m->d_qfirst, &c, (size_t)n_pairs + 2));
      G_(ms_grow(ctx, &m->d_T, &m->cap_pairs, (size_t)n_pairs, 12)); }
    { size_t c = m->cap_q; G_(ms_grow(ctx, &m->d_counts, &c, (size_t)nq)); c = m->cap_q; G_(ms_grow(ctx, &m->d_offsets, &c, (size_t)nq + 1)); c = m->cap_q; G_(ms_grow(ctx, &m->d_bsum, &c, (size_t)nq / 1024 + 2)); c = m->cap_q; G_(ms_grow(ctx, &m->d_q, &c, (size_t)nq));
      c = m->cap_q; G_(ms_grow(ctx, &m->d_wp, &c, (size_t)nq, 3)); c = m->cap_q; G_(ms_grow(ctx, &m->d_valid, &c, (size_t)nq)); G_(ms_grow(ctx, &m->d_qdesc, &m->cap_q, (size_t)nq, 4)); }
    if (!m->d_sf) G_(cs_dalloc(ctx, &m->d_sf, (size_t)32));
    // UndistortKeyPoints of every frame of the window (device to device), AssignFeaturesToGrid of the current frames
    CS_LAUNCH(ctx, "match_undistort", match_undistort, dim3((nk_all + 255) / 256), dim3(256), 0, nk_all, d_k0, make_undp(K4, dist5), m->d_keys);
    std::vector<int> kfirst((size_t)n_pairs + 1);
    for (int p = 0; p <= n_pairs; p++) kfirst[p] = first[p + 1]; // train frame of pair p: key points kfirst[p] .. kfirst[p + 1] of the window's list
    G_(cs_h2d(ctx, m->d_kfirst, kfirst.data(), (size_t)n_pairs + 1));
    G_(cs_h2d(ctx, m->d_qfirst, first.data(), (size_t)n_pairs + 1));
    G_(cs_h2d(ctx, m->d_T, Tcw, (size_t)n_pairs * 12));
    G_(cs_h2d(ctx, m->d_sf, scale_factors, (size_t)n_levels));
    G_(cs_h2d(ctx, m->d_wp, world_pos, (size_t)nq * 3));
    G_(cs_h2d(ctx, m->d_valid, valid, (size_t)nq));
    if (mp_desc) G_(cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)nq * 32));
    const unsigned long long *d_qdesc = mp_desc ? m->d_qdesc : d_d0; // NULL: a last frame's key point is matched with its own descriptor
    CS_LAUNCH(ctx, "match_grid", match_grid_batch, dim3(n_pairs), dim3(1024), 0, F, m->d_keys, m->d_kfirst, m->d_cell_start, m->d_cell_items, m->d_kp_cell);
    CS_LAUNCH(ctx, "match_project", match_project_stream, dim3((nq + 255) / 256), dim3(256), 0, nq, n_pairs, m->d_qfirst, m->d_wp, m->d_valid, d_k0, m->d_T, fx, fy, cx, cy, m->d_sf, th, F, m->d_q);
    CS_LAUNCH(ctx, "match_candidates", match_candidates_stream, dim3((nq + 3) / 4), dim3(256), 0, F, m->d_keys, d_d0, m->d_kfirst, m->d_cell_start, m->d_cell_items, nq, m->d_q, d_qdesc, 0, m->d_counts, m->d_offsets, m->d_cands);
    const int nblk = (nq + 1023) / 1024;
    CS_LAUNCH(ctx, "match_scan", match_scan_blocks, dim3(nblk), dim3(1024), 0, nq, m->d_counts, m->d_bsum);
    CS_LAUNCH(ctx, "match_scan", match_scan_fill, dim3(nblk), dim3(1024), 0, nq, nblk, m->d_counts, m->d_bsum, m->d_offsets);
    m->offsets.resize((size_t)nq + 1);
    G_(cs_d2h(ctx, m->offsets.data(), m->d_offsets, (size_t)nq + 1));
    m->keys.resize((size_t)nk_all); // (angles / levels of the candidates for the host pass; also what a caller reads as mvKeysUn)
    G_(cs_d2h(ctx, m->keys.data(), m->d_keys, (size_t)nk_all));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nq; i++) if (valid[i] && (m->keys[i].octave < 0 || m->keys[i].octave >= n_levels)) return CS_ERR_BAD_ARG;
    const long total = m->offsets[nq];
    m->last_c = total;
    if (total > m->cap_c || !m->d_cands) { if (m->d_cands) hipFree(m->d_cands); m->d_cands = nullptr; m->cap_c = total + total / 4 + 1024; G_(cs_dalloc(ctx, &m->d_cands, (size_t)m->cap_c)); }
    CS_LAUNCH(ctx, "match_candidates", match_candidates_stream, dim3((nq + 3) / 4), dim3(256), 0, F, m->d_keys, d_d0, m->d_kfirst, m->d_cell_start, m->d_cell_items, nq, m->d_q, d_qdesc, 1, m->d_counts, m->d_offsets, m->d_cands);
    m->cands.resize((size_t)std::max<long>(total, 1));
    G_(cs_d2h(ctx, m->cands.data(), m->d_cands, (size_t)total));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
#undef G_
    // the sequential greedy pass of every pair (:1397-1494), the pairs side by side
#pragma omp parallel for schedule(dynamic, 4) num_threads(std::max(1, std::min(ctx->host_threads, n_pairs)))
    for (int p = 0; p < n_pairs; p++) {
        int *tm = train_match + (first[p + 1] - first[1]);
        const cs_keypoint *tkeys = m->keys.data() + first[p + 1];
        const int q0 = first[p], q1 = first[p + 1];
        int nm = 0;
        std::vector<int> rot_items[HISTO_LENGTH];
        for (int i = q0; i < q1; i++) {
            const long b = m->offsets[i], e = m->offsets[i + 1];
            if (e == b) continue;
            int bestDist = 256, bestIdx2 = -1;
            for (long c = b; c < e; c++) {
                const int i2 = m->cands[c].x;
                if (tm[i2] >= 0 && blocks[q0 + tm[i2]]) continue;
                const int dist = m->cands[c].y;
                if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
            }
            if (bestDist <= TH_HIGH) {
                tm[bestIdx2] = i - q0;
                nm++;
                if (check_orientation) rot_items[rot_bin(m->keys[i].angle, tkeys[bestIdx2].angle)].push_back(bestIdx2);
            }
        }
        if (check_orientation) {
            int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rot_items[i].size();
            three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++)
                if (i != ind1 && i != ind2 && i != ind3)
                    for (int id : rot_items[i]) { tm[id] = -1; nm--; }
        }
        nmatches[p] = nm;
    }
    return CS_OK;
}

int cs_match_local_map(cs_ctx *ctx, cs_matcher *m, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level, const uint8_t *in_view,
                       const uint8_t *blocks, const uint8_t *mp_desc, const float *scale_factors, int n_levels, float th, float nnratio,
                       const uint8_t *train_blocked, int *train_match, int *nmatches) {
    if (!ctx || !m || n_mp < 0 || n_mp > m->max_q || !proj_xy || !view_cos || !pred_level || !in_view || !blocks || !mp_desc || !scale_factors ||
        !train_match || !nmatches)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < m->F.N; i++) train_match[i] = -1;
    *nmatches = 0;
    if (n_mp == 0) return CS_OK;
    const bool bFactor = th != 1.0;
    std::vector<Query> q((size_t)n_mp);
    for (int i = 0; i < n_mp; i++) { // :58-78 (window radius from the viewing cosine)
        Query Q{0, 0, 0, 0, 0, 0};
        if (in_view[i]) {
            if (pred_level[i] < 0 || pred_level[i] >= n_levels) return CS_ERR_BAD_ARG;
            float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            Q.x = proj_xy[i * 2]; Q.y = proj_xy[i * 2 + 1]; Q.r = r * scale_factors[pred_level[i]];
            Q.minLevel = pred_level[i] - 1; Q.maxLevel = pred_level[i]; Q.valid = 1;
        }
        q[i] = Q;
    }
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_mp * 32); if (r) return r;
    r = run_candidates(ctx, m, n_mp, true); if (r) return r;
    int nm = 0;
    for (int i = 0; i < n_mp; i++) { // :86-140
        const int b = m->offsets[i], e = m->offsets[i + 1];
        if (e == b) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int p = b; p < e; p++) {
            const int idx = m->cands[p].x;
            if (train_blocked && train_blocked[idx]) continue;
            if (train_match[idx] >= 0 && blocks[train_match[idx]]) continue;
            const int dist = m->cands[p].y;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = m->keys[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = m->keys[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            train_match[bestIdx] = i;
            nm++;
        }
    }
    *nmatches = nm;
    return CS_OK;
}

int cs_match_for_initialization(cs_ctx *ctx, cs_matcher *m, const cs_keypoint *keys1, const uint8_t *desc1, int N1, float *prev, int window_size,
                                float nnratio, int check_orientation, int *vnMatches12, int *nmatches) {
    if (!ctx || !m || !keys1 || !desc1 || N1 < 0 || N1 > m->max_q || !prev || !vnMatches12 || !nmatches) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < N1; i++) vnMatches12[i] = -1;
    *nmatches = 0;
    if (N1 == 0) return CS_OK;
    std::vector<Query> q((size_t)N1);
    for (int i = 0; i < N1; i++) { // :441-449: level-0 keypoints only, window around vbPrevMatched
        Query Q{0, 0, 0, 0, 0, 0};
        if (!(keys1[i].octave > 0)) { Q.x = prev[i * 2]; Q.y = prev[i * 2 + 1]; Q.r = (float)window_size; Q.minLevel = keys1[i].octave; Q.maxLevel = keys1[i].octave; Q.valid = 1; }
        q[i] = Q;
    }
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)N1); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, desc1, (size_t)N1 * 32); if (r) return r;
    r = run_candidates(ctx, m, N1, true); if (r) return r;
    const int N2 = m->F.N;
    int nm = 0;
    std::vector<int> vMatchedDistance((size_t)N2, INT_MAX), vnMatches21((size_t)N2, -1);
    std::vector<int> rot_items[HISTO_LENGTH];
    for (int i1 = 0; i1 < N1; i1++) { // :451-504
        const int b = m->offsets[i1], e = m->offsets[i1 + 1];
        if (e == b) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int p = b; p < e; p++) {
            const int i2 = m->cands[p].x, dist = m->cands[p].y;
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
            if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nm--; }
            vnMatches12[i1] = bestIdx2;
            vnMatches21[bestIdx2] = i1;
            vMatchedDistance[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) rot_items[rot_bin(keys1[i1].angle, m->keys[bestIdx2].angle)].push_back(i1);
        }
    }
    if (check_orientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rot_items[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rot_items[i]) if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nm--; }
        }
    }
    for (int i1 = 0; i1 < N1; i1++)
        if (vnMatches12[i1] >= 0) { prev[i1 * 2] = m->keys[vnMatches12[i1]].x; prev[i1 * 2 + 1] = m->keys[vnMatches12[i1]].y; }
    *nmatches = nm;
    return CS_OK;
}

int cs_match_fuse(cs_ctx *ctx, cs_matcher *m, const float *u_right, const float *inv_level_sigma2, int n_levels, const uint8_t *keys_static, int n_mp, const float *uv,
                  const float *ur, const int *pred_level, const uint8_t *valid, const uint8_t *mp_desc, const float *scale_factors, float th, int *best_idx,
                  int *best_dist, int *n_fused) {
    if (!ctx || !m || n_mp < 0 || n_mp > m->max_q || !u_right || !inv_level_sigma2 || n_levels < 1 || (n_mp && (!uv || !ur || !pred_level || !valid || !mp_desc)) ||
        !scale_factors || !best_idx || !best_dist || !n_fused)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    *n_fused = 0;
    if (n_mp == 0) return CS_OK;
    std::vector<Query> q((size_t)n_mp);
    for (int i = 0; i < n_mp; i++) { // :921-926: GetFeaturesInArea(u, v, th * scale[level]) without level limits
        Query Q{0, 0, 0, -1, -1, 0};
        if (valid[i]) {
            if (pred_level[i] < 0 || pred_level[i] >= n_levels) return CS_ERR_BAD_ARG;
            Q.x = uv[i * 2]; Q.y = uv[i * 2 + 1]; Q.r = th * scale_factors[pred_level[i]]; Q.valid = 1;
        }
        q[i] = Q;
    }
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_mp * 32); if (r) return r;
    r = run_candidates(ctx, m, n_mp, true); if (r) return r;
    int nf = 0;
    for (int i = 0; i < n_mp; i++) { // :934-981: the tests that need per-keypoint data of the key frame, first minimum wins
        int bestDist = 256, bestIdx = -1;
        if (valid[i]) {
            const float u = uv[i * 2], v = uv[i * 2 + 1];
            for (int p = m->offsets[i]; p < m->offsets[i + 1]; p++) {
                const int idx = m->cands[p].x;
                const cs_keypoint &kp = m->keys[idx];
                const int kpLevel = kp.octave;
                if (kpLevel < pred_level[i] - 1 || kpLevel > pred_level[i]) continue;
                if (keys_static && !keys_static[idx]) continue;
                if (kpLevel < 0 || kpLevel >= n_levels) return CS_ERR_BAD_ARG;
                if (u_right[idx] >= 0) {
                    const float ex = u - kp.x, ey = v - kp.y, er = ur[i] - u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp.x, ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
                const int dist = m->cands[p].y;
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
        if (bestDist <= TH_LOW) nf++;
    }
    *n_fused = nf;
    return CS_OK;
}

int cs_match_for_triangulation(cs_ctx *ctx, const cs_keypoint *keys1Un, const uint8_t *desc1, int N1, const int *node1, const uint8_t *skip1, const float *u_right1,
                               const cs_keypoint *keys2Un, const uint8_t *desc2, int N2, const int *node2, const uint8_t *skip2, const float *u_right2,
                               const float *F12, float ex, float ey, const float *scale_factors2, const float *level_sigma2_2, int n_levels, int only_stereo,
                               int check_orientation, int *matches12, int *nmatches) {
    if (!ctx || N1 < 0 || N2 < 0 || N2 >= (1 << 20) || !matches12 || !nmatches || !F12 || !scale_factors2 || !level_sigma2_2 || n_levels < 1 ||
        (N1 && (!keys1Un || !desc1 || !node1 || !skip1 || !u_right1)) || (N2 && (!keys2Un || !desc2 || !node2 || !skip2 || !u_right2)))
        return CS_ERR_BAD_ARG;
    *nmatches = 0;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    if (N1 == 0 || N2 == 0) return CS_OK;
    for (int i = 0; i < N2; i++) if (keys2Un[i].octave < 0 || keys2Un[i].octave >= n_levels) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    // KF2 features by node (counting sort keeps the ascending index order inside a node); node ids are compacted to 0..n_nodes-1
    std::vector<int> ids;
    for (int i = 0; i < N2; i++) if (node2[i] >= 0) ids.push_back(node2[i]);
    std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int n_nodes = (int)ids.size();
    auto compact = [&](int nd) { if (nd < 0) return -1; auto it = std::lower_bound(ids.begin(), ids.end(), nd); return (it != ids.end() && *it == nd) ? (int)(it - ids.begin()) : -1; };
    std::vector<int> start((size_t)n_nodes + 1, 0), items, n1c((size_t)N1);
    std::vector<int> n2c((size_t)N2);
    for (int i = 0; i < N2; i++) { n2c[i] = compact(node2[i]); if (n2c[i] >= 0) start[n2c[i] + 1]++; }
    for (int k = 0; k < n_nodes; k++) start[k + 1] += start[k];
    items.resize((size_t)std::max(start[n_nodes], 1));
    { std::vector<int> pos(start.begin(), start.end() - 1); for (int i = 0; i < N2; i++) if (n2c[i] >= 0) items[pos[n2c[i]]++] = i; }
    for (int i = 0; i < N1; i++) n1c[i] = compact(node1[i]);
    cs_keypoint *d_k1 = nullptr, *d_k2 = nullptr; unsigned long long *d_d1 = nullptr, *d_d2 = nullptr; int *d_n1 = nullptr, *d_st = nullptr, *d_it = nullptr, *d_m = nullptr;
    uint8_t *d_s1 = nullptr, *d_s2 = nullptr; float *d_u1 = nullptr, *d_u2 = nullptr, *d_sc = nullptr, *d_sg = nullptr;
    int r = cs_dalloc(ctx, &d_k1, (size_t)N1);
#define TA_(call) if (!r) r = (call)
    TA_(cs_dalloc(ctx, &d_k2, (size_t)N2)); TA_(cs_dalloc(ctx, &d_d1, (size_t)N1 * 4)); TA_(cs_dalloc(ctx, &d_d2, (size_t)N2 * 4));
    TA_(cs_dalloc(ctx, &d_n1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_st, (size_t)n_nodes + 1)); TA_(cs_dalloc(ctx, &d_it, items.size())); TA_(cs_dalloc(ctx, &d_m, (size_t)N1));
    TA_(cs_dalloc(ctx, &d_s1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_s2, (size_t)N2)); TA_(cs_dalloc(ctx, &d_u1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_u2, (size_t)N2));
    TA_(cs_dalloc(ctx, &d_sc, (size_t)n_levels)); TA_(cs_dalloc(ctx, &d_sg, (size_t)n_levels));
    TA_(cs_h2d(ctx, d_k1, keys1Un, (size_t)N1)); TA_(cs_h2d(ctx, d_k2, keys2Un, (size_t)N2));
    TA_(cs_h2d(ctx, (uint8_t *)d_d1, desc1, (size_t)N1 * 32)); TA_(cs_h2d(ctx, (uint8_t *)d_d2, desc2, (size_t)N2 * 32));
    TA_(cs_h2d(ctx, d_n1, n1c.data(), (size_t)N1)); TA_(cs_h2d(ctx, d_st, start.data(), start.size())); TA_(cs_h2d(ctx, d_it, items.data(), items.size()));
    TA_(cs_h2d(ctx, d_s1, skip1, (size_t)N1)); TA_(cs_h2d(ctx, d_s2, skip2, (size_t)N2)); TA_(cs_h2d(ctx, d_u1, u_right1, (size_t)N1)); TA_(cs_h2d(ctx, d_u2, u_right2, (size_t)N2));
    TA_(cs_h2d(ctx, d_sc, scale_factors2, (size_t)n_levels)); TA_(cs_h2d(ctx, d_sg, level_sigma2_2, (size_t)n_levels));
#undef TA_
    if (!r) {
        TriP P; for (int k = 0; k < 9; k++) P.F12[k] = F12[k];
        P.ex = ex; P.ey = ey; P.only_stereo = only_stereo;
        CS_LAUNCH(ctx, "match_triangulation", match_triangulation, dim3((N1 + 3) / 4), dim3(256), 0, N1, d_k1, d_d1, d_n1, d_s1, d_u1, d_k2, d_d2, d_s2, d_u2, d_st, d_it, n_nodes, P,
                  d_sc, d_sg, d_m);
        r = cs_d2h(ctx, matches12, d_m, (size_t)N1);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_k1, d_k2, d_d1, d_d2, d_n1, d_st, d_it, d_m, d_s1, d_s2, d_u1, d_u2, d_sc, d_sg};
    for (void *p : ptrs) if (p) hipFree(p);
    if (r) return r;
    int nm = 0;
    for (int i = 0; i < N1; i++) if (matches12[i] >= 0) nm++;
    if (check_orientation) { // :801-830
        std::vector<int> rotHist[HISTO_LENGTH];
        for (int i = 0; i < N1; i++) if (matches12[i] >= 0) rotHist[rot_bin(keys1Un[i].angle, keys2Un[matches12[i]].angle)].push_back(i);
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return CS_OK;
}

// Shared by both SearchByBoW overloads: distances of every non-skipped K feature to the F features of its vocabulary node.
struct BowDists {
    std::vector<int> start, items, nkc, off, dists; // node CSR over F, compact node of every K feature (-1 none), slice offsets, distances
    std::vector<std::pair<int, int>> order;         // (original node id, K index): the reference's visiting order
};
static int bow_node_dists(cs_ctx *ctx, const uint8_t *descK, int NK, const int *nodeK, const uint8_t *skipK, const uint8_t *descF, int NF, const int *nodeF, BowDists &B) {
    CS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> ids;
    for (int i = 0; i < NF; i++) if (nodeF[i] >= 0) ids.push_back(nodeF[i]);
    std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int n_nodes = (int)ids.size();
    auto compact = [&](int nd) { if (nd < 0) return -1; auto it = std::lower_bound(ids.begin(), ids.end(), nd); return (it != ids.end() && *it == nd) ? (int)(it - ids.begin()) : -1; };
    std::vector<int> &start = B.start, &items = B.items, &nkc = B.nkc, &off = B.off, &dists = B.dists;
    std::vector<int> nfc((size_t)NF);
    start.assign((size_t)n_nodes + 1, 0); nkc.assign((size_t)NK, -1); off.assign((size_t)NK + 1, 0);
    for (int i = 0; i < NF; i++) { nfc[i] = compact(nodeF[i]); if (nfc[i] >= 0) start[nfc[i] + 1]++; }
    for (int k = 0; k < n_nodes; k++) start[k + 1] += start[k];
    items.resize((size_t)std::max(start[n_nodes], 1));
    { std::vector<int> pos(start.begin(), start.end() - 1); for (int i = 0; i < NF; i++) if (nfc[i] >= 0) items[pos[nfc[i]]++] = i; }
    for (int i = 0; i < NK; i++) { nkc[i] = skipK[i] ? -1 : compact(nodeK[i]); off[i + 1] = off[i] + (nkc[i] >= 0 ? start[nkc[i] + 1] - start[nkc[i]] : 0); }
    const int total = off[NK];
    dists.resize((size_t)std::max(total, 1));
    if (total > 0) {
        unsigned long long *d_dk = nullptr, *d_df = nullptr; int *d_nk = nullptr, *d_st = nullptr, *d_it = nullptr, *d_off = nullptr, *d_di = nullptr;
        int r = cs_dalloc(ctx, &d_dk, (size_t)NK * 4);
#define BA_(call) if (!r) r = (call)
        BA_(cs_dalloc(ctx, &d_df, (size_t)NF * 4)); BA_(cs_dalloc(ctx, &d_nk, (size_t)NK)); BA_(cs_dalloc(ctx, &d_st, start.size())); BA_(cs_dalloc(ctx, &d_it, items.size()));
        BA_(cs_dalloc(ctx, &d_off, off.size())); BA_(cs_dalloc(ctx, &d_di, (size_t)total));
        BA_(cs_h2d(ctx, (uint8_t *)d_dk, descK, (size_t)NK * 32)); BA_(cs_h2d(ctx, (uint8_t *)d_df, descF, (size_t)NF * 32)); BA_(cs_h2d(ctx, d_nk, nkc.data(), (size_t)NK));
        BA_(cs_h2d(ctx, d_st, start.data(), start.size())); BA_(cs_h2d(ctx, d_it, items.data(), items.size())); BA_(cs_h2d(ctx, d_off, off.data(), off.size()));
#undef BA_
        if (!r) {
            CS_LAUNCH(ctx, "match_bow_dists", match_bow_dists, dim3((NK + 3) / 4), dim3(256), 0, NK, d_dk, d_nk, d_df, d_st, d_it, d_off, d_di);
            r = cs_d2h(ctx, dists.data(), d_di, (size_t)total);
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
        void *ptrs[] = {d_dk, d_df, d_nk, d_st, d_it, d_off, d_di};
        for (void *p : ptrs) if (p) hipFree(p);
        if (r) return r;
    }
    // the reference's order: nodes ascending (std::map), K features ascending inside a node
    for (int i = 0; i < NK; i++) if (nkc[i] >= 0) B.order.push_back(std::make_pair(nodeK[i], i));
    std::sort(B.order.begin(), B.order.end());
    return CS_OK;
}

int cs_match_by_bow(cs_ctx *ctx, const cs_keypoint *keysKF, const uint8_t *descKF, int NK, const int *nodeKF, const uint8_t *skipKF, const cs_keypoint *keysF,
                    const uint8_t *descF, int NF, const int *nodeF, const uint8_t *skipF, float nnratio, int check_orientation, int *matchesF, int *nmatches) {
    if (!ctx || NK < 0 || NF < 0 || !matchesF || !nmatches || (NK && (!keysKF || !descKF || !nodeKF || !skipKF)) || (NF && (!keysF || !descF || !nodeF))) return CS_ERR_BAD_ARG;
    *nmatches = 0;
    for (int i = 0; i < NF; i++) matchesF[i] = -1;
    if (NK == 0 || NF == 0) return CS_OK;
    BowDists B;
    if (int r = bow_node_dists(ctx, descKF, NK, nodeKF, skipKF, descF, NF, nodeF, B)) return r;
    const std::vector<int> &start = B.start, &items = B.items, &nkc = B.nkc, &off = B.off, &dists = B.dists;
    const std::vector<std::pair<int, int>> &order = B.order;
    // greedy resolve in the reference's order
    int nm = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (const auto &ok : order) {
        const int ik = ok.second, nd = nkc[ik], b = start[nd], n = start[nd + 1] - b;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int p = 0; p < n; p++) {
            const int iF = items[b + p];
            if (matchesF[iF] >= 0) continue;
            if (skipF && skipF[iF]) continue;
            const int dist = dists[off[ik] + p];
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = iF; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matchesF[bestIdxF] = ik;
            if (check_orientation) rotHist[rot_bin(keysKF[ik].angle, keysF[bestIdxF].angle)].push_back(bestIdxF);
            nm++;
        }
    }
    if (check_orientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matchesF[j] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return CS_OK;
}

int cs_match_by_bow_kf(cs_ctx *ctx, const cs_keypoint *keys1, const uint8_t *desc1, int N1, const int *node1, const uint8_t *skip1, const cs_keypoint *keys2,
                       const uint8_t *desc2, int N2, const int *node2, const uint8_t *skip2, float nnratio, int check_orientation, int *matches12, int *nmatches) {
    if (!ctx || N1 < 0 || N2 < 0 || !nmatches || (N1 && (!keys1 || !desc1 || !node1 || !skip1 || !matches12)) || (N2 && (!keys2 || !desc2 || !node2 || !skip2))) return CS_ERR_BAD_ARG;
    *nmatches = 0;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    if (N1 == 0 || N2 == 0) return CS_OK;
    BowDists B;
    if (int r = bow_node_dists(ctx, desc1, N1, node1, skip1, desc2, N2, node2, B)) return r;
    std::vector<uint8_t> matched2((size_t)N2, 0); // vbMatched2 (:556)
    std::vector<int> rotHist[HISTO_LENGTH];
    int nm = 0;
    for (const auto &ok : B.order) {
        const int i1 = ok.second, nd = B.nkc[i1], b = B.start[nd], n = B.start[nd + 1] - b;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int p = 0; p < n; p++) {
            const int i2 = B.items[b + p];
            if (matched2[i2] || skip2[i2]) continue; // :598-602
            const int dist = B.dists[B.off[i1] + p];
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) { // strict, unlike :268
            matches12[i1] = bestIdx2;
            matched2[bestIdx2] = 1;
            if (check_orientation) rotHist[rot_bin(keys1[i1].angle, keys2[bestIdx2].angle)].push_back(i1);
            nm++;
        }
    }
    if (check_orientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nm--; }
        }
    }
    *nmatches = nm;
    return CS_OK;
}

int cs_hamming_knn2(cs_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx, int *best_dist, int *second_dist) {
    if (!ctx || !q || !t || nq < 0 || nt < 0 || !best_idx || !best_dist || !second_dist) return CS_ERR_BAD_ARG;
    if (nq == 0) return CS_OK;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *dq = nullptr, *dt = nullptr; int *dres = nullptr;
    int r = cs_dalloc(ctx, &dq, (size_t)nq * 4); if (r) return r;
    r = cs_dalloc(ctx, &dt, (size_t)std::max(nt, 1) * 4); if (r) { hipFree(dq); return r; }
    r = cs_dalloc(ctx, &dres, (size_t)nq * 3); if (r) { hipFree(dq); hipFree(dt); return r; }
    r = cs_h2d(ctx, (uint8_t *)dq, q, (size_t)nq * 32);
    if (!r) r = cs_h2d(ctx, (uint8_t *)dt, t, (size_t)nt * 32);
    if (!r) {
        CS_LAUNCH(ctx, "match_knn2", match_knn2, dim3((nq + 255) / 256), dim3(256), 0, dq, nq, dt, nt, dres, dres + nq, dres + 2 * nq);
        r = cs_d2h(ctx, best_idx, dres, (size_t)nq);
        if (!r) r = cs_d2h(ctx, best_dist, dres + nq, (size_t)nq);
        if (!r) r = cs_d2h(ctx, second_dist, dres + 2 * nq, (size_t)nq);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    }
    hipFree(dq); hipFree(dt); hipFree(dres);
    return r;
}

} // extern "C"
