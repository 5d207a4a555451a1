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
// What a window search reads of a key point of the searched frame, laid out in CELL order (the entries of a cell -- and of the cells of a grid column -- lie one behind the
// other): a window's walk streams 16-byte records instead of chasing cell list -> index -> 28-byte key point record
struct CellRec { float x, y; int id, octave; };

__global__ void __launch_bounds__(1024) match_grid(FrameP F, const cs_keypoint *keys, int *cell_start /*NCELL+1*/, int *cell_items, int *kp_cell, CellRec *cell_recs) {
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
    __syncthreads();
    for (int i = tid; i < cell_start[NCELL]; i += 1024) { const int id = cell_items[i]; const cs_keypoint k = keys[id]; cell_recs[i] = CellRec{k.x, k.y, id, k.octave}; }
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

// A window query.  `pair`: which train frame of a window it searches (0 for the one-frame calls).
struct QueryS { float x, y, r; int minLevel, maxLevel, valid, pair; };

__global__ void match_project(int n, const float *world_pos, const uint8_t *valid, const int *octave, const float *T, float fx, float fy, float cx,
                              float cy, const float *scale_factors, float th, FrameP F, QueryS *q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    QueryS Q{0, 0, 0, 0, 0, 0, 0};
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

// GetFeaturesInArea (Frame.cc:404-459) for every query + the candidates' distances, in ONE pass: a wave walks its query's window twice -- it counts, takes a slice of the
// candidate arena from a global cursor, and fills the slice in the reference's candidate order (ix outer, iy inner, cell lists ascending).  Which slice a query gets depends
// on the atomics' order and on nothing else: cstart / ccount address it, and nothing reads the arena in another way.  A slice that would end past `cap` is not written and
// its count reads 0; the cursor still advances, so the host sees what the arena should have held and calls again.  The train frame of query q is key points
// kfirst[q.pair] .. of `keys` / `desc`, its cell lists at q.pair x (NCELL + 1) / kfirst[q.pair].  A candidate is (index | level << 24, distance).
// MC_G lanes per query, MC_Q queries per 256-thread workgroup, ONE cursor update per workgroup.  A window at th 15 covers 8 - 80 cells: with a whole wave per query most lanes had
// no cell, and the kernel is bound by the number of waves that wait out its seven dependent round trips (query -> frame -> cell bounds -> records, the cursor, records ->
// descriptors), not by bytes: 3.1 ms per window of 10^6 queries with 64 lanes per query whether the walk read 28-byte key point records through an index (654 MB) or
// cell-ordered 16-byte records (521 MB).  (Sixteen-wave workgroups were measured too: the same alone, 13 ms beside the region walks -- they wait for a CU with four free
// slots per SIMD.)
constexpr int MC_G = 16, MC_Q = 256 / MC_G;
__global__ void __launch_bounds__(256) match_candidates(FrameP F, const unsigned long long *desc, const int *kfirst, const int *cell_start_all, const CellRec *cell_recs_all, int nq,
                                                        const QueryS *q, const unsigned long long *qdesc, unsigned long long *cursor, long cap, long *cstart, int *ccount, int2 *cands) {
    __shared__ int s_tot[MC_Q];
    __shared__ unsigned long long s_base;
    const int grp = threadIdx.x / MC_G, qi = blockIdx.x * MC_Q + grp, lane = threadIdx.x % MC_G;
    QueryS Q{0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) Q = q[qi];
    int total = 0;
    long base = 0;
    // (the window and the walk are the same for the counting and the filling half; the workgroup's groups meet in between to share one cursor update)
    const int kb = (Q.valid && kfirst) ? kfirst[Q.pair] : 0;
    const unsigned long long *td = desc + (size_t)kb * 4;
    const int *cell_start = cell_start_all + (size_t)Q.pair * (NCELL + 1); const CellRec *recs = cell_recs_all + kb;
    const float x = Q.x, y = Q.y, r = Q.r;
    const int nMinCellX = max(0, (int)floorf((x - F.minX - r) * F.wInv));
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x - F.minX + r) * F.wInv));
    const int nMinCellY = max(0, (int)floorf((y - F.minY - r) * F.hInv));
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y - F.minY + r) * F.hInv));
    const bool window = Q.valid && !(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0);
    const bool bCheckLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
    const int ny = nMaxCellY - nMinCellY + 1, ncell = window ? (nMaxCellX - nMinCellX + 1) * ny : 0;
    unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (window && qdesc) { d0 = qdesc[(size_t)qi * 4]; d1 = qdesc[(size_t)qi * 4 + 1]; d2 = qdesc[(size_t)qi * 4 + 2]; d3 = qdesc[(size_t)qi * 4 + 3]; }
    auto passes = [&](const CellRec &kp) {
        bool ok = true;
        if (bCheckLevels) { if (kp.octave < Q.minLevel) ok = false; if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) ok = false; }
        const float distx = kp.x - x, disty = kp.y - y;
        return ok && fabsf(distx) < r && fabsf(disty) < r;
    };
    // (the groups of a wave walk in lockstep: the trip count is the longest window of the four; a group past its window does nothing)
    int ncell_w = ncell;
    for (int off = MC_G; off < 64; off <<= 1) ncell_w = max(ncell_w, __shfl_xor(ncell_w, off));
    auto walk = [&](bool fill) {
        int tot = 0;
        for (int k0 = 0; k0 < ncell_w; k0 += MC_G) {
            const int k = k0 + lane;
            int b = 0, e = 0;
            if (k < ncell) { int c = (nMinCellX + k / ny) * GRID_ROWS + nMinCellY + k % ny; b = cell_start[c]; e = cell_start[c + 1]; }
            int mine = 0;
            for (int p = b; p < e; p++) mine += passes(recs[p]);
            int inc = mine;
            for (int off = 1; off < MC_G; off <<= 1) { int t = __shfl_up(inc, off, MC_G); if (lane >= off) inc += t; }
            if (fill && mine) {
                long o = base + tot + inc - mine;
                for (int p = b; p < e; p++) {
                    const CellRec kp = recs[p];
                    if (passes(kp)) { cands[o] = make_int2(kp.id | ((kp.octave & 0xff) << 24), qdesc ? hamming256(td + (size_t)kp.id * 4, d0, d1, d2, d3) : 0); o++; }
                }
            }
            tot += __shfl(inc, MC_G - 1, MC_G);
        }
        return tot;
    };
    total = walk(false);
    if (lane == 0) s_tot[grp] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int k = 0; k < MC_Q; k++) sum += s_tot[k];
        s_base = sum ? atomicAdd(cursor, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    base = (long)s_base;
    for (int k = 0; k < grp; k++) base += s_tot[k];
    const bool fits = base + total <= cap;
    if (!fits) total = 0;
    if (total) walk(true); // (the shuffles of a walk stay inside a group of MC_G lanes: a group without candidates may sit this one out)
    if (lane == 0 && qi < nq) { cstart[qi] = base; ccount[qi] = total; }
}

// ---- the order-dependent half of the searches: claims, best / second, ratio tests, rotation histogram (match_resolve) ---------------------------------------------------
// The reference walks the queries in order and lets query i see what queries < i did to the train key points: a key point claimed by a map point with observations is
// skipped (SearchByProjection, both overloads: ORBmatcher.cc:50-142, :1397-1494), a key point matched at a smaller or equal distance is skipped and a better match
// DISPLACES the earlier one (SearchForInitialization, :451-504), a key point matched once is never matched again (SearchByBoW, :171-310, :544-677).  One workgroup of one
// wave per search ("problem": one frame pair); the train-side state lives in LDS.  64 queries are resolved at a time, a lane each:
//   every lane picks its query's winner against the state BEFORE the batch plus the claims of the LOWER lanes of the batch (mask[idx]: which lanes claim key point idx;
//   the highest lane below mine is the one whose claim I would have seen last), publishes its claim, and the batch repeats until no lane changed its claim.  Lane 0 is
//   final after the first round, lane k after round k + 1 at the latest, and a round without change is the sequential result: a claim vector in which every lane's claim
//   is its choice given the lower lanes' claims is unique (induction over the lanes).  Typical batches settle in two or three rounds (5.6 candidates per query, 64 of
//   ~2 000 train key points touched).
// Then the batch commits (the highest claiming lane of a key point owns it), records (claim, rotation bin) per query, and the next batch starts.  After the last batch:
// the three-maxima cut over the histogram (:1860-1901) and the write-back.  Everything a query needs of a candidate (index, level, distance) is in the candidate list.
enum { RV_PROJ = 0, RV_LOCAL = 1, RV_INIT = 2, RV_BOW = 3, RV_BOWKF = 4 };
struct ResolveP {
    const long *cstart; const int *ccount; const int2 *cands; // per query
    const int *qlist;        // the visiting order (query index per step) or NULL: step = query index
    const int *pfirst;       // problem p visits steps pfirst[p] .. pfirst[p + 1]; NULL: one problem, steps 0 .. n_steps
    const int *kfirst;       // problem p's train key points: kfirst[p] .. kfirst[p + 1] of tkeys / tblocked; NULL: 0 .. n_train
    int n_steps, n_train;
    const cs_keypoint *tkeys; // train key points (angle)
    const cs_keypoint *qkeys; const float *qangle; // the queries' angles: key point records or a plain array
    const uint8_t *blocks;    // per query: its map point has observations (RV_PROJ, RV_LOCAL)
    const uint8_t *tblocked;  // per train key point: never a candidate (nullable)
    float nnratio; int check_orientation;
    int *train_match;        // RV_PROJ, RV_LOCAL, RV_BOW: per train key point, at (kfirst[p] - kfirst[0]) + idx: the matched query (index within the problem) or -1
    int *q_match;            // RV_INIT, RV_BOWKF: per query, the matched train key point or -1
    int *q_rec;              // scratch, per query
    int *nmatches;           // per problem
};
constexpr int RS_NC = 8; // candidates of a query kept in registers over the rounds of its batch (the rest is re-read)

__host__ __device__ inline void three_maxima(const int *sizes, int L, int &ind1, int &ind2, int &ind3) { // ORBmatcher.cc:1860-1901
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
__host__ __device__ inline int rot_bin(float a1, float a2) { // :1483-1491
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

template <int V> __global__ void __launch_bounds__(64) match_resolve(ResolveP P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_mem[];
    __shared__ int s_cd[64], s_hist[32], s_ind[3];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int s0 = P.pfirst ? P.pfirst[p] : 0, s1 = P.pfirst ? P.pfirst[p + 1] : P.n_steps;
    const int kb = P.kfirst ? P.kfirst[p] : 0, N2 = P.kfirst ? P.kfirst[p + 1] - kb : P.n_train, out0 = P.kfirst ? kb - P.kfirst[0] : 0;
    const int qsub = P.qlist ? 0 : s0;
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(rs_mem); // lanes of the current batch that claim the key point
    int *owner = reinterpret_cast<int *>(mask + N2); // RV_PROJ / RV_LOCAL: -1 free, -2 never, else query << 1 | its map point has observations; others: -1 free, -2 never, else query
    float *tang = reinterpret_cast<float *>(owner + N2); // the train key points' angles (the rotation bin of a claim without a global round trip)
    int *mdist = reinterpret_cast<int *>(tang + N2);     // RV_INIT: vMatchedDistance
    for (int i = lane; i < N2; i += 64) {
        mask[i] = 0; owner[i] = (P.tblocked && P.tblocked[kb + i]) ? -2 : -1; tang[i] = P.check_orientation ? P.tkeys[kb + i].angle : 0.0f;
        if (V == RV_INIT) mdist[i] = INT_MAX;
    }
    if (lane < 32) s_hist[lane] = 0;
    __syncthreads();
    const unsigned long long bit = 1ull << lane, below = bit - 1;
    int n_claims = 0;
    // A batch's global reads are issued one (the first candidates) and two (start, count, flags, angle) batches ahead.  (Measured on a one-pair search of 2 000 queries:
    // 262 -> 230 us.  What is left is the rounds' walk over the lists -- a lane per query pays an LDS round trip per candidate, and the coarse levels' windows hold dozens:
    // staging a batch's lists in LDS cooperatively, or reading eight candidates' words at once, both measured slower, 368 and 329 us.)
    struct Meta { int q; long cs; int cnt; bool act, blk; float ang; };
    auto load_meta = [&](int b0) {
        Meta M{0, 0, 0, false, false, 0.0f};
        const int step = b0 + lane;
        M.act = step < s1;
        if (M.act) {
            M.q = P.qlist ? P.qlist[step] : step;
            M.cs = P.cstart[M.q]; M.cnt = P.ccount[M.q];
            if (V == RV_PROJ || V == RV_LOCAL) M.blk = P.blocks[M.q] != 0;
            if (P.check_orientation) M.ang = P.qkeys ? P.qkeys[M.q].angle : P.qangle[M.q];
        }
        return M;
    };
    Meta Mc = load_meta(s0), Mn = load_meta(s0 + 64);
    int2 rc[RS_NC], rn[RS_NC];
#pragma unroll
    for (int c = 0; c < RS_NC; c++) rc[c] = c < Mc.cnt ? P.cands[Mc.cs + c] : make_int2(0, 0);
    for (int b0 = s0; b0 < s1; b0 += 64) {
#pragma unroll
        for (int c = 0; c < RS_NC; c++) rn[c] = c < Mn.cnt ? P.cands[Mn.cs + c] : make_int2(0, 0);
        const Meta Mnn = load_meta(b0 + 128);
        const bool act = Mc.act;
        const int q = Mc.q, cnt = Mc.cnt;
        const long cs = Mc.cs;
        const bool myblk = Mc.blk;
        const unsigned long long blkmask = __ballot(myblk);
        int claim = -1, cdist = 0;
        for (;;) {
            int best = V == RV_INIT ? INT_MAX : 256, best2 = best, bidx = -1, blev = -1, blev2 = -1;
            auto visit = [&](int2 cd) {
                const int idx = cd.x & 0xffffff, dist = cd.y;
                // (frame to frame only the minimum matters, and only if it is at most TH_HIGH: a candidate that cannot become the accepted minimum is dropped before its
                //  key point's words are read -- most of a window's candidates are other corners at a Hamming distance near 128)
                if (V == RV_PROJ && (dist > TH_HIGH || dist >= best)) return;
                const unsigned long long m = mask[idx] & below;
                if (V == RV_PROJ || V == RV_LOCAL) {
                    bool blk;
                    if (m) blk = (blkmask >> (63 - __clzll((long long)m))) & 1;
                    else { const int o = owner[idx]; blk = o == -2 || (o >= 0 && (o & 1)); }
                    if (blk) return;
                } else if (V == RV_INIT) {
                    const int eff = m ? s_cd[63 - __clzll((long long)m)] : mdist[idx];
                    if (eff <= dist) return;
                } else {
                    if (m || owner[idx] != -1) return;
                }
                if (V == RV_PROJ) { if (dist < best) { best = dist; bidx = idx; } }
                else if (V == RV_LOCAL) {
                    const int lev = (int)((unsigned)cd.x >> 24);
                    if (dist < best) { best2 = best; best = dist; blev2 = blev; blev = lev; bidx = idx; }
                    else if (dist < best2) { blev2 = lev; best2 = dist; }
                } else {
                    if (dist < best) { best2 = best; best = dist; bidx = idx; }
                    else if (dist < best2) best2 = dist;
                }
            };
#pragma unroll
            for (int c = 0; c < RS_NC; c++) if (c < cnt) visit(rc[c]);
            for (int c = RS_NC; c < cnt; c++) visit(P.cands[cs + c]);
            bool accept;
            if (V == RV_PROJ) accept = best <= TH_HIGH;                                                                         // :1469
            else if (V == RV_LOCAL) accept = best <= TH_HIGH && !(blev == blev2 && (float)best > P.nnratio * (float)best2);      // :126-131
            else if (V == RV_INIT) accept = best <= TH_LOW && (float)best < (float)best2 * P.nnratio;                            // :476-478
            else if (V == RV_BOW) accept = best <= TH_LOW && (float)best < P.nnratio * (float)best2;                             // :266-268
            else accept = best < TH_LOW && (float)best < P.nnratio * (float)best2;                                               // :625-627 (strict)
            const int nc = (accept && act && bidx >= 0) ? bidx : -1;
            const bool changed = nc != claim || (V == RV_INIT && nc >= 0 && best != cdist);
            __syncthreads(); // every lane has read the masks of this round
            if (nc != claim) {
                if (claim >= 0) atomicAnd(&mask[claim], ~bit);
                if (nc >= 0) atomicOr(&mask[nc], bit);
                claim = nc;
            }
            if (V == RV_INIT) { cdist = best; s_cd[lane] = best; }
            __syncthreads();
            if (__ballot(changed) == 0) break;
        }
        // commit: the highest claiming lane of a key point is the last writer of the sequential loop
        bool top = false; int bin = 0;
        if (claim >= 0) {
            top = (mask[claim] >> lane) == 1ull;
            if (P.check_orientation) bin = rot_bin(Mc.ang, tang[claim]) & 31;
        }
        __syncthreads();
        if (claim >= 0) {
            if (top) {
                if (V == RV_PROJ || V == RV_LOCAL) owner[claim] = ((q - qsub) << 1) | (int)myblk;
                else { owner[claim] = q - qsub; if (V == RV_INIT) mdist[claim] = cdist; }
            }
            mask[claim] = 0;
            if (P.check_orientation) atomicAdd(&s_hist[bin], 1);
        }
        if (act) P.q_rec[q] = claim >= 0 ? (claim | (bin << 24)) : -1;
        n_claims += __popcll(__ballot(claim >= 0));
        __syncthreads();
        Mc = Mn; Mn = Mnn;
#pragma unroll
        for (int c = 0; c < RS_NC; c++) rc[c] = rn[c];
    }
    // the rotation histogram's three maxima and the cut (:1496-1517, :506-527, :271-292, :630-651)
    const bool cut = P.check_orientation && V != RV_LOCAL;
    if (cut && lane == 0) { int i1 = -1, i2 = -1, i3 = -1; three_maxima(s_hist, HISTO_LENGTH, i1, i2, i3); s_ind[0] = i1; s_ind[1] = i2; s_ind[2] = i3; }
    __syncthreads();
    const int ind1 = cut ? s_ind[0] : -1, ind2 = cut ? s_ind[1] : -1, ind3 = cut ? s_ind[2] : -1;
    int n_cull = 0, n_keep = 0;
    for (int step = s0 + lane; step < s1; step += 64) {
        const int q = P.qlist ? P.qlist[step] : step;
        const int rec = __hip_atomic_load(&P.q_rec[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int res = -1;
        if (rec >= 0) {
            const int idx = rec & 0xffffff, bin = rec >> 24;
            const bool culled = cut && bin != ind1 && bin != ind2 && bin != ind3;
            if (V == RV_PROJ || V == RV_LOCAL || V == RV_BOW) { if (culled) { owner[idx] = -1; n_cull++; } } // every claim of a cut bin clears the key point and counts (:1506-1512)
            else if (owner[idx] == q - qsub && !culled) { res = idx; n_keep++; }                             // a displaced or cut query has no match (:487-491, :516-522)
        }
        if (V == RV_INIT || V == RV_BOWKF) P.q_match[q] = res;
    }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) { n_cull += __shfl_xor(n_cull, off); n_keep += __shfl_xor(n_keep, off); }
    if (V == RV_PROJ || V == RV_LOCAL || V == RV_BOW)
        for (int i = lane; i < N2; i += 64) { const int o = owner[i]; P.train_match[out0 + i] = o >= 0 ? ((V == RV_BOW) ? o : (o >> 1)) : -1; }
    if (lane == 0) P.nmatches[p] = (V == RV_INIT || V == RV_BOWKF) ? n_keep : n_claims - n_cull;
}

// ORBmatcher::Fuse's search (:934-981): per map point the best key point after the tests that need per-key-point data of the key frame; first minimum wins.  A thread per map point.
__global__ void __launch_bounds__(256) match_fuse_best(int n_mp, const long *cstart, const int *ccount, const int2 *cands, const cs_keypoint *keys, const float *u_right, const float *inv_level_sigma2,
                                                       int n_levels, const uint8_t *keys_static, const float *uv, const float *ur, const int *pred_level, const uint8_t *valid, int *best_idx, int *best_dist, int *err) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mp) return;
    int bestDist = 256, bestIdx = -1;
    if (valid[i]) {
        const float u = uv[i * 2], v = uv[i * 2 + 1];
        const long b = cstart[i]; const int n = ccount[i];
        for (int p = 0; p < n; p++) {
            const int2 cd = cands[b + p];
            const int idx = cd.x & 0xffffff;
            const cs_keypoint kp = keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < pred_level[i] - 1 || kpLevel > pred_level[i]) continue;
            if (keys_static && !keys_static[idx]) continue;
            if (kpLevel < 0 || kpLevel >= n_levels) { *err = 1; break; }
            if (u_right[idx] >= 0) {
                const float ex = u - kp.x, ey = v - kp.y, er = ur[i] - u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            if (cd.y < bestDist) { bestDist = cd.y; bestIdx = idx; }
        }
    }
    best_idx[i] = bestIdx; best_dist[i] = bestDist;
}

// The orientation cut of a search whose matches are independent of each other (SearchForTriangulation, :801-830): histogram, three maxima, cut, count.  One workgroup.
__global__ void __launch_bounds__(1024) match_orient_cut(int N1, const cs_keypoint *keys1, const cs_keypoint *keys2, int *matches12, int check_orientation, int *nmatches) {
    __shared__ int s_hist[32], s_ind[3], s_n;
    const int tid = threadIdx.x;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (check_orientation)
        for (int i = tid; i < N1; i += 1024) { const int m = matches12[i]; if (m >= 0) atomicAdd(&s_hist[rot_bin(keys1[i].angle, keys2[m].angle) & 31], 1); }
    __syncthreads();
    if (tid == 0) { int i1 = -1, i2 = -1, i3 = -1; if (check_orientation) three_maxima(s_hist, HISTO_LENGTH, i1, i2, i3); s_ind[0] = i1; s_ind[1] = i2; s_ind[2] = i3; }
    __syncthreads();
    int n = 0;
    for (int i = tid; i < N1; i += 1024) {
        const int m = matches12[i];
        if (m < 0) continue;
        if (check_orientation) { const int bin = rot_bin(keys1[i].angle, keys2[m].angle) & 31; if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) { matches12[i] = -1; continue; } }
        n++;
    }
    atomicAdd(&s_n, n);
    __syncthreads();
    if (tid == 0) *nmatches = s_n;
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
// SearchByBoW (:171-310, :544-677): distances of every key-frame feature to the frame features of its vocabulary node, one wave per key-frame
// feature, written as that feature's candidate list (the node's features in ascending index, the slice length is the node's size); match_resolve claims in the reference's order.
__global__ void __launch_bounds__(256) match_bow_dists(int NK, const unsigned long long *descK, const int *nodeK, const unsigned long long *descF, const int *node_start,
                                                       const int *node_items, const long *out_off, int2 *cands) {
    const int ik = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ik >= NK) return;
    const int nd = nodeK[ik];
    if (nd < 0) return;
    const unsigned long long a0 = descK[(long)ik * 4], a1 = descK[(long)ik * 4 + 1], a2 = descK[(long)ik * 4 + 2], a3 = descK[(long)ik * 4 + 3];
    const int b = node_start[nd], n = node_start[nd + 1] - b;
    for (int p = lane; p < n; p += 64) { const int iF = node_items[b + p]; cands[out_off[ik] + p] = make_int2(iF, hamming256(descF + (long)iF * 4, a0, a1, a2, a3)); }
}


// ---- a whole window of a stream at once (cs_match_by_projection_stream): pair p = (last frame f0 + p, current frame f0 + p + 1) of the frames an extractor holds in HBM.
// Frame post-processing of every current frame and the searches of all pairs in a handful of launches.  Same arithmetic, same candidate order, same claims as the per-frame calls.
// AssignFeaturesToGrid of the current frames: one workgroup per frame, key points kfirst[p] .. kfirst[p + 1] of `keys`; cell lists relative to the frame
__global__ void __launch_bounds__(1024) match_grid_batch(FrameP F, const cs_keypoint *keys, const int *kfirst, int *cell_start_all /*P x (NCELL+1)*/, int *cell_items_all, int *kp_cell_all, CellRec *cell_recs_all) {
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
    __syncthreads();
    for (int i = tid; i < cell_start[NCELL]; i += 1024) { const int id = cell_items[i]; const cs_keypoint kk = k[id]; cell_recs_all[kb + i] = CellRec{kk.x, kk.y, id, kk.octave}; }
}// the queries of all pairs: query j belongs to pair p with qfirst[p] <= j < qfirst[p + 1]; its level is the last frame's key point's (ORBmatcher.cc:1424)
__global__ void __launch_bounds__(256) match_project_stream(int nq, int n_pairs, const int *qfirst, const float *world_pos, const uint8_t *valid, const cs_keypoint *last_keys /* raw key points of frame f0 on: query j = last_keys[j] */,
                                                           const float *T_all, float fx, float fy, float cx, float cy, const float *scale_factors, int n_levels, float th, FrameP F, QueryS *q, int *err) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    int lo = 0, hi = n_pairs; // qfirst[lo] <= i < qfirst[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qfirst[mid] <= i) lo = mid; else hi = mid; }
    const float *T = T_all + 12 * lo;
    QueryS Q{0, 0, 0, 0, 0, 0, lo};
    if (valid[i]) {
        const int o = last_keys[i].octave;
        if (o < 0 || o >= n_levels) { *err = 1; q[i] = Q; return; }
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
            if (!(u < F.minX || u > F.maxX) && !(v < F.minY || v > F.maxY)) { Q.x = u; Q.y = v; Q.r = th * scale_factors[o]; Q.minLevel = o - 1; Q.maxLevel = o + 1; Q.valid = 1; }
        }
    }
    q[i] = Q;
}
} // namespace

// a device array that grows to what a call needs (each array has its OWN capacity: ADVICE r5 -- one cap shared by arrays of different needs let the smaller one overflow)
template <class T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    int grow(cs_ctx *ctx, size_t need) {
        if (p && need <= cap) return CS_OK;
        if (p) { cs_dfree(ctx, p); p = nullptr; cap = 0; }
        const size_t c = need + need / 4 + 64;
        const int r = cs_dalloc(ctx, &p, c);
        if (r == CS_OK) cap = c;
        return r;
    }
    void release(cs_ctx *ctx) { if (p) cs_dfree(ctx, p); p = nullptr; cap = 0; }
};

struct cs_matcher {
    int max_kp = 0, max_q = 0; long max_cand = 0;
    FrameP F{};
    std::vector<cs_keypoint> keys; // host copy of mvKeysUn (what set_frame_from_orb hands back; the vbPrevMatched update of SearchForInitialization)
    cs_keypoint *d_keys = nullptr; unsigned long long *d_desc = nullptr, *d_qdesc = nullptr;
    int *d_cell_start = nullptr, *d_cell_items = nullptr, *d_kp_cell = nullptr, *d_ccount = nullptr; CellRec *d_cell_recs = nullptr;
    long *d_cstart = nullptr; unsigned long long *d_cursor = nullptr; // [0] the arena's cursor, [1] an error flag (its low word)
    QueryS *d_q = nullptr; int2 *d_cands = nullptr;
    int *d_qrec = nullptr, *d_tm = nullptr /* max_kp + 1: train_match, nmatches behind its N entries */, *d_qm = nullptr /* max_q + 1 */;
    float *d_f = nullptr, *d_ang = nullptr, *d_ur = nullptr; uint8_t *d_u8 = nullptr /* 2 x max_q: valid | blocks */, *d_tb = nullptr /* max_kp */; int *d_i = nullptr;
    std::vector<int> h_res;
    long last_q = 0, last_c = 0;
};

// the window enumeration of nq queries in m->d_q against the frame set with cs_matcher_set_frame: cursor reset + one launch, nothing waits
static int mt_candidates(cs_ctx *ctx, cs_matcher *m, int nq, bool with_desc) {
    CS_HIP(ctx, hipMemsetAsync(m->d_cursor, 0, 16, ctx->stream));
    CS_LAUNCH(ctx, "match_candidates", match_candidates, dim3((nq + MC_Q - 1) / MC_Q), dim3(256), 0, m->F, m->d_desc, (const int *)nullptr, m->d_cell_start, m->d_cell_recs, nq, m->d_q,
              with_desc ? m->d_qdesc : (const unsigned long long *)nullptr, m->d_cursor, m->max_cand, m->d_cstart, m->d_ccount, m->d_cands);
    m->last_q = nq;
    return CS_OK;
}
template <int V> static int mt_resolve(cs_ctx *ctx, const ResolveP &P, int n_problems, int n2max) {
    const size_t lds = (size_t)std::max(n2max, 1) * (V == RV_INIT ? 20 : 16) + 16;
    if (lds > 150 * 1024) { ctx->err = "match_resolve: more train key points than one workgroup's LDS holds"; return CS_ERR_CAPACITY; }
    if (lds > 48 * 1024) CS_HIP(ctx, hipFuncSetAttribute((const void *)match_resolve<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CS_LAUNCH(ctx, "match_resolve", match_resolve<V>, dim3(n_problems), dim3(64), lds, P);
    return CS_OK;
}
// results (n_res ints and the match count behind them) + the cursor, one wait
static int mt_finish(cs_ctx *ctx, cs_matcher *m, const int *d_res, int n_res, int *out, int *nmatches) {
    m->h_res.resize((size_t)n_res + 1);
    unsigned long long cur[2] = {0, 0};
    int r = cs_d2h(ctx, m->h_res.data(), d_res, (size_t)n_res + 1); if (r) return r;
    r = cs_d2h(ctx, cur, m->d_cursor, 2); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->last_c = (long)cur[0];
    if ((long)cur[0] > m->max_cand) { ctx->err = "matcher candidate capacity exceeded"; return CS_ERR_CAPACITY; }
    if (n_res) memcpy(out, m->h_res.data(), sizeof(int) * (size_t)n_res);
    *nmatches = m->h_res[(size_t)n_res];
    return CS_OK;
}

struct cs_match_stream {
    DBuf<cs_keypoint> keys; DBuf<int> cell_start, cell_items, kp_cell, kfirst, qfirst, ccount, qrec, tm, nm; DBuf<CellRec> cell_recs; DBuf<long> cstart; DBuf<QueryS> q; DBuf<int2> cands;
    DBuf<float> wp, T; DBuf<uint8_t> valid, blocks; DBuf<unsigned long long> qdesc;
    float *d_sf = nullptr; unsigned long long *d_cursor = nullptr;
    std::vector<int> h_tm;
    long last_q = 0, last_c = 0;
};
extern "C" {

void cs_matcher_destroy(cs_ctx *ctx, cs_matcher *m) {
    if (!m) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {m->d_keys, m->d_desc, m->d_qdesc, m->d_cell_start, m->d_cell_items, m->d_cell_recs, m->d_kp_cell, m->d_ccount, m->d_cstart, m->d_cursor, m->d_q, m->d_cands,
                    m->d_qrec, m->d_tm, m->d_qm, m->d_f, m->d_ang, m->d_ur, m->d_u8, m->d_tb, m->d_i};
    for (void *p : ptrs) if (p) cs_dfree(ctx, p);
    delete m;
}

int cs_matcher_last_counts(const cs_matcher *m, int *queries, long *candidates) { // workload facts of the last search, for roofline accounting
    if (!m || !queries || !candidates) return CS_ERR_BAD_ARG;
    *queries = (int)m->last_q;
    *candidates = m->last_c;
    return CS_OK;
}

int cs_matcher_create(cs_ctx *ctx, int max_keypoints, int max_queries, long max_candidates, cs_matcher **out) {
    if (!ctx || !out || max_keypoints < 1 || max_keypoints >= (1 << 24) || max_queries < 1 || max_candidates < 1) return CS_ERR_BAD_ARG;
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
    A_(cs_dalloc(ctx, &m->d_cell_recs, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_kp_cell, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_ccount, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_cstart, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_cursor, (size_t)2));
    A_(cs_dalloc(ctx, &m->d_q, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_cands, (size_t)max_candidates));
    A_(cs_dalloc(ctx, &m->d_qrec, (size_t)max_queries + 1));
    A_(cs_dalloc(ctx, &m->d_tm, (size_t)max_keypoints + 1));
    A_(cs_dalloc(ctx, &m->d_qm, (size_t)max_queries + 1));
    A_(cs_dalloc(ctx, &m->d_f, (size_t)max_queries * 3 + 64));
    A_(cs_dalloc(ctx, &m->d_ang, (size_t)max_queries));
    A_(cs_dalloc(ctx, &m->d_ur, (size_t)max_keypoints));
    A_(cs_dalloc(ctx, &m->d_u8, (size_t)max_queries * 2));
    A_(cs_dalloc(ctx, &m->d_tb, (size_t)max_keypoints));
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
    CS_LAUNCH(ctx, "match_grid", match_grid, dim3(1), dim3(1024), 0, m->F, m->d_keys, m->d_cell_start, m->d_cell_items, m->d_kp_cell, m->d_cell_recs);
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
    CS_LAUNCH(ctx, "match_grid", match_grid, dim3(1), dim3(1024), 0, m->F, m->d_keys, m->d_cell_start, m->d_cell_items, m->d_kp_cell, m->d_cell_recs);
    // mvKeysUn for the caller (and for SearchForInitialization's vbPrevMatched update); a caller that wants neither passes NULL and the frame stays on the device:
    // every search takes what it needs of a train key point from the device copy
    if (keysUn_out) {
        m->keys.resize((size_t)N);
        r = cs_d2h(ctx, m->keys.data(), m->d_keys, (size_t)N); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(keysUn_out, m->keys.data(), sizeof(cs_keypoint) * (size_t)N);
    } else m->keys.clear();
    if (n_out) *n_out = N;
    return CS_OK;
}

int cs_matcher_features_in_area(cs_ctx *ctx, cs_matcher *m, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap, int *n) {
    if (!ctx || !m || !n) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    QueryS Q{x, y, r, minLevel, maxLevel, 1, 0};
    int rr = cs_h2d(ctx, m->d_q, &Q, 1); if (rr) return rr;
    rr = mt_candidates(ctx, m, 1, false); if (rr) return rr;
    long start = 0; int cnt = 0; unsigned long long cur[2] = {0, 0};
    rr = cs_d2h(ctx, &start, m->d_cstart, 1); if (rr) return rr;
    rr = cs_d2h(ctx, &cnt, m->d_ccount, 1); if (rr) return rr;
    rr = cs_d2h(ctx, cur, m->d_cursor, 2); if (rr) return rr;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->last_c = (long)cur[0];
    if ((long)cur[0] > m->max_cand) { ctx->err = "matcher candidate capacity exceeded"; return CS_ERR_CAPACITY; }
    *n = cnt;
    if (out && cnt > 0 && cap > 0) {
        std::vector<int2> c((size_t)cnt);
        rr = cs_d2h(ctx, c.data(), m->d_cands + start, (size_t)cnt); if (rr) return rr;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < cnt && i < cap; i++) out[i] = c[i].x & 0xffffff;
    }
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
    const int N = m->F.N;
    *nmatches = 0;
    if (n_last == 0) { for (int i = 0; i < N; i++) train_match[i] = -1; return CS_OK; }
    float *d_T = m->d_f + (size_t)m->max_q * 3, *d_sf = d_T + 12;
    uint8_t *d_blocks = m->d_u8 + m->max_q;
    int r = cs_h2d(ctx, m->d_f, world_pos, (size_t)n_last * 3); if (r) return r;
    r = cs_h2d(ctx, d_T, Tcw, 12); if (r) return r;
    r = cs_h2d(ctx, d_sf, scale_factors, (size_t)n_levels); if (r) return r;
    r = cs_h2d(ctx, m->d_u8, valid, (size_t)n_last); if (r) return r;
    r = cs_h2d(ctx, d_blocks, blocks, (size_t)n_last); if (r) return r;
    r = cs_h2d(ctx, m->d_i, last_octave, (size_t)n_last); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_last * 32); if (r) return r;
    if (check_orientation) { r = cs_h2d(ctx, m->d_ang, last_angle, (size_t)n_last); if (r) return r; }
    if (train_blocked) { r = cs_h2d(ctx, m->d_tb, train_blocked, (size_t)N); if (r) return r; } // map point from before the call / KeysStatic[i2] == false, :1451-1457
    CS_LAUNCH(ctx, "match_project", match_project, dim3((n_last + 255) / 256), dim3(256), 0, n_last, m->d_f, m->d_u8, m->d_i, d_T, fx, fy, cx, cy, d_sf, th,
              m->F, m->d_q);
    r = mt_candidates(ctx, m, n_last, true); if (r) return r;
    // the greedy pass (:1397-1494) and the rotation cut (:1496-1517), on the device
    ResolveP P{};
    P.cstart = m->d_cstart; P.ccount = m->d_ccount; P.cands = m->d_cands; P.n_steps = n_last; P.n_train = N; P.tkeys = m->d_keys; P.qangle = m->d_ang; P.blocks = d_blocks;
    P.tblocked = train_blocked ? m->d_tb : nullptr; P.check_orientation = check_orientation; P.train_match = m->d_tm; P.q_rec = m->d_qrec; P.nmatches = m->d_tm + N;
    r = mt_resolve<RV_PROJ>(ctx, P, 1, N); if (r) return r;
    return mt_finish(ctx, m, m->d_tm, N, train_match, nmatches);
}


// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1373-1522) for the pairs (f0 + p, f0 + p + 1), p < n_pairs, of the frames `orb` holds in HBM, with
// Frame::UndistortKeyPoints + AssignFeaturesToGrid (Frame.cc:303-318, 546-576) of every current frame: see include/cubeslam_hip.h.
void cs_match_stream_destroy(cs_ctx *ctx, cs_match_stream *m) {
    if (!m) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    m->keys.release(ctx); m->cell_start.release(ctx); m->cell_items.release(ctx); m->cell_recs.release(ctx); m->kp_cell.release(ctx); m->kfirst.release(ctx); m->qfirst.release(ctx); m->ccount.release(ctx); m->qrec.release(ctx);
    m->tm.release(ctx); m->nm.release(ctx); m->cstart.release(ctx); m->q.release(ctx); m->cands.release(ctx); m->wp.release(ctx); m->T.release(ctx); m->valid.release(ctx); m->blocks.release(ctx); m->qdesc.release(ctx);
    if (m->d_sf) cs_dfree(ctx, m->d_sf);
    if (m->d_cursor) cs_dfree(ctx, m->d_cursor);
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
                                  int n_queries, int n_train, const float *world_pos, const uint8_t *valid, const uint8_t *blocks, const uint8_t *mp_desc, const float *Tcw, float fx, float fy, float cx, float cy,
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
    const int nk = nk_all - first[1];
    // the caller sized world_pos / valid / blocks / mp_desc for n_queries and train_match for n_train: they must be what the extractor holds NOW
    if (n_queries != nq || n_train != nk) { ctx->err = "cs_match_by_projection_stream: n_queries / n_train differ from the extractor's key point counts"; return CS_ERR_BAD_ARG; }
    for (int p = 0; p < n_pairs; p++) nmatches[p] = 0;
    for (int i = 0; i < nk; i++) train_match[i] = -1;
    m->last_q = nq; m->last_c = 0;
    if (nq == 0 || nk == 0) return CS_OK;
    if (nk_all >= (1 << 24)) { int nmax = 0; for (int p = 1; p <= n_pairs; p++) nmax = std::max(nmax, first[p + 1] - first[p]); if (nmax >= (1 << 24)) return CS_ERR_CAPACITY; }
    FrameP F; F.N = 0; F.minX = minX; F.maxX = maxX; F.minY = minY; F.maxY = maxY;
    F.wInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX); F.hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);
    int r;
#define G_(call) do { r = (call); if (r != CS_OK) return r; } while (0)
    G_(m->keys.grow(ctx, (size_t)nk_all)); G_(m->cell_items.grow(ctx, (size_t)nk_all)); G_(m->cell_recs.grow(ctx, (size_t)nk_all)); G_(m->kp_cell.grow(ctx, (size_t)nk_all)); G_(m->tm.grow(ctx, (size_t)nk));
    G_(m->cell_start.grow(ctx, (size_t)n_pairs * (NCELL + 1))); G_(m->kfirst.grow(ctx, (size_t)n_pairs + 2)); G_(m->qfirst.grow(ctx, (size_t)n_pairs + 2)); G_(m->T.grow(ctx, (size_t)n_pairs * 12)); G_(m->nm.grow(ctx, (size_t)n_pairs));
    G_(m->ccount.grow(ctx, (size_t)nq)); G_(m->cstart.grow(ctx, (size_t)nq)); G_(m->qrec.grow(ctx, (size_t)nq)); G_(m->q.grow(ctx, (size_t)nq)); G_(m->wp.grow(ctx, (size_t)nq * 3)); G_(m->valid.grow(ctx, (size_t)nq));
    G_(m->blocks.grow(ctx, (size_t)nq)); if (mp_desc) G_(m->qdesc.grow(ctx, (size_t)nq * 4));
    if (!m->d_sf) G_(cs_dalloc(ctx, &m->d_sf, (size_t)32));
    if (!m->d_cursor) G_(cs_dalloc(ctx, &m->d_cursor, (size_t)2));
    if (!m->cands.p) G_(m->cands.grow(ctx, std::max((size_t)nq * 12, (size_t)1 << 16))); // (5.6 candidates per query at th 15 on the bench stream; the cursor says what a window really needs)
    // UndistortKeyPoints of every frame of the window (device to device), AssignFeaturesToGrid of the current frames
    CS_LAUNCH(ctx, "match_undistort", match_undistort, dim3((nk_all + 255) / 256), dim3(256), 0, nk_all, d_k0, make_undp(K4, dist5), m->keys.p);
    std::vector<int> kfirst((size_t)n_pairs + 1);
    int n2max = 0;
    for (int p = 0; p <= n_pairs; p++) { kfirst[p] = first[p + 1]; if (p < n_pairs) n2max = std::max(n2max, first[p + 2] - first[p + 1]); } // train frame of pair p: key points kfirst[p] .. kfirst[p + 1] of the window's list
    G_(cs_h2d(ctx, m->kfirst.p, kfirst.data(), (size_t)n_pairs + 1));
    G_(cs_h2d(ctx, m->qfirst.p, first.data(), (size_t)n_pairs + 1));
    G_(cs_h2d(ctx, m->T.p, Tcw, (size_t)n_pairs * 12));
    G_(cs_h2d(ctx, m->d_sf, scale_factors, (size_t)n_levels));
    G_(cs_h2d(ctx, m->wp.p, world_pos, (size_t)nq * 3));
    G_(cs_h2d(ctx, m->valid.p, valid, (size_t)nq));
    G_(cs_h2d(ctx, m->blocks.p, blocks, (size_t)nq));
    if (mp_desc) G_(cs_h2d(ctx, (uint8_t *)m->qdesc.p, mp_desc, (size_t)nq * 32));
    const unsigned long long *d_qdesc = mp_desc ? m->qdesc.p : d_d0; // NULL: a last frame's key point is matched with its own descriptor
    int *d_err = reinterpret_cast<int *>(m->d_cursor + 1);
    CS_HIP(ctx, hipMemsetAsync(m->d_cursor, 0, 16, ctx->stream));
    CS_LAUNCH(ctx, "match_grid", match_grid_batch, dim3(n_pairs), dim3(1024), 0, F, m->keys.p, m->kfirst.p, m->cell_start.p, m->cell_items.p, m->kp_cell.p, m->cell_recs.p);
    CS_LAUNCH(ctx, "match_project", match_project_stream, dim3((nq + 255) / 256), dim3(256), 0, nq, n_pairs, m->qfirst.p, m->wp.p, m->valid.p, d_k0, m->T.p, fx, fy, cx, cy, m->d_sf, n_levels, th, F, m->q.p, d_err);
    // every pair's window enumeration (one launch) and its greedy pass + rotation cut (:1397-1517; one wave per pair): nothing of a candidate leaves the device.  The arena's
    // size is last call's need; a window that needs more says so through the cursor and the two launches run again on a larger one.
    unsigned long long cur[2] = {0, 0};
    for (int attempt = 0;; attempt++) {
        CS_LAUNCH(ctx, "match_candidates", match_candidates, dim3((nq + MC_Q - 1) / MC_Q), dim3(256), 0, F, d_d0, m->kfirst.p, m->cell_start.p, m->cell_recs.p, nq, m->q.p, d_qdesc, m->d_cursor, (long)m->cands.cap,
                  m->cstart.p, m->ccount.p, m->cands.p);
        ResolveP P{};
        P.cstart = m->cstart.p; P.ccount = m->ccount.p; P.cands = m->cands.p; P.pfirst = m->qfirst.p; P.kfirst = m->kfirst.p; P.tkeys = m->keys.p; P.qkeys = m->keys.p; P.blocks = m->blocks.p;
        P.check_orientation = check_orientation; P.train_match = m->tm.p; P.q_rec = m->qrec.p; P.nmatches = m->nm.p;
        G_(mt_resolve<RV_PROJ>(ctx, P, n_pairs, n2max));
        G_(cs_d2h(ctx, cur, m->d_cursor, 2));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((int)(cur[1] & 0xffffffffu)) return CS_ERR_BAD_ARG; // a valid query whose key point's level is outside the scale table
        if ((size_t)cur[0] <= m->cands.cap) break;
        if (attempt) { ctx->err = "cs_match_by_projection_stream: candidate arena"; return CS_ERR_CAPACITY; }
        G_(m->cands.grow(ctx, (size_t)cur[0]));
        CS_HIP(ctx, hipMemsetAsync(m->d_cursor, 0, 16, ctx->stream));
    }
    m->last_c = (long)cur[0];
    G_(cs_d2h(ctx, train_match, m->tm.p, (size_t)nk));
    G_(cs_d2h(ctx, nmatches, m->nm.p, (size_t)n_pairs));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
#undef G_
    return CS_OK;
}

int cs_match_local_map(cs_ctx *ctx, cs_matcher *m, int n_mp, const float *proj_xy, const float *view_cos, const int *pred_level, const uint8_t *in_view,
                       const uint8_t *blocks, const uint8_t *mp_desc, const float *scale_factors, int n_levels, float th, float nnratio,
                       const uint8_t *train_blocked, int *train_match, int *nmatches) {
    if (!ctx || !m || n_mp < 0 || n_mp > m->max_q || !proj_xy || !view_cos || !pred_level || !in_view || !blocks || !mp_desc || !scale_factors ||
        !train_match || !nmatches)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const int N = m->F.N;
    *nmatches = 0;
    if (n_mp == 0) { for (int i = 0; i < N; i++) train_match[i] = -1; return CS_OK; }
    const bool bFactor = th != 1.0;
    std::vector<QueryS> q((size_t)n_mp);
    for (int i = 0; i < n_mp; i++) { // :58-78 (window radius from the viewing cosine)
        QueryS Q{0, 0, 0, 0, 0, 0, 0};
        if (in_view[i]) {
            if (pred_level[i] < 0 || pred_level[i] >= n_levels) return CS_ERR_BAD_ARG;
            float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            Q.x = proj_xy[i * 2]; Q.y = proj_xy[i * 2 + 1]; Q.r = r * scale_factors[pred_level[i]];
            Q.minLevel = pred_level[i] - 1; Q.maxLevel = pred_level[i]; Q.valid = 1;
        }
        q[i] = Q;
    }
    uint8_t *d_blocks = m->d_u8 + m->max_q;
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_mp * 32); if (r) return r;
    r = cs_h2d(ctx, d_blocks, blocks, (size_t)n_mp); if (r) return r;
    if (train_blocked) { r = cs_h2d(ctx, m->d_tb, train_blocked, (size_t)N); if (r) return r; }
    r = mt_candidates(ctx, m, n_mp, true); if (r) return r;
    ResolveP P{}; // :86-140: best / second with their levels, the ratio rule, claims
    P.cstart = m->d_cstart; P.ccount = m->d_ccount; P.cands = m->d_cands; P.n_steps = n_mp; P.n_train = N; P.tkeys = m->d_keys; P.blocks = d_blocks;
    P.tblocked = train_blocked ? m->d_tb : nullptr; P.nnratio = nnratio; P.check_orientation = 0; P.train_match = m->d_tm; P.q_rec = m->d_qrec; P.nmatches = m->d_tm + N;
    r = mt_resolve<RV_LOCAL>(ctx, P, 1, N); if (r) return r;
    return mt_finish(ctx, m, m->d_tm, N, train_match, nmatches);
}

int cs_match_for_initialization(cs_ctx *ctx, cs_matcher *m, const cs_keypoint *keys1, const uint8_t *desc1, int N1, float *prev, int window_size,
                                float nnratio, int check_orientation, int *vnMatches12, int *nmatches) {
    if (!ctx || !m || !keys1 || !desc1 || N1 < 0 || N1 > m->max_q || !prev || !vnMatches12 || !nmatches) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    *nmatches = 0;
    if (N1 == 0) return CS_OK;
    if ((int)m->keys.size() != m->F.N) { ctx->err = "cs_match_for_initialization: the frame was set without a host copy of its key points"; return CS_ERR_BAD_ARG; }
    std::vector<QueryS> q((size_t)N1);
    std::vector<float> ang((size_t)N1);
    for (int i = 0; i < N1; i++) { // :441-449: level-0 keypoints only, window around vbPrevMatched
        QueryS Q{0, 0, 0, 0, 0, 0, 0};
        if (!(keys1[i].octave > 0)) { Q.x = prev[i * 2]; Q.y = prev[i * 2 + 1]; Q.r = (float)window_size; Q.minLevel = keys1[i].octave; Q.maxLevel = keys1[i].octave; Q.valid = 1; }
        q[i] = Q; ang[i] = keys1[i].angle;
    }
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)N1); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, desc1, (size_t)N1 * 32); if (r) return r;
    r = cs_h2d(ctx, m->d_ang, ang.data(), (size_t)N1); if (r) return r;
    r = mt_candidates(ctx, m, N1, true); if (r) return r;
    ResolveP P{}; // :451-504 (vMatchedDistance, vnMatches21 and the displacement of an earlier match) and :506-527
    P.cstart = m->d_cstart; P.ccount = m->d_ccount; P.cands = m->d_cands; P.n_steps = N1; P.n_train = m->F.N; P.tkeys = m->d_keys; P.qangle = m->d_ang;
    P.nnratio = nnratio; P.check_orientation = check_orientation; P.q_match = m->d_qm; P.q_rec = m->d_qrec; P.nmatches = m->d_qm + N1;
    r = mt_resolve<RV_INIT>(ctx, P, 1, m->F.N); if (r) return r;
    r = mt_finish(ctx, m, m->d_qm, N1, vnMatches12, nmatches); if (r) return r;
    for (int i1 = 0; i1 < N1; i1++) // :529-532
        if (vnMatches12[i1] >= 0) { prev[i1 * 2] = m->keys[vnMatches12[i1]].x; prev[i1 * 2 + 1] = m->keys[vnMatches12[i1]].y; }
    return CS_OK;
}

int cs_match_fuse(cs_ctx *ctx, cs_matcher *m, const float *u_right, const float *inv_level_sigma2, int n_levels, const uint8_t *keys_static, int n_mp, const float *uv,
                  const float *ur, const int *pred_level, const uint8_t *valid, const uint8_t *mp_desc, const float *scale_factors, float th, int *best_idx,
                  int *best_dist, int *n_fused) {
    if (!ctx || !m || n_mp < 0 || n_mp > m->max_q || !u_right || !inv_level_sigma2 || n_levels < 1 || n_levels > 32 || (n_mp && (!uv || !ur || !pred_level || !valid || !mp_desc)) ||
        !scale_factors || !best_idx || !best_dist || !n_fused)
        return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    *n_fused = 0;
    if (n_mp == 0) return CS_OK;
    const int N = m->F.N;
    std::vector<QueryS> q((size_t)n_mp);
    for (int i = 0; i < n_mp; i++) { // :921-926: GetFeaturesInArea(u, v, th * scale[level]) without level limits
        QueryS Q{0, 0, 0, -1, -1, 0, 0};
        if (valid[i]) {
            if (pred_level[i] < 0 || pred_level[i] >= n_levels) return CS_ERR_BAD_ARG;
            Q.x = uv[i * 2]; Q.y = uv[i * 2 + 1]; Q.r = th * scale_factors[pred_level[i]]; Q.valid = 1;
        }
        q[i] = Q;
    }
    float *d_uv = m->d_f, *d_urq = m->d_f + (size_t)n_mp * 2, *d_inv = m->d_f + (size_t)m->max_q * 3 + 12;
    int *d_err = reinterpret_cast<int *>(m->d_cursor + 1);
    int r = cs_h2d(ctx, m->d_q, q.data(), (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, (uint8_t *)m->d_qdesc, mp_desc, (size_t)n_mp * 32); if (r) return r;
    r = cs_h2d(ctx, d_uv, uv, (size_t)n_mp * 2); if (r) return r;
    r = cs_h2d(ctx, d_urq, ur, (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, d_inv, inv_level_sigma2, (size_t)n_levels); if (r) return r;
    r = cs_h2d(ctx, m->d_i, pred_level, (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, m->d_u8, valid, (size_t)n_mp); if (r) return r;
    r = cs_h2d(ctx, m->d_ur, u_right, (size_t)N); if (r) return r;
    if (keys_static) { r = cs_h2d(ctx, m->d_tb, keys_static, (size_t)N); if (r) return r; }
    r = mt_candidates(ctx, m, n_mp, true); if (r) return r;
    // :934-981: the tests that need per-keypoint data of the key frame, first minimum wins
    CS_LAUNCH(ctx, "match_fuse_best", match_fuse_best, dim3((n_mp + 255) / 256), dim3(256), 0, n_mp, m->d_cstart, m->d_ccount, m->d_cands, m->d_keys, m->d_ur, d_inv, n_levels,
              keys_static ? m->d_tb : (const uint8_t *)nullptr, d_uv, d_urq, m->d_i, m->d_u8, m->d_qm, m->d_qrec, d_err);
    unsigned long long cur[2] = {0, 0};
    r = cs_d2h(ctx, best_idx, m->d_qm, (size_t)n_mp); if (r) return r;
    r = cs_d2h(ctx, best_dist, m->d_qrec, (size_t)n_mp); if (r) return r;
    r = cs_d2h(ctx, cur, m->d_cursor, 2); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->last_c = (long)cur[0];
    if ((long)cur[0] > m->max_cand) { ctx->err = "matcher candidate capacity exceeded"; return CS_ERR_CAPACITY; }
    if ((int)(cur[1] & 0xffffffffu)) return CS_ERR_BAD_ARG; // a candidate's level outside the sigma table
    int nf = 0;
    for (int i = 0; i < n_mp; i++) if (best_dist[i] <= TH_LOW) nf++;
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
    TA_(cs_dalloc(ctx, &d_n1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_st, (size_t)n_nodes + 1)); TA_(cs_dalloc(ctx, &d_it, items.size())); TA_(cs_dalloc(ctx, &d_m, (size_t)N1 + 1));
    TA_(cs_dalloc(ctx, &d_s1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_s2, (size_t)N2)); TA_(cs_dalloc(ctx, &d_u1, (size_t)N1)); TA_(cs_dalloc(ctx, &d_u2, (size_t)N2));
    TA_(cs_dalloc(ctx, &d_sc, (size_t)n_levels)); TA_(cs_dalloc(ctx, &d_sg, (size_t)n_levels));
    TA_(cs_h2d(ctx, d_k1, keys1Un, (size_t)N1)); TA_(cs_h2d(ctx, d_k2, keys2Un, (size_t)N2));
    TA_(cs_h2d(ctx, (uint8_t *)d_d1, desc1, (size_t)N1 * 32)); TA_(cs_h2d(ctx, (uint8_t *)d_d2, desc2, (size_t)N2 * 32));
    TA_(cs_h2d(ctx, d_n1, n1c.data(), (size_t)N1)); TA_(cs_h2d(ctx, d_st, start.data(), start.size())); TA_(cs_h2d(ctx, d_it, items.data(), items.size()));
    TA_(cs_h2d(ctx, d_s1, skip1, (size_t)N1)); TA_(cs_h2d(ctx, d_s2, skip2, (size_t)N2)); TA_(cs_h2d(ctx, d_u1, u_right1, (size_t)N1)); TA_(cs_h2d(ctx, d_u2, u_right2, (size_t)N2));
    TA_(cs_h2d(ctx, d_sc, scale_factors2, (size_t)n_levels)); TA_(cs_h2d(ctx, d_sg, level_sigma2_2, (size_t)n_levels));
#undef TA_
    std::vector<int> res((size_t)N1 + 1);
    if (!r) {
        TriP P; for (int k = 0; k < 9; k++) P.F12[k] = F12[k];
        P.ex = ex; P.ey = ey; P.only_stereo = only_stereo;
        CS_LAUNCH(ctx, "match_triangulation", match_triangulation, dim3((N1 + 3) / 4), dim3(256), 0, N1, d_k1, d_d1, d_n1, d_s1, d_u1, d_k2, d_d2, d_s2, d_u2, d_st, d_it, n_nodes, P,
                  d_sc, d_sg, d_m);
        CS_LAUNCH(ctx, "match_orient_cut", match_orient_cut, dim3(1), dim3(1024), 0, N1, d_k1, d_k2, d_m, check_orientation, d_m + N1); // :801-830
        r = cs_d2h(ctx, res.data(), d_m, (size_t)N1 + 1);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_k1, d_k2, d_d1, d_d2, d_n1, d_st, d_it, d_m, d_s1, d_s2, d_u1, d_u2, d_sc, d_sg};
    for (void *p : ptrs) if (p) cs_dfree(ctx, p);
    if (r) return r;
    memcpy(matches12, res.data(), sizeof(int) * (size_t)N1);
    *nmatches = res[(size_t)N1];
    return CS_OK;
}

} // extern "C"

// Both SearchByBoW overloads: the K side's features are visited by (vocabulary node ascending, index ascending) -- the reference's std::map / vector order -- and each one's
// candidates are the F side's features of its node in ascending index.  The distances (one wave per K feature) and the greedy claims (match_resolve) run on the device; the
// host only lays out the node lists.  V = RV_BOW: out = matchesF[NF] (K feature per F feature); RV_BOWKF: out = matches12[NK] (F feature per K feature).
template <int V> static int bow_search(cs_ctx *ctx, const cs_keypoint *keysK, const uint8_t *descK, int NK, const int *nodeK, const uint8_t *skipK, const cs_keypoint *keysF, const uint8_t *descF, int NF,
                                       const int *nodeF, const uint8_t *skipF, float nnratio, int check_orientation, int *out, int *nmatches) {
    CS_HIP(ctx, hipSetDevice(ctx->device));
    if (NF >= (1 << 24)) return CS_ERR_CAPACITY;
    std::vector<int> ids;
    for (int i = 0; i < NF; i++) if (nodeF[i] >= 0) ids.push_back(nodeF[i]);
    std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int n_nodes = (int)ids.size();
    auto compact = [&](int nd) { if (nd < 0) return -1; auto it = std::lower_bound(ids.begin(), ids.end(), nd); return (it != ids.end() && *it == nd) ? (int)(it - ids.begin()) : -1; };
    std::vector<int> start((size_t)n_nodes + 1, 0), items, nkc((size_t)NK, -1), nfc((size_t)NF), cnt((size_t)NK, 0);
    std::vector<long> off((size_t)NK + 1, 0);
    for (int i = 0; i < NF; i++) { nfc[i] = compact(nodeF[i]); if (nfc[i] >= 0) start[nfc[i] + 1]++; }
    for (int k = 0; k < n_nodes; k++) start[k + 1] += start[k];
    items.resize((size_t)std::max(start[n_nodes], 1));
    { std::vector<int> pos(start.begin(), start.end() - 1); for (int i = 0; i < NF; i++) if (nfc[i] >= 0) items[pos[nfc[i]]++] = i; }
    for (int i = 0; i < NK; i++) { nkc[i] = skipK[i] ? -1 : compact(nodeK[i]); cnt[i] = nkc[i] >= 0 ? start[nkc[i] + 1] - start[nkc[i]] : 0; off[i + 1] = off[i] + cnt[i]; }
    const long total = off[NK];
    // the reference's order: nodes ascending (std::map), K features ascending inside a node
    std::vector<std::pair<int, int>> order;
    for (int i = 0; i < NK; i++) if (nkc[i] >= 0) order.push_back(std::make_pair(nodeK[i], i));
    std::sort(order.begin(), order.end());
    const int n_steps = (int)order.size();
    const int n_out = V == RV_BOW ? NF : NK;
    *nmatches = 0;
    if (n_steps == 0 || total == 0) return CS_OK; // (out is pre-filled with -1 by the callers)
    std::vector<int> qlist((size_t)n_steps);
    for (int s = 0; s < n_steps; s++) qlist[s] = order[s].second;
    cs_keypoint *d_kk = nullptr, *d_kf = nullptr; unsigned long long *d_dk = nullptr, *d_df = nullptr; int *d_nk = nullptr, *d_st = nullptr, *d_it = nullptr, *d_cnt = nullptr, *d_ql = nullptr, *d_rec = nullptr, *d_out = nullptr;
    long *d_off = nullptr; int2 *d_ca = nullptr; uint8_t *d_sf = nullptr;
    int r = cs_dalloc(ctx, &d_dk, (size_t)NK * 4);
#define BA_(call) if (!r) r = (call)
    BA_(cs_dalloc(ctx, &d_df, (size_t)NF * 4)); BA_(cs_dalloc(ctx, &d_nk, (size_t)NK)); BA_(cs_dalloc(ctx, &d_st, start.size())); BA_(cs_dalloc(ctx, &d_it, items.size()));
    BA_(cs_dalloc(ctx, &d_off, off.size())); BA_(cs_dalloc(ctx, &d_ca, (size_t)total)); BA_(cs_dalloc(ctx, &d_cnt, (size_t)NK)); BA_(cs_dalloc(ctx, &d_ql, (size_t)n_steps)); BA_(cs_dalloc(ctx, &d_rec, (size_t)NK));
    BA_(cs_dalloc(ctx, &d_out, (size_t)n_out + 1)); BA_(cs_dalloc(ctx, &d_kk, (size_t)NK)); BA_(cs_dalloc(ctx, &d_kf, (size_t)NF)); if (skipF) BA_(cs_dalloc(ctx, &d_sf, (size_t)NF));
    BA_(cs_h2d(ctx, (uint8_t *)d_dk, descK, (size_t)NK * 32)); BA_(cs_h2d(ctx, (uint8_t *)d_df, descF, (size_t)NF * 32)); BA_(cs_h2d(ctx, d_nk, nkc.data(), (size_t)NK));
    BA_(cs_h2d(ctx, d_st, start.data(), start.size())); BA_(cs_h2d(ctx, d_it, items.data(), items.size())); BA_(cs_h2d(ctx, d_off, off.data(), off.size()));
    BA_(cs_h2d(ctx, d_cnt, cnt.data(), (size_t)NK)); BA_(cs_h2d(ctx, d_ql, qlist.data(), (size_t)n_steps)); BA_(cs_h2d(ctx, d_kk, keysK, (size_t)NK)); BA_(cs_h2d(ctx, d_kf, keysF, (size_t)NF));
    if (skipF) BA_(cs_h2d(ctx, d_sf, skipF, (size_t)NF));
    std::vector<int> res((size_t)n_out + 1, -1);
    if (!r && V == RV_BOWKF) { hipError_t e = hipMemsetAsync(d_out, 0xff, sizeof(int) * (size_t)NK, ctx->stream); if (e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; } } // K features outside the visiting order have no match
    if (!r) {
        CS_LAUNCH(ctx, "match_bow_dists", match_bow_dists, dim3((NK + 3) / 4), dim3(256), 0, NK, d_dk, d_nk, d_df, d_st, d_it, d_off, d_ca);
        ResolveP P{};
        P.cstart = d_off; P.ccount = d_cnt; P.cands = d_ca; P.qlist = d_ql; P.n_steps = n_steps; P.n_train = NF; P.tkeys = d_kf; P.qkeys = d_kk; P.tblocked = d_sf; P.nnratio = nnratio;
        P.check_orientation = check_orientation; P.train_match = d_out; P.q_match = d_out; P.q_rec = d_rec; P.nmatches = d_out + n_out;
        r = mt_resolve<V>(ctx, P, 1, NF);
    }
    BA_(cs_d2h(ctx, res.data(), d_out, (size_t)n_out + 1));
#undef BA_
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    void *ptrs[] = {d_kk, d_kf, d_dk, d_df, d_nk, d_st, d_it, d_cnt, d_ql, d_rec, d_out, d_off, d_ca, d_sf};
    for (void *p : ptrs) if (p) cs_dfree(ctx, p);
    if (r) return r;
    memcpy(out, res.data(), sizeof(int) * (size_t)n_out);
    *nmatches = res[(size_t)n_out];
    return CS_OK;
}

extern "C" {

int cs_match_by_bow(cs_ctx *ctx, const cs_keypoint *keysKF, const uint8_t *descKF, int NK, const int *nodeKF, const uint8_t *skipKF, const cs_keypoint *keysF,
                    const uint8_t *descF, int NF, const int *nodeF, const uint8_t *skipF, float nnratio, int check_orientation, int *matchesF, int *nmatches) {
    if (!ctx || NK < 0 || NF < 0 || !matchesF || !nmatches || (NK && (!keysKF || !descKF || !nodeKF || !skipKF)) || (NF && (!keysF || !descF || !nodeF))) return CS_ERR_BAD_ARG;
    *nmatches = 0;
    for (int i = 0; i < NF; i++) matchesF[i] = -1;
    if (NK == 0 || NF == 0) return CS_OK;
    return bow_search<RV_BOW>(ctx, keysKF, descKF, NK, nodeKF, skipKF, keysF, descF, NF, nodeF, skipF, nnratio, check_orientation, matchesF, nmatches); // :171-310
}

int cs_match_by_bow_kf(cs_ctx *ctx, const cs_keypoint *keys1, const uint8_t *desc1, int N1, const int *node1, const uint8_t *skip1, const cs_keypoint *keys2,
                       const uint8_t *desc2, int N2, const int *node2, const uint8_t *skip2, float nnratio, int check_orientation, int *matches12, int *nmatches) {
    if (!ctx || N1 < 0 || N2 < 0 || !nmatches || (N1 && (!keys1 || !desc1 || !node1 || !skip1 || !matches12)) || (N2 && (!keys2 || !desc2 || !node2 || !skip2))) return CS_ERR_BAD_ARG;
    *nmatches = 0;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    if (N1 == 0 || N2 == 0) return CS_OK;
    return bow_search<RV_BOWKF>(ctx, keys1, desc1, N1, node1, skip1, keys2, desc2, N2, node2, skip2, nnratio, check_orientation, matches12, nmatches); // :544-677 (vbMatched2 = a claimed key point; skip2 = no usable map point)
}

int cs_hamming_knn2(cs_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx, int *best_dist, int *second_dist) {
    if (!ctx || !q || !t || nq < 0 || nt < 0 || !best_idx || !best_dist || !second_dist) return CS_ERR_BAD_ARG;
    if (nq == 0) return CS_OK;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *dq = nullptr, *dt = nullptr; int *dres = nullptr;
    int r = cs_dalloc(ctx, &dq, (size_t)nq * 4); if (r) return r;
    r = cs_dalloc(ctx, &dt, (size_t)std::max(nt, 1) * 4); if (r) { cs_dfree(ctx, dq); return r; }
    r = cs_dalloc(ctx, &dres, (size_t)nq * 3); if (r) { cs_dfree(ctx, dq); cs_dfree(ctx, dt); return r; }
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
    cs_dfree(ctx, dq); cs_dfree(ctx, dt); cs_dfree(ctx, dres);
    return r;
}

} // extern "C"
