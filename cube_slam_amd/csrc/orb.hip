// orb.hip -- ORB_SLAM2::ORBextractor on MI355X (gfx950).
//
// Replaces ORBextractor::operator() (reference orb_object_slam/src/ORBextractor.cc:1036-1099) and what it calls.
// Built with -ffp-contract=off (the float rotation px*b + py*a feeds cvRound).  OpenCV semantics assumed: DESIGN.md section 7.
//
//   orb_resize       level l from l-1: cv::resize INTER_LINEAR 8-bit fixed point, coefficient tables built on the host
//   orb_fast_score   S(p) = max over the 16 nine-pixel arcs of min(v - ring) / min(ring - v): p is a FAST-9/16 corner for
//                    threshold t iff S > t and cv::FAST's cornerScore is S - 1 -- one threshold-free map serves both the
//                    iniThFAST pass and the minThFAST fallback (:809-817)
//   orb_cells        per 30x30-ish cell: strict 3x3 NMS inside the cell's own FAST window, threshold fallback per cell,
//                    count (pass 0) / ordered emit (pass 1) with ballot prefix sums
//   orb_scan_*       exclusive scans of the cell counts -> one compact candidate array for the whole batch
//   (host)           DistributeOctTree, index based (orb_quadtree.h)
//   orb_angle        IC_Angle: one wave per keypoint, a patch row per lane, integer moments, cv::fastAtan2
//   orb_blur         7x7 sigma-2 Gaussian, 8-bit fixed point, LDS tile
//   orb_desc         steered rBRIEF: one wave per keypoint, lane = test, 4 x ballot -> 4 x u64
#include "common.h"
#include "orb_quadtree.h"

#include <cfloat>
#include <chrono>
#include <cmath>
#include <omp.h>

namespace {

constexpr int MAXL = 16;
constexpr int HALF_PATCH = 15, PATCH = 31, EDGE_T = 19, MINB = EDGE_T - 3; // ORBextractor.cc:70-72,776
constexpr int TW = 64, TH = 16;
constexpr int BLUR_ROWS = 64; // output rows per workgroup of the row-streaming kernels (orb_fast_score, orb_blur)

struct Lvl {
    int w, h;
    long off;                 // byte offset of this level inside one frame's pyramid
    int nCols, nRows, wCell, hCell, maxBX, maxBY;
    int cell_off;             // first cell of this level inside one frame's cell array
    int tab_x, tab_y;         // offsets of the resize tables
    float scale; int patch;   // mvScaleFactor[level], scaledPatchSize
};
struct Pyr {
    int nlevels, cells_per_frame, ini_th, min_th;
    long frame_stride;
    int umax[HALF_PATCH + 1];
    int gk[7];
    int tile_off[MAXL + 1], blur_off[MAXL + 1]; // first workgroup of every level in the grids of orb_fast_score / orb_blur (a grid of max-tiles x levels was 61 % workgroups that found nothing to do)
    Lvl l[MAXL];
};

__constant__ int c_ring[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
__device__ const int d_pattern[1024] = {
#include "orb_pattern.inc"
};

// Four neighbouring output pixels x RS_ROWS output rows per thread.  The four source columns span at most six bytes of a source row
// (scale 1.2), so each source row is fetched with one unaligned 8-byte window and the (p, p+1) pairs are cut out with
// v_perm_b32 and multiplied with v_dot2_u32_u16; a result row leaves as one dword.  The column tables are read once per thread and the windows of all RS_ROWS rows
// are requested before the first one is used: a thread per pixel is a chain of three dependent memory round trips (level
// descriptor -> tables -> pixels) with a handful of instructions in between, and the launch was bound by that latency (95 us for
// the 27 Mpixel level 1, 12x its memory time).  Products are 24-bit multiplies (pixel < 2^8, coefficients <= 2^11, r >> 4 < 2^15): a
// plain int product is v_mul_lo_u32, a quarter-rate instruction.  Same integer arithmetic as cv::resize's 8-bit fixed-point path.
#ifndef ORB_RS_ROWS
#define ORB_RS_ROWS 8
#endif
constexpr int RS_ROWS = ORB_RS_ROWS;
__global__ void __launch_bounds__(256) orb_resize(Pyr P, int level, uint8_t *__restrict__ pyr, const int *__restrict__ xofs, const short *__restrict__ ialpha, const int *__restrict__ yofs, const short *__restrict__ ibeta) {
    const Lvl &D = P.l[level], &S = P.l[level - 1];
    // a wave works on ONE group of rows: said so to the compiler, the row tables (yofs, ibeta) and the row base addresses are scalar loads / registers and the pixel
    // windows of all rows leave as one burst of vector loads instead of a load-wait chain per row (level 1 of 1 024 frames: 485 -> 371 us).  The kernel is bound by the
    // instructions its waves issue, whatever their lanes do: measured and dropped -- sixteen rows per thread (464 us), the row's last narrow tile folded into the previous
    // tile's lanes (492) or packed four row groups to a wave (312, but every other level slower through the second copy of the body)
    const int dx = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, dy0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * RS_ROWS;
    if (dx >= D.w || dy0 >= D.h) return;
    uint8_t *base = pyr + (long)blockIdx.z * P.frame_stride;
    const uint8_t *src = base + S.off;
    const int nrow = min(RS_ROWS, D.h - dy0);
    int sx[4], a0[4], a1[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int x = min(dx + c, D.w - 1);
        const int aa = reinterpret_cast<const int *>(ialpha)[D.tab_x + x]; // the coefficient pair of a column as one dword
        sx[c] = xofs[D.tab_x + x]; a0[c] = (short)(aa & 0xffff); a1[c] = aa >> 16;
    }
    const bool fast = dx + 3 < D.w && sx[3] - sx[0] <= 4 && sx[0] + 7 < S.w; // one 8-byte window per source row, always inside the row
    if (fast) {
        uint32_t lo0[RS_ROWS], hi0[RS_ROWS], lo1[RS_ROWS], hi1[RS_ROWS]; int b0[RS_ROWS], b1[RS_ROWS];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            const int dy = min(dy0 + r, D.h - 1);
            const int sy = yofs[D.tab_y + dy], sy1 = min(sy + 1, S.h - 1);
            const int bb = reinterpret_cast<const int *>(ibeta)[D.tab_y + dy]; // (uniform over the wave: a scalar load)
            b0[r] = (short)(bb & 0xffff); b1[r] = bb >> 16;
            const uint8_t *row0 = src + (long)sy * S.w + sx[0], *row1 = src + (long)sy1 * S.w + sx[0];
            lo0[r] = load_u32_unaligned(row0); hi0[r] = load_u32_unaligned(row0 + 4);
            lo1[r] = load_u32_unaligned(row1); hi1[r] = load_u32_unaligned(row1 + 4);
        }
        // pixel pair (p, p+1) of column c = bytes k, k+1 of the window, k = sx[c] - sx[0] in 0..4: one v_perm_b32 spreads them into
        // two 16-bit halves, one v_dot2_u32_u16 against (alpha0, alpha1) is the horizontal interpolation
        typedef unsigned short rs_v2u16 __attribute__((ext_vector_type(2)));
        uint32_t sel[4]; rs_v2u16 ap[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { sel[c] = 0x0c010c00u + (uint32_t)(sx[c] - sx[0]) * 0x00010001u; ap[c] = __builtin_bit_cast(rs_v2u16, (uint32_t)a0[c] | ((uint32_t)a1[c] << 16)); }
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            if (r >= nrow) break;
            uint32_t packed = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_v2u16, __builtin_amdgcn_perm(hi0[r], lo0[r], sel[c])), ap[c], 0u, false);
                const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_v2u16, __builtin_amdgcn_perm(hi1[r], lo1[r], sel[c])), ap[c], 0u, false);
                const uint32_t v = ((__umul24((uint32_t)b0[r], r0 >> 4) >> 16) + (__umul24((uint32_t)b1[r], r1 >> 4) >> 16) + 2u) >> 2;
                packed |= (v & 255u) << (8 * c);
            }
            *reinterpret_cast<cs_u32_unaligned *>(base + D.off + (long)(dy0 + r) * D.w + dx) = packed;
        }
        return;
    }
    for (int r = 0; r < nrow; r++) { // row ends and degenerate tables: pixel by pixel
        const int dy = dy0 + r, sy = yofs[D.tab_y + dy], sy1 = min(sy + 1, S.h - 1);
        const int b0 = ibeta[(D.tab_y + dy) * 2], b1 = ibeta[(D.tab_y + dy) * 2 + 1];
        const uint8_t *row0 = src + (long)sy * S.w, *row1 = src + (long)sy1 * S.w;
        for (int c = 0; c < 4 && dx + c < D.w; c++) {
            const int sxc = sx[c], sx1 = min(sxc + 1, S.w - 1);
            const int r0 = row0[sxc] * a0[c] + row0[sx1] * a1[c];
            const int r1 = row1[sxc] * a0[c] + row1[sx1] * a1[c];
            base[D.off + (long)dy * D.w + dx + c] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// grid (tiles of the largest level, level, frame)
// FAST-9/16 corner score S for every pixel of every level (64x16 LDS tiles).  Only scores above tq = min(iniThFAST, minThFAST)
// are ever looked at (orb_cells masks the rest to 0, like cv::FAST's score buffer) and the map is zero-filled before the launch: a
// 9-arc of the 16-ring always contains two neighbouring compass points, so unless two neighbouring compass pixels are both darker
// (or both brighter) than the centre by more than tq the score cannot exceed tq and the min/max network is skipped.  (A
// row-streaming variant like orb_blur was tried: 715 us vs 383 us, the per-column byte extraction costs more than the tile loads.)
__global__ void __launch_bounds__(256) orb_fast_score(Pyr P, const uint8_t *pyr, uint8_t *smap) {
    int lvq = 0;
    while (lvq + 1 < P.nlevels && (int)blockIdx.x >= P.tile_off[lvq + 1]) lvq++;
    const int lv_u = __builtin_amdgcn_readfirstlane(lvq), tile = (int)blockIdx.x - P.tile_off[lv_u];
    const Lvl &L = P.l[lv_u];
    const int tiles_x = (L.w + TW - 1) / TW;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    __shared__ uint32_t g32[TH + 6][(TW + 8) / 4];
    __shared__ unsigned short s_list[TW * TH];
    __shared__ int s_n;
    uint8_t (*g)[TW + 8] = reinterpret_cast<uint8_t (*)[TW + 8]>(g32);
    const int tq = min(P.ini_th, P.min_th);
    const uint8_t *img = pyr + (long)blockIdx.z * P.frame_stride + L.off;
    if (threadIdx.x == 0) s_n = 0;
    // tile + 3-pixel halo as dwords (unaligned loads); a dword that leaves the image row falls back to guarded bytes
    for (int i = threadIdx.x; i < (TH + 6) * ((TW + 8) / 4); i += 256) {
        const int ly = i / ((TW + 8) / 4), k4 = i % ((TW + 8) / 4);
        const int X = tx0 + 4 * k4 - 3, Y = ty0 + ly - 3;
        uint32_t v = 0;
        if (Y >= 0 && Y < L.h) {
            const uint8_t *row = img + (long)Y * L.w;
            if (X >= 0 && X + 3 < L.w) v = load_u32_unaligned(row + X);
            else
                for (int c = 0; c < 4; c++) if (X + c >= 0 && X + c < L.w) v |= (uint32_t)row[X + c] << (8 * c);
        }
        g32[ly][k4] = v;
    }
    __syncthreads();
    // pass 1: the compass test on every pixel; the few that pass are appended to the workgroup's list, so that the min/max network
    // below runs with full waves instead of once per wave that holds a single candidate
    const int lx = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (the row of a wave is uniform: its address arithmetic is the scalar unit's)
    static_assert(TH == 16, "four rows per wave");
    unsigned long long cm[4];
    const int x = tx0 + lx;
    const unsigned long long lt = (1ull << lx) - 1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = wv + 4 * r, y = ty0 + ly;
        // Two neighbouring compass points both darker than v - tq  <=>  the largest of the four neighbouring pairs' minima of (v - p) exceeds tq; both brighter: the smallest
        // of the pairs' maxima is below -tq.  Written as min / max on purpose: the boolean form compiled to five branches and twenty-five mask operations per row.
        const int v = g[ly + 3][lx + 3];
        const int d0 = v - (int)g[ly + 6][lx + 3], d4 = v - (int)g[ly + 3][lx + 6], d8 = v - (int)g[ly][lx + 3], d12 = v - (int)g[ly + 3][lx];
        const int dark = max(max(min(d0, d4), min(d4, d8)), max(min(d8, d12), min(d12, d0)));
        const int bright = min(min(max(d0, d4), max(d4, d8)), min(max(d8, d12), max(d12, d0)));
        const bool inb = (unsigned)(x - 3) < (unsigned)(L.w - 6) & (unsigned)(y - 3) < (unsigned)(L.h - 6); // (levels are wider and higher than 6)
        const bool cand = inb & ((dark > tq) | (bright < -tq));
        if (!cand && x < L.w && y < L.h) smap[(long)blockIdx.z * P.frame_stride + L.off + (long)y * L.w + x] = 0; // every pixel of the map is written here or below: no fill pass in front of the launch
        cm[r] = __ballot(cand);
    }
    const int c0 = __popcll(cm[0]), c1 = __popcll(cm[1]), c2 = __popcll(cm[2]), c3 = __popcll(cm[3]);
    if (c0 + c1 + c2 + c3) { // one slot request per wave for its four rows
        int base = 0;
        if (lx == 0) base = atomicAdd(&s_n, c0 + c1 + c2 + c3);
        base = __builtin_amdgcn_readfirstlane(base);
        const int off[4] = {base, base + c0, base + c0 + c1, base + c0 + c1 + c2};
#pragma unroll
        for (int r = 0; r < 4; r++)
            if ((cm[r] >> lx) & 1ull) s_list[off[r] + __popcll(cm[r] & lt)] = (unsigned short)((wv + 4 * r) << 8 | lx);
    }
    __syncthreads();
    const int n = s_n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ly = s_list[i] >> 8, px = s_list[i] & 255;
        const int v = g[ly + 3][px + 3];
        // the min / max network of cornerScore on PAIRS: P[j] = (d[j], d[j + 8]) in the two 16-bit halves of a register (|d| <= 255), so that a packed min / max serves two
        // arcs; element j + 8 of a packed array is element j with its halves swapped, which the instruction's operand select does for nothing
        typedef short v2s __attribute__((ext_vector_type(2)));
        auto swp = [](v2s a) -> v2s { return __builtin_shufflevector(a, a, 1, 0); };
        v2s Pk[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int ra = g[ly + 3 + c_ring[k][1]][px + 3 + c_ring[k][0]], rb = g[ly + 3 + c_ring[k + 8][1]][px + 3 + c_ring[k + 8][0]];
            v2s r; r.x = (short)(v - ra); r.y = (short)(v - rb);
            Pk[k] = r;
        }
        auto at = [&](const v2s (&X)[8], int k) -> v2s { k &= 15; return k < 8 ? X[k] : swp(X[k - 8]); };
        v2s lo2[8], hi2[8], lo4[8], hi4[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { lo2[k] = __builtin_elementwise_min(Pk[k], at(Pk, k + 1)); hi2[k] = __builtin_elementwise_max(Pk[k], at(Pk, k + 1)); }
#pragma unroll
        for (int k = 0; k < 8; k++) { lo4[k] = __builtin_elementwise_min(lo2[k], at(lo2, k + 2)); hi4[k] = __builtin_elementwise_max(hi2[k], at(hi2, k + 2)); }
        v2s sdp = {-1000, -1000}, sbp = {1000, 1000};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const v2s lo9 = __builtin_elementwise_min(__builtin_elementwise_min(lo4[k], at(lo4, k + 4)), swp(Pk[k])); // min over ring[k .. k+8] and ring[k+8 .. k+16]
            const v2s hi9 = __builtin_elementwise_max(__builtin_elementwise_max(hi4[k], at(hi4, k + 4)), swp(Pk[k]));
            sdp = __builtin_elementwise_max(sdp, lo9);  // darker arc: v - p > t on the whole arc
            sbp = __builtin_elementwise_min(sbp, hi9);  // brighter arc: p - v > t  <=>  max(v - p) < -t
        }
        const int sd = max((int)sdp.x, (int)sdp.y), sb = min((int)sbp.x, (int)sbp.y);
        const int S = max(max(sd, -sb), 0);
        smap[(long)blockIdx.z * P.frame_stride + L.off + (long)(ty0 + ly) * L.w + tx0 + px] = S > tq ? (uint8_t)S : (uint8_t)0;
    }
}

// One wave per cell.  pass 0: survivors count + chosen threshold; pass 1: ordered emit at cell_base[cell] + rank.
constexpr int CELL_MASK_WORDS = 32; // 64-pixel chunks of a cell's FAST window (at most 42 x 42 pixels = 28 chunks)
// (Four cells per workgroup, one wave each, was measured: 1.23 / 1.66 ms against 0.81 / 1.15 for single-wave workgroups -- not bound by dispatch.)
constexpr int CELLS_PER_WG = 1;
__global__ void __launch_bounds__(64 * CELLS_PER_WG) orb_cells(Pyr P, const uint8_t *smap, int pass, int *cell_count, const int *cell_base, float *cand, unsigned long long *cell_mask) {
    __shared__ uint8_t s_all[CELLS_PER_WG][64][64];
    uint8_t (*s)[64] = s_all[threadIdx.x >> 6];
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int cell_in_frame = blockIdx.x * CELLS_PER_WG + (threadIdx.x >> 6);
    if (cell_in_frame >= P.cells_per_frame) return;
    int c = cell_in_frame, lv = 0;
    while (lv + 1 < P.nlevels && c >= P.l[lv + 1].cell_off) lv++;
    const Lvl &L = P.l[lv];
    c -= L.cell_off;
    if (c >= L.nCols * L.nRows) return;
    const int ci = c / L.nCols, cj = c % L.nCols;
    const int cell = f * P.cells_per_frame + cell_in_frame;
    // cell window handed to cv::FAST (:790-807); FAST itself skips a 3-px frame of that window
    const int iniX = MINB + cj * L.wCell, iniY = MINB + ci * L.hCell;
    int maxX = iniX + L.wCell + 6, maxY = iniY + L.hCell + 6;
    bool skip = iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6;
    if (maxX > L.maxBX) maxX = L.maxBX;
    if (maxY > L.maxBY) maxY = L.maxBY;
    const int ax0 = iniX + 3, ay0 = iniY + 3, aw = maxX - 3 - ax0, ah = maxY - 3 - ay0;
    if (skip || aw <= 0 || ah <= 0) { if (pass == 0 && lane == 0) cell_count[cell] = 0; return; }
    const uint8_t *S = smap + (long)f * P.frame_stride + L.off;
    if (pass == 1 && (cell_count[cell] & 0xffffff) == 0) return; // nothing to emit
    if (pass == 1 && aw * ah <= 64 * CELL_MASK_WORDS) { // the survivors of pass 0 are on record: ordered emit from the masks, the score is the map's byte
        // lane = chunk of 64 pixels: one load for all masks, ranks from a wave scan of the popcounts, then every lane walks its own few bits
        const int nch = (aw * ah + 63) >> 6;
        unsigned long long my = lane < nch ? cell_mask[(long)cell * CELL_MASK_WORDS + lane] : 0ull;
        const int cnt = __popcll(my);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; } // (chunks live in lanes 0 .. 27)
        long o = (long)cell_base[cell] + incl - cnt;
        while (my) {
            const int b = __ffsll((long long)my) - 1;
            my &= my - 1;
            const int p = lane * 64 + b, ly = p / aw, lx = p % aw;
            cand[o * 3 + 0] = (float)(ax0 + lx - MINB); // view column + j*wCell  (:821-826)
            cand[o * 3 + 1] = (float)(ay0 + ly - MINB);
            cand[o * 3 + 2] = (float)((int)S[(long)(ay0 + ly) * L.w + ax0 + lx] - 1); // cornerScore
            o++;
        }
        return;
    }
    bool any = false;
    { // the window's scores + a one-pixel frame of zeros, four pixels per lane and step (unaligned dwords; a dword that leaves the row falls back to bytes)
        const int dw = (aw + 2 + 3) >> 2; // dwords per tile row
        uint32_t *s32 = reinterpret_cast<uint32_t *>(&s[0][0]);
        for (int i = lane; i < (ah + 2) * dw; i += 64) {
            const int ly = i / dw, k4 = i - ly * dw, lx0 = 4 * k4;
            const int X0 = ax0 + lx0 - 1, Y = ay0 + ly - 1;
            uint32_t v = 0;
            if (ly >= 1 && ly <= ah) {
                const uint8_t *row = S + (long)Y * L.w;
                if (X0 >= 0 && X0 + 3 < L.w) v = load_u32_unaligned(row + X0);
                else
                    for (int c = 0; c < 4; c++) if (X0 + c >= 0 && X0 + c < L.w) v |= (uint32_t)row[X0 + c] << (8 * c);
                // columns outside [1, aw] of the tile are the frame: zero
                uint32_t keepm = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) if (lx0 + c >= 1 && lx0 + c <= aw) keepm |= 0xffu << (8 * c);
                v &= keepm;
            }
            s32[ly * 16 + k4] = v; // a tile row is 64 bytes
            any = any || v != 0;
        }
    }
    if (!__any(any)) { // the score map only holds scores above min(iniTh, minTh): an all-zero cell has no corner at either threshold
        if (pass == 0 && lane == 0) cell_count[cell] = 0 | (P.min_th << 24);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the wave's own LDS writes, read by its other lanes (in-order LDS queue: only the compiler is held back)
    __builtin_amdgcn_wave_barrier();
    int th = P.ini_th, total = 0;
    unsigned long long my_mask = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (pass == 1) { th = cell_count[cell] >> 24; }
        const int base = pass == 1 ? cell_base[cell] : 0;
        total = 0;
        for (int p0 = 0; p0 < aw * ah; p0 += 64) {
            const int p = p0 + lane;
            bool keep = false;
            int ly = 0, lx = 0, sc = 0;
            if (p < aw * ah) {
                ly = p / aw + 1; lx = p % aw + 1;
                const int v = s[ly][lx];
                if (v > th) {
                    sc = v - 1; // cornerScore
                    keep = true;
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            if (dx == 0 && dy == 0) continue;
                            const int q = s[ly + dy][lx + dx];
                            const int qs = q > th ? q - 1 : 0;
                            keep = keep && sc > qs;
                        }
                }
            }
            unsigned long long m = __ballot(keep);
            if (lane == (p0 >> 6)) my_mask = m; // chunk c's survivors in lane c (the second attempt overwrites the first)
            if (pass == 1 && keep) {
                long o = (long)base + total + __popcll(m & ((1ull << lane) - 1));
                cand[o * 3 + 0] = (float)(ax0 + lx - 1 - MINB); // view column + j*wCell  (:821-826)
                cand[o * 3 + 1] = (float)(ay0 + ly - 1 - MINB);
                cand[o * 3 + 2] = (float)sc;
            }
            total += __popcll(m);
        }
        if (pass == 1 || total > 0 || attempt == 1) break;
        th = P.min_th; // :813-817
    }
    if (pass == 0 && lane == 0) cell_count[cell] = total | (th << 24);
    if (pass == 0 && aw * ah <= 64 * CELL_MASK_WORDS && lane < CELL_MASK_WORDS) cell_mask[(long)cell * CELL_MASK_WORDS + lane] = my_mask; // one 256-byte store per cell
}

// exclusive scan of the cell counts of one (frame, level) -> cell_base (relative), level_total
__global__ void __launch_bounds__(64) orb_scan_cells(Pyr P, const int *cell_count, int *cell_base, int *level_total) {
    const int f = blockIdx.y, lv = blockIdx.x, lane = threadIdx.x;
    const Lvl &L = P.l[lv];
    const int n = L.nCols * L.nRows, c0 = f * P.cells_per_frame + L.cell_off;
    int run = 0;
    for (int b = 0; b < n; b += 64) {
        int i = b + lane;
        int v = i < n ? (cell_count[c0 + i] & 0xffffff) : 0, inc = v;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (i < n) cell_base[c0 + i] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) level_total[f * P.nlevels + lv] = run;
}
// exclusive scan over all (frame, level) totals (single wave) and rebase of the cells
__global__ void __launch_bounds__(64) orb_scan_levels(int n, const int *level_total, int *level_base) {
    const int lane = threadIdx.x;
    int run = 0;
    for (int b = 0; b < n; b += 64) {
        int i = b + lane;
        int v = i < n ? level_total[i] : 0, inc = v;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (i < n) level_base[i] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) level_base[n] = run;
}
__global__ void orb_rebase_cells(Pyr P, int n_frames, const int *level_base, int *cell_base) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * P.cells_per_frame) return;
    int f = i / P.cells_per_frame, c = i % P.cells_per_frame, lv = 0;
    while (lv + 1 < P.nlevels && c >= P.l[lv + 1].cell_off) lv++;
    cell_base[i] += level_base[f * P.nlevels + lv];
}

struct SelKP { float x, y, response; int level, frame; }; // level coordinates (minBorder already added)

__device__ __forceinline__ float fast_atan2f(float y, float x) { // cv::fastAtan2
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// IC_Angle (:74-101): one wave per keypoint, lane = patch row v = lane-15
// IC_Angle (ORBextractor.cc:76-103): half a wave per key point, a lane per patch row; the row's 31 bytes arrive as eight unaligned dwords, the sums are byte
// dot products (weights u + 15 >= 0 for the unsigned dot: sum u val = sum (u + 15) val - 15 sum val).  Integer sums: any order is exact.
__device__ __forceinline__ uint32_t low_bytes(int n) { return n >= 4 ? 0xffffffffu : (n <= 0 ? 0u : ((1u << (8 * n)) - 1)); } // bytes [0, n) set
__global__ void __launch_bounds__(256) orb_angle(Pyr P, const uint8_t *pyr, const SelKP *sel, int n, float *angle) {
    static_assert(PATCH == 31 && HALF_PATCH == 15, "a patch row is 31 bytes around the centre");
    const int k = blockIdx.x * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
    const bool live = k < n;
    const SelKP kp = sel[live ? k : n - 1];
    const Lvl &L = P.l[kp.level];
    const uint8_t *img = pyr + (long)kp.frame * P.frame_stride + L.off;
    const int cx = __float2int_rn(kp.x), cy = __float2int_rn(kp.y);
    int m10 = 0, m01 = 0;
    if (lane < PATCH) {
        const int v = lane - HALF_PATCH, d = P.umax[v < 0 ? -v : v];
        const uint8_t *row = img + (long)(cy + v) * L.w + cx - HALF_PATCH; // byte j of the row is u = j - 15
        uint32_t wd[8];
#pragma unroll
        for (int q = 0; q < 8; q++) wd[q] = load_u32_unaligned(row + 4 * q); // (byte 31, u = 16, is inside the level's 19-pixel frame and masked below)
        const int lo = HALF_PATCH - d, hi = HALF_PATCH + d; // bytes lo .. hi take part
        uint32_t s1 = 0, sw = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t m = low_bytes(hi + 1 - 4 * q) & ~low_bytes(lo - 4 * q);
            const uint32_t x = wd[q] & m;
            const uint32_t wts = (uint32_t)(4 * q) * 0x01010101u + 0x03020100u; // u + 15 = j of the four bytes
            s1 = __builtin_amdgcn_udot4(x, 0x01010101u, s1, false);
            sw = __builtin_amdgcn_udot4(x, wts, sw, false);
        }
        m10 = (int)sw - HALF_PATCH * (int)s1;
        m01 = v * (int)s1;
    }
    for (int off = 16; off > 0; off >>= 1) { m10 += __shfl_xor(m10, off); m01 += __shfl_xor(m01, off); }
    if (lane == 0 && live) angle[k] = fast_atan2f((float)m01, (float)m10);
}

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on u8 (:1078-1079): two 8-bit fixed point passes, (v + 2^15) >> 16
// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in 8-bit fixed point, row streaming: ONE WAVE owns a 256-column strip and
// BLUR_ROWS output rows, every lane four neighbouring columns.  Each input row is loaded once (rows of the next group already in
// flight), goes through an LDS line for the horizontal 7-tap pass, and the last seven horizontal results of each column stay in
// registers for the vertical pass; the four results of a lane leave as one dword store (byte stores, 64 B per wave, made this
// kernel 10x slower: 1.45 ms vs 0.13 ms without the store).  No workgroup barrier: a single wave orders its own LDS traffic.
// Same integer arithmetic as the tile version: sum_h = sum g[x+t-3] k[t]; out = (sum_t h[y+t-3] k[t] + 2^15) >> 16, clamped.
__global__ void __launch_bounds__(64) orb_blur(Pyr P, const uint8_t *pyr, uint8_t *blur) {
    int lvq = 0;
    while (lvq + 1 < P.nlevels && (int)blockIdx.x >= P.blur_off[lvq + 1]) lvq++;
    const int lv_u = __builtin_amdgcn_readfirstlane(lvq), blk = (int)blockIdx.x - P.blur_off[lv_u];
    const Lvl &L = P.l[lv_u];
    const int strips = (L.w + 255) / 256;
    const int sx = (blk % strips) * 256, y0 = (blk / strips) * BLUR_ROWS, tid = threadIdx.x;
    const int rows = min(BLUR_ROWS, L.h - y0);
    __shared__ uint32_t line32[(256 + 16) / 4]; // bytes: [0,3) left halo pad .. laid out so that column sx + c sits at byte 4 + c
    uint8_t *line = reinterpret_cast<uint8_t *>(line32);
    const uint8_t *img = pyr + (long)blockIdx.z * P.frame_stride + L.off;
    uint8_t *out = blur + (long)blockIdx.z * P.frame_stride + L.off;
    int xc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) xc[c] = reflect101(sx + 4 * tid + c, L.w);
    const int xh = tid < 3 ? reflect101(sx - 3 + tid, L.w) : reflect101(sx + 256 + (tid - 3), L.w); // halo columns, lanes 0..5
    const int x = sx + 4 * tid;
    const bool inner = x + 3 < L.w; // the lane's four columns exist (x >= 0 always): one unaligned dword load per row
    int k[7];
#pragma unroll
    for (int t = 0; t < 7; t++) k[t] = P.gk[t];
    const uint32_t kA = (uint32_t)k[0] | (uint32_t)k[1] << 8 | (uint32_t)k[2] << 16 | (uint32_t)k[3] << 24, kB = (uint32_t)k[4] | (uint32_t)k[5] << 8 | (uint32_t)k[6] << 16;
    int ring[4][7];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int t = 0; t < 7; t++) ring[c][t] = 0;
    constexpr int G = 4; // input rows in flight per lane
    const int total = rows + 6;
    uint32_t cur[G], nxt[G]; uint8_t curh[G], nxth[G];
    auto fetch = [&](int r0, uint32_t (&a)[G], uint8_t (&hh)[G]) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            a[u] = 0; hh[u] = 0;
            if (r0 + u < total) {
                const uint8_t *row = img + (long)reflect101(y0 + r0 + u - 3, L.h) * L.w;
                a[u] = inner ? load_u32_unaligned(row + x) : ((uint32_t)row[xc[0]] | ((uint32_t)row[xc[1]] << 8) | ((uint32_t)row[xc[2]] << 16) | ((uint32_t)row[xc[3]] << 24));
                if (tid < 6) hh[u] = row[xh];
            }
        }
    };
    fetch(0, cur, curh);
    for (int r0 = 0; r0 < total; r0 += G) {
        fetch(r0 + G, nxt, nxth);
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int r = r0 + u;
            if (r < total) {
                line32[1 + tid] = cur[u];                          // columns sx + 4 tid .. + 3 at bytes 4 + 4 tid ..
                if (tid < 3) line[1 + tid] = curh[u];              // sx - 3 .. sx - 1 at bytes 1..3
                else if (tid < 6) line[4 + 256 + (tid - 3)] = curh[u]; // sx + 256 .. + 2
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t w0 = line32[tid], w1 = line32[tid + 1], w2 = line32[tid + 2];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next row overwrites the line
                // column sx + 4 tid + c: taps at bytes c + 1 .. c + 7 of the 12-byte window w0 w1 w2.  The 8-bit weights ride in two
                // dwords (kA = k0..k3, kB = k4..k6, 0): two byte-aligns and two v_dot4_u32_u8 per column; the vertical taps are
                // 24-bit multiply-adds (h < 2^16, k < 2^8).  A plain int product compiles to v_mul_lo_u32, a quarter-rate instruction
                // -- 52 of them per row made this kernel four times slower than its memory traffic.
                uint32_t packed = 0;
                const uint32_t lo4[4] = {win4<1>(w0, w1, w2), win4<2>(w0, w1, w2), win4<3>(w0, w1, w2), win4<4>(w0, w1, w2)};
                const uint32_t hi4[4] = {win4<5>(w0, w1, w2), win4<6>(w0, w1, w2), win4<7>(w0, w1, w2), win4<8>(w0, w1, w2)};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t h = __builtin_amdgcn_udot4(lo4[c], kA, __builtin_amdgcn_udot4(hi4[c], kB, 0u, false), false);
#pragma unroll
                    for (int t = 0; t < 6; t++) ring[c][t] = ring[c][t + 1];
                    ring[c][6] = (int)h;
                    uint32_t sum = 1u << 15;
#pragma unroll
                    for (int t = 0; t < 7; t++) sum = __umul24((uint32_t)ring[c][t], (uint32_t)k[t]) + sum;
                    const uint32_t v = sum >> 16; // <= 255: the weights sum to 256 in each direction
                    packed |= (v > 255u ? 255u : v) << (8 * c);
                }
                if (r >= 6 && x < L.w) {
                    uint8_t *dst = out + (long)(y0 + r - 6) * L.w + x;
                    if (x + 3 < L.w) *reinterpret_cast<uint32_t *>(dst) = packed; // may be unaligned: row starts are not multiples of 4
                    else for (int c = 0; c < 4 && x + c < L.w; c++) dst[c] = (uint8_t)(packed >> (8 * c));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; u++) { cur[u] = nxt[u]; curh[u] = nxth[u]; }
    }
}

// float sin/cos of the keypoint angle: evaluated in double with +,-,* only and rounded to float, so that host oracle
// and device agree bit for bit
__device__ __forceinline__ void sincos_f(float angle, float &so, float &co) {
    const double x = (double)angle;
    const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HI = 1.57079632679489655800, PIO2_LO = 6.12323399573676603587e-17;
    const double kd = floor(x * TWO_OVER_PI + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double r2 = r * r;
    const double sp = -1.0 / 6.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 5040.0 + r2 * (1.0 / 362880.0 + r2 * (-1.0 / 39916800.0 + r2 * (1.0 / 6227020800.0 + r2 * (-1.0 / 1307674368000.0))))));
    const double cp = -1.0 / 2.0 + r2 * (1.0 / 24.0 + r2 * (-1.0 / 720.0 + r2 * (1.0 / 40320.0 + r2 * (-1.0 / 3628800.0 + r2 * (1.0 / 479001600.0 + r2 * (-1.0 / 87178291200.0 + r2 * (1.0 / 20922789888000.0)))))));
    const double s = r + r * r2 * sp;
    const double c = 1.0 + r2 * cp;
    double ss, cc;
    switch (k & 3) {
    case 0: ss = s; cc = c; break;
    case 1: ss = c; cc = -s; break;
    case 2: ss = -s; cc = -c; break;
    default: ss = -c; cc = s; break;
    }
    so = (float)ss; co = (float)cc;
}

// computeOrbDescriptor (:104-150) + keypoint finalisation (:1085-1096).  One wave per keypoint; lane l evaluates tests
// 64r + l (r = 0..3), each ballot is 8 descriptor bytes.
__global__ void __launch_bounds__(256) orb_desc(Pyr P, const uint8_t *blur, const SelKP *sel, const float *angle, int n, cs_keypoint *kps,
                                                unsigned long long *desc) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n) return;
    const SelKP kp = sel[k];
    const Lvl &L = P.l[kp.level];
    const uint8_t *img = blur + (long)kp.frame * P.frame_stride + L.off;
    const float ang = angle[k];
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    sincos_f(ang * factorPI, b, a);
    const long cidx = (long)__float2int_rn(kp.y) * L.w + __float2int_rn(kp.x);
    const long last = (long)L.w * L.h - 1;
    // The 512 test points lie within 18.4 pixels of the centre (bit_pattern_31_), so every rotated, rounded offset is inside a 37 x 37 window: the wave stages
    // that window in LDS with coalesced dword loads -- byte (ry, rx) = img[clamp(cidx + ry * w + rx)], the flat, clamped addressing of the tests themselves --
    // and gathers from there (eight 64-line gathers per key point through the texture path were this kernel's time).
    constexpr int DR = 18, DW = 10; // window radius; dwords per staged row (40 bytes for 37)
    __shared__ __attribute__((aligned(4))) uint8_t s_patch[4][2 * DR + 1][4 * DW];
    uint8_t (*pt)[4 * DW] = s_patch[threadIdx.x >> 6];
    for (int j = lane; j < (2 * DR + 1) * DW; j += 64) {
        const int row = j / DW, q = j - row * DW;
        const long o0 = cidx + (long)(row - DR) * L.w - DR + 4 * q;
        uint32_t v;
        if (o0 >= 0 && o0 + 3 <= last) v = load_u32_unaligned(img + o0);
        else {
            v = 0;
            for (int c = 0; c < 4; c++) { long o = o0 + c; o = o < 0 ? 0 : (o > last ? last : o); v |= (uint32_t)img[o] << (8 * c); } // unchecked addressing of the reference, clamped
        }
        reinterpret_cast<uint32_t *>(&pt[row][0])[q] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // the wave's own LDS writes, read by its other lanes
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 4; r++) {
        const int t = r * 64 + lane;
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float px = (float)d_pattern[4 * t + 2 * e], py = (float)d_pattern[4 * t + 2 * e + 1];
            val[e] = pt[__float2int_rn(px * b + py * a) + DR][__float2int_rn(px * a - py * b) + DR];
        }
        unsigned long long m = __ballot(val[0] < val[1]);
        if (lane == 0) desc[(long)k * 4 + r] = m;
    }
    if (lane == 0) {
        cs_keypoint o;
        o.x = kp.x; o.y = kp.y;
        if (kp.level != 0) { o.x = kp.x * L.scale; o.y = kp.y * L.scale; }
        o.size = (float)L.patch; o.angle = ang; o.response = kp.response; o.octave = kp.level; o.class_id = -1;
        kps[k] = o;
    }
}

static inline int h_cvRound(double v) { return (int)std::lrint(v); }
static inline int h_cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int h_cvCeil(double v) { int i = (int)v; return i + (i < v); }

// ------------------------------------------------------------------------------------------------ keypoint distribution on the device
// ORBextractor::DistributeOctTree / ExtractorNode::DivideNode (reference ORBextractor.cc:483-538, :540-763), one workgroup per
// (frame, level).  Same selection and the same output order as the host version (orb_quadtree.h), re-expressed without a linked
// list: a pass of the first phase replaces the list by [children of the last expandable node, n4..n1, ..., children of the first
// expandable node] ++ [nodes that are not expandable, in order] (validated against the sequential list surgery on random inputs);
// the second phase (largest nodes first until the quota is reached) is sequential by nature and runs in wave 0, with the point
// partition of every split done by the wave.  Node records live in global memory (int4: x0|y0<<16, x1|y1<<16, begin, end), the
// lists in LDS.  Equal-size ties of the second phase are broken by creation order.
struct QtLevel { int N, minX, maxX, minY, maxY, node_off, capn, slot_off; };
struct QtParams { QtLevel l[MAXL]; int nlevels, nodes_per_frame, slots_per_frame; };

__device__ __forceinline__ int qt_quadrant(const float *K, int idx, int mx, int my) {
    const float x = K[idx * 3], y = K[idx * 3 + 1];
    return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}
// stable 4-way partition of perm[begin, end) by one wave; c[4] = sizes.  cls(idx) -> 0..3
template <class F> __device__ __forceinline__ void qt_wave_partition(int *perm, int *tmp, int begin, int end, F cls, int (&c)[4]) {
    const int lane = threadIdx.x & 63;
    int cnt[4] = {0, 0, 0, 0};
    for (int p0 = begin; p0 < end; p0 += 64) { // counts
        const int p = p0 + lane;
        const int q = p < end ? cls(perm[p]) : -1;
#pragma unroll
        for (int k = 0; k < 4; k++) cnt[k] += __popcll(__ballot(q == k));
    }
    int pos[4] = {begin, begin + cnt[0], begin + cnt[0] + cnt[1], begin + cnt[0] + cnt[1] + cnt[2]};
    for (int p0 = begin; p0 < end; p0 += 64) {
        const int p = p0 + lane;
        const int v = p < end ? perm[p] : 0;
        const int q = p < end ? cls(v) : -1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(q == k);
            if (q == k) tmp[pos[k] + __popcll(m & ((1ull << lane) - 1))] = v;
            pos[k] += __popcll(m);
        }
    }
    __threadfence_block();
    for (int p = begin + lane; p < end; p += 64) perm[p] = tmp[p];
    __threadfence_block();
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = cnt[k];
}
__device__ __forceinline__ void qt_child_box(const int4 nd, int k, int &x0, int &y0, int &x1, int &y1) { // DivideNode :483-538
    const int px0 = nd.x & 0xffff, py0 = nd.x >> 16, px1 = nd.y & 0xffff, py1 = nd.y >> 16;
    const int halfX = (int)ceilf(static_cast<float>(px1 - px0) / 2), halfY = (int)ceilf(static_cast<float>(py1 - py0) / 2);
    const int mx = px0 + halfX, my = py0 + halfY;
    x0 = (k & 1) ? mx : px0; x1 = (k & 1) ? px1 : mx;
    y0 = (k & 2) ? my : py0; y1 = (k & 2) ? py1 : my;
}
__global__ void __launch_bounds__(256) orb_quadtree(QtParams Q, const int *level_base, const float *cand, int *perm_all, int *tmp_all, int4 *nodes_all, SelKP *slots, int *slot_cnt,
                                                   int *status) {
    extern __shared__ int qsh[];
    // Frame in x, level in y.  The dispatcher deals consecutive workgroups to the eight XCDs in turn: with the eight LEVELS in x every level-0 workgroup -- the largest
    // tree of a frame -- landed on the same XCD and the launch took as long as that XCD needed (2.29 ms per 1 024 frames against 0.93 with the frames in x; 4.98 against
    // 1.55 ms per 512 frames of 1241 x 376 with 2 000 features).  One wave per small level was measured on top: slower.
    const int lv = blockIdx.y, f = blockIdx.x, fl = f * Q.nlevels + lv, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const QtLevel L = Q.l[lv];
    const int b0 = level_base[fl], n = level_base[fl + 1] - b0;
    const float *K = cand + (long)b0 * 3;
    int *perm = perm_all + b0, *tmp = tmp_all + b0;
    int4 *nodes = nodes_all + (long)f * Q.nodes_per_frame + L.node_off;
    SelKP *out = slots + (long)f * Q.slots_per_frame + L.slot_off;
    const int N = L.N, CAPL = 4 * N + 16, CAPV = N + 8, CAPN = L.capn;
    // LDS carve-up (ints)
    short *listA = (short *)qsh, *listB = listA + CAPL, *Elist = listB + CAPL, *front = Elist + CAPL;          // 4 x CAPL shorts
    unsigned long long *cnt64 = (unsigned long long *)(front + CAPL);                                            // CAPV
    unsigned long long *vkA = cnt64 + CAPV, *vkB = vkA + CAPV;                                                    // 2 x CAPV keys (size << 32 | id)
    int *P = (int *)(vkB + CAPV);                                                                                // CAPV + 1 prefix of child counts
    int *Pv = P + CAPV + 1;                                                                                      // CAPV + 1 prefix of children with > 1 point
    unsigned *dead = (unsigned *)(Pv + CAPV + 1);                                                                // (CAPN + 31) / 32
    short *order = (short *)(dead + (CAPN + 31) / 32);                                                           // CAPV
    __shared__ int s_i[16];
    __shared__ int s_w[8];
    if (tid == 0) slot_cnt[fl] = 0;
    const int nIni = (int)roundf(static_cast<float>(L.maxX - L.minX) / (L.maxY - L.minY));
    if (n <= 0 || nIni < 1) return;
    if (nIni > 4) { if (tid == 0) *status = 2; return; } // host path handles panoramas
    const float hX = static_cast<float>(L.maxX - L.minX) / nIni;
    for (int i = tid; i < n; i += 256) perm[i] = i;
    for (int i = tid; i < (CAPN + 31) / 32; i += 256) dead[i] = 0;
    __syncthreads();
    int n_nodes = 0, n_list = 0; // uniform across the workgroup (recomputed identically by every thread from LDS broadcasts)
    short *list = listA, *nlist = listB;
    // ---- root nodes (:551-586): stable bucket by x / hX, wave 0
    if (wave == 0) {
        int c[4];
        qt_wave_partition(perm, tmp, 0, n, [&](int idx) { int b = (int)(K[idx * 3] / hX); return b >= nIni ? nIni - 1 : b; }, c);
        if (lane == 0) {
            int beg = 0, cntn = 0;
            for (int b = 0; b < nIni; b++) {
                if (c[b] > 0) {
                    const int x0 = (int)(hX * static_cast<float>(b)), x1 = (int)(hX * static_cast<float>(b + 1));
                    nodes[cntn] = make_int4(x0 | (0 << 16), x1 | ((L.maxY - L.minY) << 16), beg, beg + c[b]);
                    listA[cntn] = (short)cntn;
                    cntn++;
                }
                beg += c[b];
            }
            s_i[0] = cntn;
        }
    }
    __threadfence_block();
    __syncthreads();
    n_nodes = s_i[0]; n_list = n_nodes;
    bool finish = false, overflow = false;
    int n_vk = 0; // entries in vkA
    while (!finish) {
        const int prev_size = n_list;
        // ---- a. expandable nodes (more than one point) keep list order in Elist; the others go to the tail of the new list later
        //      ordered compaction by the whole workgroup
        int nE = 0, nKeepBefore = 0;
        {
            int baseE = 0;
            for (int i0 = 0; i0 < n_list; i0 += 256) {
                const int i = i0 + tid;
                int id = -1; bool ex = false;
                if (i < n_list) { id = list[i]; const int4 nd = nodes[id]; ex = (nd.w - nd.z) > 1; }
                const unsigned long long m = __ballot(ex);
                if (lane == 0) s_w[wave] = __popcll(m);
                __syncthreads();
                int wb = 0;
                for (int w = 0; w < wave; w++) wb += s_w[w];
                const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
                if (ex) Elist[baseE + wb + __popcll(m & ((1ull << lane) - 1))] = (short)id;
                baseE += tot;
                __syncthreads();
            }
            nE = baseE;
            (void)nKeepBefore;
        }
        if (nE >= CAPV) { overflow = true; break; }
        // ---- b. split the points of every expandable node (one wave per node), child sizes packed 4 x 16 bit
        for (int e = wave; e < nE; e += 4) {
            const int4 nd = nodes[Elist[e]];
            int x0, y0, x1, y1;
            qt_child_box(nd, 0, x0, y0, x1, y1); // (x1, y1) of child 0 = the split point (mx, my)
            const int mx = x1, my = y1;
            int c[4];
            qt_wave_partition(perm, tmp, nd.z, nd.w, [&](int idx) { return qt_quadrant(K, idx, mx, my); }, c);
            if (lane == 0) cnt64[e] = (unsigned long long)c[0] | ((unsigned long long)c[1] << 16) | ((unsigned long long)c[2] << 32) | ((unsigned long long)c[3] << 48);
        }
        __threadfence_block();
        __syncthreads();
        // ---- c. prefix sums over the expandable nodes (children, children with more than one point): wave 0, 64 at a time
        if (wave == 0) {
            int runK = 0, runV = 0;
            for (int e0 = 0; e0 < nE; e0 += 64) {
                const int e = e0 + lane;
                int k = 0, v = 0;
                if (e < nE) { const unsigned long long c = cnt64[e]; for (int q = 0; q < 4; q++) { const int s = (int)((c >> (16 * q)) & 0xffff); k += s > 0; v += s > 1; } }
                int ik = k, iv = v;
                for (int d = 1; d < 64; d <<= 1) { const int a = __shfl_up(ik, d), b2 = __shfl_up(iv, d); if (lane >= d) { ik += a; iv += b2; } }
                if (e < nE) { P[e] = runK + ik - k; Pv[e] = runV + iv - v; }
                runK += __shfl(ik, 63); runV += __shfl(iv, 63);
            }
            if (lane == 0) { P[nE] = runK; Pv[nE] = runV; }
        }
        __syncthreads();
        const int T = P[nE], nV = Pv[nE];
        const int n_keep = n_list - nE;
        if (n_nodes + T > CAPN || T + n_keep > CAPL) { overflow = true; break; }
        // ---- d. child records, new list, (size, id) entries in creation order
        for (int e = tid; e < nE; e += 256) {
            const int4 nd = nodes[Elist[e]];
            const unsigned long long c = cnt64[e];
            int beg = nd.z, j = 0, jv = 0;
            int kk = 0;
            for (int q = 0; q < 4; q++) kk += ((c >> (16 * q)) & 0xffff) > 0;
            for (int q = 0; q < 4; q++) {
                const int s = (int)((c >> (16 * q)) & 0xffff);
                if (s > 0) {
                    int x0, y0, x1, y1;
                    qt_child_box(nd, q, x0, y0, x1, y1);
                    const int id = n_nodes + P[e] + j;
                    nodes[id] = make_int4(x0 | (y0 << 16), x1 | (y1 << 16), beg, beg + s);
                    nlist[(T - P[e] - kk) + (kk - 1 - j)] = (short)id;
                    if (s > 1) { if (Pv[e] + jv < CAPV) vkA[Pv[e] + jv] = ((unsigned long long)s << 32) | (unsigned)id; jv++; }
                    j++;
                }
                beg += s;
            }
        }
        // non-expandable nodes keep their order behind the children: ordered compaction again
        {
            int baseK = T;
            for (int i0 = 0; i0 < n_list; i0 += 256) {
                const int i = i0 + tid;
                int id = -1; bool kp = false;
                if (i < n_list) { id = list[i]; const int4 nd = nodes[id]; kp = (nd.w - nd.z) <= 1; }
                const unsigned long long m = __ballot(kp);
                if (lane == 0) s_w[wave] = __popcll(m);
                __syncthreads();
                int wb = 0;
                for (int w = 0; w < wave; w++) wb += s_w[w];
                const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
                if (kp) nlist[baseK + wb + __popcll(m & ((1ull << lane) - 1))] = (short)id;
                baseK += tot;
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
        n_nodes += T;
        n_list = T + n_keep;
        n_vk = nV < CAPV ? nV : CAPV;
        { short *t = list; list = nlist; nlist = t; }
        if (n_list >= N || n_list == prev_size) finish = true;
        else if (n_list + nV * 3 > N) {
            if (nV >= CAPV) { overflow = true; break; }
            // ---- second phase (:676-741), wave 0: split the largest nodes first until the quota is reached
            if (wave == 0) {
                int cur = n_list, nfront = 0, nn = n_nodes, nvk = n_vk;
                unsigned long long *vk = vkA, *vn = vkB;
                bool fin = false, ovf = false;
                while (!fin && !ovf) {
                    const int ps = cur, m = nvk;
                    // rank sort ascending by (size, id): order[rank] = entry
                    for (int i = lane; i < m; i += 64) {
                        const unsigned long long ki = vk[i];
                        int rnk = 0;
                        for (int j = 0; j < m; j++) rnk += vk[j] < ki;
                        order[rnk] = (short)i;
                    }
                    __threadfence_block();
                    int nvn = 0;
                    for (int j = m - 1; j >= 0; j--) {
                        const int id = (int)(vk[order[j]] & 0xffffffffu);
                        const int4 nd = nodes[id];
                        int x0, y0, x1, y1;
                        qt_child_box(nd, 0, x0, y0, x1, y1);
                        const int mx = x1, my = y1;
                        int c[4];
                        qt_wave_partition(perm, tmp, nd.z, nd.w, [&](int idx) { return qt_quadrant(K, idx, mx, my); }, c);
                        int kk = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
                        if (nn + kk > CAPN || nfront + kk > CAPL || nvn + kk > CAPV) { ovf = true; break; }
                        if (lane == 0) {
                            int beg = nd.z, jn = 0;
                            for (int q = 0; q < 4; q++) {
                                if (c[q] > 0) {
                                    qt_child_box(nd, q, x0, y0, x1, y1);
                                    nodes[nn + jn] = make_int4(x0 | (y0 << 16), x1 | (y1 << 16), beg, beg + c[q]);
                                    front[nfront + jn] = (short)(nn + jn);
                                    jn++;
                                }
                                beg += c[q];
                            }
                            dead[id >> 5] |= 1u << (id & 31);
                        }
                        { int jn = 0; for (int q = 0; q < 4; q++) if (c[q] > 0) { if (c[q] > 1) { if (lane == 0) vn[nvn] = ((unsigned long long)c[q] << 32) | (unsigned)(nn + jn); nvn++; } jn++; } }
                        nn += kk; nfront += kk; cur += kk - 1;
                        __threadfence_block();
                        if (cur >= N) break;
                    }
                    if (cur >= N || cur == ps) fin = true;
                    { unsigned long long *t = vk; vk = vn; vn = t; }
                    nvk = nvn;
                }
                // final list: pushed children newest first, then the survivors of the first-phase list
                int o = 0;
                if (!ovf) {
                    for (int k0 = nfront - 1; k0 >= 0; k0 -= 64) {
                        const int k = k0 - lane;
                        int id = -1; bool al = false;
                        if (k >= 0) { id = front[k]; al = !((dead[id >> 5] >> (id & 31)) & 1u); }
                        const unsigned long long mm = __ballot(al);
                        if (al) nlist[o + __popcll(mm & ((1ull << lane) - 1))] = (short)id;
                        o += __popcll(mm);
                    }
                    for (int i0 = 0; i0 < n_list; i0 += 64) {
                        const int i = i0 + lane;
                        int id = -1; bool al = false;
                        if (i < n_list) { id = list[i]; al = !((dead[id >> 5] >> (id & 31)) & 1u); }
                        const unsigned long long mm = __ballot(al);
                        if (al) nlist[o + __popcll(mm & ((1ull << lane) - 1))] = (short)id;
                        o += __popcll(mm);
                    }
                }
                if (lane == 0) { s_i[1] = o; s_i[2] = ovf ? 1 : 0; }
            }
            __threadfence_block();
            __syncthreads();
            if (s_i[2]) { overflow = true; break; }
            n_list = s_i[1];
            { short *t = list; list = nlist; nlist = t; }
            finish = true;
        }
    }
    if (overflow) { if (tid == 0) { *status = 1; slot_cnt[fl] = 0; } return; }
    // ---- best response per node, first wins ties (:744-760); keypoint in level coordinates (:841-844)
    for (int i = tid; i < n_list; i += 256) {
        const int4 nd = nodes[list[i]];
        int best = perm[nd.z];
        float mxr = K[best * 3 + 2];
        for (int p = nd.z + 1; p < nd.w; p++) { const int idx = perm[p]; const float rr = K[idx * 3 + 2]; if (rr > mxr) { best = idx; mxr = rr; } }
        out[i] = SelKP{K[best * 3] + L.minX, K[best * 3 + 1] + L.minY, K[best * 3 + 2], lv, f};
    }
    if (tid == 0) slot_cnt[fl] = n_list;
}
// slotted selections -> contiguous level-major list per frame
__global__ void __launch_bounds__(256) orb_compact_sel(QtParams Q, const SelKP *slots, const int *slot_cnt, const int *sel_base, SelKP *sel) {
    const int lv = blockIdx.x, f = blockIdx.y, fl = f * Q.nlevels + lv;
    const SelKP *src = slots + (long)f * Q.slots_per_frame + Q.l[lv].slot_off;
    const int n = slot_cnt[fl], b = sel_base[fl];
    for (int i = threadIdx.x; i < n; i += 256) sel[b + i] = src[i];
}

} // namespace

struct cs_orb {
    int nfeatures = 0, nlevels = 0, W = 0, H = 0, max_frames = 0, n_frames = 0;
    float scaleFactor = 0;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel;
    Pyr P{};
    int max_tiles = 0, max_blur_blocks = 0;
    long cand_cap = 0, cand_alloc = 0; // candidates of all frames: the worst case, and what d_cand / d_qperm / d_qtmp hold room for (grown to a batch's count, cs_orb_run)
    // device
    uint8_t *d_pyr = nullptr, *d_smap = nullptr, *d_blur = nullptr;
    unsigned long long *d_cell_mask = nullptr; // per cell: the NMS survivors of orb_cells' counting pass, one bit per pixel of the cell's window
    int *d_xofs = nullptr, *d_yofs = nullptr, *d_cell_count = nullptr, *d_cell_base = nullptr, *d_level_total = nullptr, *d_level_base = nullptr;
    short *d_ialpha = nullptr, *d_ibeta = nullptr;
    float *d_cand = nullptr, *d_angle = nullptr;
    SelKP *d_sel = nullptr;
    // device keypoint distribution (orb_quadtree): scratch + slotted results
    QtParams Q{}; bool gpu_quadtree = false; size_t qt_lds = 0;
    int *d_qperm = nullptr, *d_qtmp = nullptr, *d_slot_cnt = nullptr, *d_sel_base = nullptr, *d_qstatus = nullptr; int4 *d_qnodes = nullptr; SelKP *d_slots = nullptr;
    std::vector<int> sel_base;
    bool cand_valid = false; // host copy of the candidates is current (device quadtree skips the copy)
    cs_keypoint *d_kps = nullptr;
    unsigned long long *d_desc = nullptr;
    long sel_cap = 0;
    // host results of the last run
    std::vector<int> level_base;          // n_frames*nlevels + 1
    std::vector<cs_orb_host::Cand> cand;  // compact candidates
    std::vector<SelKP> sel;
    std::vector<int> frame_first;         // n_frames + 1 offsets into sel
};

int cs_orb_device_frame(const cs_orb *e, int frame, const cs_keypoint **d_kps, const unsigned long long **d_desc, int *n) {
    if (!e || frame < 0 || frame >= e->n_frames || (int)e->frame_first.size() <= frame + 1) return CS_ERR_BAD_ARG;
    const int b0 = e->frame_first[frame];
    *d_kps = e->d_kps + b0; *d_desc = e->d_desc + (size_t)b0 * 4; *n = e->frame_first[frame + 1] - b0;
    return CS_OK;
}

extern "C" {

void cs_orb_destroy(cs_ctx *ctx, cs_orb *e) {
    if (!e) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {e->d_cell_mask, e->d_pyr, e->d_smap, e->d_blur, e->d_xofs, e->d_yofs, e->d_cell_count, e->d_cell_base, e->d_level_total, e->d_level_base,
                    e->d_ialpha, e->d_ibeta, e->d_cand, e->d_angle, e->d_sel, e->d_kps, e->d_desc, e->d_qperm, e->d_qtmp, e->d_slot_cnt, e->d_sel_base, e->d_qstatus,
                    e->d_qnodes, e->d_slots};
    for (void *p : ptrs) if (p) hipFree(p);
    delete e;
}

int cs_orb_create(cs_ctx *ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width, int height,
                  int max_frames, cs_orb **out) {
    if (!ctx || !out || nfeatures < 1 || nlevels < 1 || nlevels > MAXL || !(scaleFactor > 1.0f) || width < 64 || height < 64 || max_frames < 1 ||
        iniThFAST < 0 || iniThFAST > 255 || minThFAST < 0 || minThFAST > 255)
        return CS_ERR_BAD_ARG;
    *out = nullptr;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_orb *e = new (std::nothrow) cs_orb();
    if (!e) return CS_ERR_NOMEM;
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->W = width; e->H = height; e->max_frames = max_frames; e->scaleFactor = scaleFactor;
    // constructor tables, ORBextractor.cc:412-471
    e->mvScaleFactor.resize(nlevels); e->mvLevelSigma2.resize(nlevels); e->mvInvScaleFactor.resize(nlevels); e->mvInvLevelSigma2.resize(nlevels);
    e->mvScaleFactor[0] = 1.0f; e->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * scaleFactor; e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i]; }
    for (int i = 0; i < nlevels; i++) { e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i]; e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i]; }
    e->mnFeaturesPerLevel.resize(nlevels);
    {
        float factor = 1.0f / scaleFactor;
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) { e->mnFeaturesPerLevel[level] = h_cvRound(nDesired); sum += e->mnFeaturesPerLevel[level]; nDesired *= factor; }
        e->mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    }
    Pyr &P = e->P;
    P.nlevels = nlevels; P.ini_th = iniThFAST; P.min_th = minThFAST;
    {
        int v, v0, vmax = h_cvFloor(HALF_PATCH * std::sqrt(2.f) / 2 + 1), vmin = h_cvCeil(HALF_PATCH * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v <= vmax; ++v) P.umax[v] = h_cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (P.umax[v0] == P.umax[v0 + 1]) ++v0; P.umax[v] = v0; ++v0; }
    }
    { // cv::getGaussianKernel(7, 2, CV_32F) scaled to 8-bit fixed point
        float cf[7];
        double scale2X = -0.5 / (2.0 * 2.0), sum = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)std::exp(scale2X * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); P.gk[i] = h_cvRound(cf[i] * 256.f); }
    }
    std::vector<int> xofs, yofs;
    std::vector<short> ialpha, ibeta;
    long off = 0;
    int cells = 0;
    e->max_tiles = 0;
    long cand_per_frame = 0;
    auto sat_short = [](float v) { int i = h_cvRound(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); };
    for (int l = 0; l < nlevels; l++) {
        Lvl &L = P.l[l];
        float scale = e->mvInvScaleFactor[l];
        L.w = h_cvRound((float)width * scale); L.h = h_cvRound((float)height * scale); // :1105-1106
        L.off = off; off += ((long)L.w * L.h + 63) / 64 * 64;
        L.maxBX = L.w - EDGE_T + 3; L.maxBY = L.h - EDGE_T + 3; // :778-779
        const float fw = (float)(L.maxBX - MINB), fh = (float)(L.maxBY - MINB);
        L.nCols = (int)(fw / 30.f); L.nRows = (int)(fh / 30.f);
        if (L.nCols < 1 || L.nRows < 1) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 1; }
        else { L.wCell = (int)std::ceil(fw / L.nCols); L.hCell = (int)std::ceil(fh / L.nRows); }
        if (L.wCell > 58 || L.hCell > 58) { delete e; return CS_ERR_CAPACITY; }
        L.cell_off = cells; cells += L.nCols * L.nRows;
        L.scale = e->mvScaleFactor[l]; L.patch = (int)(PATCH * e->mvScaleFactor[l]);
        L.tab_x = (int)xofs.size(); L.tab_y = (int)yofs.size();
        if (l > 0) { // cv::resize INTER_LINEAR coefficient tables (imgwarp.cpp), source = level l-1
            const int sw = P.l[l - 1].w, sh = P.l[l - 1].h;
            const double scale_x = 1. / ((double)L.w / sw), scale_y = 1. / ((double)L.h / sh);
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = h_cvFloor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
                xofs.push_back(sx); ialpha.push_back(sat_short((1.f - fx) * 2048)); ialpha.push_back(sat_short(fx * 2048));
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = h_cvFloor(fy);
                fy -= sy;
                if (sy < 0) { fy = 0; sy = 0; }
                if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
                yofs.push_back(sy); ibeta.push_back(sat_short((1.f - fy) * 2048)); ibeta.push_back(sat_short(fy * 2048));
            }
        }
        e->max_tiles = std::max(e->max_tiles, ((L.w + TW - 1) / TW) * ((L.h + TH - 1) / TH));
        e->max_blur_blocks = std::max(e->max_blur_blocks, ((L.w + 255) / 256) * ((L.h + BLUR_ROWS - 1) / BLUR_ROWS));
        P.tile_off[l + 1] = P.tile_off[l] + ((L.w + TW - 1) / TW) * ((L.h + TH - 1) / TH);
        P.blur_off[l + 1] = P.blur_off[l] + ((L.w + 255) / 256) * ((L.h + BLUR_ROWS - 1) / BLUR_ROWS);
        cand_per_frame += (long)((L.w + 1) / 2) * ((L.h + 1) / 2);
    }
    P.frame_stride = off; P.cells_per_frame = cells;
    e->cand_cap = cand_per_frame * max_frames;
    e->sel_cap = (long)(nfeatures + 4 * nlevels + 64) * max_frames * 2;
#define A_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_orb_destroy(ctx, e); return r__; } } while (0)
    A_(cs_dalloc(ctx, &e->d_pyr, (size_t)off * max_frames));
    A_(cs_dalloc(ctx, &e->d_smap, (size_t)off * max_frames));
    A_(cs_dalloc(ctx, &e->d_blur, (size_t)off * max_frames));
    A_(cs_dalloc(ctx, &e->d_xofs, xofs.size()));
    A_(cs_dalloc(ctx, &e->d_yofs, yofs.size()));
    A_(cs_dalloc(ctx, &e->d_ialpha, ialpha.size()));
    A_(cs_dalloc(ctx, &e->d_ibeta, ibeta.size()));
    A_(cs_dalloc(ctx, &e->d_cell_count, (size_t)cells * max_frames));
    A_(cs_dalloc(ctx, &e->d_cell_mask, (size_t)cells * max_frames * CELL_MASK_WORDS));
    A_(cs_dalloc(ctx, &e->d_cell_base, (size_t)cells * max_frames));
    A_(cs_dalloc(ctx, &e->d_level_total, (size_t)nlevels * max_frames));
    A_(cs_dalloc(ctx, &e->d_level_base, (size_t)nlevels * max_frames + 1));
    // (d_cand and the quadtree's two permutation arrays are sized by the candidates a batch really has, known before they are written: cs_orb_run.  Their worst case,
    // a corner in every other pixel of every level, would be 4.9 MB per frame)
    A_(cs_dalloc(ctx, &e->d_sel, (size_t)e->sel_cap));
    A_(cs_dalloc(ctx, &e->d_angle, (size_t)e->sel_cap));
    A_(cs_dalloc(ctx, &e->d_kps, (size_t)e->sel_cap));
    A_(cs_dalloc(ctx, &e->d_desc, (size_t)e->sel_cap * 4));
    { // device quadtree plan: per level quota, node pool and result slots; LDS need of the largest level
        QtParams &Q = e->Q;
        Q.nlevels = nlevels;
        int node_off = 0, slot_off = 0, maxN = 0;
        bool ok = true;
        for (int l = 0; l < nlevels; l++) {
            const Lvl &L = e->P.l[l];
            QtLevel &q = Q.l[l];
            q.N = e->mnFeaturesPerLevel[l]; q.minX = MINB; q.maxX = L.maxBX; q.minY = MINB; q.maxY = L.maxBY;
            q.capn = 12 * std::max(q.N, 1) + 64; q.node_off = node_off; q.slot_off = slot_off;
            node_off += q.capn; slot_off += 4 * std::max(q.N, 1) + 16;
            maxN = std::max(maxN, q.N);
            if (L.maxBX >= 32768 || L.maxBY >= 32768) ok = false;
        }
        Q.nodes_per_frame = node_off; Q.slots_per_frame = slot_off;
        const int CAPL = 4 * std::max(maxN, 1) + 16, CAPV = std::max(maxN, 1) + 8, CAPN = 12 * std::max(maxN, 1) + 64;
        e->qt_lds = (size_t)8 * CAPL + (size_t)8 * CAPV * 3 + (size_t)4 * (CAPV + 1) * 2 + (size_t)4 * ((CAPN + 31) / 32) + (size_t)2 * CAPV + 64;
        const char *force = getenv("CUBESLAM_ORB_QUADTREE"); // "host": keep DistributeOctTree on the host (tests compare both)
        e->gpu_quadtree = ok && maxN >= 1 && CAPN < 32768 && e->qt_lds <= 150 * 1024 && !(force && !strcmp(force, "host"));
        if (e->gpu_quadtree) {
            A_(cs_dalloc(ctx, &e->d_qnodes, (size_t)Q.nodes_per_frame * max_frames)); A_(cs_dalloc(ctx, &e->d_slots, (size_t)Q.slots_per_frame * max_frames));
            A_(cs_dalloc(ctx, &e->d_slot_cnt, (size_t)nlevels * max_frames)); A_(cs_dalloc(ctx, &e->d_sel_base, (size_t)nlevels * max_frames + 1));
            A_(cs_dalloc(ctx, &e->d_qstatus, (size_t)1));
        }
    }
    A_(cs_h2d(ctx, e->d_xofs, xofs.data(), xofs.size()));
    A_(cs_h2d(ctx, e->d_yofs, yofs.data(), yofs.size()));
    A_(cs_h2d(ctx, e->d_ialpha, ialpha.data(), ialpha.size()));
    A_(cs_h2d(ctx, e->d_ibeta, ibeta.data(), ibeta.size()));
    { hipError_t er = hipStreamSynchronize(ctx->stream); if (er != hipSuccess) { ctx->err = hipGetErrorString(er); cs_orb_destroy(ctx, e); return CS_ERR_HIP; } }
#undef A_
    *out = e;
    return CS_OK;
}

int cs_orb_get_table(const cs_orb *e, int which, void *out) {
    if (!e || !out) return CS_ERR_BAD_ARG;
    const std::vector<float> *t = which == 0 ? &e->mvScaleFactor : which == 1 ? &e->mvInvScaleFactor : which == 2 ? &e->mvLevelSigma2 : which == 3 ? &e->mvInvLevelSigma2 : nullptr;
    if (t) { memcpy(out, t->data(), t->size() * sizeof(float)); return CS_OK; }
    if (which == 4) { memcpy(out, e->mnFeaturesPerLevel.data(), e->mnFeaturesPerLevel.size() * sizeof(int)); return CS_OK; }
    return CS_ERR_BAD_ARG;
}

int cs_orb_upload(cs_ctx *ctx, cs_orb *e, const uint8_t *gray, int n_frames, int stride) {
    if (!ctx || !e || !gray || n_frames < 1 || n_frames > e->max_frames || stride < e->W) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    e->n_frames = n_frames;
    for (int f = 0; f < n_frames; f++) // level 0 = the image itself (copyMakeBorder only adds the border, :1119-1123)
        CS_HIP(ctx, hipMemcpy2DAsync(e->d_pyr + (long)f * e->P.frame_stride, (size_t)e->W, gray + (size_t)f * stride * e->H, (size_t)stride, (size_t)e->W,
                                     (size_t)e->H, hipMemcpyHostToDevice, ctx->stream));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

// The frames of a run from DEVICE memory (n_frames x height rows of `width` bytes, one frame behind the other): a copy on the context's stream, nothing waits.  The
// streaming front-end (cs_frontend_stream_*) feeds the extractor from the ring slot a copy stream filled.
int cs_orb_geometry(const cs_orb *e, int *width, int *height, int *max_frames) { // (library-internal: the streaming runner checks its ring against the objects it feeds)
    if (!e) return CS_ERR_BAD_ARG;
    *width = e->W; *height = e->H; *max_frames = e->max_frames;
    return CS_OK;
}
int cs_orb_set_frames_device(cs_ctx *ctx, cs_orb *e, const uint8_t *d_gray, int n_frames) {
    if (!ctx || !e || !d_gray || n_frames < 1 || n_frames > e->max_frames) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    e->n_frames = n_frames;
    const size_t fb = (size_t)e->W * e->H; // level 0 of frame f sits at f * frame_stride of the pyramid arena, rows of W bytes
    CS_HIP(ctx, hipMemcpy2DAsync(e->d_pyr, (size_t)e->P.frame_stride, d_gray, fb, fb, (size_t)n_frames, hipMemcpyDeviceToDevice, ctx->stream));
    return CS_OK;
}

int cs_orb_run(cs_ctx *ctx, cs_orb *e) {
    if (!ctx || !e || e->n_frames < 1) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const Pyr &P = e->P;
    const int F = e->n_frames, NL = P.nlevels;
    // ---- GPU phase A: pyramid, FAST score map, per-cell NMS + ordered compaction
    for (int l = 1; l < NL; l++) {
        CS_LAUNCH(ctx, "orb_resize", orb_resize, dim3((P.l[l].w + 255) / 256, (P.l[l].h + 4 * RS_ROWS - 1) / (4 * RS_ROWS), F), dim3(256), 0, P, l, e->d_pyr, e->d_xofs, e->d_ialpha,
                  e->d_yofs, e->d_ibeta);
    }
    CS_LAUNCH(ctx, "orb_fast_score", orb_fast_score, dim3(P.tile_off[NL], 1, F), dim3(256), 0, P, e->d_pyr, e->d_smap);
    CS_LAUNCH(ctx, "orb_cells", orb_cells, dim3((P.cells_per_frame + CELLS_PER_WG - 1) / CELLS_PER_WG, F), dim3(64 * CELLS_PER_WG), 0, P, e->d_smap, 0, e->d_cell_count, e->d_cell_base, e->d_cand, e->d_cell_mask);
    CS_LAUNCH(ctx, "orb_scan", orb_scan_cells, dim3(NL, F), dim3(64), 0, P, e->d_cell_count, e->d_cell_base, e->d_level_total);
    CS_LAUNCH(ctx, "orb_scan", orb_scan_levels, dim3(1), dim3(64), 0, F * NL, e->d_level_total, e->d_level_base);
    CS_LAUNCH(ctx, "orb_scan", orb_rebase_cells, dim3((F * P.cells_per_frame + 255) / 256), dim3(256), 0, P, F, e->d_level_base, e->d_cell_base);
    e->level_base.resize((size_t)F * NL + 1);
    int r = cs_d2h(ctx, e->level_base.data(), e->d_level_base, e->level_base.size()); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long total = e->level_base[(size_t)F * NL];
    if (total > e->cand_cap) { ctx->err = "ORB candidate capacity exceeded"; return CS_ERR_CAPACITY; }
    if (total > e->cand_alloc) { // the second pass of orb_cells writes `total` candidates: room for them and a quarter more
        void *old[] = {e->d_cand, e->d_qperm, e->d_qtmp};
        for (void *q : old) if (q) hipFree(q);
        e->d_cand = nullptr; e->d_qperm = nullptr; e->d_qtmp = nullptr; e->cand_alloc = 0;
        const long cap = std::min<long>(e->cand_cap, total + total / 4 + 4096);
        r = cs_dalloc(ctx, &e->d_cand, (size_t)cap * 3); if (r) return r;
        if (e->gpu_quadtree) { r = cs_dalloc(ctx, &e->d_qperm, (size_t)cap); if (r) return r; r = cs_dalloc(ctx, &e->d_qtmp, (size_t)cap); if (r) return r; }
        e->cand_alloc = cap;
    }
    CS_LAUNCH(ctx, "orb_cells", orb_cells, dim3((P.cells_per_frame + CELLS_PER_WG - 1) / CELLS_PER_WG, F), dim3(64 * CELLS_PER_WG), 0, P, e->d_smap, 1, e->d_cell_count, e->d_cell_base, e->d_cand, e->d_cell_mask);
    bool on_device = e->gpu_quadtree;
    if (on_device) {
        // ---- DistributeOctTree on the device: no candidate round trip; the blur runs behind it
        CS_HIP(ctx, hipMemsetAsync(e->d_qstatus, 0, sizeof(int), ctx->stream));
        if (e->qt_lds > 64 * 1024) CS_HIP(ctx, hipFuncSetAttribute((const void *)orb_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->qt_lds));
        CS_LAUNCH(ctx, "orb_quadtree", orb_quadtree, dim3(F, NL), dim3(256), e->qt_lds, e->Q, e->d_level_base, e->d_cand, e->d_qperm, e->d_qtmp, e->d_qnodes, e->d_slots,
                  e->d_slot_cnt, e->d_qstatus);
        CS_LAUNCH(ctx, "orb_scan", orb_scan_levels, dim3(1), dim3(64), 0, F * NL, e->d_slot_cnt, e->d_sel_base);
        e->sel_base.resize((size_t)F * NL + 1);
        int qstatus = 0;
        r = cs_d2h(ctx, e->sel_base.data(), e->d_sel_base, e->sel_base.size()); if (r) return r;
        r = cs_d2h(ctx, &qstatus, e->d_qstatus, 1); if (r) return r;
        CS_LAUNCH(ctx, "orb_blur", orb_blur, dim3(P.blur_off[NL], 1, F), dim3(64), 0, P, e->d_pyr, e->d_blur);
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        e->cand_valid = false;
        if (qstatus != 0) on_device = false; // panorama / pool overflow: redo this batch on the host
        else {
            const int n = e->sel_base[(size_t)F * NL];
            e->frame_first.assign((size_t)F + 1, 0);
            for (int f = 0; f <= F; f++) e->frame_first[f] = e->sel_base[(size_t)std::min(f, F) * NL];
            if (n > e->sel_cap) { ctx->err = "ORB keypoint capacity exceeded"; return CS_ERR_CAPACITY; }
            if (n == 0) return CS_OK;
            CS_LAUNCH(ctx, "orb_compact_sel", orb_compact_sel, dim3(NL, F), dim3(256), 0, e->Q, e->d_slots, e->d_slot_cnt, e->d_sel_base, e->d_sel);
            CS_LAUNCH(ctx, "orb_angle", orb_angle, dim3((n + 7) / 8), dim3(256), 0, P, e->d_pyr, e->d_sel, n, e->d_angle);
            CS_LAUNCH(ctx, "orb_desc", orb_desc, dim3((n + 3) / 4), dim3(256), 0, P, e->d_blur, e->d_sel, e->d_angle, n, e->d_kps, e->d_desc);
            CS_HIP(ctx, hipGetLastError());
            return CS_OK;
        }
    }
    const bool blur_done = e->gpu_quadtree; // the device attempt already ran the blur
    e->cand.resize((size_t)std::max<long>(total, 1));
    r = cs_d2h(ctx, (float *)e->cand.data(), e->d_cand, (size_t)total * 3); if (r) return r;
    hipEvent_t ev_cand = ctx->get_event();
    CS_HIP(ctx, hipEventRecord(ev_cand, ctx->stream));
    // the blur does not depend on the selection: queued behind the candidate copy, it overlaps the host quadtree
    if (!blur_done) CS_LAUNCH(ctx, "orb_blur", orb_blur, dim3(P.blur_off[NL], 1, F), dim3(64), 0, P, e->d_pyr, e->d_blur);
    CS_HIP(ctx, hipEventSynchronize(ev_cand));
    e->cand_valid = true;
    ctx->pool.push_back(ev_cand);
    // ---- host: DistributeOctTree per (frame, level), ORBextractor.cc:831-832
    const auto t_host0 = std::chrono::steady_clock::now();
    e->sel.clear();
    e->frame_first.assign((size_t)F + 1, 0);
    {
        std::vector<std::vector<SelKP>> per((size_t)F * NL);
    cs_omp_prepare();
#pragma omp parallel num_threads(std::max(1, std::min(ctx->host_threads, F * NL / 4)))
        {
            cs_orb_host::QuadTree qt;
            std::vector<int> idx;
#pragma omp for schedule(dynamic, 1)
            for (int fl = 0; fl < F * NL; fl++) {
                const int f = fl / NL, l = fl % NL;
                const Lvl &L = P.l[l];
                const long b0 = e->level_base[fl], n = e->level_base[fl + 1] - b0;
                if (n <= 0 || L.nCols == 0) continue;
                const cs_orb_host::Cand *K = e->cand.data() + b0;
                qt.distribute(K, (int)n, MINB, L.maxBX, MINB, L.maxBY, e->mnFeaturesPerLevel[l], idx);
                std::vector<SelKP> &o = per[fl];
                o.reserve(idx.size());
                for (int id : idx) o.push_back(SelKP{K[id].x + MINB, K[id].y + MINB, K[id].response, l, f}); // :841-844
            }
        }
        for (int f = 0; f < F; f++) {
            e->frame_first[f] = (int)e->sel.size();
            for (int l = 0; l < NL; l++) e->sel.insert(e->sel.end(), per[(size_t)f * NL + l].begin(), per[(size_t)f * NL + l].end());
        }
        e->frame_first[F] = (int)e->sel.size();
    }
    const int n = (int)e->sel.size();
    if (ctx->timing) { // host section accounted like a kernel ("host_" prefix), in ms
        auto &rec = ctx->timings["host_orb_quadtree"];
        rec.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
        rec.count++;
        ctx->timings["host_omp_threads"].total_ms = ctx->host_threads;
        ctx->timings["host_omp_threads"].count = 1;
    }
    if (n > e->sel_cap) { ctx->err = "ORB keypoint capacity exceeded"; return CS_ERR_CAPACITY; }
    if (n == 0) return CS_OK;
    // ---- GPU phase B: orientation, descriptors
    r = cs_h2d(ctx, e->d_sel, e->sel.data(), (size_t)n); if (r) return r;
    CS_LAUNCH(ctx, "orb_angle", orb_angle, dim3((n + 7) / 8), dim3(256), 0, P, e->d_pyr, e->d_sel, n, e->d_angle);
    CS_LAUNCH(ctx, "orb_desc", orb_desc, dim3((n + 3) / 4), dim3(256), 0, P, e->d_blur, e->d_sel, e->d_angle, n, e->d_kps, e->d_desc);
    CS_HIP(ctx, hipGetLastError());
    return CS_OK;
}

int cs_orb_read(cs_ctx *ctx, cs_orb *e, cs_keypoint *kps, uint8_t *desc, int cap_per_frame, int *counts) {
    if (!ctx || !e || !kps || !desc || !counts || cap_per_frame < 1) return CS_ERR_BAD_ARG;
    int status = CS_OK;
    for (int f = 0; f < e->n_frames; f++) {
        const int b0 = e->frame_first[f], n = e->frame_first[f + 1] - b0;
        counts[f] = std::min(n, cap_per_frame);
        if (n > cap_per_frame) status = CS_ERR_CAPACITY;
        int r = cs_d2h(ctx, kps + (size_t)f * cap_per_frame, e->d_kps + b0, (size_t)counts[f]); if (r) return r;
        r = cs_d2h(ctx, desc + (size_t)f * cap_per_frame * 32, (const uint8_t *)(e->d_desc + (size_t)b0 * 4), (size_t)counts[f] * 32); if (r) return r;
    }
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return status;
}

// the key points and descriptors of every frame of the last run, one frame behind the other (two copies instead of two per frame): frame f is
// first[f] .. first[f + 1] (first: n_frames + 1 entries).  *total = key points of the run; CS_ERR_CAPACITY (nothing copied) when cap_total is too small.
int cs_orb_read_packed(cs_ctx *ctx, cs_orb *e, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total) {
    if (!ctx || !e || !first || !total || (int)e->frame_first.size() < e->n_frames + 1) return CS_ERR_BAD_ARG;
    const int b0 = e->frame_first[0];
    const long n = e->frame_first[(size_t)e->n_frames] - b0;
    *total = n;
    for (int f = 0; f <= e->n_frames; f++) first[f] = e->frame_first[(size_t)f] - b0;
    if (!kps || !desc) return CS_OK; // size query
    if (n > cap_total) return CS_ERR_CAPACITY;
    if (n == 0) return CS_OK;
    int r = cs_d2h(ctx, kps, e->d_kps + b0, (size_t)n); if (r) return r;
    r = cs_d2h(ctx, desc, (const uint8_t *)(e->d_desc + (size_t)b0 * 4), (size_t)n * 32); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

// cs_orb_read_packed's copies on a stream of the caller's choice, nothing waits (the streaming front-end reads a step's results on its own copy stream behind an event)
int cs_orb_read_packed_on(cs_orb *e, void *stream, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total) {
    if (!e || !first || !total || !kps || !desc || (int)e->frame_first.size() < e->n_frames + 1) return CS_ERR_BAD_ARG;
    const int b0 = e->frame_first[0];
    const long n = e->frame_first[(size_t)e->n_frames] - b0;
    *total = n;
    for (int f = 0; f <= e->n_frames; f++) first[f] = e->frame_first[(size_t)f] - b0;
    if (n > cap_total) return CS_ERR_CAPACITY;
    if (n == 0) return CS_OK;
    if (hipMemcpyAsync(kps, e->d_kps + b0, sizeof(cs_keypoint) * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return CS_ERR_HIP;
    if (hipMemcpyAsync(desc, e->d_desc + (size_t)b0 * 4, (size_t)n * 32, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return CS_ERR_HIP;
    return CS_OK;
}

int cs_orb_extract(cs_ctx *ctx, cs_orb *e, const uint8_t *gray, int n_frames, int stride, cs_keypoint *kps, uint8_t *desc, int cap_per_frame,
                   int *counts) {
    int r = cs_orb_upload(ctx, e, gray, n_frames, stride);
    if (r == CS_OK) r = cs_orb_run(ctx, e);
    if (r == CS_OK) r = cs_orb_read(ctx, e, kps, desc, cap_per_frame, counts);
    return r;
}

int cs_orb_get_level(cs_ctx *ctx, cs_orb *e, int frame, int level, int blurred, uint8_t *out, int *w, int *h) {
    if (!ctx || !e || frame < 0 || frame >= e->n_frames || level < 0 || level >= e->nlevels || !w || !h) return CS_ERR_BAD_ARG;
    const Lvl &L = e->P.l[level];
    *w = L.w; *h = L.h;
    if (out) {
        int r = cs_d2h(ctx, out, (blurred ? e->d_blur : e->d_pyr) + (long)frame * e->P.frame_stride + L.off, (size_t)L.w * L.h); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CS_OK;
}

int cs_orb_get_candidates(cs_ctx *ctx, cs_orb *e, int frame, int level, float *xyr, int cap, int *n) {
    if (!ctx || !e || frame < 0 || frame >= e->n_frames || level < 0 || level >= e->nlevels || !n) return CS_ERR_BAD_ARG;
    const size_t fl = (size_t)frame * e->nlevels + level;
    if (e->level_base.size() <= fl + 1) return CS_ERR_BAD_ARG;
    const long b0 = e->level_base[fl], cnt = e->level_base[fl + 1] - b0;
    *n = (int)cnt;
    if (xyr && !e->cand_valid) { // the device quadtree does not copy the candidates to the host: fetch them now
        const long total = e->level_base.back();
        e->cand.resize((size_t)std::max<long>(total, 1));
        int r = cs_d2h(ctx, (float *)e->cand.data(), e->d_cand, (size_t)total * 3); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        e->cand_valid = true;
    }
    if (xyr) for (long i = 0; i < cnt && i < cap; i++) { xyr[i * 3] = e->cand[b0 + i].x; xyr[i * 3 + 1] = e->cand[b0 + i].y; xyr[i * 3 + 2] = e->cand[b0 + i].response; }
    return CS_OK;
}

} // extern "C"
