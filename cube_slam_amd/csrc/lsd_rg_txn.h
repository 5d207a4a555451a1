// lsd_rg_txn.h -- one seed's turn of the LSD region stage as a resumable transaction against the shared owner map (see lsd_regions.hip for
// the scheme).  The same source runs on the device (one transaction per lane, the lanes of a wave step together) and on the host in
// tools/lsd_sim/txn_sim.cpp, which interleaves a thousand transactions step by step to check the protocol and count its work.
//
// Owner word of a pixel: rank of the holding seed << 32 | tag, all ones = free.  Tags of one seed decrease from execution to execution (and
// from the first growth to the re-growth of refine), so an atomic min with (rank, new tag) takes a pixel from any higher rank AND renews the
// seed's own claim from an earlier execution.  A re-execution therefore never gives its pixels up while it runs: what it does not claim
// again is released at its end, and higher ranks only ever see a net change.
//
// Marks (tile of the pixel, rank of the marker; a transaction runs again when a tile of its read box carries a mark below its rank):
//   * a pixel taken from a higher rank, at once;
//   * a pixel released at the end that the final footprint does not hold (left over from an earlier execution, from the first growth of
//     refine, or cut by the radius reduction): somebody may have seen it taken;
//   * every pixel of the old and the new footprint when the two differ as sequences.
// A mark goes to the tiles of the pixel's 3x3 neighbourhood, and a transaction remembers the tiles of the pixels it expanded (every test
// it made is in the neighbourhood of one of them), so a long thin region only depends on what happens along it, not on its bounding box.
#pragma once
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>

#include "glibc_sincosf.h"

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ inline
#else
#define RG_HD inline
#endif

namespace rg {
#if !defined(__HIPCC__)
using std::max;
using std::min;
#endif
typedef unsigned long long u64;
constexpr double NOTDEF = -1024.0, PI_ = 3.1415926535897932384626433832795, DEG_TO_RADS = PI_ / 180, M_3_2_PI_ = (3 * PI_) / 2, M_2__PI_ = 2 * PI_; // lsd.cpp:54-55
constexpr double ANG_TH = 22.5, DENSITY_TH = 0.7;
constexpr u64 FREE = ~0ull;
constexpr int CAP = 4096;   // pixels of one region (three scratch lists per lane); a larger region sends the frame to the host stage
constexpr int TILE = 8;   // marks: TILE x TILE pixels
constexpr int INF = INT_MAX;

#if defined(__HIP_DEVICE_COMPILE__)
// the owner map of a frame is only touched by the frame's workgroup: workgroup-scope atomics (coherent through the CU's L1)
__device__ __forceinline__ u64 ld64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u64 min64(u64 *p, u64 v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool cas64(u64 *p, u64 e, u64 d) { return __hip_atomic_compare_exchange_strong(p, &e, d, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int min32(int *p, int v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int add32(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
inline u64 ld64(const u64 *p) { return *p; }
inline u64 min64(u64 *p, u64 v) { u64 o = *p; if (v < o) *p = v; return o; }
inline bool cas64(u64 *p, u64 e, u64 d) { if (*p != e) return false; *p = d; return true; }
inline int min32(int *p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline int add32(int *p, int v) { int o = *p; *p = o + v; return o; }
#endif

RG_HD float fast_atan2(float y, float x) { // cv::fastAtan2, the polynomial of lsd.hip / the oracle
    const float p1 = 0.9997878412794807f * (float)(180 / PI_), p3 = -0.3258083974640975f * (float)(180 / PI_), p5 = 0.1555786518463281f * (float)(180 / PI_),
                p7 = -0.04432655554792128f * (float)(180 / PI_);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct Frame { // one frame's view of the buffers
    int w, h, ne;
    const int *caddr;          // defined pixels in address order: rank -> address
    const double *ang, *mod;   // dense maps
    u64 *own;                  // w*h owner words
    int *fp_off, *fp_cnt, *fp_cap, *fp_nt; // per rank: slot of the last execution in the pool (offset, footprint pixels, reserved ints, tiles after the pixels)
    unsigned *execs;           // per rank: executions so far (tags)
    uint8_t *flag;             // per rank: bit0 active (holds its seed pixel), bit1 line candidate (rectangle stage passed, nothing taken away)
    double *reg_angle;         // per rank: region angle of a line candidate
    int *pool; int pool_cap; int *pool_head;
    int *chg; int tw;          // tile marks of the running round
    int *status;               // [0] rounds, [1] failure flag, [2] executions, [3] lane steps
    int min_reg_size;
};

RG_HD bool aligned_ang(double a, double theta, double prec) { // isAligned lsd.cpp:1138-1154 on a fetched angle
    if (a == NOTDEF) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI_) { n_theta -= M_2__PI_; if (n_theta < 0) n_theta = -n_theta; }
    return n_theta <= prec;
}
RG_HD double dist(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }
RG_HD double angle_diff_signed(double a, double b) { double diff = a - b; while (diff <= -PI_) diff += M_2__PI_; while (diff > PI_) diff -= M_2__PI_; return diff; }

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

// region2rect + get_theta (lsd.cpp:690-784) over the pixel list L[0..n) in list order
RG_HD void region2rect(const Frame &F, const int *L, int n, double reg_angle, double prec, double p, Rect &rec) {
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < n; ++i) { const int q = L[i]; const double wgt = F.mod[q]; x += double(q % F.w) * wgt; y += double(q / F.w) * wgt; sum += wgt; }
    x /= sum; y /= sum;
    double Ixx = 0, Iyy = 0, Ixy = 0;
    for (int i = 0; i < n; ++i) {
        const int q = L[i];
        const double weight = F.mod[q], dx = double(q % F.w) - x, dy = double(q / F.w) - y;
        Ixx += dy * dy * weight; Iyy += dx * dx * weight; Ixy -= dx * dy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? double(fast_atan2(float(lambda - Ixx), float(Ixy))) : double(fast_atan2(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += PI_;
    const double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < n; ++i) {
        const int q = L[i];
        const double rdx = double(q % F.w) - x, rdy = double(q / F.w) - y;
        const double l = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (ww > w_max) w_max = ww; else if (ww < w_min) w_min = ww;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

RG_HD void mark(const Frame &F, int q, int lo) { // the tiles of the pixels that may have tested q
    const int x = q % F.w, y = q / F.w;
    const int tx0 = max(x - 1, 0) / TILE, tx1 = min(x + 1, F.w - 1) / TILE, ty0 = max(y - 1, 0) / TILE, ty1 = min(y + 1, F.h - 1) / TILE;
    min32(&F.chg[ty0 * F.tw + tx0], lo);
    if (tx1 != tx0) min32(&F.chg[ty0 * F.tw + tx1], lo);
    if (ty1 != ty0) { min32(&F.chg[ty1 * F.tw + tx0], lo); if (tx1 != tx0) min32(&F.chg[ty1 * F.tw + tx1], lo); }
}
RG_HD uint32_t rank_of(u64 wd) { return (uint32_t)(wd >> 32); }

enum Phase { PH_IDLE = 0, PH_OLD, PH_SEED, PH_GROW, PH_POST, PH_FINAL, PH_DONE };

struct Txn {
    int phase = PH_IDLE;
    int s = 0, saddr = 0;
    u64 word = 0;          // (rank, tag) of the growth in progress
    int n = 0, i = 0, q = 0, j = 0, ne = 0, np1 = 0;
    double reg_angle = 0, prec = 0;
    float sumdx = 0, sumdy = 0;
    int nt = 0, t_last = -1, t_prev = -1; // tiles of the expanded pixels (consecutive repeats dropped)
    bool overflow = false, candidate = false, second = false, lost = false;
    int *L = nullptr, *E = nullptr, *P1 = nullptr, *TL = nullptr; // this lane's scratch lists: region, effective old footprint, first growth of refine, tiles
    long n_exec = 0, n_steps = 0;
#if defined(RG_PROFILE)
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned profn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
};

RG_HD void start_growth(const Frame &F, Txn &T, double prec) {
    T.n = 1; T.i = 0; T.q = T.saddr; T.L[0] = T.saddr;
    T.reg_angle = F.ang[T.saddr];
    T.sumdx = float(cos(T.reg_angle)); T.sumdy = float(sin(T.reg_angle)); // :651-652: doubles here
    T.prec = prec;
    T.phase = PH_GROW;
}

// one pixel of region_grow (lsd.cpp:637-688): the owners and angles of the 3x3 neighbourhood are fetched together (nine independent
// loads), then tested in the reference's order
RG_HD void grow_step(const Frame &F, Txn &T) {
    const int s = T.s, q = T.q, px = q % F.w, py = q / F.w;
    const int tile = (py / TILE) * F.tw + px / TILE;
    if (tile != T.t_last && tile != T.t_prev) { if (T.nt < CAP) T.TL[T.nt++] = tile; else T.overflow = true; T.t_prev = T.t_last; T.t_last = tile; }
    u64 ow[9];
    double an[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int xx = px + (k % 3) - 1, yy = py + (k / 3) - 1;
        const bool ok = xx >= 0 && xx < F.w && yy >= 0 && yy < F.h;
        const int c = ok ? xx + yy * F.w : q;
        ow[k] = ok ? ld64(&F.own[c]) : 0;
        an[k] = ok ? F.ang[c] : NOTDEF; // (outside the image: never aligned)
    }
    int qn = (T.i + 1 < T.n) ? T.L[T.i + 1] : -1; // the next pixel of the list, unless this step appends it
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (rank_of(ow[k]) < (uint32_t)s || ow[k] == T.word) continue; // outside, a lower rank's, or already in this region: used
        if (!aligned_ang(an[k], T.reg_angle, T.prec)) continue;
        const int c = px + (k % 3) - 1 + (py + (k / 3) - 1) * F.w;
        const u64 old = min64(&F.own[c], T.word);
        if (rank_of(old) < (uint32_t)s || old == T.word) continue; // a lower rank got there first
        if (rank_of(old) > (uint32_t)s && old != FREE) mark(F, c, s); // taken away from a higher rank
        if (T.n >= CAP) { T.overflow = true; continue; }
        T.L[T.n] = c;
        if (T.n == T.i + 1) qn = c;
        ++T.n;
        T.sumdx += glibc_sincosf::cosf_(float(an[k])); // :676-677 cos(float), sin(float)
        T.sumdy += glibc_sincosf::sinf_(float(an[k]));
        T.reg_angle = fast_atan2(T.sumdy, T.sumdx) * DEG_TO_RADS;
    }
    T.q = qn;
    ++T.i;
}

RG_HD void release_leftover(const Frame &F, const Txn &T, int q) {
    const u64 wd = ld64(&F.own[q]);
    if (rank_of(wd) == (uint32_t)T.s && wd != T.word && cas64(&F.own[q], wd, FREE)) mark(F, q, T.s);
}

// flsd :490-502 between region_grow and rect_improve: rectangle, density, refine (:786-832), reduce_region_radius (:834-871)
RG_HD void post_growth(const Frame &F, Txn &T) {
    const double prec = PI_ * ANG_TH / 180, p = ANG_TH / 180;
    const int s = T.s;
    Rect rec;
    if (!T.second) {
        T.candidate = false;
        T.phase = PH_FINAL;
        if (T.n < F.min_reg_size) return;
        region2rect(F, T.L, T.n, T.reg_angle, prec, p, rec);
        const double density = double(T.n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        T.candidate = true;
        if (density >= DENSITY_TH) return;
        // refine: the tolerance from the pixels near the seed, then the region grows again (its pixels stay claimed under the old tag meanwhile)
        const double xc = double(T.saddr % F.w), yc = double(T.saddr / F.w), ang_c = F.ang[T.saddr];
        double sum = 0, s_sum = 0;
        int cnt = 0;
        for (int i = 0; i < T.n; ++i) {
            const int q = T.L[i];
            T.P1[i] = q;
            if (dist(xc, yc, double(q % F.w), double(q / F.w)) < rec.width) { const double ang_d = angle_diff_signed(F.ang[q], ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++cnt; }
        }
        T.np1 = T.n;
        const double mean_angle = sum / double(cnt);
        const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / double(cnt) + mean_angle * mean_angle);
        T.word -= 1;
        T.second = true;
        const u64 old = min64(&F.own[T.saddr], T.word);
        if (rank_of(old) < (uint32_t)s) { T.lost = true; T.candidate = false; T.n = 0; return; } // the seed went to a lower rank meanwhile
        start_growth(F, T, tau);
        return;
    }
    T.phase = PH_FINAL;
    if (T.n < 2) { T.candidate = false; return; }
    region2rect(F, T.L, T.n, T.reg_angle, prec, p, rec);
    double density = double(T.n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= DENSITY_TH) return;
    const double xc = double(T.saddr % F.w), yc = double(T.saddr / F.w);
    const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc), r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
    double radSq = r1 > r2 ? r1 : r2;
    while (density < DENSITY_TH) {
        radSq *= 0.75 * 0.75;
        for (int i = 0; i < T.n; ++i) {
            const int q = T.L[i];
            const double ddx = double(q % F.w) - xc, ddy = double(q / F.w) - yc;
            if (ddx * ddx + ddy * ddy > radSq) {
                if (cas64(&F.own[q], T.word, FREE)) mark(F, q, s);
                T.L[i] = T.L[T.n - 1]; T.L[T.n - 1] = q; // std::swap(reg[i], reg[reg_size - 1])
                --T.n; --i;
            }
        }
        if (T.n < 2) { T.candidate = false; return; }
        region2rect(F, T.L, T.n, T.reg_angle, prec, p, rec);
        density = double(T.n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
}

RG_HD void finish(const Frame &F, Txn &T) {
    const int s = T.s;
    int m = 0;
    for (int i = 0; i < T.n; i++) if (ld64(&F.own[T.L[i]]) == T.word) T.L[m++] = T.L[i];
    const bool intact = m == T.n; // nothing was taken away while it ran
    bool same = m == T.ne;
    for (int i = 0; same && i < m; i++) same = T.L[i] == T.E[i];
    for (int i = 0; i < T.ne; i++) release_leftover(F, T, T.E[i]);
    for (int i = 0; i < T.np1; i++) release_leftover(F, T, T.P1[i]);
    if (!same) { for (int i = 0; i < T.ne; i++) mark(F, T.E[i], s); for (int i = 0; i < m; i++) mark(F, T.L[i], s); }
    T.phase = PH_IDLE;
    F.fp_cnt[s] = 0; F.fp_nt[s] = 0;
    if (T.overflow) { F.status[1] = 1; return; }
    int off = F.fp_off[s];
    if (m + T.nt > F.fp_cap[s]) { // a larger slot in the pool (slots are not given back)
        int cap = 8;
        while (cap < m + T.nt) cap *= 2;
        off = add32(F.pool_head, cap);
        if (off + cap > F.pool_cap) { F.status[1] = 1; return; }
        F.fp_off[s] = off; F.fp_cap[s] = cap;
    }
    for (int i = 0; i < m; i++) F.pool[off + i] = T.L[i];
    for (int i = 0; i < T.nt; i++) F.pool[off + m + i] = T.TL[i];
    F.fp_cnt[s] = m; F.fp_nt[s] = T.nt;
    F.reg_angle[s] = T.reg_angle;
    F.flag[s] = T.lost ? 0 : (uint8_t)(1 | ((T.candidate && intact) ? 2 : 0));
}

// One step of the lane's transaction; fetches the next dirty seed of the round when idle.  ctl[2] is the round's work counter.
RG_HD void step(const Frame &F, Txn &T, const int *dirty, int n_dirty, int *ctl) {
    T.n_steps++;
#if defined(RG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    const int ph0 = T.phase;
    const bool first = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __ffsll((long long)__ballot(T.phase == ph0)) - 1);
    const unsigned long long c0 = wall_clock64();
    struct Acc { Txn &T; int ph; bool first; unsigned long long c0; __device__ ~Acc() { if (first) { T.prof[ph] += wall_clock64() - c0; T.profn[ph]++; } } } acc{T, ph0, first, c0};
#endif
    switch (T.phase) {
    case PH_IDLE: {
        const int k = add32(&ctl[2], 1);
        if (k >= n_dirty) { T.phase = PH_DONE; return; }
        const int s = dirty[k];
        T.s = s; T.saddr = F.caddr[s] & 0x7fffffff; // (bit 31 is lsd_emit's single-pixel flag for the host stage)
        const unsigned e = F.execs[s]++;
        T.word = ((u64)(uint32_t)s << 32) | (u64)(0xFFFFFFF0u - 2u * e);
        T.j = 0; T.ne = 0; T.np1 = 0; T.n = 0;
        T.overflow = false; T.candidate = false; T.second = false; T.lost = false;
        T.nt = 0; T.t_last = -1; T.t_prev = -1;
        T.n_exec++;
        T.phase = PH_OLD;
        return;
    }
    case PH_OLD: { // what it still holds of its previous footprint
        const int off = F.fp_off[T.s], cnt = F.fp_cnt[T.s];
        for (int k = 0; k < 8 && T.j < cnt; k++, T.j++) { const int q = F.pool[off + T.j]; if (rank_of(ld64(&F.own[q])) == (uint32_t)T.s) T.E[T.ne++] = q; }
        if (T.j >= cnt) T.phase = PH_SEED;
        return;
    }
    case PH_SEED: {
        F.flag[T.s] = 0;
        const u64 old = min64(&F.own[T.saddr], T.word);
        if (rank_of(old) < (uint32_t)T.s) { T.lost = true; T.phase = PH_FINAL; return; } // the seed pixel belongs to an earlier region
        if (rank_of(old) > (uint32_t)T.s && old != FREE) mark(F, T.saddr, T.s);
        start_growth(F, T, PI_ * ANG_TH / 180);
        return;
    }
    case PH_GROW:
        if (T.i < T.n) grow_step(F, T);
        else T.phase = PH_POST;
        return;
    case PH_POST: post_growth(F, T); return;
    case PH_FINAL: finish(F, T); return;
    default: return;
    }
}

// does seed i run in the next round?  (after a round, before the tile marks are cleared)
RG_HD bool is_dirty(const Frame &F, int i) {
    const uint32_t o = rank_of(ld64(&F.own[F.caddr[i] & 0x7fffffff]));
    if (F.flag[i] & 1) {
        if (o != (uint32_t)i) return true;
        const int *tl = F.pool + F.fp_off[i] + F.fp_cnt[i];
        for (int k = 0, nt = F.fp_nt[i]; k < nt; k++) if (F.chg[tl[k]] < i) return true;
        return false;
    }
    return o > (uint32_t)i || F.fp_cnt[i] > 0; // free or held by a higher rank: it is a seed (or it still has a footprint to give back)
}
// first round: seeds without an aligned defined neighbour of lower rank (the others are almost always swallowed by an earlier region)
RG_HD bool is_initial(const Frame &F, int i) {
    const int s = F.caddr[i] & 0x7fffffff, x = s % F.w, y = s / F.w;
    const double a = F.ang[s], prec = PI_ * ANG_TH / 180;
    const int nb[4][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}};
    for (int k = 0; k < 4; k++) { const int xx = x + nb[k][0], yy = y + nb[k][1]; if (xx < 0 || yy < 0 || xx >= F.w) continue; if (aligned_ang(F.ang[xx + yy * F.w], a, prec)) return false; }
    return true;
}
} // namespace rg
