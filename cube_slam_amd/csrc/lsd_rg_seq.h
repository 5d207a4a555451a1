// lsd_rg_seq.h -- the LSD region stage (flsd's seed loop, region_grow, region2rect, refine, reduce_region_radius; lsd.cpp:464-871) as ONE WAVE
// PER FRAME that walks the reference's sequence exactly: the seeds in raster order, every region seeing the marks of all regions before it.
// Nothing is speculated; the 64 lanes serve the latency of the sequence instead:
//   * the seed loop fetches the state of 64 seeds at a time and keeps it current in registers (every accepted pixel is compared with the 64
//     seed addresses), so a seed that an earlier region swallowed costs no memory round trip;
//   * region_grow works on two 8 x 8 windows of the pixel map held in the lanes' registers (one 4-byte load per lane: the map is ONE FLOAT per pixel, the level-line
//     angle in float degrees while the pixel is defined and unused; cos / sin of a window's pixels are computed in place at the fetch, glibc's values); a list pixel inside a
//     window is expanded without touching memory, its nine tests in the reference's order; an accepted pixel is struck from the windows and
//     changes the region angle for the tests after it -- one ballot + two lane reads + the fastAtan2 polynomial per accepted pixel.  The
//     windows outlive the region: the wave is the only writer of its frame's map, and the next seed is usually next door;
//   * the ordered double sums of region2rect / refine take their terms from the lanes (products computed in parallel, added in list order).
// Throughput comes from frames, not from inside a frame: a frame is a chain of dependent instructions (about 90 per accepted pixel), so thousands
// of frames are resident, four to a SIMD.  No LDS, every wave-uniform value marked as such (W::uni): the bookkeeping runs on the scalar unit.
// Written once for the device and for a host model (tools/lsd_sim/seq_sim.cpp) that runs the 64 lanes as loops: per-lane
// values are PerLane<T>, per-lane code sits in W::each bodies, everything else is wave-uniform.
// A body must not read what another lane's part of the SAME body writes (on the device the lanes run it together).
#pragma once
#include "glibc_sincosf.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ inline
#else
#define RG_HD inline
#endif

namespace rg { // arithmetic shared by the sequential stage and the rectangle kernel (lsd_regions.hip)
typedef unsigned long long u64;
constexpr double NOTDEF = -1024.0, PI_ = 3.1415926535897932384626433832795, DEG_TO_RADS = PI_ / 180, M_3_2_PI_ = (3 * PI_) / 2, M_2__PI_ = 2 * PI_; // lsd.cpp:54-55
constexpr double ANG_TH = 22.5, DENSITY_TH = 0.7;

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
RG_HD float fast_atan2_1(float y, float x) { // fast_atan2 with ONE division: the two branches divide min by max + eps -- the same IEEE operations on the same operands, bit for bit the same result
    const float p1 = 0.9997878412794807f * (float)(180 / PI_), p3 = -0.3258083974640975f * (float)(180 / PI_), p5 = 0.1555786518463281f * (float)(180 / PI_),
                p7 = -0.04432655554792128f * (float)(180 / PI_);
    const float ax = fabsf(x), ay = fabsf(y);
    const bool hi = ax >= ay;
    const float c = (hi ? ay : ax) / ((hi ? ax : ay) + (float)DBL_EPSILON), c2 = c * c;
    const float q = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = hi ? q : 90.f - q;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
RG_HD double dist(double x1, double y1, double x2, double y2) { return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }
RG_HD double angle_diff_signed(double a, double b) { double diff = a - b; while (diff <= -PI_) diff += M_2__PI_; while (diff > PI_) diff -= M_2__PI_; return diff; }

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };
} // namespace rg

namespace rgs {
using rg::u64;
constexpr float NOTDEF_F = -1024.0f;
constexpr int CAP = 32768; // pixels of one region (its list lives in global memory); a larger region sends the batch to the host stage


struct Frame {
    int w, h, ne;
    const int *caddr;   // defined pixels in address order (bit 31: lsd_emit's "stays alone as a seed" flag)
    float *fre;         // dense w*h, the walk's own: the level-line angle in float degrees while the pixel is defined and unused, NOTDEF_F otherwise (used = NOTDEF_F)
    const float *ang;   // dense w*h, read-only: the angle whatever the use (lsd_gradient's map) -- what a release puts back
    const double *mod;  // dense gradient norms
    const float *seed_cs; // per rank: float(cos(angle)), float(sin(angle)) of the pixel's angle as a double -- what a seed starts its sums with (:651-652)
    double *rect; int cand_cap; int *cand_cnt; // the rectangles (12 doubles each, rg::Rect) that reach rect_improve, in seed order
    unsigned long long *prof; // RGS_PROFILE
    int *status;        // [0] region_grow calls, [1] failure (capacity), [2] regions at the rectangle stage, [3] window fetches
    int min_reg_size;
    int list_cap;       // pixels of one region before the frame gives up (<= CAP)
};

#if defined(__HIPCC__)
#define RGS_FN __host__ __device__ inline
#else
#define RGS_FN inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
template <class T> struct PerLane {
    T v;
    __device__ __forceinline__ T &operator[](int) { return v; }
    __device__ __forceinline__ const T &operator[](int) const { return v; }
};
struct Wave { // the 64 lanes of the calling wave
    template <class Fn> static __device__ __forceinline__ void each(Fn f) { f(int(threadIdx.x & 63)); }
    static __device__ __forceinline__ u64 ballot(const PerLane<bool> &p) { return __ballot(p.v); }
    static __device__ __forceinline__ int bc(const PerLane<int> &x, int l) { return __builtin_amdgcn_readlane(x.v, l); }
    static __device__ __forceinline__ float bc(const PerLane<float> &x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x.v), l)); }
    static __device__ __forceinline__ double bc(const PerLane<double> &x, int l) {
        const long long b = __double_as_longlong(x.v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ double uni(double v) { // every lane holds v: say so (branches on it become scalar, what they assign stays in scalar registers)
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ double vmax(const PerLane<double> &x) { double m = x.v; for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o)); return uni(m); }
    static __device__ __forceinline__ double vmin(const PerLane<double> &x) { double m = x.v; for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o)); return uni(m); }
    // memory written by one lane and read by another lane of the wave: its memory instructions execute in order, so only the compiler has to be held back
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
    static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); } // a value every lane holds (tells the compiler so)
};
__device__ __forceinline__ void st_free(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float ld_free(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ctz64(u64 m) { return __ffsll((long long)m) - 1; }
#else
template <class T> struct PerLane {
    T v[64];
    RGS_FN T &operator[](int l) { return v[l]; }
    RGS_FN const T &operator[](int l) const { return v[l]; }
};
struct Wave { // host model: the lanes of a body run one after the other
    template <class Fn> static RGS_FN void each(Fn f) { for (int l = 0; l < 64; l++) f(l); }
    static RGS_FN u64 ballot(const PerLane<bool> &p) { u64 m = 0; for (int l = 0; l < 64; l++) if (p.v[l]) m |= 1ull << l; return m; }
    template <class T> static RGS_FN T bc(const PerLane<T> &x, int l) { return x.v[l]; }
    static RGS_FN double vmax(const PerLane<double> &x) { double m = x.v[0]; for (int l = 1; l < 64; l++) m = fmax(m, x.v[l]); return m; }
    static RGS_FN double vmin(const PerLane<double> &x) { double m = x.v[0]; for (int l = 1; l < 64; l++) m = fmin(m, x.v[l]); return m; }
    static RGS_FN void sync() {}
    static RGS_FN int uni(int v) { return v; }
    static RGS_FN double uni(double v) { return v; }
};
RGS_FN void st_free(float *p, float v) { *p = v; }
RGS_FN float ld_free(const float *p) { return *p; }
RGS_FN int ctz64(u64 m) { return __builtin_ctzll(m); }
#endif

// region_grow's pixel list (packed x | y << 16): global memory for the passes over a finished region, and the newest 64 entries also in one register
// across the lanes (entry k in lane k % 64) -- the growth reads the list a few entries behind its end, so it reads lanes, not memory.  No LDS: a wave
// that sits on a CU for its frame's 100 ms must not keep the workgroups of the other streams' kernels from finding room.
#if !defined(RGS_RING)
#define RGS_RING 64 // (the host model also runs with a ring of 4 to walk the memory path)
#endif
struct List {
    int *glob;
    PerLane<int> ring;
};
template <class W> RGS_FN int list_head(const List &L, int i, int n) { return n - i <= RGS_RING ? W::bc(L.ring, i & (RGS_RING - 1)) : W::uni(L.glob[i]); }

// isAligned lsd.cpp:1138-1154 on an angle in radians (or FAR), without branches: the same IEEE operations in the same order -- |theta - a|, then
// |that - 2 pi| if it exceeds 3 pi / 2 -- the conditional negations written as fabs
RGS_FN bool aligned_rad(double a, double theta, double prec) {
    const double d = fabs(theta - a), d2 = fabs(d - rg::M_2__PI_);
    return (d > rg::M_3_2_PI_ ? d2 : d) <= prec;
}

#if defined(RGS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
// (development) wall-clock ticks (10 ns) and entries per part, added up over all frames without waiting for the result
#define RGS_T0(k) const unsigned long long rgs_t_##k = wall_clock64()
#define RGS_T1(k) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&F.prof[2 * (k)], wall_clock64() - rgs_t_##k); atomicAdd(&F.prof[2 * (k) + 1], 1ull); } } while (0)
#else
#define RGS_T0(k)
#define RGS_T1(k)
#endif

struct Seeds { // the 64 seeds the seed loop is looking at
    PerLane<int> sa; // address, -1 past the end
    u64 freem;       // defined and unused, kept current while regions grow
};

// An 8 x 8 window of pixel records in the lanes' registers: lane l holds pixel (wx + l % 8, wy + l / 8).  The wave is the only writer of its frame's
// map, so a window stays valid across regions and seeds as long as every mark the wave makes is also made in the lanes (strike) and a release
// (refine) drops it.  A pixel whose 3 x 3 neighbourhood lies inside a window is expanded without touching memory: consecutive seeds are
// neighbours in raster order and most regions are a handful of pixels, so most region_grow calls need no fetch at all.
struct Win {
    int wx, wy;
    PerLane<double> ar;        // level-line angle in radians (the map value exactly, :566) while the pixel is defined and unused; FAR otherwise (used, undefined, outside)
    PerLane<float> pc, ps;     // cos, sin of float(angle)
    u64 am; bool am_ok;        // lanes aligned with the current region angle / tolerance, while am_ok
};
struct Wins { Win a, b; }; // a: the window used last; b: the one before (replaced on a miss)
template <class W> RGS_FN void win_swap(Wins &V) {
    const int wx = V.a.wx, wy = V.a.wy; V.a.wx = V.b.wx; V.a.wy = V.b.wy; V.b.wx = wx; V.b.wy = wy;
    const u64 am = V.a.am; V.a.am = V.b.am; V.b.am = am;
    const bool ok = V.a.am_ok; V.a.am_ok = V.b.am_ok; V.b.am_ok = ok;
    W::each([&](int l) { const double r = V.a.ar[l]; V.a.ar[l] = V.b.ar[l]; V.b.ar[l] = r; const float c = V.a.pc[l]; V.a.pc[l] = V.b.pc[l]; V.b.pc[l] = c; const float q = V.a.ps[l]; V.a.ps[l] = V.b.ps[l]; V.b.ps[l] = q; });
}
constexpr int WIN_NONE = -(1 << 20);
constexpr double FAR = 1e300; // an "angle" no region angle is within any tolerance of

RGS_FN bool win_covers(const Win &w, int px, int py) { return (unsigned)(px - w.wx - 1) < 6u && (unsigned)(py - w.wy - 1) < 6u; } // the pixel's 3 x 3 neighbourhood is inside
template <class W> RGS_FN void win_fetch(const Frame &F, Win &w, int wx, int wy) {
    w.wx = wx; w.wy = wy; w.am_ok = false;
    W::each([&](int l) {
        const int xx = wx + (l & 7), yy = wy + (l >> 3);
        w.ar[l] = FAR; w.pc[l] = 0; w.ps[l] = 0;
        if (xx >= 0 && xx < F.w && yy >= 0 && yy < F.h) {
            const float fd = ld_free(&F.fre[xx + yy * F.w]);
            const bool fr = fd != NOTDEF_F;
            const double a = double(fr ? fd : 0.f) * rg::DEG_TO_RADS; // the map value (:566)
            float sn, cc;
            glibc_sincosf::sincosf_pos(float(a), &sn, &cc); // cos(float(angle)), sin(float(angle)) :676-677 with glibc's values, for the 64 pixels at once
            w.ar[l] = fr ? a : FAR; w.pc[l] = cc; w.ps[l] = sn;
        }
    });
#if defined(RGS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    { PerLane<bool> z; W::each([&](int l) { z[l] = w.ar[l] == 12345.0; }); if (W::ballot(z)) w.wx = WIN_NONE; } // (the data has to arrive inside the timed part)
#endif
}
template <class W> RGS_FN void win_strike(Win &w, int x, int y) { // pixel (x, y) is used now
    const int lx = x - w.wx, ly = y - w.wy;
    if (lx < 0 || lx > 7 || ly < 0 || ly > 7) return;
    const int lane = ly * 8 + lx;
    W::each([&](int l) { if (l == lane) w.ar[l] = FAR; });
    w.am &= ~(1ull << lane);
}

// the pixels of region_grow's list are packed x | y << 16
RGS_FN int xy_pack(int x, int y) { return x | (y << 16); }
RGS_FN int xy_addr(int xy, int w) { return (xy & 0xffff) + (xy >> 16) * w; }

// one pixel of region_grow (lsd.cpp:660-686) against the window that covers it; other: the second window, kept current
template <class W> RGS_FN void expand(const Frame &F, Win &w, Win &other, int px, int py, List &L, int &n, double &reg_angle, double prec, float &sumdx, float &sumdy, Seeds &S, bool &overflow) {
    const int b0 = (py - 1 - w.wy) * 8 + (px - 1 - w.wx); // lane of the first neighbour; the nine in the reference's order (yy outer, xx inner) are ascending lanes
    const u64 nb = 0x070707ull << b0;
    int cur = 0;
    for (;;) {
        if (!w.am_ok) { // the tests of all 64 pixels at once; they hold until the region angle changes, so a pixel that accepts nothing costs no vector work
            PerLane<bool> al;
            W::each([&](int l) { al[l] = aligned_rad(w.ar[l], reg_angle, prec); });
            w.am = W::ballot(al); w.am_ok = true;
        }
        const u64 m = w.am & nb & (~0ull << cur);
        if (!m) break;
        const int l0 = ctz64(m);
        const int cx = w.wx + (l0 & 7), cy = w.wy + (l0 >> 3), cs = cx + cy * F.w;
        const float cc = W::bc(w.pc, l0), ss = W::bc(w.ps, l0);
        if (n >= F.list_cap) { overflow = true; return; }
        PerLane<bool> hit;
        W::each([&](int l) {
            if (l == l0) { st_free(&F.fre[cs], NOTDEF_F); L.glob[n] = xy_pack(cx, cy); w.ar[l] = FAR; }
            if (l == (n & (RGS_RING - 1))) L.ring[l] = xy_pack(cx, cy);
            hit[l] = S.sa[l] == cs;
        });
        S.freem &= ~W::ballot(hit);
        win_strike<W>(other, cx, cy);
        ++n;
        sumdx += cc; sumdy += ss; // cos(float(angle)), sin(float(angle)) :676-677, from lsd_emit
        reg_angle = rg::fast_atan2_1(sumdy, sumdx) * rg::DEG_TO_RADS;
        w.am_ok = false; other.am_ok = false;
        if (l0 == 63) break;
        cur = l0 + 1;
    }
}

// region_grow lsd.cpp:637-688 from the pixel (sx, sy) (angle sdeg, float degrees)
template <class W> RGS_FN void grow(const Frame &F, Wins &V, List &L, int &n, double &reg_angle, double prec, int sx, int sy, float sdeg, float scos, float ssin, Seeds &S, bool &overflow, int &fetches) {
    n = 1;
    reg_angle = double(sdeg) * rg::DEG_TO_RADS;
    float sumdx = scos, sumdy = ssin; // cos / sin of the seed angle as a double (:651-652), computed once per pixel by lsd_rg_scatter
    const int saddr = sx + sy * F.w;
    V.a.am_ok = false; V.b.am_ok = false;
    {
        PerLane<bool> hit;
        W::each([&](int l) { if (l == 0) { L.glob[0] = xy_pack(sx, sy); L.ring[l] = xy_pack(sx, sy); st_free(&F.fre[saddr], NOTDEF_F); } hit[l] = S.sa[l] == saddr; });
        S.freem &= ~W::ballot(hit);
        win_strike<W>(V.a, sx, sy); win_strike<W>(V.b, sx, sy);
    }
    for (int i = 0; i < n && !overflow; ++i) {
        const int q = list_head<W>(L, i, n), px = q & 0xffff, py = q >> 16;
        if (!win_covers(V.a, px, py)) {
            if (!win_covers(V.b, px, py)) { // fetch around the pixel, leaning away from the seed (the region grows outwards), over the window used longest ago
                const int ox = px > sx ? 2 : (px < sx ? 5 : (i == 0 ? 2 : 3)), oy = py > sy ? 2 : (py < sy ? 5 : 3);
                RGS_T0(0);
                win_fetch<W>(F, V.b, px - ox, py - oy);
                RGS_T1(0);
                fetches++;
            }
            win_swap<W>(V);
        }
        RGS_T0(1);
        expand<W>(F, V.a, V.b, px, py, L, n, reg_angle, prec, sumdx, sumdy, S, overflow);
        RGS_T1(1);
    }
}

// region2rect + get_theta lsd.cpp:690-784 over L[0..n): the terms come from the lanes, the sums run in list order
template <class W> RGS_FN void to_rect(const Frame &F, const List &L, int n, double reg_angle, double prec, double p, rg::Rect &rec) {
    W::sync();
    double x = 0, y = 0, sum = 0;
    for (int b = 0; b < n; b += 64) {
        PerLane<double> xw, yw, wg;
        W::each([&](int l) {
            const int idx = b + l;
            xw[l] = 0; yw[l] = 0; wg[l] = 0;
            if (idx < n) { const int q = L.glob[idx], qx = q & 0xffff, qy = q >> 16; const double m = F.mod[qx + qy * F.w]; wg[l] = m; xw[l] = double(qx) * m; yw[l] = double(qy) * m; }
        });
        const int cnt = n - b < 64 ? n - b : 64;
        for (int j = 0; j < cnt; j++) { x += W::bc(xw, j); y += W::bc(yw, j); sum += W::bc(wg, j); }
    }
    x /= sum; y /= sum;
    double Ixx = 0, Iyy = 0, Ixy = 0;
    for (int b = 0; b < n; b += 64) {
        PerLane<double> t1, t2, t3;
        W::each([&](int l) {
            const int idx = b + l;
            t1[l] = 0; t2[l] = 0; t3[l] = 0;
            if (idx < n) { const int q = L.glob[idx], qx = q & 0xffff, qy = q >> 16; const double m = F.mod[qx + qy * F.w], dx = double(qx) - x, dy = double(qy) - y; t1[l] = dy * dy * m; t2[l] = dx * dx * m; t3[l] = dx * dy * m; }
        });
        const int cnt = n - b < 64 ? n - b : 64;
        for (int j = 0; j < cnt; j++) { Ixx += W::bc(t1, j); Iyy += W::bc(t2, j); Ixy -= W::bc(t3, j); }
    }
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? double(rg::fast_atan2(float(lambda - Ixx), float(Ixy))) : double(rg::fast_atan2(float(Ixy), float(lambda - Iyy)));
    theta *= rg::DEG_TO_RADS;
    if (fabs(rg::angle_diff_signed(theta, reg_angle)) > prec) theta += rg::PI_;
    const double dx = cos(theta), dy = sin(theta);
    // l_max / w_max only grow from 0 and l_min / w_min only fall from 0 (the reference's else-if never matters: a new maximum is positive): plain extrema
    PerLane<double> lmx, lmn, wmx, wmn;
    W::each([&](int l) { lmx[l] = 0; lmn[l] = 0; wmx[l] = 0; wmn[l] = 0; });
    for (int b = 0; b < n; b += 64)
        W::each([&](int l) {
            const int idx = b + l;
            if (idx < n) {
                const int q = L.glob[idx];
                const double rdx = double(q & 0xffff) - x, rdy = double(q >> 16) - y, ll = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
                if (ll > lmx[l]) lmx[l] = ll;
                if (ll < lmn[l]) lmn[l] = ll;
                if (ww > wmx[l]) wmx[l] = ww;
                if (ww < wmn[l]) wmn[l] = ww;
            }
        });
    const double l_max = W::vmax(lmx), l_min = W::vmin(lmn), w_max = W::vmax(wmx), w_min = W::vmin(wmn);
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

// refine lsd.cpp:786-832 + reduce_region_radius :834-871.  released: pixels went back to "unused" (the seed loop re-reads its 64 seeds)
template <class W> RGS_FN bool refine(const Frame &F, Wins &V, List &L, int &n, double &reg_angle, double prec, double p, rg::Rect &rec, int sx, int sy, float sdeg, float scos, float ssin, Seeds &S, bool &overflow, int &fetches,
                                      bool &released) {
    double density = double(n) / (rg::dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= rg::DENSITY_TH) return true;
    released = true;
    V.a.wx = WIN_NONE; V.b.wx = WIN_NONE; // the lanes' copies do not see the release
    const double xc = double(sx), yc = double(sy), ang_c = double(sdeg) * rg::DEG_TO_RADS;
    double sum = 0, s_sum = 0;
    int cnt = 0;
    W::sync();
    for (int b = 0; b < n; b += 64) {
        PerLane<double> d, dd; PerLane<bool> near;
        W::each([&](int l) {
            const int idx = b + l;
            near[l] = false; d[l] = 0; dd[l] = 0;
            if (idx < n) {
                const int q = L.glob[idx], qx = q & 0xffff, qy = q >> 16;
                const float dg = F.ang[qx + qy * F.w];
                st_free(&F.fre[qx + qy * F.w], dg); // :800 used = NOTUSED
                if (rg::dist(xc, yc, double(qx), double(qy)) < rec.width) { const double a = rg::angle_diff_signed(double(dg) * rg::DEG_TO_RADS, ang_c); d[l] = a; dd[l] = a * a; near[l] = true; }
            }
        });
        u64 m = W::ballot(near);
        while (m) { const int j = ctz64(m); m &= m - 1; sum += W::bc(d, j); s_sum += W::bc(dd, j); ++cnt; }
    }
    const double mean_angle = sum / double(cnt);
    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / double(cnt) + mean_angle * mean_angle);
    grow<W>(F, V, L, n, reg_angle, tau, sx, sy, sdeg, scos, ssin, S, overflow, fetches);
    if (overflow) return false;
    if (n < 2) return false;
    to_rect<W>(F, L, n, reg_angle, prec, p, rec);
    density = double(n) / (rg::dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= rg::DENSITY_TH) return true;
    const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc), r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
    double radSq = r1 > r2 ? r1 : r2;
    while (density < rg::DENSITY_TH) {
        radSq *= 0.75 * 0.75;
        W::sync();
        for (int i = 0; i < n; ++i) { // the swaps reorder the list: one pixel after the other, like the reference
            const int q = W::uni(L.glob[i]);
            const double ddx = double(q & 0xffff) - xc, ddy = double(q >> 16) - yc;
            if (ddx * ddx + ddy * ddy > radSq) {
                const int last = W::uni(L.glob[n - 1]);
                W::each([&](int l) { if (l == 0) { const int ad = xy_addr(q, F.w); st_free(&F.fre[ad], F.ang[ad]); L.glob[i] = last; L.glob[n - 1] = q; } });
                W::sync();
                --n; --i;
            }
        }
        V.a.wx = WIN_NONE; V.b.wx = WIN_NONE;
        if (n < 2) return false;
        to_rect<W>(F, L, n, reg_angle, prec, p, rec);
        density = double(n) / (rg::dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
}

// flsd's seed loop lsd.cpp:477-505 up to rect_improve: the rectangles that reach it are left for lsd_rg_improve
template <class W> RGS_FN void run_frame(const Frame &F, List &L) {
    const double prec = rg::PI_ * rg::ANG_TH / 180, p = rg::ANG_TH / 180;
    int n_grow = 0, n_reg = 0, fetches = 0, n_cand = 0;
    bool overflow = false;
    Wins V;
    V.a.wx = WIN_NONE; V.a.wy = 0; V.b.wx = WIN_NONE; V.b.wy = 0; V.a.am = 0; V.b.am = 0; V.a.am_ok = false; V.b.am_ok = false;
    W::each([&](int l) { V.a.ar[l] = FAR; V.a.pc[l] = 0; V.a.ps[l] = 0; V.b.ar[l] = FAR; V.b.pc[l] = 0; V.b.ps[l] = 0; });
    RGS_T0(5);
    for (int i0 = 0; i0 < F.ne && !overflow; i0 += 64) {
        RGS_T0(4);
        Seeds S;
        PerLane<int> iso, sxy; PerLane<float> dg, sc, ss; PerLane<bool> fr;
        W::each([&](int l) {
            const int idx = i0 + l;
            S.sa[l] = -1; iso[l] = 0; sxy[l] = 0; dg[l] = NOTDEF_F; sc[l] = 0; ss[l] = 0; fr[l] = false;
            if (idx < F.ne) { const int ca = F.caddr[idx]; S.sa[l] = ca & 0x7fffffff; iso[l] = ca < 0; { const int yy = S.sa[l] / F.w; sxy[l] = xy_pack(S.sa[l] - yy * F.w, yy); } sc[l] = F.seed_cs[2 * idx]; ss[l] = F.seed_cs[2 * idx + 1]; dg[l] = F.ang[S.sa[l]]; fr[l] = ld_free(&F.fre[S.sa[l]]) != NOTDEF_F; }
        });
        S.freem = W::ballot(fr);
        RGS_T1(4);
        bool reload = false;
        int pos = 0;
        while (pos < 64) {
            if (reload) { // a refinement gave pixels back: some of the seeds ahead may be free again
                W::each([&](int l) { fr[l] = S.sa[l] >= 0 && ld_free(&F.fre[S.sa[l]]) != NOTDEF_F; });
                S.freem = W::ballot(fr);
                reload = false;
            }
            const u64 m = S.freem & (~0ull << pos);
            if (!m) break;
            const int j = ctz64(m);
            pos = j + 1;
            const int saddr = W::bc(S.sa, j), sq = W::bc(sxy, j), sx = sq & 0xffff, sy = sq >> 16;
            if (W::bc(iso, j)) { // no neighbour is aligned with this pixel's own angle: a region of one pixel
                W::each([&](int l) { if (l == j) st_free(&F.fre[saddr], NOTDEF_F); });
                win_strike<W>(V.a, sx, sy); win_strike<W>(V.b, sx, sy);
                S.freem &= ~(1ull << j);
                continue;
            }
            const float sdeg = W::bc(dg, j), scos = W::bc(sc, j), ssin = W::bc(ss, j);
            int n; double reg_angle;
            RGS_T0(2);
            grow<W>(F, V, L, n, reg_angle, prec, sx, sy, sdeg, scos, ssin, S, overflow, fetches);
            RGS_T1(2);
            n_grow++;
            if (overflow) break;
            if (n < F.min_reg_size) continue;
            n_reg++;
            rg::Rect rec;
            RGS_T0(3);
            to_rect<W>(F, L, n, reg_angle, prec, p, rec);
            bool released = false;
            const bool ok = refine<W>(F, V, L, n, reg_angle, prec, p, rec, sx, sy, sdeg, scos, ssin, S, overflow, fetches, released);
            if (released) reload = true;
            RGS_T1(3);
            if (overflow) break;
            if (!ok) continue;
            if (n_cand >= F.cand_cap) { overflow = true; break; }
            W::each([&](int l) {
                if (l == 0) { double *o = F.rect + (size_t)n_cand * 12; o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y; o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy; o[10] = rec.prec; o[11] = rec.p; }
            });
            ++n_cand;
        }
    }
    RGS_T1(5);
    W::each([&](int l) { if (l == 0) { F.status[0] = n_grow; F.status[1] = overflow ? 1 : 0; F.status[2] = n_reg; F.status[3] = fetches; *F.cand_cnt = n_cand; } });
}
} // namespace rgs
