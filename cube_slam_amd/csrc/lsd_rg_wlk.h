// lsd_rg_wlk.h -- the LSD region stage (flsd's seed loop, region_grow, region2rect, refine, reduce_region_radius; lsd.cpp:464-871) as a WORKGROUP OF TWO
// ROLES.  The reference's sequence inside a frame is untouched (seeds in raster order, every region seeing the marks of all regions before it, every
// accepted pixel changing the angle the next test uses), so the result is exact by construction; what is new against round 4's lane-per-frame walk (every
// phase of the state machine in one loop) is WHO executes which phase:
//
//   * WALKER waves: one lane per frame, and a lane only ever seeds and grows.  An iteration is one memory round trip (the 3 x 3 neighbourhood of the
//     list pixel: three 12-byte loads from the frame's one-float-per-pixel map, or four seed candidates) followed by the alignment tests of the
//     eight neighbours and AT MOST `ACC` accepted pixels; a list pixel with more aligned neighbours simply stays for another iteration, with its
//     neighbourhood in registers.  So the body a wave executes per iteration is tests + one accept + one seed step -- about a quarter of that earlier loop's, where
//     eight predicated accept slots and six pass phases ran in every iteration because some lane of the 64 was always in each of them.
//   * RECTANGLE waves: when a region reaches the rectangle stage (min_reg_size pixels; ~400 of a frame's ~15 000 growths) the lane PARKS: it posts the
//     region through a mailbox in LDS and polls for the answer.  A rectangle wave of the same workgroup takes the job with all 64 lanes -- region2rect's
//     and refine's ordered double sums fed from the lanes exactly like lsd_rg_seq.h, reduce_region_radius as a compaction that leaves the list in
//     the order of the reference's swaps -- and answers "go on with the seeds" or "grow again from the seed with tolerance tau" (refine, :811-815).
//
// Both roles sit on ONE CU: the hand-over needs workgroup-scope ordering only (the waves share the CU's L1; no agent-scope release / acquire, no
// cross-XCD traffic), and no wave ever waits for a wave of another workgroup.
// Written once for the device and for a host model (tools/lsd_sim/wlk_sim.cpp: the lanes as loops, the mailbox served after every iteration).
#pragma once
#include "glibc_sincosf.h"
#include "lsd_rg_seq.h"
#include <cstring>
#if defined(RGL_STATS)
#include <cstdio>
#endif

namespace rgl { // the batch layout of the one-lane-per-frame walk and its memory helpers
using rg::u64;
using rgs::NOTDEF_F;
using rgs::PerLane;

struct Ent { int xy; float deg; };             // one pixel of a region: x | y << 16, level-line angle (float degrees)
constexpr int CAP = 32768;                     // pixels of one region; a larger one sends the batch to the host stage


struct Batch { // wave-uniform; every pointer is the slice of the launch's frames (all offsets inside a slice fit 32 bits: the host cuts a batch into slices)
    int F, w, h, npx;
    int ang_stride, list_stride, rect_stride; // elements from one frame's map / list / rectangles to the next
    int ang_head;                             // floats in front of the slice's frame 0 (>= w + 2, NOTDEF_F like the gaps between the frames)
    const int *order;                         // the frames of the slice sorted by their number of defined pixels (frames of similar work share a wave), or NULL
    const int *caddr; const int *frame_base;  // defined pixels in address order (bit 31: "stays alone as a seed"), all frames one after the other (4 entries of slack behind the last)
    float *ang;                               // dense: the level-line angle in float degrees while the pixel is defined and unused, NOTDEF_F otherwise; frame f at ang_head + f * ang_stride
    const double *mod;                        // dense gradient norms, frame f at f * npx
    const float *seed_cs;                     // per defined pixel: float(cos(angle)), float(sin(angle)) of the angle as a double (:651-652)
    Ent *list; int list_cap;                  // frame f at f * list_stride (8 entries of slack behind a list)
    double *rect; int cand_cap; int *cand_cnt; // the rectangles that reach rect_improve, per frame in seed order
    int *status;                              // per frame: [0] region_grow calls from the seed loop, [1] failure (capacity), [2] regions at the rectangle stage, [3] walker iterations of the frame
    int min_reg_size; int max_iters;
};

#if defined(__HIP_DEVICE_COMPILE__)
struct LWave {
    template <class Fn> static __device__ __forceinline__ void each(Fn f) { f(int(threadIdx.x & 63)); }
    static __device__ __forceinline__ bool any(const PerLane<bool> &p) { return __ballot(p.v) != 0; }
};
__device__ __forceinline__ void st_ang(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ Ent ld_ent(const Ent *p) { const int2 v = *reinterpret_cast<const int2 *>(p); Ent e; e.xy = v.x; e.deg = __int_as_float(v.y); return e; }
__device__ __forceinline__ void st_ent(Ent *p, int xy, float deg) { *reinterpret_cast<int2 *>(p) = make_int2(xy, __float_as_int(deg)); }
struct I4 { int a, b, c, d; };
__device__ __forceinline__ I4 ld_i4(const int *p) { const int4 v = *reinterpret_cast<const int4 *>(p); return I4{v.x, v.y, v.z, v.w}; } // (4-byte aligned: gfx950 global memory is in unaligned-access mode)
__device__ __forceinline__ void ld_f3(const float *p, float *o) { const float3 v = *reinterpret_cast<const float3 *>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
__device__ __forceinline__ void ld_ent2(const Ent *p, int *xy) { const int4 v = *reinterpret_cast<const int4 *>(p); xy[0] = v.x; xy[1] = v.z; }
#else
struct LWave {
    template <class Fn> static RGS_FN void each(Fn f) { for (int l = 0; l < 64; l++) f(l); }
    static RGS_FN bool any(const PerLane<bool> &p) { for (int l = 0; l < 64; l++) if (p.v[l]) return true; return false; }
};
RGS_FN void st_ang(float *p, float v) { *p = v; }
RGS_FN Ent ld_ent(const Ent *p) { return *p; }
RGS_FN void st_ent(Ent *p, int xy, float deg) { p->xy = xy; p->deg = deg; }
struct I4 { int a, b, c, d; };
RGS_FN I4 ld_i4(const int *p) { I4 v; memcpy(&v, p, 16); return v; }
RGS_FN void ld_f3(const float *p, float *o) { memcpy(o, p, 12); }
RGS_FN void ld_ent2(const Ent *p, int *xy) { xy[0] = p[0].xy; xy[1] = p[1].xy; }
#endif

// byte-offset addressing from a wave-uniform base: the 32-bit offset lets the device use its base + offset form (no 64-bit vector address arithmetic)
template <class T> RGS_FN T *at(T *base, unsigned idx) { return (T *)((char *)base + (size_t)(idx * (unsigned)sizeof(T))); }

RGS_FN int div_w(int a, int w, float rcp_w) { // a / w for 0 <= a < 2^24 (a frame's pixel address), without the integer division sequence
    int q = int(float(a) * rcp_w);
    int r = a - q * w;
    if (r < 0) { q--; r += w; }
    if (r >= w) { q++; }
    return q;
}

} // namespace rgl


namespace rgw {
using rg::u64;
using rgl::at;
using rgl::Batch;
using rgl::Ent;
using rgs::NOTDEF_F;
using rgs::PerLane;

constexpr int SEED_B = 4;
enum : int { W_SEED = 0, W_GROW, W_WAIT, W_DONE };
enum : int { AF_CHECK = 0, AF_REGROWN };
enum : int { ACT_CONT = 0, ACT_REGROW, ACT_FAIL };

// a walker lane's mailbox (LDS on the device): the region it parked with, and the rectangle wave's answer
struct Slot {
    int state;                     // 0: idle, 1: posted, 2: answered
    int fl, n, after, sx, sy, n_cand;
    float sdeg;
    double reg_angle;
    int action, pad; double tau;   // the answer (n_cand is updated in place)
};
template <int NS> struct Mail {
    Slot slot[NS];
    int ring_id[NS], ring_tick[NS]; // the posted slots in posting order: ticket t sits in ring_id[t % NS] once ring_tick[t % NS] == t + 1
    int tail, head, walkers_done;
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// every memory operation of the calling wave has completed (the waves of a workgroup share their CU's L1, so that is all a hand-over inside it needs)
__device__ __forceinline__ void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ void wg_release_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); } // (between two LDS stores of one lane that another wave reads in that order)
#else
RGS_FN int lds_ld(const int *p) { return *p; }
RGS_FN void lds_st(int *p, int v) { *p = v; }
RGS_FN int lds_add(int *p, int v) { const int o = *p; *p += v; return o; }
RGS_FN void wg_release() {}
RGS_FN void wg_acquire() {}
RGS_FN void wg_release_lds() {}
#endif

#if defined(RGW_PROF) && defined(__HIPCC__)
// (development) shader-clock ticks per part of a walker iteration, summed over all walker waves: [0] issuing the loads, [1] waiting for them, [2] growth, [3] seeds, [4] mailbox, [5] iterations;
// rectangle waves: [8] ticks in jobs, [9] jobs, [10] ticks waiting for a ticket
__device__ unsigned long long g_rgw_prof[16];
#endif
#if defined(RGW_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define RGW_TICK(k) do { const unsigned long long rgw_now = clock64(); rgw_acc[k] += rgw_now - rgw_t; rgw_t = rgw_now; } while (0)
#else
#define RGW_TICK(k)
#endif

// =========================================================================================================================================
// The rectangle stage of one parked region, by a whole wave (W = rgs::Wave: the lanes of the calling wave / loops on the host)
// =========================================================================================================================================
struct FrameRefs { float *ang; const double *mod; Ent *list; double *rect; int w; };

// region2rect + get_theta lsd.cpp:690-784 over list[0 .. n): the terms come from the lanes, the sums run in list order
template <class W> RGS_FN void to_rect(const FrameRefs &F, int n, double reg_angle, double prec, double p, rg::Rect &rec) {
    W::sync();
    double x = 0, y = 0, sum = 0;
    for (int b = 0; b < n; b += 64) {
        PerLane<double> xw, yw, wg;
        W::each([&](int l) {
            const int idx = b + l;
            xw[l] = 0; yw[l] = 0; wg[l] = 0;
            if (idx < n) { const int q = F.list[idx].xy, qx = q & 0xffff, qy = q >> 16; const double m = F.mod[qx + qy * F.w]; wg[l] = m; xw[l] = double(qx) * m; yw[l] = double(qy) * m; }
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
            if (idx < n) { const int q = F.list[idx].xy, qx = q & 0xffff, qy = q >> 16; const double m = F.mod[qx + qy * F.w], dx = double(qx) - x, dy = double(qy) - y; t1[l] = dy * dy * m; t2[l] = dx * dx * m; t3[l] = dx * dy * m; }
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
                const int q = F.list[idx].xy;
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

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int clz64(u64 m) { return __clzll((long long)m); }
#else
RGS_FN int clz64(u64 m) { return __builtin_clzll(m); }
#endif

// One round of reduce_region_radius' loop :849-859: "for i: if list[i] is outside the radius, mark it unused, put list[n - 1] in its place, --n, look at i again".
// What that loop leaves: every entry outside is marked unused; the entries inside keep their places; a place < n' (n' = the number of entries inside) that held an
// entry outside receives an entry inside from behind n' -- the k-th such place in ascending order the k-th such entry in DESCENDING order (the loop always pulls
// the list's last entry, and an entry pulled in that is outside itself is pulled over at once).  So: one pass that marks and counts, one that fills the holes.
template <class W> RGS_FN int shrink_round(const FrameRefs &F, int n, double xc, double yc, double radSq) {
    W::sync();
    int n2 = 0;
    for (int b = 0; b < n; b += 64) {
        PerLane<bool> in;
        W::each([&](int l) {
            const int idx = b + l;
            in[l] = false;
            if (idx < n) {
                const Ent e = rgl::ld_ent(&F.list[idx]);
                const double ddx = double(e.xy & 0xffff) - xc, ddy = double(e.xy >> 16) - yc;
                if (ddx * ddx + ddy * ddy > radSq) rgl::st_ang(&F.ang[rgs::xy_addr(e.xy, F.w)], e.deg); // :852 used = NOTUSED
                else in[l] = true;
            }
        });
        n2 += (int)__builtin_popcountll(W::ballot(in));
    }
    if (n2 == n) return n;
    // holes: entries outside at places < n2, ascending; sources: entries inside at places >= n2, descending
    int hb = 0; u64 hmask = 0; int hbase = 0;
    int sb = n2 + ((n - 1 - n2) / 64) * 64 + 64; u64 smask = 0; // (the next source chunk is [sb - 64, sb))
    PerLane<int> sxy; PerLane<float> sdg;
    W::each([&](int l) { sxy[l] = 0; sdg[l] = 0; });
    for (;;) {
        while (!hmask && hb < n2) {
            PerLane<bool> out;
            W::each([&](int l) {
                const int idx = hb + l;
                out[l] = false;
                if (idx < n2) { const int q = F.list[idx].xy; const double ddx = double(q & 0xffff) - xc, ddy = double(q >> 16) - yc; out[l] = ddx * ddx + ddy * ddy > radSq; }
            });
            hmask = W::ballot(out); hbase = hb; hb += 64;
        }
        if (!hmask) break;
        while (!smask) { // (as many sources as holes: there is one)
            sb -= 64;
            PerLane<bool> in;
            W::each([&](int l) {
                const int idx = sb + l;
                in[l] = false;
                if (idx >= n2 && idx < n) { const Ent e = rgl::ld_ent(&F.list[idx]); sxy[l] = e.xy; sdg[l] = e.deg; const double ddx = double(e.xy & 0xffff) - xc, ddy = double(e.xy >> 16) - yc; in[l] = !(ddx * ddx + ddy * ddy > radSq); }
            });
            smask = W::ballot(in);
        }
        const int hole = hbase + rgs::ctz64(hmask); hmask &= hmask - 1;
        const int sl = 63 - clz64(smask); smask &= ~(1ull << sl);
        const int vxy = W::bc(sxy, sl); const float vdg = W::bc(sdg, sl);
        W::each([&](int l) { if (l == 0) rgl::st_ent(&F.list[hole], vxy, vdg); });
    }
    W::sync();
    return n2;
}

// A parked region: region2rect, the density test, and what follows it in refine :786-832 / reduce_region_radius :834-871 up to the next region_grow (the
// walker's) or the rectangle.  Returns the action; tau / n_cand are the answer's other fields.
template <class W> RGS_FN int rect_job(const Batch &B, int fl, int n, int after, int sx, int sy, float sdeg, double reg_angle, int &n_cand, double &tau) {
    const double prec = rg::PI_ * rg::ANG_TH / 180, p = rg::ANG_TH / 180;
    FrameRefs F;
    F.ang = B.ang + (size_t)B.ang_head + (size_t)fl * (size_t)B.ang_stride; F.mod = B.mod + (size_t)fl * (size_t)B.npx; F.list = B.list + (size_t)fl * (size_t)B.list_stride; F.rect = B.rect + (size_t)fl * (size_t)B.rect_stride; F.w = B.w;
    rg::Rect rec;
    to_rect<W>(F, n, reg_angle, prec, p, rec);
    double density = double(n) / (rg::dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    const double xc = double(sx), yc = double(sy);
    if (density < rg::DENSITY_TH) {
        if (after == AF_CHECK) { // refine :798-815: every pixel of the region unused again, the spread of the angles near the seed -> tau; the walker grows again
            const double ang_c = double(sdeg) * rg::DEG_TO_RADS;
            double sum = 0, s_sum = 0;
            int cnt = 0;
            for (int b = 0; b < n; b += 64) {
                PerLane<double> d, dd; PerLane<bool> near;
                W::each([&](int l) {
                    const int idx = b + l;
                    near[l] = false; d[l] = 0; dd[l] = 0;
                    if (idx < n) {
                        const Ent e = rgl::ld_ent(&F.list[idx]);
                        const int qx = e.xy & 0xffff, qy = e.xy >> 16;
                        rgl::st_ang(&F.ang[qx + qy * F.w], e.deg); // :800 used = NOTUSED
                        if (rg::dist(xc, yc, double(qx), double(qy)) < rec.width) { const double a = rg::angle_diff_signed(double(e.deg) * rg::DEG_TO_RADS, ang_c); d[l] = a; dd[l] = a * a; near[l] = true; }
                    }
                });
                u64 m = W::ballot(near);
                while (m) { const int j = rgs::ctz64(m); m &= m - 1; sum += W::bc(d, j); s_sum += W::bc(dd, j); ++cnt; }
            }
            const double mean_angle = sum / double(cnt);
            tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / double(cnt) + mean_angle * mean_angle);
            W::each([&](int l) { if (l == 0) rgl::st_ang(&F.ang[sx + sy * F.w], NOTDEF_F); }); // (region_grow marks its seed :648)
            return ACT_REGROW;
        }
        // reduce_region_radius :836-869
        const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc), r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < rg::DENSITY_TH) {
            radSq *= 0.75 * 0.75;
            n = shrink_round<W>(F, n, xc, yc, radSq);
            if (n < 2) return ACT_CONT; // :862
            to_rect<W>(F, n, reg_angle, prec, p, rec);
            density = double(n) / (rg::dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
    }
    if (n_cand >= B.cand_cap) return ACT_FAIL;
    {
        const int k = n_cand;
        W::each([&](int l) {
            if (l == 0) { double *o = F.rect + (size_t)k * 12; o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y; o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy; o[10] = rec.prec; o[11] = rec.p; }
        });
    }
    ++n_cand;
    return ACT_CONT;
}

// a rectangle wave's turn at ticket t of the mailbox (the caller waited for ring_tick[t % NS] == t + 1)
template <class W, int NS> RGS_FN void serve_ticket(const Batch &B, Mail<NS> &M, int t) {
    wg_acquire();
    const int id = W::uni(lds_ld(&M.ring_id[t % NS]));
    Slot &s = M.slot[id];
    const int fl = W::uni(s.fl), n = W::uni(s.n), after = W::uni(s.after), sx = W::uni(s.sx), sy = W::uni(s.sy);
    int n_cand = W::uni(s.n_cand);
    const float sdeg = s.sdeg; const double reg_angle = W::uni(s.reg_angle);
    double tau = 0;
    const int act = rect_job<W>(B, fl, n, after, sx, sy, sdeg, reg_angle, n_cand, tau);
    W::each([&](int l) { if (l == 0) { s.action = act; s.tau = tau; s.n_cand = n_cand; } });
    wg_release(); // the marks, the list and the rectangle are in memory before the walker reads the answer
    W::each([&](int l) { if (l == 0) lds_st(&s.state, 2); });
}

// =========================================================================================================================================
// The walker: one lane per frame, seeds and growth only
// =========================================================================================================================================
struct WSt {
    int phase, mode, valid, start;
    unsigned ao, lo, cb; int fl, ne;
    int si, par, pref, seed_idx;
    int ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3;
    int n, i, fcnt, fq0, fq1, fq2, fq3;
    double reg_angle, prec; float sumdx, sumdy;
    int sx, sy; float sdeg, scos, ssin;
    float nb[9]; int kpos, base, need_nb;
    int n_cand, n_grow, n_reg, fail, it_done, cap;
#if defined(RGL_STATS)
    long ph_iters[4], accepts, stays;
#endif
};
struct WIt { // one iteration's loads (declared inside the loop WITHOUT initial values: a load's destination that is merged with a default at the end of a branch makes the compiler wait there)
    int cur[SEED_B], adx[SEED_B], cval[SEED_B]; float sfd[SEED_B];
    int refill, n_issue, rq[2];
};

RGS_FN unsigned aligned_mask(const float *nb, double reg_angle, double prec) { // the eight neighbours in the reference's order (yy outer, xx inner :660-668); bit 4 = the list pixel itself stays clear
    unsigned m = 0; // (no short-circuit: eight straight-line tests, no branch per neighbour)
#pragma unroll
    for (int sl = 0; sl < 9; sl++) {
        if (sl == 4) continue;
        const float fd = nb[sl];
        const bool al = rgs::aligned_rad(double(fd) * rg::DEG_TO_RADS, reg_angle, prec);
        m |= (unsigned)((fd != NOTDEF_F) & al) << sl;
    }
    return m;
}
RGS_FN int ctz32(unsigned m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)m) - 1;
#else
    return __builtin_ctz(m);
#endif
}

// M: the mailbox policy -- Mail<NS> &mail(), int slot_of(lane), void after_iteration() (the host model serves the posted regions there)
template <class W, int ACC, class MP> RGS_FN void run_walker(const Batch &B, int f0, MP &mp) {
    const double PREC = rg::PI_ * rg::ANG_TH / 180;
    const float rcp_w = 1.0f / float(B.w);
    const int cb0 = B.frame_base[0];
    float *const angw = B.ang;
    const int *const caddrw = B.caddr;
    const float *const scsw = B.seed_cs;
    Ent *const listw = B.list;
    const int Bw = B.w, list_cap = B.list_cap, min_reg_size = B.min_reg_size, max_iters = B.max_iters;
    auto &M = mp.mail();

    PerLane<WSt> st;
    W::each([&](int l) {
        WSt &s = st[l];
        s.valid = f0 + l < B.F;
        s.fl = s.valid ? (B.order ? B.order[f0 + l] : f0 + l) : 0;
        s.ao = (unsigned)B.ang_head + (unsigned)s.fl * (unsigned)B.ang_stride; s.lo = (unsigned)s.fl * (unsigned)B.list_stride;
        s.cb = 0; s.ne = 0;
        if (s.valid) { const int b = B.frame_base[s.fl]; s.cb = (unsigned)(b - cb0); s.ne = B.frame_base[s.fl + 1] - b; }
        s.phase = (s.valid && s.ne > 0) ? W_SEED : W_DONE;
        s.mode = 0; s.start = 0;
        s.si = 0; s.par = 0; s.pref = 0; s.seed_idx = 0;
        s.ca0 = s.ca1 = s.ca2 = s.ca3 = s.cb0 = s.cb1 = s.cb2 = s.cb3 = 0;
        if (s.phase == W_SEED) { const rgl::I4 a = rgl::ld_i4(at(caddrw, s.cb)), b = rgl::ld_i4(at(caddrw, s.cb + (unsigned)SEED_B)); s.ca0 = a.a; s.ca1 = a.b; s.ca2 = a.c; s.ca3 = a.d; s.cb0 = b.a; s.cb1 = b.b; s.cb2 = b.c; s.cb3 = b.d; }
        s.n = 0; s.i = 0; s.fcnt = 0; s.fq0 = s.fq1 = s.fq2 = s.fq3 = 0; s.reg_angle = 0; s.prec = PREC; s.sumdx = 0; s.sumdy = 0;
        s.sx = 0; s.sy = 0; s.sdeg = 0; s.scos = 0; s.ssin = 0;
        for (int u = 0; u < 9; u++) s.nb[u] = NOTDEF_F;
        s.kpos = 0; s.base = 0; s.need_nb = 0;
        s.n_cand = 0; s.n_grow = 0; s.n_reg = 0; s.fail = 0; s.it_done = 0; s.cap = list_cap;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(s.cap)); // (a vector register from here on: as a kernel argument the compiler re-reads it from scalar memory inside the loop)
#endif
#if defined(RGL_STATS)
        for (int q = 0; q < 4; q++) s.ph_iters[q] = 0;
        s.accepts = 0; s.stays = 0;
#endif
        M.slot[mp.slot_of(l)].fl = s.fl;
    });

#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0f70); // (the first seed addresses: nothing is in flight when the loop starts, or every iteration would carry the waits of the first)
#endif
    int iters = 0;
#if defined(RGW_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long rgw_acc[6] = {0, 0, 0, 0, 0, 0}, rgw_t = clock64();
#endif
    for (;;) {
        {
            PerLane<bool> on;
            W::each([&](int l) { WSt &s = st[l]; on[l] = s.phase != W_DONE; if (on[l]) s.it_done = iters + 1;
#if defined(RGL_STATS)
                s.ph_iters[s.phase]++;
#endif
            });
            if (!rgl::LWave::any(on)) break;
        }
        if (++iters > max_iters) W::each([&](int l) { WSt &s = st[l]; if (s.phase != W_DONE && s.phase != W_WAIT) { s.fail = 1; s.phase = W_DONE; } }); // (a parked lane first takes its answer)

#if defined(__HIP_DEVICE_COMPILE__)
        PerLane<WIt> itv;
#else
        PerLane<WIt> itv; memset(&itv, 0, sizeof(itv));
#endif
        // ---- every load of the iteration, none depending on another
        PerLane<bool> growing, seeding, waiting;
        W::each([&](int l) {
            WSt &s = st[l]; WIt &t = itv[l];
            growing[l] = s.phase == W_GROW; seeding[l] = s.phase == W_SEED; waiting[l] = s.phase == W_WAIT;
            if (seeding[l]) {
                t.cur[0] = s.par ? s.cb0 : s.ca0; t.cur[1] = s.par ? s.cb1 : s.ca1; t.cur[2] = s.par ? s.cb2 : s.ca2; t.cur[3] = s.par ? s.cb3 : s.ca3;
                for (int k = 0; k < SEED_B; k++) {
                    t.cval[k] = s.si + k < s.ne;
                    t.adx[k] = t.cval[k] ? (t.cur[k] & 0x7fffffff) : 0;
                    t.sfd[k] = *at(angw, s.ao + (unsigned)t.adx[k]);
                }
            }
            if (s.pref) { // seed addresses two batches ahead (entries behind the frame's last are never looked at)
                const bool both = s.pref == 2;
                if (both || s.par == 1) { const rgl::I4 a = rgl::ld_i4(at(caddrw, s.cb + (unsigned)(both ? s.si : s.si + SEED_B))); s.ca0 = a.a; s.ca1 = a.b; s.ca2 = a.c; s.ca3 = a.d; }
                if (both || s.par == 0) { const rgl::I4 b = rgl::ld_i4(at(caddrw, s.cb + (unsigned)(s.si + SEED_B))); s.cb0 = b.a; s.cb1 = b.b; s.cb2 = b.c; s.cb3 = b.d; }
                s.pref = 0;
            }
            if (growing[l]) {
                if (s.start && s.mode == 0) { const float *cs = at(scsw, 2u * (s.cb + (unsigned)s.seed_idx)); s.scos = cs[0]; s.ssin = cs[1]; } // a new region: what its seed starts the sums with (:651-652)
                t.refill = s.fcnt <= 1 && s.i + s.fcnt < s.n;
                t.n_issue = s.n;
                if (t.refill) rgl::ld_ent2(at(listw, s.lo + (unsigned)(s.i + s.fcnt)), t.rq);
                if (s.need_nb) {
                    const int q = s.fq0, px = q & 0xffff, py = q >> 16;
                    s.base = py * Bw + px;
                    // the 3 x 3 neighbourhood, three rows of three floats from column px - 1 (a frame's last row and column are undefined, "row -1" / "column -1" fall on them)
                    // (straight into the lane's state: the list pixel before has left)
                    rgl::ld_f3(at(angw, s.ao + (unsigned)(s.base - Bw - 1)), s.nb);
                    rgl::ld_f3(at(angw, s.ao + (unsigned)(s.base - 1)), s.nb + 3);
                    rgl::ld_f3(at(angw, s.ao + (unsigned)(s.base + Bw - 1)), s.nb + 6);
                    s.kpos = 0; s.need_nb = 0;
                }
            }
        });
        RGW_TICK(0);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): nothing is in flight across the back edge
#endif
        RGW_TICK(1);

        PerLane<bool> park;
        W::each([&](int l) { park[l] = false; });
        // ---- region_grow :660-686: the list pixel's eight neighbours in the reference's order, at most ACC of them accepted per iteration
        if (rgl::LWave::any(growing)) {
            W::each([&](int l) {
                WSt &s = st[l]; WIt &t = itv[l];
                if (!growing[l]) return;
                if (s.start) { s.sumdx = s.scos; s.sumdy = s.ssin; s.start = 0; } // :651-652
                if (t.refill) { // (the list's end is not in the registers, so no entry joins them in this iteration)
                    const int r0 = s.i + s.fcnt, got = t.n_issue - r0 < 2 ? t.n_issue - r0 : 2;
                    if (s.fcnt == 0) { s.fq0 = t.rq[0]; s.fq1 = t.rq[1]; } else { s.fq1 = t.rq[0]; s.fq2 = t.rq[1]; }
                    s.fcnt += got;
                }
                const int q = s.fq0, px = q & 0xffff, py = q >> 16;
                unsigned m = aligned_mask(s.nb, s.reg_angle, s.prec) & (~0u << s.kpos);
#pragma unroll
                for (int a = 0; a < ACC; a++) {
                    if (m && !s.fail) { // :669-683
                        if (s.n >= s.cap) { s.fail = 1; break; } // the region outgrew its list: the frame gives up
                        const int sl = ctz32(m);
                        float fd = s.nb[0];
#pragma unroll
                        for (int u = 1; u < 9; u++) if (sl == u) fd = s.nb[u];
                        const int r3 = (sl * 11) >> 5, dy = r3 - 1, dx = sl - 3 * r3 - 1, cxy = rgs::xy_pack(px + dx, py + dy);
                        rgl::st_ang(at(angw, s.ao + (unsigned)(s.base + dy * Bw + dx)), NOTDEF_F);
                        rgl::st_ent(at(listw, s.lo + (unsigned)s.n), cxy, fd);
                        if (s.i + s.fcnt == s.n && s.fcnt < 4) { // the list's end is in the registers: so is this entry
                            if (s.fcnt == 1) s.fq1 = cxy; else if (s.fcnt == 2) s.fq2 = cxy; else if (s.fcnt == 3) s.fq3 = cxy; else s.fq0 = cxy;
                            s.fcnt++;
                        }
                        s.n++;
                        float cc, sn;
                        glibc_sincosf::sincosf_pos(float(double(fd) * rg::DEG_TO_RADS), &sn, &cc); // cos(float(angle)), sin(float(angle)) :676-677 with glibc's values
                        s.sumdx += cc; s.sumdy += sn;
                        s.reg_angle = rg::fast_atan2_1(s.sumdy, s.sumdx) * rg::DEG_TO_RADS;
                        s.kpos = sl + 1;
                        m = aligned_mask(s.nb, s.reg_angle, s.prec) & (~0u << s.kpos);
#if defined(RGL_STATS)
                        s.accepts++;
#endif
                    }
                }
                if (s.fail) { s.phase = W_DONE; return; }
                if (m) {
#if defined(RGL_STATS)
                    s.stays++;
#endif
                    return; // more aligned neighbours: the pixel stays for another iteration
                }
                // the next list pixel
                s.i++; s.fq0 = s.fq1; s.fq1 = s.fq2; s.fq2 = s.fq3; s.fcnt--; s.need_nb = 1;
                if (s.i >= s.n) { // the region is complete
                    if (s.mode == 0) {
                        s.n_grow++;
                        if (s.n < min_reg_size) s.phase = W_SEED; // :489
                        else { s.n_reg++; park[l] = true; }
                    } else {
                        if (s.n < 2) s.phase = W_SEED; // :817
                        else park[l] = true;
                    }
                }
            });
        }

        RGW_TICK(2);
        // ---- the seed loop :477-487: four candidates
        if (rgl::LWave::any(seeding)) {
            W::each([&](int l) {
                WSt &s = st[l]; WIt &t = itv[l];
                if (!seeding[l]) return;
                int k0 = -1;
                for (int k = 0; k < SEED_B; k++)
                    if (k0 < 0 && t.cval[k] && t.sfd[k] != NOTDEF_F) {
                        if (t.cur[k] < 0) rgl::st_ang(at(angw, s.ao + (unsigned)t.adx[k]), NOTDEF_F); // a region of one pixel (flagged by lsd_emit): used, nothing else
                        else k0 = k;
                    }
                if (k0 >= 0) { // region_grow from this seed :637-657
                    int sadx = t.adx[0]; float sdeg = t.sfd[0];
                    for (int k = 1; k < SEED_B; k++) if (k0 == k) { sadx = t.adx[k]; sdeg = t.sfd[k]; }
                    s.sy = rgl::div_w(sadx, Bw, rcp_w); s.sx = sadx - s.sy * Bw; s.sdeg = sdeg;
                    s.seed_idx = s.si + k0;
                    rgl::st_ang(at(angw, s.ao + (unsigned)sadx), NOTDEF_F);
                    rgl::st_ent(at(listw, s.lo), rgs::xy_pack(s.sx, s.sy), s.sdeg); // the seed's entry
                    s.n = 1; s.i = 0; s.fcnt = 1; s.fq0 = rgs::xy_pack(s.sx, s.sy); s.need_nb = 1; s.start = 1;
                    s.reg_angle = double(s.sdeg) * rg::DEG_TO_RADS; s.prec = PREC; s.mode = 0;
                    s.phase = W_GROW;
                    s.si = s.seed_idx + 1; s.par = 0; s.pref = 2;
                } else {
                    s.si += SEED_B; s.par ^= 1; s.pref = 1;
                    if (s.si >= s.ne) { s.phase = W_DONE; s.pref = 0; }
                }
            });
        }

        RGW_TICK(3);
        // ---- parked regions: the answers that have arrived, then this iteration's new posts
        if (rgl::LWave::any(waiting)) {
            PerLane<bool> got;
            W::each([&](int l) { got[l] = waiting[l] && lds_ld(&M.slot[mp.slot_of(l)].state) == 2; });
            if (rgl::LWave::any(got)) {
                wg_acquire();
                W::each([&](int l) {
                    if (!got[l]) return;
                    WSt &s = st[l]; Slot &b = M.slot[mp.slot_of(l)];
                    const int act = b.action;
                    s.n_cand = b.n_cand;
                    lds_st(&b.state, 0);
                    if (act == ACT_FAIL) { s.fail = 1; s.phase = W_DONE; }
                    else if (act == ACT_CONT) s.phase = W_SEED;
                    else { // :811-815 grow again from the seed with the tolerance tau
                        s.n = 1; s.i = 0; s.fcnt = 1; s.fq0 = rgs::xy_pack(s.sx, s.sy); s.need_nb = 1; s.start = 1;
                        s.reg_angle = double(s.sdeg) * rg::DEG_TO_RADS; s.prec = b.tau; s.mode = 1;
                        s.phase = W_GROW;
                    }
                });
            }
        }
        if (rgl::LWave::any(park)) {
            W::each([&](int l) {
                if (!park[l]) return;
                WSt &s = st[l]; Slot &b = M.slot[mp.slot_of(l)];
                b.n = s.n; b.after = s.mode == 0 ? AF_CHECK : AF_REGROWN; b.sx = s.sx; b.sy = s.sy; b.sdeg = s.sdeg; b.n_cand = s.n_cand; b.reg_angle = s.reg_angle;
                lds_st(&b.state, 1);
                s.phase = W_WAIT;
            });
            wg_release(); // the region's marks and list entries are in memory, the slots are written
            W::each([&](int l) {
                if (!park[l]) return;
                const int t = lds_add(&M.tail, 1);
                lds_st(&M.ring_id[t % mp.ring_size()], mp.slot_of(l));
                wg_release_lds();
                lds_st(&M.ring_tick[t % mp.ring_size()], t + 1);
            });
        }
        mp.after_iteration(B);
        RGW_TICK(4);
    }
#if defined(RGW_PROF) && defined(__HIP_DEVICE_COMPILE__)
    rgw_acc[5] = (unsigned long long)iters;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) atomicAdd(&g_rgw_prof[k], rgw_acc[k]);
#endif
    W::each([&](int l) {
        const WSt &s = st[l];
        if (s.valid) { int *o = B.status + 4 * (size_t)s.fl; o[0] = s.n_grow; o[1] = s.fail; o[2] = s.n_reg; o[3] = s.it_done; B.cand_cnt[s.fl] = s.n_cand;
#if defined(RGL_STATS)
            printf("  frame %d iterations by phase: seed %ld grow %ld (stays %ld) wait %ld; accepts %ld\n", s.fl, s.ph_iters[0], s.ph_iters[1], s.stays, s.ph_iters[2], s.accepts);
#endif
        }
    });
}
} // namespace rgw
