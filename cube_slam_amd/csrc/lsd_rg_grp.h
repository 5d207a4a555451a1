// lsd_rg_grp.h -- the LSD region stage (flsd's seed loop, region_grow, region2rect, refine, reduce_region_radius; lsd.cpp:464-871) with SEVERAL
// FRAMES PER WAVE: a group of G = 8 * P lanes walks one frame's sequence, 64 / G frames share a wave.  The sequence inside a frame is the
// reference's, untouched (seeds in raster order, every region sees the marks of all regions before it, every accepted pixel changes the angle the
// next test uses), so the result is exact by construction; what changes against lsd_rg_seq.h (one wave per frame) is what an issued instruction
// buys: there a frame's bookkeeping ran on the scalar unit and its vector instructions served one frame (14.3 M issued instructions per frame);
// here the bookkeeping of 64 / G frames is ONE vector instruction stream (the state of a frame is replicated in its G lanes) and only the
// neighbour tests are spread over the lanes (about 3.5 M issued instructions per frame with G = 8):
//   * region_grow: lane j of a group loads neighbour j of the list pixel -- ONE FLOAT, the level-line angle while the pixel is defined and
//     unused; the whole map of a frame is 4 bytes a pixel, so that the rows a walk is busy with stay in the caches with thousands of frames in
//     flight.  With P = 2 the second eight lanes load the neighbours of the NEXT list pixel in the same round trip.  The eight (sixteen)
//     alignment tests are one ballot, taken in ascending lane order = the reference's order; an accepted pixel is broadcast inside the group
//     (three DPP steps), changes the region angle -- cosf / sinf of its angle are computed in place (glibc's values, glibc_sincosf.h) -- is
//     struck from the lanes that hold it and the remaining lanes are tested again: the loop runs once per accepted pixel of the busiest group;
//   * the region list (8 bytes per pixel: coordinates, angle) lives in global memory, its next G entries also in one register across the
//     group's lanes (refilled by one load per lane when it runs low);
//   * the seed loop tests G seeds per round trip (their addresses are fetched two batches ahead, while the previous region grows);
//   * region2rect / refine / reduce_region_radius are sequential passes over the list, eight entries per iteration, every lane of the group
//     computing the same ordered double sums; the gradient norms are gathered from the dense map one iteration after the coordinates.
// A wave runs ONE loop: every iteration each frame does one step of whatever phase it is in (a state machine per group); ALL loads of an
// iteration are issued before the first use, so an iteration costs one memory round trip however many phases are active in the wave.
// Written once for the device and for a host model (tools/lsd_sim/grp_sim.cpp: the 64 lanes as loops): per-lane values are PerLane<T>,
// per-lane code sits in W::each bodies, cross-lane traffic goes through W::gmask / W::gpick / W::any.
#pragma once
#include "glibc_sincosf.h"
#include "lsd_rg_seq.h"
#include <cstring>
#if defined(RGG_STATS)
#include <cstdio>
#endif

namespace rgg {
using rg::u64;
using rgs::NOTDEF_F;
using rgs::PerLane;

struct Ent { int xy; float deg; };             // one pixel of a region: x | y << 16, level-line angle (float degrees)
constexpr int CAP = 32768;                     // pixels of one region; a larger one sends the batch to the host stage
constexpr int PASS_B = 8;                      // list entries per iteration of a pass
constexpr int SHRINK_B = 4;                    // list entries per iteration of reduce_region_radius (an iteration ends at the first entry it removes)
enum : int { PH_SEED = 0, PH_GROW, PH_P0, PH_P1, PH_P2, PH_P3, PH_STAT, PH_SHRINK, PH_DONE };
enum : int { AF_CHECK = 0, AF_REGROWN, AF_SHRUNK };


#if defined(RGG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
// (development) shader-clock ticks per part of the iteration, summed per wave: prof[16 * wave + k]
#define RGG_T(k) do { const unsigned long long rgg_now = clock64(); rgg_acc[k] += rgg_now - rgg_t; rgg_t = rgg_now; } while (0)
#define RGG_C(k) do { rgg_acc[k] += 1; } while (0)
#else
#define RGG_T(k)
#define RGG_C(k)
#endif
struct Batch { // wave-uniform; every pointer is the slice of the launch's frames (all offsets inside a slice fit 32 bits: the host cuts a batch into slices)
    int F, w, h, npx;
    int ang_stride, list_stride, rect_stride; // elements from one frame's map / list / rectangles to the next (padded: not a multiple of a large power of two)
    const int *order;                         // the frames of the slice sorted by their number of defined pixels (frames of similar work share a wave), or NULL
    const int *caddr; const int *frame_base;  // defined pixels in address order (bit 31: "stays alone as a seed"), all frames one after the other
    float *ang;                               // dense: the level-line angle in float degrees while the pixel is defined and unused, NOTDEF_F otherwise; frame f at f * ang_stride
    const double *mod;                        // dense gradient norms, frame f at f * npx
    const float *seed_cs;                     // per defined pixel: float(cos(angle)), float(sin(angle)) of the angle as a double (:651-652)
    Ent *list; int list_cap;                  // frame f at f * list_stride
    double *rect; int cand_cap; int *cand_cnt; // the rectangles that reach rect_improve, per frame in seed order
    int *status;                              // per frame: [0] region_grow calls from the seed loop, [1] failure (capacity), [2] regions at the rectangle stage, [3] iterations of the frame
    int min_reg_size; int max_iters;
    unsigned long long *prof; // RGG_PROFILE
};

struct St { // a frame's walk; identical in the G lanes of its group except ring / c0 / c1
    int phase, after, mode, valid;
    unsigned ao, mo, lo, cb, ro; // element offsets of the frame inside the slice's map, norms, lists, caddr / seed_cs, rectangles
    int fl, ne;
    int si, c0, c1, par, pref, seed_idx; // seed cursor; c[par] = caddr[si + jj], c[par ^ 1] = caddr[si + G + jj]; pref: 1 = fetch c[par ^ 1], 2 = fetch both (par = 0) in the next load stage; the rank of the region's seed
    int n, i, rhi, ring;     // region size, list cursor, ring holds entries [i, rhi)
    double reg_angle, prec; float sumdx, sumdy;
    int sx, sy; float sdeg, scos, ssin;
    int k;
    int pxy[PASS_B];         // the coordinates of the pass block whose norms the next load stage gathers
    double a0, a1, a2, cx, cy, theta, dx, dy, lmin, lmax, wmin, wmax;
    double width; int scnt;  // the last rectangle's width (refine's pass needs it); the pass of refine sums into a0 / a1
    double radSq;
    int n_cand, n_grow, n_reg, fail, it_done;
    int cap; // list_cap, per lane: as a scalar the compiler re-reads it from the kernel arguments inside region_grow's loop (a scalar-memory round trip per accepted pixel)
#if defined(RGG_STATS)
    long ph_iters[9], acc_rounds;
#endif
};
struct It { // one iteration's loads and what follows from them.  Declared inside the loop WITHOUT initial values: a load's destination that is merged with a
            // default at the end of a branch makes the compiler wait for the load there, and the loads of an iteration would run one after the other
    int live, act, isfree, cur, nxy, naddr, inb, refill, do_ld, rl, n_issue, cval, adx, ccur;
    float fd, sfd; double arad;
    Ent ent[PASS_B]; Ent elast; double pm[PASS_B];
};

#if defined(__HIP_DEVICE_COMPILE__)
template <int G> struct GWave {
    template <class Fn> static __device__ __forceinline__ void each(Fn f) { f(int(threadIdx.x & 63)); }
    static __device__ __forceinline__ bool any(const PerLane<bool> &p) { return __ballot(p.v) != 0; }
    static __device__ __forceinline__ PerLane<unsigned> gmask(const PerLane<bool> &p) { // the G predicate bits of the lane's group
        const u64 b = __ballot(p.v);
        PerLane<unsigned> r; r.v = (unsigned)(b >> (threadIdx.x & 63 & ~(G - 1))) & ((1u << G) - 1u);
        return r;
    }
    static __device__ __forceinline__ unsigned gor(unsigned x) { // OR over the lanes of a group, result in every lane
        x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
        x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
        x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true); // row_half_mirror
        if (G == 16) x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true); // row_mirror
        return x;
    }
    static __device__ __forceinline__ PerLane<unsigned> gpick(const PerLane<unsigned> &v, const PerLane<bool> &sel) { // the value of the group's selected lane (at most one), 0 without one
        PerLane<unsigned> r; r.v = gor(sel.v ? v.v : 0u);
        return r;
    }
    static __device__ __forceinline__ int ctz(unsigned m) { return __ffs((int)m) - 1; }
};
__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ void st_ang(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ Ent ld_ent(const Ent *p) { const int2 v = *reinterpret_cast<const int2 *>(p); Ent e; e.xy = v.x; e.deg = __int_as_float(v.y); return e; }
__device__ __forceinline__ void st_ent(Ent *p, int xy, float deg) { *reinterpret_cast<int2 *>(p) = make_int2(xy, __float_as_int(deg)); }
#else
template <int G> struct GWave {
    template <class Fn> static RGS_FN void each(Fn f) { for (int l = 0; l < 64; l++) f(l); }
    static RGS_FN bool any(const PerLane<bool> &p) { for (int l = 0; l < 64; l++) if (p.v[l]) return true; return false; }
    static RGS_FN PerLane<unsigned> gmask(const PerLane<bool> &p) {
        PerLane<unsigned> r;
        for (int l = 0; l < 64; l++) { unsigned m = 0; const int b = l & ~(G - 1); for (int q = 0; q < G; q++) if (p.v[b + q]) m |= 1u << q; r.v[l] = m; }
        return r;
    }
    static RGS_FN PerLane<unsigned> gpick(const PerLane<unsigned> &v, const PerLane<bool> &sel) {
        PerLane<unsigned> r;
        for (int l = 0; l < 64; l++) { unsigned m = 0; const int b = l & ~(G - 1); for (int q = 0; q < G; q++) if (sel.v[b + q]) m |= v.v[b + q]; r.v[l] = m; }
        return r;
    }
    static RGS_FN int ctz(unsigned m) { return __builtin_ctz(m); }
};
RGS_FN unsigned f2u(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
RGS_FN float u2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
RGS_FN void st_ang(float *p, float v) { *p = v; }
RGS_FN Ent ld_ent(const Ent *p) { return *p; }
RGS_FN void st_ent(Ent *p, int xy, float deg) { p->xy = xy; p->deg = deg; }
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

// the walk of the frames order[f0 ..] (or f0 ..) of a slice: one wave
template <int P, class W> RGS_FN void run_wave(const Batch &B, int f0) {
    constexpr int G = 8 * P;
    const double PREC = rg::PI_ * rg::ANG_TH / 180, PP = rg::ANG_TH / 180;
    const float rcp_w = 1.0f / float(B.w);
    const int cb0 = B.frame_base[0];
    float *const angw = B.ang;
    const double *const modw = B.mod;
    const int *const caddrw = B.caddr;
    const float *const scsw = B.seed_cs;
    Ent *const listw = B.list;
    double *const rectw = B.rect;
    const int Bw = B.w, Bh = B.h, list_cap = B.list_cap, min_reg_size = B.min_reg_size, cand_cap = B.cand_cap, max_iters = B.max_iters;

    PerLane<St> st;
    W::each([&](int l) {
        St &s = st[l];
        const int g = l / G, jj = l % G;
        s.valid = f0 + g < B.F;
        s.fl = s.valid ? (B.order ? B.order[f0 + g] : f0 + g) : 0;
        s.ao = (unsigned)s.fl * (unsigned)B.ang_stride; s.mo = (unsigned)s.fl * (unsigned)B.npx; s.lo = (unsigned)s.fl * (unsigned)B.list_stride; s.ro = (unsigned)s.fl * (unsigned)B.rect_stride;
        s.cb = 0; s.ne = 0;
        if (s.valid) { const int b = B.frame_base[s.fl]; s.cb = (unsigned)(b - cb0); s.ne = B.frame_base[s.fl + 1] - b; }
        s.phase = (s.valid && s.ne > 0) ? PH_SEED : PH_DONE;
        s.after = AF_CHECK; s.mode = 0;
        s.si = 0; s.c0 = 0; s.c1 = 0; s.par = 0; s.pref = 0; s.seed_idx = 0;
        if (jj < s.ne) s.c0 = *at(caddrw, s.cb + (unsigned)jj);
        if (G + jj < s.ne) s.c1 = *at(caddrw, s.cb + (unsigned)(G + jj));
        s.n = 0; s.i = 0; s.rhi = 0; s.ring = 0; s.reg_angle = 0; s.prec = PREC; s.sumdx = 0; s.sumdy = 0;
        s.sx = 0; s.sy = 0; s.sdeg = 0; s.scos = 0; s.ssin = 0; s.k = 0;
        for (int u = 0; u < PASS_B; u++) s.pxy[u] = 0;
        s.a0 = s.a1 = s.a2 = s.cx = s.cy = s.theta = s.dx = s.dy = s.lmin = s.lmax = s.wmin = s.wmax = 0;
        s.width = 0; s.scnt = 0; s.radSq = 0;
        s.n_cand = 0; s.n_grow = 0; s.n_reg = 0; s.fail = 0; s.it_done = 0; s.cap = list_cap;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(s.cap)); // (a vector register from here on: the compiler must not see a kernel argument in it)
#endif
#if defined(RGG_STATS)
        for (int q = 0; q < 9; q++) s.ph_iters[q] = 0;
        s.acc_rounds = 0;
#endif
    });

    int iters = 0;
#if defined(RGG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long rgg_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, rgg_t = clock64();
#endif
    for (;;) {
        {
            PerLane<bool> on;
            W::each([&](int l) { St &s = st[l]; on[l] = s.phase != PH_DONE; if (on[l]) s.it_done = iters + 1;
#if defined(RGG_STATS)
                s.ph_iters[s.phase]++;
#endif
            });
            if (!W::any(on)) break;
        }
        if (++iters > max_iters) { W::each([&](int l) { St &s = st[l]; if (s.phase != PH_DONE) { s.fail = 1; s.phase = PH_DONE; } }); break; }

#if defined(__HIP_DEVICE_COMPILE__)
        PerLane<It> itv;
#else
        PerLane<It> itv; memset(&itv, 0, sizeof(itv));
#endif
        RGG_T(0);
        // ---- the list pixels this iteration expands: entries i .. i + P - 1 from the ring
        PerLane<unsigned> qpick[P];
        for (int p = 0; p < P; p++) {
            PerLane<unsigned> rv; PerLane<bool> sel;
            W::each([&](int l) { const St &s = st[l]; rv[l] = (unsigned)s.ring; sel[l] = s.phase == PH_GROW && (l % G) == ((s.i + p) & (G - 1)); });
            qpick[p] = W::gpick(rv, sel);
        }
        // ---- EVERY load of the iteration, and no load anywhere else: none depends on another, and what is fetched ahead for later iterations (seed addresses,
        // a new region's seed sums, the next pass block's coordinates) is issued here too, so that the one wait in front of the first use covers it (a load
        // issued at the end of an iteration would be waited for at the top of the next).  Lanes outside a load's range read a valid address of their frame
        // and ignore the value.
        PerLane<bool> growing, seeding;
        W::each([&](int l) {
            St &s = st[l]; It &t = itv[l];
            const int jj = l % G, p = jj >> 3, j = jj & 7;
            growing[l] = s.phase == PH_GROW; seeding[l] = s.phase == PH_SEED;
            if (seeding[l]) {
                t.cval = s.si + jj < s.ne;
                t.ccur = s.par ? s.c1 : s.c0;
                t.adx = t.cval ? (t.ccur & 0x7fffffff) : 0;
                t.sfd = *at(angw, s.ao + (unsigned)t.adx);
            }
            if (s.pref) { // seed addresses two batches ahead
                const int last = s.ne - 1, ia = s.si + jj, ib = s.si + G + jj;
                const bool both = s.pref == 2;
                if (both || s.par == 1) s.c0 = *at(caddrw, s.cb + (unsigned)(both ? (ia < last ? ia : last) : (ib < last ? ib : last)));
                if (both || s.par == 0) s.c1 = *at(caddrw, s.cb + (unsigned)(ib < last ? ib : last));
                s.pref = 0;
            }
            if (s.phase >= PH_P0 && s.phase <= PH_STAT) { // the pass block k .. k + 7 (P0: only its coordinates, for the norms the next iteration gathers)
                const bool wrap = s.phase == PH_P1 && s.k + PASS_B >= s.n; // the last block of region2rect's first pass: get_theta starts over
                const int base = s.phase == PH_P0 ? 0 : ((s.phase == PH_P1 || s.phase == PH_P2) ? (wrap ? 0 : s.k + PASS_B) : s.k);
                for (int u = 0; u < PASS_B; u++) { const int idx = base + u < s.n ? base + u : s.n - 1; t.ent[u] = ld_ent(at(listw, s.lo + (unsigned)idx)); }
                if (s.phase == PH_P1 || s.phase == PH_P2)
                    for (int u = 0; u < PASS_B; u++) t.pm[u] = *at(modw, s.mo + (unsigned)rgs::xy_addr(s.pxy[u], Bw)); // (pxy: the block k .. k + 7, clamped entries repeat the last pixel)
            }
            if (s.phase == PH_SHRINK) { // (k < n holds while the phase lasts)
                for (int u = 0; u < SHRINK_B; u++) { const int idx = s.k + u < s.n ? s.k + u : s.n - 1; t.ent[u] = ld_ent(at(listw, s.lo + (unsigned)idx)); }
                t.elast = ld_ent(at(listw, s.lo + (unsigned)(s.n - 1)));
            }
            if (growing[l]) {
                if (s.i == 0 && s.mode == 0) { const float *cs = at(scsw, 2u * (s.cb + (unsigned)s.seed_idx)); s.scos = cs[0]; s.ssin = cs[1]; } // a new region: what its seed starts the sums with (:651-652)
                const int avail = s.rhi - s.i;
                t.live = avail < P ? avail : P;
                t.act = p < t.live;
                const int q = (int)qpick[p < P ? p : 0][l];
                const int px = q & 0xffff, py = q >> 16;
                const int i9 = j + (j >= 4), dy = ((i9 * 11) >> 5) - 1, dx = i9 - 3 * (dy + 1) - 1; // the 3 x 3 neighbourhood without its centre (the list pixel itself is used), yy outer, xx inner like :665-668
                const int nx = px + dx, ny = py + dy;
                t.inb = t.act && (unsigned)nx < (unsigned)Bw && (unsigned)ny < (unsigned)Bh;
                t.naddr = t.inb ? nx + ny * Bw : 0; t.nxy = t.inb ? rgs::xy_pack(nx, ny) : -1;
                t.fd = *at(angw, s.ao + (unsigned)t.naddr);
                t.refill = avail < 2 * P && s.rhi < s.n;
                t.n_issue = s.n;
                const int e = s.i + ((jj - s.i) & (G - 1)); // the entry of [i, i + G) this lane's ring slot holds
                t.do_ld = t.refill && e >= s.rhi && e < s.n;
                if (t.refill) t.rl = at(listw, s.lo + (unsigned)(t.do_ld ? e : 0))->xy;
            }
        });
        RGG_T(1);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0), on every path through the iteration: nothing is in flight across the back edge, so no load of the next iteration's stage waits for a register
#endif

        RGG_T(2);
        // ---- region_grow :660-686: the neighbours of the live list pixels, accepted one after the other in lane order
        if (W::any(growing)) {
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!growing[l]) { t.isfree = 0; t.cur = G; t.arad = 0; return; }
                if (s.i == 0) { s.sumdx = s.scos; s.sumdy = s.ssin; } // :651-652
                t.cur = 0; t.isfree = t.inb && t.fd != NOTDEF_F; t.arad = double(t.fd) * rg::DEG_TO_RADS;
            });
            for (;;) {
                PerLane<bool> ok;
                W::each([&](int l) { const St &s = st[l]; It &t = itv[l]; ok[l] = t.isfree && (l % G) >= t.cur && rgs::aligned_rad(t.arad, s.reg_angle, s.prec); });
                if (!W::any(ok)) break;
                RGG_C(8);
                const PerLane<unsigned> m = W::gmask(ok);
#if defined(RGG_STATS)
                W::each([&](int l) { if (m[l]) st[l].acc_rounds++; });
#endif
                PerLane<bool> isacc; PerLane<unsigned> vd, vq;
                W::each([&](int l) {
                    It &t = itv[l];
                    const bool have = m[l] != 0;
                    const int first = have ? W::ctz(m[l]) : 0;
                    isacc[l] = have && (l % G) == first;
                    vd[l] = f2u(t.fd); vq[l] = (unsigned)t.nxy;
                });
                const PerLane<unsigned> pd = W::gpick(vd, isacc), pq = W::gpick(vq, isacc);
                W::each([&](int l) {
                    St &s = st[l]; It &t = itv[l];
                    if (m[l] == 0) return;
                    const int jj = l % G, first = W::ctz(m[l]);
                    if (s.n >= s.cap) { s.fail = 1; t.isfree = 0; t.act = 0; t.cur = G; return; } // the region outgrew its list: the frame gives up (every lane of the group takes this branch)
                    const int cxy = (int)pq[l];
                    const float cdeg = u2f(pd[l]);
                    if (jj == first) { // :669-674
                        st_ang(at(angw, s.ao + (unsigned)t.naddr), NOTDEF_F);
                        st_ent(at(listw, s.lo + (unsigned)s.n), cxy, cdeg);
                    }
                    if (t.nxy == cxy) t.isfree = 0; // (with P = 2 the pixel may also be a neighbour of the other list pixel)
                    if (s.rhi == s.n && s.n - s.i < G) { if (jj == (s.n & (G - 1))) s.ring = cxy; s.rhi++; }
                    s.n++;
                    float cc, sn;
                    glibc_sincosf::sincosf_pos(float(double(cdeg) * rg::DEG_TO_RADS), &sn, &cc); // cos(float(angle)), sin(float(angle)) :676-677 with glibc's values
                    s.sumdx += cc; s.sumdy += sn;
                    s.reg_angle = rg::fast_atan2_1(s.sumdy, s.sumdx) * rg::DEG_TO_RADS;
                    t.cur = first + 1;
                });
            }
            RGG_T(3);
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!growing[l]) return;
                if (s.fail) { s.phase = PH_DONE; return; }
                if (s.i == 0) st_ent(at(listw, s.lo), rgs::xy_pack(s.sx, s.sy), s.sdeg); // the seed's entry
                const int i_old = s.i;
                s.i += t.live;
                if (t.refill) { if (t.do_ld) s.ring = t.rl; const int top = i_old + G; s.rhi = t.n_issue < top ? t.n_issue : top; }
                if (s.i >= s.n) { // the region is complete
                    if (s.mode == 0) {
                        s.n_grow++;
                        if (s.n < min_reg_size) s.phase = PH_SEED; // :489
                        else { s.n_reg++; s.phase = PH_P0; s.after = AF_CHECK; }
                    } else {
                        if (s.n < 2) s.phase = PH_SEED; // :817
                        else { s.phase = PH_P0; s.after = AF_REGROWN; }
                    }
                }
            });
        }

        RGG_T(4);
        // ---- the seed loop :477-487: G seeds at a time
        if (W::any(seeding)) {
            RGG_C(9);
            PerLane<bool> grower, issel; PerLane<unsigned> va, vd;
            W::each([&](int l) { It &t = itv[l]; grower[l] = seeding[l] && t.cval && t.sfd != NOTDEF_F && t.ccur >= 0; });
            const PerLane<unsigned> mg = W::gmask(grower);
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                issel[l] = false; va[l] = 0; vd[l] = 0;
                if (!seeding[l]) return;
                const int jj = l % G, j0 = mg[l] ? W::ctz(mg[l]) : G;
                if (t.cval && t.sfd != NOTDEF_F && t.ccur < 0 && jj < j0) st_ang(at(angw, s.ao + (unsigned)t.adx), NOTDEF_F); // a region of one pixel (flagged by lsd_emit): used, nothing else
                issel[l] = jj == j0;
                va[l] = (unsigned)t.adx; vd[l] = f2u(t.sfd);
            });
            const PerLane<unsigned> pa = W::gpick(va, issel), pd = W::gpick(vd, issel);
            W::each([&](int l) {
                St &s = st[l];
                if (!seeding[l]) return;
                const int jj = l % G;
                if (mg[l]) { // region_grow from this seed :637-657
                    const int j0 = W::ctz(mg[l]), idx0 = s.si + j0, sadx = (int)pa[l];
                    s.sy = div_w(sadx, Bw, rcp_w); s.sx = sadx - s.sy * Bw; s.sdeg = u2f(pd[l]);
                    s.seed_idx = idx0;
                    if (jj == j0) st_ang(at(angw, s.ao + (unsigned)sadx), NOTDEF_F);
                    s.n = 1; s.i = 0; s.rhi = 1; if (jj == 0) s.ring = rgs::xy_pack(s.sx, s.sy);
                    s.reg_angle = double(s.sdeg) * rg::DEG_TO_RADS; s.prec = PREC; s.mode = 0;
                    s.phase = PH_GROW;
                    s.si = idx0 + 1; s.par = 0; s.pref = 2;
                } else {
                    s.si += G; s.par ^= 1; s.pref = 1;
                    if (s.si >= s.ne) { s.phase = PH_DONE; s.pref = 0; }
                }
            });
        }

        RGG_T(5);
        // ---- the passes over a finished region (region2rect :690-746, get_theta :748-784, refine :786-832, reduce_region_radius :834-871)
        PerLane<bool> passing;
        W::each([&](int l) { const St &s = st[l]; passing[l] = (s.phase >= PH_P0 && s.phase <= PH_SHRINK) && !growing[l] && !seeding[l]; });
        if (W::any(passing)) {
            RGG_C(10);
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!passing[l]) return;
                const int jj = l % G;
                bool rect_done = false;
                rg::Rect r = rg::Rect{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (s.phase == PH_P0) { // the first block's coordinates are here: the next iteration gathers its norms
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = t.ent[u].xy;
                    s.phase = PH_P1; s.k = 0; s.a0 = 0; s.a1 = 0; s.a2 = 0;
                } else if (s.phase == PH_P1) { // :692-702
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) { const int xy = s.pxy[u]; const double m = t.pm[u]; s.a0 += double(xy & 0xffff) * m; s.a1 += double(xy >> 16) * m; s.a2 += m; }
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = t.ent[u].xy;
                    s.k += PASS_B;
                    if (s.k >= s.n) { s.cx = s.a0 / s.a2; s.cy = s.a1 / s.a2; s.a0 = 0; s.a1 = 0; s.a2 = 0; s.k = 0; s.phase = PH_P2; }
                } else if (s.phase == PH_P2) { // :755-766
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) { const int xy = s.pxy[u]; const double dx = double(xy & 0xffff) - s.cx, dy = double(xy >> 16) - s.cy, m = t.pm[u]; s.a0 += dy * dy * m; s.a1 += dx * dx * m; s.a2 -= dx * dy * m; }
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = t.ent[u].xy;
                    s.k += PASS_B;
                    if (s.k >= s.n) {
                        const double Ixx = s.a0, Iyy = s.a1, Ixy = s.a2;
                        const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
                        double theta = (fabs(Ixx) > fabs(Iyy)) ? double(rg::fast_atan2(float(lambda - Ixx), float(Ixy))) : double(rg::fast_atan2(float(Ixy), float(lambda - Iyy)));
                        theta *= rg::DEG_TO_RADS;
                        if (fabs(rg::angle_diff_signed(theta, s.reg_angle)) > PREC) theta += rg::PI_;
                        s.theta = theta; s.dx = cos(theta); s.dy = sin(theta);
                        s.lmin = 0; s.lmax = 0; s.wmin = 0; s.wmax = 0; s.k = 0; s.phase = PH_P3;
                    }
                } else if (s.phase == PH_P3) { // :714-728
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) {
                        const Ent &e = t.ent[u];
                        const double rdx = double(e.xy & 0xffff) - s.cx, rdy = double(e.xy >> 16) - s.cy, ll = rdx * s.dx + rdy * s.dy, ww = -rdx * s.dy + rdy * s.dx;
                        if (ll > s.lmax) s.lmax = ll; else if (ll < s.lmin) s.lmin = ll;
                        if (ww > s.wmax) s.wmax = ww; else if (ww < s.wmin) s.wmin = ww;
                    }
                    s.k += PASS_B;
                    if (s.k >= s.n) {
                        r.x1 = s.cx + s.lmin * s.dx; r.y1 = s.cy + s.lmin * s.dy; r.x2 = s.cx + s.lmax * s.dx; r.y2 = s.cy + s.lmax * s.dy;
                        r.width = s.wmax - s.wmin; r.x = s.cx; r.y = s.cy; r.theta = s.theta; r.dx = s.dx; r.dy = s.dy; r.prec = PREC; r.p = PP;
                        if (r.width < 1.0) r.width = 1.0;
                        s.width = r.width;
                        rect_done = true;
                    }
                } else if (s.phase == PH_STAT) { // :798-810
                    const double xc = double(s.sx), yc = double(s.sy), ang_c = double(s.sdeg) * rg::DEG_TO_RADS;
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) {
                        const Ent &e = t.ent[u];
                        const int qx = e.xy & 0xffff, qy = e.xy >> 16;
                        if (jj == u) st_ang(at(angw, s.ao + (unsigned)(qx + qy * Bw)), e.deg); // :800 used = NOTUSED
                        if (rg::dist(xc, yc, double(qx), double(qy)) < s.width) { const double a = rg::angle_diff_signed(double(e.deg) * rg::DEG_TO_RADS, ang_c); s.a0 += a; s.a1 += a * a; ++s.scnt; }
                    }
                    s.k += PASS_B;
                    if (s.k >= s.n) { // :811-815: grow again from the seed with the tolerance tau
                        const double mean_angle = s.a0 / double(s.scnt);
                        const double tau = 2.0 * sqrt((s.a1 - 2.0 * mean_angle * s.a0) / double(s.scnt) + mean_angle * mean_angle);
                        if (jj == 0) st_ang(at(angw, s.ao + (unsigned)(s.sx + s.sy * Bw)), NOTDEF_F); // (the lane that released the seed's entry)
                        s.n = 1; s.i = 0; s.rhi = 1; if (jj == 0) s.ring = rgs::xy_pack(s.sx, s.sy);
                        s.reg_angle = ang_c; s.prec = tau; s.mode = 1;
                        s.phase = PH_GROW;
                    }
                } else { // PH_SHRINK :849-859: the swaps reorder the list, one entry after the other; an iteration ends at the first entry it removes
                    const double xc = double(s.sx), yc = double(s.sy);
                    bool stop = false;
                    for (int u = 0; u < SHRINK_B; u++) if (!stop && s.k < s.n) {
                        const Ent cur = t.ent[u];
                        const double ddx = double(cur.xy & 0xffff) - xc, ddy = double(cur.xy >> 16) - yc;
                        if (ddx * ddx + ddy * ddy > s.radSq) {
                            if (jj == 0) st_ang(at(angw, s.ao + (unsigned)rgs::xy_addr(cur.xy, Bw)), cur.deg);
                            st_ent(at(listw, s.lo + (unsigned)s.k), t.elast.xy, t.elast.deg); // (every lane of the group stores it: its own later loads of the entry follow its own store)
                            --s.n; stop = true; // the entry that took its place is looked at by the next iteration
                        } else ++s.k;
                    }
                    if (s.k >= s.n) {
                        if (s.n < 2) s.phase = PH_SEED; // :862
                        else { s.phase = PH_P0; s.after = AF_SHRUNK; }
                    }
                }
                if (rect_done) {
                    const double density = double(s.n) / (rg::dist(r.x1, r.y1, r.x2, r.y2) * r.width);
                    bool emit = false, shrink = false;
                    if (density >= rg::DENSITY_TH) emit = true;
                    else if (s.after == AF_CHECK) { s.phase = PH_STAT; s.k = 0; s.a0 = 0; s.a1 = 0; s.scnt = 0; }
                    else if (s.after == AF_REGROWN) { // :836-842
                        const double xc = double(s.sx), yc = double(s.sy);
                        const double r1 = (r.x1 - xc) * (r.x1 - xc) + (r.y1 - yc) * (r.y1 - yc), r2 = (r.x2 - xc) * (r.x2 - xc) + (r.y2 - yc) * (r.y2 - yc);
                        s.radSq = r1 > r2 ? r1 : r2;
                        shrink = true;
                    } else shrink = true;
                    if (shrink) { s.radSq *= 0.75 * 0.75; s.k = 0; s.phase = PH_SHRINK; }
                    if (emit) {
                        if (s.n_cand >= cand_cap) { s.fail = 1; s.phase = PH_DONE; }
                        else {
                            if (jj == 0) { double *o = at(rectw, s.ro + (unsigned)s.n_cand * 12u); o[0] = r.x1; o[1] = r.y1; o[2] = r.x2; o[3] = r.y2; o[4] = r.width; o[5] = r.x; o[6] = r.y; o[7] = r.theta; o[8] = r.dx; o[9] = r.dy; o[10] = r.prec; o[11] = r.p; }
                            ++s.n_cand;
                            s.phase = PH_SEED;
                        }
                    }
                }
            });
        }
        RGG_T(6);
    }
#if defined(RGG_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    rgg_acc[11] = iters;
    if (B.prof && (threadIdx.x & 63) == 0) for (int k = 0; k < 16; k++) B.prof[16 * (size_t)(f0 / (64 / G)) + k] = rgg_acc[k];
#endif
    W::each([&](int l) {
        const St &s = st[l];
        if (s.valid && (l % G) == 0) { int *o = B.status + 4 * (size_t)s.fl; o[0] = s.n_grow; o[1] = s.fail; o[2] = s.n_reg; o[3] = s.it_done; B.cand_cnt[s.fl] = s.n_cand;
#if defined(RGG_STATS)
            printf("  frame %d iterations by phase: seed %ld grow %ld p0 %ld p1 %ld p2 %ld p3 %ld stat %ld shrink %ld; accept rounds %ld\n", s.fl, s.ph_iters[0], s.ph_iters[1], s.ph_iters[2], s.ph_iters[3], s.ph_iters[4], s.ph_iters[5], s.ph_iters[6], s.ph_iters[7], s.acc_rounds);
#endif
        }
    });
}
} // namespace rgg
