// lsd_rg_lpf.h -- the LSD region stage (flsd's seed loop, region_grow, region2rect, refine, reduce_region_radius; lsd.cpp:464-871) with ONE LANE PER
// FRAME: each of a wave's 64 lanes runs the reference's sequence on its own frame as a state machine -- seeds in raster order, every region seeing
// the marks of all regions before it, every accepted pixel changing the angle the next test uses; nothing inside a frame is reordered, so the
// result is exact by construction.  What changes against one wave per frame (lsd_rg_seq.h) is what an issued instruction buys: there a frame cost
// 14.3 M issued instructions (7.2 M of them on the scalar unit, which was the limit); here every instruction of the walk is a vector instruction
// that serves 64 frames -- about 2.5 M per frame -- no lane ever talks to another, and 1024 frames hold 16 wave slots instead of 1024.
//
// A wave runs ONE loop.  Per iteration every frame does one step of the phase it is in:
//   SEED     four seed candidates (their addresses were fetched while the previous region grew): used ones are skipped, the ones lsd_emit flagged
//            as "no neighbour aligned with my own angle" are marked used, the first other one starts a region;
//   GROW     one list pixel: its 3 x 3 neighbourhood is three 12-byte loads from the frame's map -- ONE FLOAT per pixel, the level-line angle
//            while the pixel is defined and unused; no bounds are tested: the last row and column of a frame are undefined by construction
//            (lsd_gradient) and the frames are laid out so that "row -1" and "column -1" fall on such pixels -- and its eight neighbours are
//            tested in the reference's order, every accepted one changing the region angle (cosf / sinf of its angle computed in place with
//            glibc's values, glibc_sincosf.h); the list's next entries sit in four registers of the lane;
//   P0 .. P3, STAT, SHRINK   region2rect / get_theta / refine / reduce_region_radius as sequential passes over the list, eight (four) entries
//            per iteration, the gradient norms gathered from the dense map one iteration after the coordinates.
// ALL loads of an iteration are issued before the first use (and no load anywhere else), so an iteration costs one memory round trip however
// many phases are active in the wave; with 64 frames in a wave every phase is active nearly always -- an iteration is the sum of the phase bodies,
// about 1200 vector instructions -- which is why the frames of a launch are sorted by their work (frames of similar work finish together).
// Written once for the device and for a host model (tools/lsd_sim/lpf_sim.cpp: the lanes as loops).
#pragma once
#include "glibc_sincosf.h"
#include "lsd_rg_seq.h"
#include <cstring>
#if defined(RGL_STATS)
#include <cstdio>
#endif

namespace rgl {
using rg::u64;
using rgs::NOTDEF_F;
using rgs::PerLane;

struct Ent { int xy; float deg; };             // one pixel of a region: x | y << 16, level-line angle (float degrees)
constexpr int CAP = 32768;                     // pixels of one region; a larger one sends the batch to the host stage
constexpr int PASS_B = 8;                      // list entries per iteration of a pass
constexpr int SHRINK_B = 4;                    // list entries per iteration of reduce_region_radius (an iteration ends at the first entry it removes)
constexpr int SEED_B = 4;                      // seed candidates per iteration
enum : int { PH_SEED = 0, PH_GROW, PH_P0, PH_P1, PH_P2, PH_P3, PH_STAT, PH_SHRINK, PH_DONE };
enum : int { AF_CHECK = 0, AF_REGROWN, AF_SHRUNK };


struct Batch { // wave-uniform; every pointer is the slice of the launch's frames (all offsets inside a slice fit 32 bits: the host cuts a batch into slices)
    int F, w, h, npx;
    int ang_stride, list_stride, rect_stride; // elements from one frame's map / list / rectangles to the next
    int ang_head;                             // floats in front of the slice's frame 0 (>= w + 2, NOTDEF_F like the gaps between the frames)
    const int *order;                         // the frames of the slice sorted by their number of defined pixels (frames of similar work share a wave), or NULL
    const int *caddr; const int *frame_base;  // defined pixels in address order (bit 31: "stays alone as a seed"), all frames one after the other (SEED_B entries of slack behind the last)
    float *ang;                               // dense: the level-line angle in float degrees while the pixel is defined and unused, NOTDEF_F otherwise; frame f at ang_head + f * ang_stride
    const double *mod;                        // dense gradient norms, frame f at f * npx
    const float *seed_cs;                     // per defined pixel: float(cos(angle)), float(sin(angle)) of the angle as a double (:651-652)
    Ent *list; int list_cap;                  // frame f at f * list_stride (PASS_B entries of slack behind a list)
    double *rect; int cand_cap; int *cand_cnt; // the rectangles that reach rect_improve, per frame in seed order
    int *status;                              // per frame: [0] region_grow calls from the seed loop, [1] failure (capacity), [2] regions at the rectangle stage, [3] iterations of the frame
    int min_reg_size; int max_iters;
};

struct St { // a frame's walk
    int phase, after, mode, valid;
    unsigned ao, mo, lo, cb, ro; // element offsets of the frame inside the slice's map, norms, lists, caddr / seed_cs, rectangles
    int fl, ne;
    int si, par, pref, seed_idx; // seed cursor; the set `par` of ca / cb holds caddr[si .. si + 3], the other caddr[si + 4 ..]; pref: 1 = fetch the other set, 2 = fetch both (par = 0) in the next load stage
    int ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3; // (scalars, not arrays: a load into an array element through a pointer keeps the array in scratch memory)
    int n, i, fcnt, fq[4];   // region size, list cursor, the list entries i .. i + fcnt - 1 (coordinates)
    double reg_angle, prec; float sumdx, sumdy;
    int sx, sy; float sdeg, scos, ssin;
    int k;
    int pxy[PASS_B];         // the coordinates of the pass block whose norms the next load stage gathers
    double a0, a1, a2, cx, cy, theta, dx, dy, lmin, lmax, wmin, wmax;
    double width; int scnt;  // the last rectangle's width (refine's pass needs it); the pass of refine sums into a0 / a1
    double radSq;
    int n_cand, n_grow, n_reg, fail, it_done, cap;
#if defined(RGL_STATS)
    long ph_iters[9], accepts;
#endif
};
struct It { // one iteration's loads and what follows from them.  Declared inside the loop WITHOUT initial values: a load's destination that is merged with a
            // default at the end of a branch makes the compiler wait for the load there, and the loads of an iteration would run one after the other
    int cur[SEED_B], adx[SEED_B], cval[SEED_B]; float sfd[SEED_B];
    float nb[9]; int base, refill, n_issue, rq[2];
    Ent ent[PASS_B]; Ent elast; double pm[PASS_B];
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

// the walk of the frames order[f0 .. f0 + 63] (or f0 ..) of a slice: one wave
template <class W> RGS_FN void run_wave(const Batch &B, int f0) {
    const double PREC = rg::PI_ * rg::ANG_TH / 180, PP = rg::ANG_TH / 180;
    const float rcp_w = 1.0f / float(B.w);
    const int cb0 = B.frame_base[0];
    float *const angw = B.ang;
    const double *const modw = B.mod;
    const int *const caddrw = B.caddr;
    const float *const scsw = B.seed_cs;
    Ent *const listw = B.list;
    double *const rectw = B.rect;
    const int Bw = B.w, list_cap = B.list_cap, min_reg_size = B.min_reg_size, cand_cap = B.cand_cap, max_iters = B.max_iters;

    PerLane<St> st;
    W::each([&](int l) {
        St &s = st[l];
        s.valid = f0 + l < B.F;
        s.fl = s.valid ? (B.order ? B.order[f0 + l] : f0 + l) : 0;
        s.ao = (unsigned)B.ang_head + (unsigned)s.fl * (unsigned)B.ang_stride; s.mo = (unsigned)s.fl * (unsigned)B.npx; s.lo = (unsigned)s.fl * (unsigned)B.list_stride; s.ro = (unsigned)s.fl * (unsigned)B.rect_stride;
        s.cb = 0; s.ne = 0;
        if (s.valid) { const int b = B.frame_base[s.fl]; s.cb = (unsigned)(b - cb0); s.ne = B.frame_base[s.fl + 1] - b; }
        s.phase = (s.valid && s.ne > 0) ? PH_SEED : PH_DONE;
        s.after = AF_CHECK; s.mode = 0;
        s.si = 0; s.par = 0; s.pref = 0; s.seed_idx = 0;
        s.ca0 = s.ca1 = s.ca2 = s.ca3 = s.cb0 = s.cb1 = s.cb2 = s.cb3 = 0;
        if (s.phase == PH_SEED) { const I4 a = ld_i4(at(caddrw, s.cb)), b = ld_i4(at(caddrw, s.cb + (unsigned)SEED_B)); s.ca0 = a.a; s.ca1 = a.b; s.ca2 = a.c; s.ca3 = a.d; s.cb0 = b.a; s.cb1 = b.b; s.cb2 = b.c; s.cb3 = b.d; }
        s.n = 0; s.i = 0; s.fcnt = 0; s.fq[0] = s.fq[1] = s.fq[2] = s.fq[3] = 0; s.reg_angle = 0; s.prec = PREC; s.sumdx = 0; s.sumdy = 0;
        s.sx = 0; s.sy = 0; s.sdeg = 0; s.scos = 0; s.ssin = 0; s.k = 0;
        for (int u = 0; u < PASS_B; u++) s.pxy[u] = 0;
        s.a0 = s.a1 = s.a2 = s.cx = s.cy = s.theta = s.dx = s.dy = s.lmin = s.lmax = s.wmin = s.wmax = 0;
        s.width = 0; s.scnt = 0; s.radSq = 0;
        s.n_cand = 0; s.n_grow = 0; s.n_reg = 0; s.fail = 0; s.it_done = 0; s.cap = list_cap;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(s.cap)); // (a vector register from here on: as a kernel argument the compiler re-reads it from scalar memory inside the loop)
#endif
#if defined(RGL_STATS)
        for (int q = 0; q < 9; q++) s.ph_iters[q] = 0;
        s.accepts = 0;
#endif
    });

    int iters = 0;
    for (;;) {
        {
            PerLane<bool> on;
            W::each([&](int l) { St &s = st[l]; on[l] = s.phase != PH_DONE; if (on[l]) s.it_done = iters + 1;
#if defined(RGL_STATS)
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
        // ---- EVERY load of the iteration, and no load anywhere else: none depends on another, and what is fetched ahead for later iterations (seed addresses,
        // a new region's seed sums, the next pass block's coordinates) is issued here too, so that the one wait below covers it
        PerLane<bool> growing, seeding;
        W::each([&](int l) {
            St &s = st[l]; It &t = itv[l];
            growing[l] = s.phase == PH_GROW; seeding[l] = s.phase == PH_SEED;
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
                if (both || s.par == 1) { const I4 a = ld_i4(at(caddrw, s.cb + (unsigned)(both ? s.si : s.si + SEED_B))); s.ca0 = a.a; s.ca1 = a.b; s.ca2 = a.c; s.ca3 = a.d; }
                if (both || s.par == 0) { const I4 b = ld_i4(at(caddrw, s.cb + (unsigned)(s.si + SEED_B))); s.cb0 = b.a; s.cb1 = b.b; s.cb2 = b.c; s.cb3 = b.d; }
                s.pref = 0;
            }
            if (s.phase >= PH_P0 && s.phase <= PH_STAT) { // the pass block k .. k + 7 (P0: only its coordinates, for the norms the next iteration gathers)
                const bool wrap = s.phase == PH_P1 && s.k + PASS_B >= s.n; // the last block of region2rect's first pass: get_theta starts over
                const int base = s.phase == PH_P0 ? 0 : ((s.phase == PH_P1 || s.phase == PH_P2) ? (wrap ? 0 : s.k + PASS_B) : s.k);
                for (int u = 0; u < PASS_B; u++) t.ent[u] = ld_ent(at(listw, s.lo + (unsigned)(base + u))); // (entries behind the region's last are read and not used: PASS_B entries of slack behind a list)
                if (s.phase == PH_P1 || s.phase == PH_P2)
                    for (int u = 0; u < PASS_B; u++) t.pm[u] = *at(modw, s.mo + (unsigned)rgs::xy_addr(s.pxy[u], Bw)); // (pxy: the block k .. k + 7, the entries behind the last repeat the seed)
            }
            if (s.phase == PH_SHRINK) { // (k < n holds while the phase lasts)
                for (int u = 0; u < SHRINK_B; u++) t.ent[u] = ld_ent(at(listw, s.lo + (unsigned)(s.k + u)));
                t.elast = ld_ent(at(listw, s.lo + (unsigned)(s.n - 1)));
            }
            if (growing[l]) {
                if (s.i == 0 && s.mode == 0) { const float *cs = at(scsw, 2u * (s.cb + (unsigned)s.seed_idx)); s.scos = cs[0]; s.ssin = cs[1]; } // a new region: what its seed starts the sums with (:651-652)
                const int q = s.fq[0], px = q & 0xffff, py = q >> 16;
                t.base = py * Bw + px;
                // the 3 x 3 neighbourhood, three rows of three floats from column px - 1 (a frame's last row and column are undefined, "row -1" / "column -1" fall on them)
                ld_f3(at(angw, s.ao + (unsigned)(t.base - Bw - 1)), t.nb);
                ld_f3(at(angw, s.ao + (unsigned)(t.base - 1)), t.nb + 3);
                ld_f3(at(angw, s.ao + (unsigned)(t.base + Bw - 1)), t.nb + 6);
                t.refill = s.fcnt <= 1 && s.i + s.fcnt < s.n;
                t.n_issue = s.n;
                if (t.refill) ld_ent2(at(listw, s.lo + (unsigned)(s.i + s.fcnt)), t.rq);
            }
        });
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0), on every path through the iteration: nothing is in flight across the back edge, so no load of the next iteration's stage waits for a register
#endif

        // ---- region_grow :660-686: the eight neighbours of the list pixel in the reference's order (yy outer, xx inner; the centre is the list pixel itself: used)
        if (W::any(growing)) {
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!growing[l]) return;
                if (s.i == 0) { s.sumdx = s.scos; s.sumdy = s.ssin; } // :651-652
                const int q = s.fq[0], px = q & 0xffff, py = q >> 16;
                const int i_old = s.i, fcnt_old = s.fcnt;
#pragma unroll
                for (int sl = 0; sl < 9; sl++) {
                    if (sl == 4) continue;
                    const float fd = t.nb[sl];
                    if (fd != NOTDEF_F && !s.fail && rgs::aligned_rad(double(fd) * rg::DEG_TO_RADS, s.reg_angle, s.prec)) { // :669-683
                        if (s.n >= s.cap) { s.fail = 1; continue; } // the region outgrew its list: the frame gives up
                        const int dy = sl / 3 - 1, dx = sl % 3 - 1, cxy = rgs::xy_pack(px + dx, py + dy);
                        st_ang(at(angw, s.ao + (unsigned)(t.base + dy * Bw + dx)), NOTDEF_F);
                        st_ent(at(listw, s.lo + (unsigned)s.n), cxy, fd);
                        if (s.i + s.fcnt == s.n && s.fcnt < 4) { // the list's end is in the registers: so is this entry
                            if (s.fcnt == 1) s.fq[1] = cxy; else if (s.fcnt == 2) s.fq[2] = cxy; else if (s.fcnt == 3) s.fq[3] = cxy; else s.fq[0] = cxy;
                            s.fcnt++;
                        }
                        s.n++;
                        float cc, sn;
                        glibc_sincosf::sincosf_pos(float(double(fd) * rg::DEG_TO_RADS), &sn, &cc); // cos(float(angle)), sin(float(angle)) :676-677 with glibc's values
                        s.sumdx += cc; s.sumdy += sn;
                        s.reg_angle = rg::fast_atan2_1(s.sumdy, s.sumdx) * rg::DEG_TO_RADS;
#if defined(RGL_STATS)
                        s.accepts++;
#endif
                    }
                }
                if (s.fail) { s.phase = PH_DONE; return; }
                if (i_old == 0) st_ent(at(listw, s.lo), rgs::xy_pack(s.sx, s.sy), s.sdeg); // the seed's entry
                // the next list pixel
                s.i++; s.fq[0] = s.fq[1]; s.fq[1] = s.fq[2]; s.fq[2] = s.fq[3]; s.fcnt--;
                if (t.refill) { // (no entry joined the registers in this iteration: the list's end was not in them)
                    const int r0 = i_old + fcnt_old, got = t.n_issue - r0 < 2 ? t.n_issue - r0 : 2;
                    if (s.fcnt == 0) { s.fq[0] = t.rq[0]; s.fq[1] = t.rq[1]; } else { s.fq[1] = t.rq[0]; s.fq[2] = t.rq[1]; }
                    s.fcnt += got;
                }
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

        // ---- the seed loop :477-487: four candidates
        if (W::any(seeding)) {
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!seeding[l]) return;
                int k0 = -1;
                for (int k = 0; k < SEED_B; k++)
                    if (k0 < 0 && t.cval[k] && t.sfd[k] != NOTDEF_F) {
                        if (t.cur[k] < 0) st_ang(at(angw, s.ao + (unsigned)t.adx[k]), NOTDEF_F); // a region of one pixel (flagged by lsd_emit): used, nothing else
                        else k0 = k;
                    }
                if (k0 >= 0) { // region_grow from this seed :637-657
                    int sadx = t.adx[0]; float sdeg = t.sfd[0];
                    for (int k = 1; k < SEED_B; k++) if (k0 == k) { sadx = t.adx[k]; sdeg = t.sfd[k]; }
                    s.sy = div_w(sadx, Bw, rcp_w); s.sx = sadx - s.sy * Bw; s.sdeg = sdeg;
                    s.seed_idx = s.si + k0;
                    st_ang(at(angw, s.ao + (unsigned)sadx), NOTDEF_F);
                    s.n = 1; s.i = 0; s.fcnt = 1; s.fq[0] = rgs::xy_pack(s.sx, s.sy);
                    s.reg_angle = double(s.sdeg) * rg::DEG_TO_RADS; s.prec = PREC; s.mode = 0;
                    s.phase = PH_GROW;
                    s.si = s.seed_idx + 1; s.par = 0; s.pref = 2;
                } else {
                    s.si += SEED_B; s.par ^= 1; s.pref = 1;
                    if (s.si >= s.ne) { s.phase = PH_DONE; s.pref = 0; }
                }
            });
        }

        // ---- the passes over a finished region (region2rect :690-746, get_theta :748-784, refine :786-832, reduce_region_radius :834-871)
        PerLane<bool> passing;
        W::each([&](int l) { const St &s = st[l]; passing[l] = (s.phase >= PH_P0 && s.phase <= PH_SHRINK) && !growing[l] && !seeding[l]; });
        if (W::any(passing)) {
            W::each([&](int l) {
                St &s = st[l]; It &t = itv[l];
                if (!passing[l]) return;
                const int seed_xy = rgs::xy_pack(s.sx, s.sy);
                bool rect_done = false;
                rg::Rect r = rg::Rect{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (s.phase == PH_P0) { // the first block's coordinates are here: the next iteration gathers its norms
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = u < s.n ? t.ent[u].xy : seed_xy;
                    s.phase = PH_P1; s.k = 0; s.a0 = 0; s.a1 = 0; s.a2 = 0;
                } else if (s.phase == PH_P1) { // :692-702
                    const bool wrap = s.k + PASS_B >= s.n; const int nb = wrap ? 0 : s.k + PASS_B;
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) { const int xy = s.pxy[u]; const double m = t.pm[u]; s.a0 += double(xy & 0xffff) * m; s.a1 += double(xy >> 16) * m; s.a2 += m; }
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = nb + u < s.n ? t.ent[u].xy : seed_xy;
                    s.k += PASS_B;
                    if (s.k >= s.n) { s.cx = s.a0 / s.a2; s.cy = s.a1 / s.a2; s.a0 = 0; s.a1 = 0; s.a2 = 0; s.k = 0; s.phase = PH_P2; }
                } else if (s.phase == PH_P2) { // :755-766
                    for (int u = 0; u < PASS_B; u++) if (s.k + u < s.n) { const int xy = s.pxy[u]; const double dx = double(xy & 0xffff) - s.cx, dy = double(xy >> 16) - s.cy, m = t.pm[u]; s.a0 += dy * dy * m; s.a1 += dx * dx * m; s.a2 -= dx * dy * m; }
                    for (int u = 0; u < PASS_B; u++) s.pxy[u] = s.k + PASS_B + u < s.n ? t.ent[u].xy : seed_xy;
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
                        st_ang(at(angw, s.ao + (unsigned)(qx + qy * Bw)), e.deg); // :800 used = NOTUSED
                        if (rg::dist(xc, yc, double(qx), double(qy)) < s.width) { const double a = rg::angle_diff_signed(double(e.deg) * rg::DEG_TO_RADS, ang_c); s.a0 += a; s.a1 += a * a; ++s.scnt; }
                    }
                    s.k += PASS_B;
                    if (s.k >= s.n) { // :811-815: grow again from the seed with the tolerance tau
                        const double mean_angle = s.a0 / double(s.scnt);
                        const double tau = 2.0 * sqrt((s.a1 - 2.0 * mean_angle * s.a0) / double(s.scnt) + mean_angle * mean_angle);
                        st_ang(at(angw, s.ao + (unsigned)(s.sx + s.sy * Bw)), NOTDEF_F);
                        s.n = 1; s.i = 0; s.fcnt = 1; s.fq[0] = seed_xy;
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
                            st_ang(at(angw, s.ao + (unsigned)rgs::xy_addr(cur.xy, Bw)), cur.deg);
                            st_ent(at(listw, s.lo + (unsigned)s.k), t.elast.xy, t.elast.deg);
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
                            double *o = at(rectw, s.ro + (unsigned)s.n_cand * 12u);
                            o[0] = r.x1; o[1] = r.y1; o[2] = r.x2; o[3] = r.y2; o[4] = r.width; o[5] = r.x; o[6] = r.y; o[7] = r.theta; o[8] = r.dx; o[9] = r.dy; o[10] = r.prec; o[11] = r.p;
                            ++s.n_cand;
                            s.phase = PH_SEED;
                        }
                    }
                }
            });
        }
    }
    W::each([&](int l) {
        const St &s = st[l];
        if (s.valid) { int *o = B.status + 4 * (size_t)s.fl; o[0] = s.n_grow; o[1] = s.fail; o[2] = s.n_reg; o[3] = s.it_done; B.cand_cnt[s.fl] = s.n_cand;
#if defined(RGL_STATS)
            printf("  frame %d iterations by phase: seed %ld grow %ld p0 %ld p1 %ld p2 %ld p3 %ld stat %ld shrink %ld; accepts %ld\n", s.fl, s.ph_iters[0], s.ph_iters[1], s.ph_iters[2], s.ph_iters[3], s.ph_iters[4], s.ph_iters[5], s.ph_iters[6], s.ph_iters[7], s.accepts);
#endif
        }
    });
}
} // namespace rgl
