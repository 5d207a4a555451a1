// lsd_regions.hip -- LSD region growing, rectangle fitting and NFA validation on the device (reference line_lbd/libs/lsd.cpp:464-1155:
// flsd's seed loop, region_grow, region2rect, get_theta, refine, reduce_region_radius, rect_improve, rect_nfa, nfa).
//
// The reference is a greedy SEQUENCE: seeds are visited in raster order (lsd.cpp:477-480 walks `list` by index, see LsdHost::run in lsd.hip),
// every region sees the `used` marks of all regions before it, and every accepted pixel changes the angle the next test uses.  The device
// keeps the sequence and takes its parallelism from the FRAMES:
//
//   * lsd_rg_seq: one wave per frame walks the frame's seeds exactly like the reference (lsd_rg_seq.h: 8 x 8 windows of the pixel map in
//     the lanes' registers, the region list's tail in a register, ordered double sums fed from the lanes).  A frame takes ~100-180 ms
//     whatever the batch -- the chain of dependent instructions per accepted pixel, not memory -- so thousands of frames have to be
//     resident: sixteen waves (frames) per workgroup fill a CU and leave the other CUs empty for the kernels of the other streams.
//     Output: the rectangles that reach rect_improve (after refine / reduce_region_radius), per frame in seed order.
//   * lsd_rg_improve: eight rectangles per wave for rect_improve's first rect_nfa (two thirds are accepted there), the others improved one at a time by
//     the whole wave; the rows of a rectangle spread over lanes (the row limits advance by integer steps, lsd.cpp:1057-1095, so they have a closed form),
//     the five variants of a rect_improve loop walked and their nfa() evaluated side by side, log_gamma of the pixel counts from a table.
//   * the segments leave in the reference's emission order (frames in order, seeds in order).
//
// tools/lsd_sim/seq_sim.cpp compiles lsd_rg_seq.h for the host (the lanes as loops) and demands the oracle's sequential result: the same
// `used` map and the same rectangles bit for bit (tests/test_lsd_regions_cpu.py); tests/test_lsd_gpu.py demands byte-identical KeyLines
// from the device.  An earlier formulation -- a speculative fixed point over an owner map with one transaction per seed, parallel inside a
// frame -- was exact as well but 3-10 times slower than either this or the host stage and is gone (DESIGN.md 7.3b keeps what it taught).
//
// Arithmetic: doubles as in the reference, no FMA contraction (the Makefile's -ffp-contract=off); cos(float) / sin(float) of the pixel angles
// are glibc's cosf / sinf restated (glibc_sincosf.h: equal on every float of the domain); fastAtan2 is the same polynomial as the host's.
// cos / sin of the seed and rectangle angles and the transcendental functions of nfa() are the device library's doubles, which may differ
// from glibc in the last bit: that reaches the output only through a float rounding of an end point (about 1e-8 per line) or through a
// rectangle whose log-NFA is within 1e-15 of the threshold or of a competing variant.
#include "common.h"

#include <cfloat>
#include <climits>
#include <cmath>
#include <vector>

#include "lsd_regions.h"
#include "lsd_rg_seq.h"
#include "lsd_rg_wlk.h"

namespace {
constexpr int PAD_PX = 272, PAD_LIST = 272, PAD_RECT = 34; // (lsd_rg_wlk) 1088 / 2176 / 272 bytes between the frames' maps / lists / rectangles: the strides are not multiples of a large power of two (64 lanes, 64 frames: one channel otherwise)
constexpr double PI_ = rg::PI_, LOG_EPS = 0.0, LSD_SCALE = 0.8;
struct AngMap { int w, h; const float *deg; }; // a frame's level-line angles as lsd_gradient leaves them: float degrees, NOTDEF_F where the gradient is below the threshold; the map value is deg * DEG_TO_RADS in double

// ---- rect_improve / rect_nfa / nfa: one wave per rectangle -------------------------------------------------------------------------------
__device__ double rg_log_gamma(double x) { // lsd.cpp:70,124-160
    if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
    const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0;
    for (int n = 0; n < 7; ++n) { a -= log(x + double(n)); b += q[n] * pow(x, double(n)); }
    return a + log(b);
}
#ifndef RG_IMPROVE_WAVES
#ifndef RG_IMPROVE_WAVES
#define RG_IMPROVE_WAVES 3
#endif // waves per SIMD the register allocation of lsd_rg_improve aims at (latency-bound: pixel gathers and double transcendentals; measured 9.1 / 7.2 / 8.3 ms per 1024 frames at 2 / 3 / 4)
#endif
constexpr int LG_N = 32768; // log_gamma of the integers below this: a table filled by the same function (its arguments are pixel counts; each call costs 16 log + 14 pow)
__global__ void __launch_bounds__(256) lsd_rg_lgamma_table(double *t) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < LG_N) t[i] = i > 0 ? rg_log_gamma(double(i)) : 0.0; }
__device__ __forceinline__ double rg_lgam_int(int x, const double *lgt) { return (lgt && x > 0 && x < LG_N) ? lgt[x] : rg_log_gamma(double(x)); }
__device__ double rg_nfa(int n, int k, double p, double LOG_NT, const double *lgt) { // :1100-1136
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - double(n) * log10(p);
    const double p_term = p / (1 - p);
    const double log1term = (double(n) + 1) - rg_lgam_int(k + 1, lgt) - rg_lgam_int(n - k + 1, lgt) + double(k) * log(p) + double(n - k) * log(1.0 - p);
    double term = exp(log1term);
    const double RELATIVE_ERROR_FACTOR = 100.0;
    auto double_equal = [&](double a, double b) { if (a == b) return true; double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b); double abs_max = aa > bb ? aa : bb; if (abs_max < DBL_MIN) abs_max = DBL_MIN; return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON); };
    if (double_equal(term, 0)) { if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT; else return -LOG_NT; }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = double(n - i + 1) / double(i), mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - LOG_NT;
}
#if defined(RGI_PROF)
__device__ unsigned long long g_rgi_prof[4]; // ticks in rectangle walks, in nfa(), calls of each
#define RGI_T0 const unsigned long long rgi_t0 = wall_clock64()
#define RGI_T1(k) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&g_rgi_prof[k], wall_clock64() - rgi_t0); atomicAdd(&g_rgi_prof[2 + (k)], 1ull); } } while (0)
#else
#define RGI_T0
#define RGI_T1(k)
#endif
// What rect_nfa's row walk needs of a rectangle: the first and last row inside the image and, per row, the closed form of the reference's stepping limits.
struct RectSpan {
    int mnX, lmY, rmY, flstep, slstep, frstep, srstep, y_first, y_hi;
    // The four corners ordered like the reference's std::sort by (x, y) -- a sorting network gives the same order for distinct keys, ties are equal points -- and its
    // selection of the lowest / leftmost / rightmost / remaining corner, written on VALUES: indexing the sorted arrays with run-time indices cost ~600 compare / select
    // instructions per walk, a third of this kernel.  In the sorted order the first corner not taken has the smallest x (the reference's strict `>` keeps the first of
    // equals), so leftmost = corner 1 if the lowest is corner 0, else corner 0; the rightmost is the later of the remaining two unless the earlier has the larger x.
    __device__ __forceinline__ void set(const rg::Rect &rec, int h) {
        const double half_width = rec.width / 2.0, dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        int x0 = int(rec.x1 - dyhw), y0 = int(rec.y1 + dxhw), x1 = int(rec.x2 - dyhw), y1 = int(rec.y2 + dxhw);
        int x2 = int(rec.x2 + dyhw), y2 = int(rec.y2 - dxhw), x3 = int(rec.x1 + dyhw), y3 = int(rec.y1 - dxhw);
        auto cswap = [](int &xa, int &ya, int &xb, int &yb) {
            const bool sw = xa > xb || (xa == xb && ya > yb);
            const int tx = sw ? xb : xa, ty = sw ? yb : ya; xb = sw ? xa : xb; yb = sw ? ya : yb; xa = tx; ya = ty;
        };
        cswap(x0, y0, x1, y1); cswap(x2, y2, x3, y3); cswap(x0, y0, x2, y2); cswap(x1, y1, x3, y3); cswap(x1, y1, x2, y2);
        int mnI = 0, mnY = y0; mnX = x0; // min_y: the first corner with the smallest y
        { bool c = mnY > y1; mnI = c ? 1 : mnI; mnX = c ? x1 : mnX; mnY = c ? y1 : mnY; c = mnY > y2; mnI = c ? 2 : mnI; mnX = c ? x2 : mnX; mnY = c ? y2 : mnY; c = mnY > y3; mnI = c ? 3 : mnI; mnX = c ? x3 : mnX; mnY = c ? y3 : mnY; }
        const int mxY = max(max(y0, y1), max(y2, y3));
        const bool m0 = mnI == 0, le1 = mnI <= 1, is3 = mnI == 3;
        const int lmX = m0 ? x1 : x0; lmY = m0 ? y1 : y0;                                                 // leftmost
        const int iX = le1 ? x2 : x1, iY = le1 ? y2 : y1, jX = is3 ? x2 : x3, jY = is3 ? y2 : y3;         // the two that are left, in order
        const bool rj = iX < jX;
        const int rmX = rj ? jX : iX, tlX = rj ? iX : jX; rmY = rj ? jY : iY;                             // rightmost; of the last one only x is ever used
        // integer divisions and the tailp->x comparisons are the reference's (:1057-1065); the steps are integers
        flstep = (mnY != lmY) ? (mnX - lmX) / (mnY - lmY) : 0;
        slstep = (lmY != tlX) ? (lmX - tlX) / (lmY - tlX) : 0;
        frstep = (mnY != rmY) ? (mnX - rmX) / (mnY - rmY) : 0;
        srstep = (rmY != tlX) ? (rmX - tlX) / (rmY - tlX) : 0;
        y_hi = min(mxY, h - 1); y_first = max(mnY, 0);
    }
    // the limits move after every row INSIDE the image (the reference `continue`s past the stepping for the others): by the first step while
    // y < leftmost.y (rightmost.y), by the second from then on -- a closed form per row, so the lanes take rows independently
    __device__ __forceinline__ bool row(int y, int w, int &lx, int &rx) const {
        const long r = (long)y - y_first; // rows stepped before this one: y_first .. y-1
        auto adv = [&](int first, int second, int ysw) -> long { // sum over y' in [y_first, y) of (y' >= ysw ? second : first)
            long nf = (long)ysw - y_first; if (nf < 0) nf = 0; if (nf > r) nf = r;
            return nf * first + (r - nf) * second;
        };
        const long lxl = max((long)mnX + adv(flstep, slstep, lmY), 0L), rxl = min((long)mnX + adv(frstep, srstep, rmY), (long)w - 1);
        lx = (int)lxl; rx = (int)rxl; // both inside [0, w) when the row is not empty
        return rxl >= lxl;
    }
};
// the pixels [lx + g, rx] of a row in steps of G against the tolerances precs[0 .. NP): four requested before the first is tested
template <int NP> __device__ __forceinline__ void rg_row_count(const float *row, int lx, int rx, int g, int G, double theta, const double *precs, int *alg_pts) {
    for (int x = lx + g; x <= rx; x += 4 * G) {
        float ad[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int xx = x + u * G; ad[u] = row[xx <= rx ? xx : rx]; } // (a valid address either way; a value past the end is not counted)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = x + u * G <= rx && ad[u] != rgs::NOTDEF_F;
            const double a = double(ad[u]) * rg::DEG_TO_RADS;
            const double d = fabs(theta - a), d2 = fabs(d - rg::M_2__PI_), nt = d > rg::M_3_2_PI_ ? d2 : d; // isAligned :1138-1154
            for (int k = 0; k < NP; k++) alg_pts[k] += in && nt <= precs[k];
        }
    }
}
// The pixel walk of rect_nfa (:977-1098).  total = pixels of the rectangle inside the image; algs[k] = those aligned with rec.theta within precs[k] (several
// tolerances share a walk: rect_improve's first and last loop only halve the tolerance of an unchanged rectangle).  Every lane gets the sums.  The counts are
// integers, so the pixels may be visited in any order: G = 64 / rows lanes share a row (the rectangles that reach this stage hold 28 pixels in 7 rows on average).
template <int NP> __device__ void rg_rect_count(const AngMap &F, const rg::Rect &rec, const double *precs, int lane, int &total, int *algs) {
    RGI_T0;
    RectSpan S; S.set(rec, F.h);
    int total_pts = 0, alg_pts[NP];
    for (int k = 0; k < NP; k++) alg_pts[k] = 0;
    const int rows = S.y_hi - S.y_first + 1;
    const int G = (rows > 0 && rows < 64) ? 64 / rows : 1, RB = 64 / G; // lanes per row, rows per pass (64 % G lanes idle)
    const int lrow = lane / G, g = lane - lrow * G;
    for (int yb = S.y_first; yb <= S.y_hi; yb += RB) {
        const int y = yb + lrow;
        int lx, rx;
        if (y > S.y_hi || lrow >= RB || !S.row(y, F.w, lx, rx)) continue;
        if (g == 0) total_pts += rx - lx + 1;
        rg_row_count<NP>(F.deg + (size_t)y * F.w, lx, rx, g, G, rec.theta, precs, alg_pts);
    }
    for (int off = 32; off > 0; off >>= 1) {
        total_pts += __shfl_xor(total_pts, off);
        for (int k = 0; k < NP; k++) alg_pts[k] += __shfl_xor(alg_pts[k], off);
    }
    total = total_pts;
    for (int k = 0; k < NP; k++) algs[k] = alg_pts[k];
    RGI_T1(0);
}
// The walks of up to five rectangles at once (rect_improve's loops try five variants of a rectangle that do not depend on each other): eight lanes per rectangle,
// lane s of them takes the rows y_first + s, y_first + s + 8, ...  One preamble, one round trip and one (three-step) reduction instead of five of each -- with 28
// pixels per rectangle a walk is all fixed cost.  rs: this lane's copy of the variants (every lane holds the same five).
__device__ void rg_rect_count_five(const AngMap &F, const rg::Rect *rs, int cnt, int lane, int *tot, int *alg) {
    RGI_T0;
    const int grp = lane >> 3, sub = lane & 7;
    const bool on = grp < cnt;
    const rg::Rect rec = rs[on ? grp : 0];
    RectSpan S; S.set(rec, F.h);
    int total_pts = 0, alg_pts[1] = {0};
    if (on)
        for (int y = S.y_first + sub; y <= S.y_hi; y += 8) {
            int lx, rx;
            if (!S.row(y, F.w, lx, rx)) continue;
            total_pts += rx - lx + 1;
            rg_row_count<1>(F.deg + (size_t)y * F.w, lx, rx, 0, 1, rec.theta, &rec.prec, alg_pts);
        }
    for (int off = 1; off < 8; off <<= 1) { total_pts += __shfl_xor(total_pts, off); alg_pts[0] += __shfl_xor(alg_pts[0], off); }
    for (int n = 0; n < 5; n++) { tot[n] = __shfl(total_pts, 8 * n); alg[n] = __shfl(alg_pts[0], 8 * n); }
    RGI_T1(0);
}
// nfa() of up to five (n, k, p) triples at once: lane v computes triple v -- the loops inside nfa are sequential, the triples independent
__device__ void rg_nfa5(const int *n, const int *k, const double *p, int cnt, double LOG_NT, const double *lgt, int lane, double *out) {
    int nn = n[0], kk = k[0]; double pp = p[0];
    for (int v = 1; v < 5; v++) if (v < cnt && lane == v) { nn = n[v]; kk = k[v]; pp = p[v]; }
    const double r = rg_nfa(nn, kk, pp, LOG_NT, lgt);
    for (int v = 0; v < 5; v++) out[v] = __shfl(r, v);
}
// rect_improve lsd.cpp:873-975.  The five variants of each of its loops do not depend on each other's result (only the best is remembered), so
// their pixel walks run one after the other -- or as one walk where only the tolerance changes -- and their five nfa() side by side in five lanes.
// log_nfa: rect_nfa of the rectangle as it comes (the kernel computes it for eight rectangles at once).
__device__ double rg_rect_improve(const AngMap &F, rg::Rect &rec, double LOG_NT, int lane, const double *lgt, double log_nfa) {
    const double delta = 0.5, delta_2 = delta / 2.0;
    if (log_nfa > LOG_EPS) return log_nfa;
    rg::Rect rs[5];
    int tot[5], alg[5], cnt; double ps[5], v[5];
    auto tolerances = [&]() { // r.p /= 2 five times on the rectangle `rec`: one walk, five counts
        rg::Rect r = rec; double precs[5];
        for (int n = 0; n < 5; ++n) { r.p /= 2; r.prec = r.p * PI_; rs[n] = r; precs[n] = r.prec; ps[n] = r.p; }
        int t; rg_rect_count<5>(F, rec, precs, lane, t, alg);
        for (int n = 0; n < 5; ++n) tot[n] = t;
        cnt = 5;
    };
    auto take_best = [&]() {
        RGI_T0;
        rg_nfa5(tot, alg, ps, cnt, LOG_NT, lgt, lane, v);
        RGI_T1(1);
        for (int n = 0; n < cnt; ++n) if (v[n] > log_nfa) { log_nfa = v[n]; rec = rs[n]; }
    };
    auto walks = [&]() { if (cnt) rg_rect_count_five(F, rs, cnt, lane, tot, alg); for (int n = 0; n < cnt; ++n) ps[n] = rs[n].p; };
    tolerances(); take_best();
    if (log_nfa > LOG_EPS) return log_nfa;
    rg::Rect r = rec;
    cnt = 0;
    for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.width -= delta; rs[cnt++] = r; }
    walks(); if (cnt) take_best();
    if (log_nfa > LOG_EPS) return log_nfa;
    for (int side = 0; side < 2; side++) {
        r = rec;
        cnt = 0;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
            if (side == 0) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
            else { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
            r.width -= delta;
            rs[cnt++] = r;
        }
        walks(); if (cnt) take_best();
        if (log_nfa > LOG_EPS) return log_nfa;
    }
    if ((rec.width - delta) >= 0.5) { tolerances(); take_best(); } // (the width does not change in the last loop: all five or none)
    return log_nfa;
}

// ---- the sequential stage: one wave per frame (lsd_rg_seq.h) ------------------------------------------------------------------------------
struct SeqParams {
    int F, w, h;
    const int *caddr; const int *frame_base; const float *cdeg; const float2 *ccs; const double *mod; const float *ang;
    float *ang32; float *seed_cs; int *glist; double *rect; size_t rect_stride; int cand_cap; int *cand_cnt; int *status;
    int min_reg_size, list_cap;
    unsigned long long *prof;
    size_t pix_stride; // elements from one frame's map (ang32: the walk's own copy of the angles, a float per pixel: the angle while the pixel is defined and unused) to the next
};
__global__ void __launch_bounds__(256) lsd_rg_fill32(float4 *ang, size_t n4) { // lsd_rg_wlk's map: one float per pixel
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) ang[i] = make_float4(rgs::NOTDEF_F, rgs::NOTDEF_F, rgs::NOTDEF_F, rgs::NOTDEF_F);
}
// the defined pixels' records from lsd_emit's compact lists, and what each would start a region with as a seed
__global__ void __launch_bounds__(256) lsd_rg_scatter(SeqParams P) {
    const int f = blockIdx.y, base = P.frame_base[f], ne = P.frame_base[f + 1] - base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ne) return;
    const int q = P.caddr[base + i] & 0x7fffffff;
    const float d = P.cdeg[base + i];
    P.ang32[(size_t)f * P.pix_stride + q] = d;
    const double a = double(d) * rg::DEG_TO_RADS; // the map value (:566); region_grow starts a region's sums with cos / sin of it as a double (:651-652)
    reinterpret_cast<float2 *>(P.seed_cs)[base + i] = make_float2(float(cos(a)), float(sin(a)));
}
// A workgroup is a bundle of independent waves, one frame each (no LDS, no barrier).  Sixteen waves fill a CU (4 a SIMD, 128 VGPRs each): the frames
// of a batch then sit on F / 16 CUs and leave the others EMPTY -- cuboid_sweep_score's workgroups need a whole CU (160 KB of LDS, 2 x 240 VGPRs a
// SIMD) and would otherwise wait for a frame's 100 ms to pass.
__device__ __forceinline__ void lsd_rg_seq_body(const SeqParams &P) {
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (f >= P.F) return;
#ifdef RGS_PRIO
    __builtin_amdgcn_s_setprio(RGS_PRIO); // (experiment: the walk's instructions ahead of the other waves of its SIMD)
#endif
    const int base = P.frame_base[f];
    rgs::Frame Fr;
    Fr.w = P.w; Fr.h = P.h; Fr.ne = P.frame_base[f + 1] - base;
    Fr.caddr = P.caddr + base; Fr.fre = P.ang32 + (size_t)f * P.pix_stride; Fr.ang = P.ang + (size_t)f * P.w * P.h; Fr.mod = P.mod + (size_t)f * P.w * P.h; Fr.seed_cs = P.seed_cs + 2 * (size_t)base;
    Fr.rect = P.rect + (size_t)f * P.rect_stride; Fr.cand_cap = P.cand_cap; Fr.cand_cnt = P.cand_cnt + f;
    Fr.status = P.status + 4 * f; Fr.min_reg_size = P.min_reg_size; Fr.list_cap = P.list_cap; Fr.prof = P.prof ? P.prof + 16 * (size_t)f : nullptr;
    rgs::List L;
    L.glob = P.glist + (size_t)f * rgs::CAP; L.ring[0] = 0;
    rgs::run_frame<rgs::Wave>(Fr, L);
}
// (Smaller workgroups do not pack: the dispatcher spreads them over the emptiest CUs.  A 96-VGPR build in workgroups of ten frames, meant to sit two
// to a CU, took one CU each, 206 CUs for two detectors, and cuboid_sweep_score waited 27 ms per launch.)
#ifdef RGS_WAVES // (experiment: the walk squeezed into fewer registers -- 6: 80 VGPRs + 32 B of scratch per lane, 8: 64 + 92 B -- to leave more of a SIMD's file to the kernels beside it)
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(RGS_WAVES, RGS_WAVES))) lsd_rg_seq(SeqParams P) { lsd_rg_seq_body(P); }
#else
__global__ void __launch_bounds__(1024) lsd_rg_seq(SeqParams P) { lsd_rg_seq_body(P); }
#endif
// A launch walks up to WLK_SLICES slices of a batch side by side (inside a slice every offset fits 32 bits: lsd_rg_wlk.h addresses base + offset).
constexpr int WLK_SLICES = 8;
struct WlkLaunch { rgl::Batch slice[WLK_SLICES]; int n_slices, waves_per_slice; };
// Two roles per workgroup (lsd_rg_wlk.h): WK walker waves (one lane per frame: seeds and growth), the other waves take the regions the walkers park (rectangle stage),
// through a mailbox in LDS.  A workgroup never waits for another one; its roles share a CU, so workgroup-scope ordering is all the hand-over needs.
template <int NS> struct WlkMail { // the walkers' side of the mailbox policy
    rgw::Mail<NS> &m; int base;
    __device__ __forceinline__ rgw::Mail<NS> &mail() { return m; }
    __device__ __forceinline__ int slot_of(int l) const { return base + l; }
    __device__ __forceinline__ int ring_size() const { return NS; }
    __device__ __forceinline__ void after_iteration(const rgl::Batch &) {}
};
template <int WK, int NWAVES, int ACC> __global__ void __launch_bounds__(64 * NWAVES) lsd_rg_wlk(WlkLaunch L, int blocks_per_slice) {
    constexpr int NS = WK * 64;
    __shared__ rgw::Mail<NS> mail;
    for (int k = threadIdx.x; k < NS; k += 64 * NWAVES) { mail.slot[k].state = 0; mail.slot[k].fl = 0; mail.ring_id[k] = 0; mail.ring_tick[k] = 0; }
    if (threadIdx.x == 0) { mail.tail = 0; mail.head = 0; mail.walkers_done = 0; }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = blockIdx.x / blocks_per_slice;
    const rgl::Batch &B = L.slice[c];
    if (wv < WK) {
        const int f0 = ((blockIdx.x - c * blocks_per_slice) * WK + wv) * 64;
        if (f0 < B.F) { WlkMail<NS> mp{mail, wv * 64}; rgw::run_walker<rgl::LWave, ACC>(B, f0, mp); }
        rgw::wg_release();
        if (lane == 0) rgw::lds_add(&mail.walkers_done, 1);
        return;
    }
#if defined(RGW_PROF)
    unsigned long long rgw_job = 0, rgw_jobs = 0, rgw_idle = 0;
#endif
    for (;;) { // a rectangle wave: the next ticket, its region, the answer
#if defined(RGW_PROF)
        const unsigned long long rgw_t0 = clock64();
#endif
        int t = 0;
        if (lane == 0) t = rgw::lds_add(&mail.head, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        bool quit = false;
        while (__builtin_amdgcn_readfirstlane(rgw::lds_ld(&mail.ring_tick[t % NS])) != t + 1) {
            if (__builtin_amdgcn_readfirstlane(rgw::lds_ld(&mail.walkers_done)) == WK && __builtin_amdgcn_readfirstlane(rgw::lds_ld(&mail.tail)) <= t) { quit = true; break; } // (a walker is done when its last region has been answered)
            __builtin_amdgcn_s_sleep(4);
        }
        if (quit) break;
#if defined(RGW_PROF)
        const unsigned long long rgw_t1 = clock64();
#endif
        rgw::serve_ticket<rgs::Wave, NS>(B, mail, t);
#if defined(RGW_PROF)
        rgw_idle += rgw_t1 - rgw_t0; rgw_job += clock64() - rgw_t1; rgw_jobs++;
#endif
    }
#if defined(RGW_PROF)
    if (lane == 0) { atomicAdd(&rgw::g_rgw_prof[8], rgw_job); atomicAdd(&rgw::g_rgw_prof[9], rgw_jobs); atomicAdd(&rgw::g_rgw_prof[10], rgw_idle); }
#endif
}
// the frames' rectangle lists one after the other (frames in order, seeds in order): cand_base[f] = rectangles of the frames before f
__global__ void __launch_bounds__(1024) lsd_rg_cand_scan(const int *cand_cnt, int F, int *cand_base) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (F + 1023) / 1024;
    int sum = 0;
    for (int k = 0; k < per; k++) { const int f = t * per + k; if (f < F) sum += cand_cnt[f]; }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int v = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - sum;
    for (int k = 0; k < per; k++) { const int f = t * per + k; if (f < F) { cand_base[f] = run; run += cand_cnt[f]; } }
    if (t == 1023) cand_base[F] = part[1023];
}
// rect_improve + the NFA test (lsd.cpp:873-975, :503-505) of one rectangle per wave; line[k] / has[k] in the order of the scan above
// Eight rectangles per wave.  rect_improve starts with rect_nfa of the rectangle as it is, and two thirds of the rectangles are accepted there: that first walk (28
// pixels in 7 rows on average) and its nfa() run for eight rectangles at once, eight lanes each -- every lane of a group evaluates its rectangle's nfa(), which costs
// the wave what one evaluation costs.  The rectangles that go on are then improved one after the other by the whole wave.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RG_IMPROVE_WAVES, RG_IMPROVE_WAVES))) lsd_rg_improve(SeqParams P, const int *cand_base, int n_cand, const double *lgt, float4 *line, uint8_t *has) {
    const int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8, lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    if (base >= n_cand) return;
    const double LOG_NT = 5 * (log10(double(P.w)) + log10(double(P.h))) / 2 + log10(11.0);
    auto frame_of = [&](int wv) { int lo = 0, hi = P.F; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cand_base[mid] <= wv) lo = mid; else hi = mid; } return lo; }; // cand_base[f] <= wv < cand_base[f + 1]
    auto load_rect = [&](int wv, int f, rg::Rect &rec) {
        const double *o = P.rect + (size_t)f * P.rect_stride + (size_t)(wv - cand_base[f]) * 12;
        rec.x1 = o[0]; rec.y1 = o[1]; rec.x2 = o[2]; rec.y2 = o[3]; rec.width = o[4]; rec.x = o[5]; rec.y = o[6]; rec.theta = o[7]; rec.dx = o[8]; rec.dy = o[9]; rec.prec = o[10]; rec.p = o[11];
    };
    auto emit = [&](int wv, rg::Rect rec, double log_nfa) {
        const bool ok = log_nfa > LOG_EPS;
        if (ok) {
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= LSD_SCALE; rec.y1 /= LSD_SCALE; rec.x2 /= LSD_SCALE; rec.y2 /= LSD_SCALE;
            line[wv] = make_float4(float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2));
        }
        has[wv] = ok ? 1 : 0;
    };
    // ---- rect_nfa of eight rectangles
    const int mine = base + grp;
    const bool on = mine < n_cand;
    const int f1 = frame_of(on ? mine : n_cand - 1);
    rg::Rect r1;
    load_rect(on ? mine : n_cand - 1, f1, r1);
    double log1;
    {
        RGI_T0;
        RectSpan S; S.set(r1, P.h);
        const float *deg = P.ang + (size_t)f1 * P.w * P.h;
        int tot = 0, alg[1] = {0};
        if (on)
            for (int y = S.y_first + sub; y <= S.y_hi; y += 8) {
                int lx, rx;
                if (!S.row(y, P.w, lx, rx)) continue;
                tot += rx - lx + 1;
                rg_row_count<1>(deg + (size_t)y * P.w, lx, rx, 0, 1, r1.theta, &r1.prec, alg);
            }
        for (int off = 1; off < 8; off <<= 1) { tot += __shfl_xor(tot, off); alg[0] += __shfl_xor(alg[0], off); }
        RGI_T1(0);
        { RGI_T0; log1 = rg_nfa(tot, alg[0], r1.p, LOG_NT, lgt); RGI_T1(1); }
    }
    if (on && sub == 0 && log1 > LOG_EPS) emit(mine, r1, log1);
    // ---- the others, one at a time
    unsigned long long todo = __ballot(on && sub == 0 && !(log1 > LOG_EPS));
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int wv = base + (l >> 3), f = __shfl(f1, l);
        const double first = __shfl(log1, l);
        rg::Rect rec;
        load_rect(wv, f, rec);
        const AngMap Fr = {P.w, P.h, P.ang + (size_t)f * P.w * P.h};
        const double log_nfa = rg_rect_improve(Fr, rec, LOG_NT, lane, lgt, first);
        if (lane == 0) emit(wv, rec, log_nfa);
    }
}
} // namespace

// ---- host side of the sequential stage ----------------------------------------------------------------------------------------------------
struct LsdSeq {
    int F = 0, w = 0, h = 0; int cand_cap = 0; size_t cap_lines = 0;
    float *d_pix = nullptr; bool pix_borrowed = false; // lsd_rg_seq's map, a float per pixel (borrowed: the caller's scratch, not freed here)
    rgl::Ent *d_elist = nullptr; float *d_ang32 = nullptr; int *d_order = nullptr; // the lane-per-frame walk (lsd_rg_wlk): region lists of 8-byte entries, a float per pixel, the frames sorted by work
    int *d_glist = nullptr, *d_cand_cnt = nullptr, *d_cand_base = nullptr, *d_status = nullptr, *d_frame_base = nullptr;
    double *d_rect = nullptr, *d_lgt = nullptr;
    uint8_t *d_has = nullptr;
    float4 *d_line = nullptr;
    std::vector<int> h_base, h_status; std::vector<uint8_t> h_has; std::vector<float4> h_line;
};
void lsd_seq_destroy(LsdSeq *r) {
    if (!r) return;
    void *ptrs[] = {r->d_elist, r->d_ang32, r->d_order, r->pix_borrowed ? nullptr : r->d_pix, r->d_glist, r->d_cand_cnt, r->d_cand_base, r->d_status, r->d_frame_base, r->d_rect, r->d_lgt, r->d_has, r->d_line};
    for (void *p : ptrs) if (p) hipFree(p);
    delete r;
}
// The region stage of F frames, one wave per frame.  lines[f] = x1 y1 x2 y2 floats in the reference's emission order.  CS_ERR_CAPACITY: a region
// outgrew the wave's list (rgs::CAP pixels) or a frame its rectangle list -- the caller then runs the host stage for the batch.
int lsd_seq_run(cs_ctx *ctx, LsdSeq **handle, int F, int w, int h, const float *d_ang, const double *d_mod, const int *d_caddr, const float *d_cdeg, const float2 *d_ccs, const int *frame_base,
                std::vector<std::vector<float>> &lines, long *stats /* [0] region_grow calls, [1] window fetches, [2] rectangles at rect_improve, [3] regions at the rectangle stage */,
                void (*before_seq)(void *), void (*after_seq)(void *), void *gate_arg /* the front-end runner's phase gate: called in front of the lsd_rg_seq launch and once it has left the GPU; may be NULL */,
                int grp_p /* 0: lsd_rg_seq, one wave per frame; 65: lsd_rg_wlk, walker waves with one lane per frame + rectangle waves */,
                int waves_per_workgroup /* frames per workgroup of lsd_rg_seq: 16 packs a batch onto F / 16 CUs and leaves the others empty; 4 spreads it over the chip (the alternating runner, where
                                           every CU is busy anyway: 128 -> 104 ms per launch there) */,
                void *scratch, size_t scratch_bytes /* memory the caller has no use for while the stage runs: lsd_rg_seq's pixel records go there when it is large enough */,
                bool pix_ready /* lsd_emit<true> already left the pixel records at the head of `scratch` and the seeds' cos / sin in d_ccs: no fill, no scatter */,
                bool walk_bg /* lsd_rg_seq on the context's lowest-priority background stream (cs_ctx::bg_begin): the walk paces itself (it takes its ~100 ms whatever runs beside it),
                                so the context's own stream can have the priority of the short kernels in front of it and behind it */) {
    LsdSeq *r = *handle;
    if (w > 0xffff || h > 0x7fff) return CS_ERR_CAPACITY; // (the region list packs x | y << 16)
    int max_ne = 0;
    for (int f = 0; f < F; f++) max_ne = std::max(max_ne, frame_base[f + 1] - frame_base[f]);
    int rc;
#define RA_(call) do { rc = (call); if (rc != CS_OK) return rc; } while (0)
    if (!r || r->F < F || r->w != w || r->h != h) {
        lsd_seq_destroy(r);
        r = new LsdSeq();
        *handle = r;
        r->F = F; r->w = w; r->h = h; r->cand_cap = 4096;
        // (d_pix, d_glist: lsd_rg_seq's; d_ang32, d_elist: lsd_rg_wlk's -- allocated by the mode that runs)
        RA_(cs_dalloc(ctx, &r->d_order, (size_t)F)); RA_(cs_dalloc(ctx, &r->d_rect, (size_t)F * (r->cand_cap * 12 + PAD_RECT)));
        RA_(cs_dalloc(ctx, &r->d_cand_cnt, (size_t)F)); RA_(cs_dalloc(ctx, &r->d_cand_base, (size_t)F + 1));
        RA_(cs_dalloc(ctx, &r->d_status, (size_t)F * 4)); RA_(cs_dalloc(ctx, &r->d_frame_base, (size_t)F + 1));
        RA_(cs_dalloc(ctx, &r->d_lgt, (size_t)LG_N));
        CS_LAUNCH(ctx, "lsd_rg_lgamma_table", lsd_rg_lgamma_table, dim3(LG_N / 256), dim3(256), 0, r->d_lgt);
    }
    RA_(cs_h2d(ctx, r->d_frame_base, frame_base, (size_t)F + 1));
    SeqParams S;
    S.F = F; S.w = w; S.h = h; S.caddr = d_caddr; S.frame_base = r->d_frame_base; S.cdeg = d_cdeg; S.ccs = d_ccs; S.mod = d_mod; S.ang = d_ang;
    S.ang32 = nullptr; S.seed_cs = reinterpret_cast<float *>(const_cast<float2 *>(d_ccs)); /* in place: lsd_rg_scatter reads a pixel's cos / sin (floats of the float angle) and leaves the seed's (of the double angle) in the same 8 bytes */ S.glist = r->d_glist; S.rect = r->d_rect; S.cand_cap = r->cand_cap; S.cand_cnt = r->d_cand_cnt; S.status = r->d_status;
    const double LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    S.min_reg_size = int(-LOG_NT / std::log10(rg::ANG_TH / 180));
    S.list_cap = rgs::CAP;
    if (const char *e = getenv("CUBESLAM_LSD_SEQ_CAP")) S.list_cap = std::max(2, std::min(rgs::CAP, atoi(e))); // (tests: a small cap walks the fallback to the host stage)
    if (grp_p != 0 && grp_p != 65) return CS_ERR_BAD_ARG;
    const bool grp = grp_p == 65, pad = grp; // walker + rectangle waves (lsd_rg_wlk.h): a float per pixel, 8-byte list entries, padded strides
    const size_t rect_stride = (size_t)r->cand_cap * 12 + (pad ? PAD_RECT : 0), list_stride = (size_t)rgl::CAP + (pad ? PAD_LIST : 0);
    S.pix_stride = (size_t)w * h + (pad ? PAD_PX : 0); S.rect_stride = rect_stride;
    S.prof = nullptr;
#if defined(RGS_PROFILE)
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) hipMalloc((void **)&d_prof, 4096 * 16 * sizeof(unsigned long long));
    hipMemsetAsync(d_prof, 0, 4096 * 16 * sizeof(unsigned long long), ctx->stream);
    S.prof = d_prof;
#endif
    const size_t npx = (size_t)F * S.pix_stride;
    if (grp) {
        const size_t head = (size_t)((w + 8 + 3) & ~3); // undefined pixels in front of frame 0 (the walkers read "row -1" without a test)
        if (!r->d_ang32) RA_(cs_dalloc(ctx, &r->d_ang32, head + (size_t)r->F * ((size_t)w * h + PAD_PX) + 16));
        if (!r->d_elist) RA_(cs_dalloc(ctx, &r->d_elist, (size_t)r->F * ((size_t)rgl::CAP + PAD_LIST) + 16));
        S.ang32 = r->d_ang32 + head;
        CS_LAUNCH(ctx, "lsd_rg_fill", lsd_rg_fill32, dim3((unsigned)(((npx + head) / 4 + 1 + 255) / 256)), dim3(256), 0, reinterpret_cast<float4 *>(r->d_ang32), (npx + head) / 4 + 1);
    } else {
        if (!r->d_pix) {
            if (scratch && scratch_bytes >= (size_t)r->F * w * h * sizeof(float) + 64) { r->d_pix = static_cast<float *>(scratch); r->pix_borrowed = true; }
            else RA_(cs_dalloc(ctx, &r->d_pix, (size_t)r->F * w * h + 16));
        }
        if (!r->d_glist) RA_(cs_dalloc(ctx, &r->d_glist, (size_t)r->F * rgs::CAP));
        S.ang32 = r->d_pix; S.glist = r->d_glist;
        if (pix_ready && !r->pix_borrowed) { ctx->err = "lsd_seq_run: the pixel records were announced in the scratch buffer, but the stage does not use it for this batch"; return CS_ERR_BAD_ARG; }
        if (!pix_ready) CS_LAUNCH(ctx, "lsd_rg_fill", lsd_rg_fill32, dim3((unsigned)((npx / 4 + 1 + 255) / 256)), dim3(256), 0, reinterpret_cast<float4 *>(r->d_pix), npx / 4 + 1);
    }
    if (!pix_ready) CS_LAUNCH(ctx, "lsd_rg_scatter", lsd_rg_scatter, dim3((max_ne + 255) / 256, F), dim3(256), 0, S);
    int wpb = std::max(1, std::min(16, waves_per_workgroup)); // waves (= frames) per workgroup
    if (const char *e = getenv("CUBESLAM_LSD_SEQ_WPB")) wpb = std::max(1, std::min(16, atoi(e)));
    if (before_seq) before_seq(gate_arg);
    if (grp) {
        const size_t npx = (size_t)w * h, head = (size_t)((w + 8 + 3) & ~3);
        int chunk = (int)std::min<size_t>(4096, 0xffffffffull / (npx * sizeof(double))); // (the norms: the widest per-pixel array the walk reads) // every offset of a launch fits 32 bits (lsd_rg_wlk.h addresses base + offset)
        chunk = std::min(chunk, 1024);
        if (chunk < 1) return CS_ERR_CAPACITY;
        std::vector<int> order((size_t)F);
        for (int c0 = 0; c0 < F; c0 += chunk) { // per launch: its frames sorted by their number of defined pixels, so that frames of similar work share a wave
            const int fc = std::min(chunk, F - c0);
            for (int k = 0; k < fc; k++) order[(size_t)c0 + k] = k;
            std::stable_sort(order.begin() + c0, order.begin() + c0 + fc, [&](int a, int b) { return frame_base[c0 + a + 1] - frame_base[c0 + a] > frame_base[c0 + b + 1] - frame_base[c0 + b]; });
        }
        RA_(cs_h2d(ctx, r->d_order, order.data(), (size_t)F));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (order is a local)
        for (int l0 = 0; l0 < F; l0 += chunk * WLK_SLICES) { // one launch per WLK_SLICES slices
            WlkLaunch L;
            L.n_slices = 0; L.waves_per_slice = (chunk + 63) / 64;
            for (int c0 = l0; c0 < F && L.n_slices < WLK_SLICES; c0 += chunk) {
                const int fc = std::min(chunk, F - c0);
                rgl::Batch &B = L.slice[L.n_slices++];
                B.F = fc; B.w = w; B.h = h; B.npx = (int)npx; B.order = r->d_order + c0; B.ang_stride = (int)S.pix_stride; B.list_stride = (int)list_stride; B.rect_stride = (int)rect_stride; B.ang_head = (int)head;
                B.caddr = d_caddr + frame_base[c0]; B.frame_base = r->d_frame_base + c0; B.ang = r->d_ang32 + (size_t)c0 * S.pix_stride; B.mod = d_mod + (size_t)c0 * npx; B.seed_cs = S.seed_cs + 2 * (size_t)frame_base[c0];
                B.list = r->d_elist + (size_t)c0 * list_stride; B.list_cap = std::min(S.list_cap, rgl::CAP); B.rect = r->d_rect + (size_t)c0 * rect_stride; B.cand_cap = r->cand_cap; B.cand_cnt = r->d_cand_cnt + c0; B.status = r->d_status + 4 * (size_t)c0;
                B.min_reg_size = S.min_reg_size; B.max_iters = (int)std::min<size_t>(64 * npx, 0x7fffffff);
            }
            int wk = 1, nw = 8, acc = 1; // walker waves / waves per workgroup / accepted pixels per iteration (measured on 1 024 distinct frames: 454 ms a pass for 1,8,1; 466 for 2,8,1 on half the CUs; 505 for 1,8,2)
            if (const char *e = getenv("CUBESLAM_LSD_WLK")) sscanf(e, "%d,%d,%d", &wk, &nw, &acc);
            const int bps = (L.waves_per_slice + wk - 1) / wk, groups = L.n_slices * bps;
#define WLK_(WK, NW, AC) else if (wk == WK && nw == NW && acc == AC) CS_LAUNCH(ctx, "lsd_rg_wlk", (lsd_rg_wlk<WK, NW, AC>), dim3(groups), dim3(64 * NW), 0, L, bps)
            if (false) {}
            WLK_(1, 8, 1); WLK_(1, 8, 2); WLK_(1, 4, 1); WLK_(2, 8, 1); WLK_(2, 16, 1); WLK_(4, 16, 1); WLK_(8, 16, 1);
            else { ctx->err = "CUBESLAM_LSD_WLK: no such shape (walkers,waves,accepts)"; return CS_ERR_BAD_ARG; }
#undef WLK_
#if defined(RGW_PROF)
            { unsigned long long h[16]; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(rgw::g_rgw_prof), sizeof h); unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(rgw::g_rgw_prof), z, sizeof z);
              const double it = (double)std::max<unsigned long long>(h[5], 1);
              fprintf(stderr, "[rgw prof] walker iterations %.0f (all waves); clocks per iteration: issue %.0f wait %.0f grow %.0f seed %.0f mail %.0f = %.0f; rectangle waves: %.0f jobs, %.0f clocks each, waiting %.0f clocks per job\n", it, h[0] / it, h[1] / it, h[2] / it,
                      h[3] / it, h[4] / it, (h[0] + h[1] + h[2] + h[3] + h[4]) / it, (double)h[9], h[9] ? (double)h[8] / h[9] : 0.0, h[9] ? (double)h[10] / h[9] : 0.0); }
#endif
        }
    } else if (walk_bg) {
        CS_HIP(ctx, ctx->bg_begin());
        cs_ctx::pending_ev pe;
        if (ctx->timing) { pe.name = "lsd_rg_seq"; pe.a = ctx->get_event(); pe.b = ctx->get_event(); hipEventRecord(pe.a, ctx->bg_stream); }
        hipLaunchKernelGGL(lsd_rg_seq, dim3((F + wpb - 1) / wpb), dim3(64 * wpb), 0, ctx->bg_stream, S);
        if (ctx->timing) { hipEventRecord(pe.b, ctx->bg_stream); ctx->pending.push_back(pe); }
        CS_HIP(ctx, ctx->bg_end());
    } else
    CS_LAUNCH(ctx, "lsd_rg_seq", lsd_rg_seq, dim3((F + wpb - 1) / wpb), dim3(64 * wpb), 0, S);
    CS_LAUNCH(ctx, "lsd_rg_cand_scan", lsd_rg_cand_scan, dim3(1), dim3(1024), 0, r->d_cand_cnt, F, r->d_cand_base);
    r->h_base.resize((size_t)F + 1); r->h_status.resize((size_t)F * 4);
    RA_(cs_d2h(ctx, r->h_base.data(), r->d_cand_base, (size_t)F + 1));
    RA_(cs_d2h(ctx, r->h_status.data(), r->d_status, (size_t)F * 4));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (after_seq) after_seq(gate_arg);
#if defined(RGS_PROFILE)
    { std::vector<unsigned long long> all((size_t)F * 16); hipMemcpy(all.data(), S.prof, all.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long hp[16] = {0}; for (int f = 0; f < F; f++) for (int k = 0; k < 16; k++) hp[k] += all[(size_t)f * 16 + k];
      const char *nm[6] = {"fetch", "expand", "grow", "rect+refine", "seed batch", "frame"};
      for (int k = 0; k < 6; k++) fprintf(stderr, "[rgs prof] %-12s entries/frame %9.0f  ms/frame %8.2f  us/entry %6.2f\n", nm[k], double(hp[2 * k + 1]) / F, hp[2 * k] / 1e5 / F, hp[2 * k + 1] ? hp[2 * k] / 100.0 / hp[2 * k + 1] : 0.0); }
#endif
    long grows = 0, fetches = 0, nreg = 0;
    bool bad = false;
    for (int f = 0; f < F; f++) { grows += r->h_status[4 * f]; bad = bad || r->h_status[4 * f + 1] != 0; nreg += r->h_status[4 * f + 2]; fetches += r->h_status[4 * f + 3]; }
    const int n_cand = r->h_base[F];
    if (stats) { stats[0] = grows; stats[1] = fetches; stats[2] = n_cand; stats[3] = nreg; }
    if (bad) return CS_ERR_CAPACITY;
    lines.assign((size_t)F, {});
    if (n_cand == 0) return CS_OK;
    if ((size_t)n_cand > r->cap_lines) {
        if (r->d_line) hipFree(r->d_line); if (r->d_has) hipFree(r->d_has);
        r->d_line = nullptr; r->d_has = nullptr; r->cap_lines = 0;
        const size_t cap = (size_t)n_cand + n_cand / 4 + 1024;
        RA_(cs_dalloc(ctx, &r->d_line, cap)); RA_(cs_dalloc(ctx, &r->d_has, cap));
        r->cap_lines = cap;
    }
    CS_LAUNCH(ctx, "lsd_rg_improve", lsd_rg_improve, dim3((n_cand + 31) / 32), dim3(256), 0, S, r->d_cand_base, n_cand, r->d_lgt, r->d_line, r->d_has); // eight rectangles per wave
#if defined(RGI_PROF)
    { unsigned long long h[4]; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rgi_prof), sizeof h); unsigned long long z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_rgi_prof), z, sizeof z);
      fprintf(stderr, "[rgi prof] %d rectangles: walks %.0f calls %.2f us each = %.1f us per rectangle; nfa %.0f calls %.2f us each = %.1f us per rectangle\n", n_cand, (double)h[2], h[2] ? h[0] / 100.0 / h[2] : 0.0,
              h[0] / 100.0 / n_cand, (double)h[3], h[3] ? h[1] / 100.0 / h[3] : 0.0, h[1] / 100.0 / n_cand); }
#endif
    r->h_has.resize((size_t)n_cand); r->h_line.resize((size_t)n_cand);
    RA_(cs_d2h(ctx, r->h_has.data(), r->d_has, (size_t)n_cand));
    RA_(cs_d2h(ctx, r->h_line.data(), r->d_line, (size_t)n_cand));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
#undef RA_
#pragma omp parallel for schedule(static) num_threads(std::max(1, std::min(ctx->host_threads, F)))
    for (int f = 0; f < F; f++)
        for (int k = r->h_base[f]; k < r->h_base[f + 1]; k++)
            if (r->h_has[k]) { const float4 v = r->h_line[k]; lines[f].push_back(v.x); lines[f].push_back(v.y); lines[f].push_back(v.z); lines[f].push_back(v.w); }
    return CS_OK;
}
