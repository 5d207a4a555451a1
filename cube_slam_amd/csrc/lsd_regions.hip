// lsd_regions.hip -- LSD region growing, rectangle fitting and NFA validation on the device (reference line_lbd/libs/lsd.cpp:464-1155:
// flsd's seed loop, region_grow, region2rect, get_theta, refine, reduce_region_radius, rect_improve, rect_nfa, nfa).
//
// The reference is a greedy SEQUENCE: seeds are visited in raster order (lsd.cpp:477-480 walks `list` by index, see LsdHost::run in lsd.hip),
// every region sees the `used` marks of all regions before it, and every accepted pixel changes the angle the next test uses.  The device
// formulation keeps the sequential result exactly and finds the parallelism ACROSS regions, as a speculative fixed point:
//
//   * rank = position of a seed in raster order.  The owner word of a pixel holds the rank of the seed that claims it (all ones = free).
//     A region of rank s may take any pixel that no LOWER rank holds (a pixel held by a higher rank is taken away from it): at the fixed
//     point, "a lower rank holds q" says exactly what the reference's used[q] says when seed s has its turn.
//   * a transaction = one seed's whole turn (grow; from min_reg_size pixels on rectangle + refine, which may re-grow with a tighter tolerance
//     and cut the radius), run by ONE lane against the live owner map; claims are 64-bit atomic mins.  lsd_rg_txn.h holds the transaction
//     as a resumable state machine (one pixel of region_grow per step, so the 64 lanes of a wave advance 64 regions together instead of
//     waiting for the longest) and the rules that decide when a transaction must run again: tile marks below its rank on the tiles it read.
//     Marks live in LDS; one workgroup per frame runs the rounds of its frame with workgroup barriers only.
//   * by induction on the rank the fixed point is unique and equals the sequential result (the lowest rank depends on nothing; once the
//     ranks < s are final nothing below s changes, so s's next run is final).  tools/lsd_sim/txn_sim.cpp compiles the same transaction
//     source for the host and interleaves a thousand lanes step by step against the oracle's sequential algorithm: identical owner map
//     and line candidates, 20-35 rounds and about 12-20 times the sequential pixel steps on textured 640x480 frames.
//   * first round: seeds without an aligned lower-ranked neighbour (the others are almost always swallowed by an earlier region; if not
//     they start in the next round -- any start order converges to the same result).
//
// After the fixed point every region that passed the rectangle stage goes to lsd_rg_lines: one WAVE per region, region2rect in the
// reference's summation order (all lanes redundantly), rect_improve / rect_nfa with the rectangle's rows spread over the lanes (the row
// limits advance by integer steps, lsd.cpp:1057-1095, so they have a closed form), nfa on every lane.  Lines are compacted in seed order,
// like the reference emits them.
//
// Arithmetic: doubles as in the reference, no FMA contraction (the Makefile's -ffp-contract=off); cos(float) / sin(float) of the pixel angles
// are glibc's cosf / sinf restated (glibc_sincosf.h: equal on every float of the domain); fastAtan2 is the same polynomial as the host's.
// cos / sin of the rectangle angle and the transcendental functions of nfa() are the device library's doubles, which may differ from glibc
// in the last bit: that reaches the output only through a float rounding of an end point (about 1e-8 per line) or through a rectangle
// whose log-NFA is within 1e-15 of the threshold or of a competing variant.
#include "common.h"

#include <cfloat>
#include <climits>
#include <cmath>
#include <vector>

#include "lsd_regions.h"
#include "lsd_rg_txn.h"
#include "lsd_rg_seq.h"

namespace {
constexpr double PI_ = rg::PI_, LOG_EPS = 0.0, LSD_SCALE = 0.8;
constexpr int RG_LISTS = 4; // scratch lists of a lane (region, old footprint, first growth, tiles)

struct RgParams {
    int F, w, h, lanes;
    const int *caddr; const int *frame_base; // device copies
    const double *ang, *mod;
    rg::u64 *own; int *fp_off, *fp_cnt, *fp_cap, *fp_nt; unsigned *execs; uint8_t *flag; double *reg_angle;
    int *pool; int pool_per_frame; int *pool_head; int *dirty; int *scratch; int *status;
    int min_reg_size, max_rounds;
    unsigned long long *prof; // RG_PROFILE: wall-clock ticks (100 MHz) and entries per phase, summed over the waves' first active lanes
};

__device__ __forceinline__ rg::Frame rg_frame(const RgParams &P, int f, int *chg) {
    const int base = P.frame_base[f];
    rg::Frame Fr;
    Fr.w = P.w; Fr.h = P.h; Fr.ne = P.frame_base[f + 1] - base;
    Fr.caddr = P.caddr + base;
    Fr.ang = P.ang + (size_t)f * P.w * P.h; Fr.mod = P.mod + (size_t)f * P.w * P.h;
    Fr.own = P.own + (size_t)f * P.w * P.h;
    Fr.fp_off = P.fp_off + base; Fr.fp_cnt = P.fp_cnt + base; Fr.fp_cap = P.fp_cap + base; Fr.fp_nt = P.fp_nt + base; Fr.execs = P.execs + base;
    Fr.flag = P.flag + base; Fr.reg_angle = P.reg_angle + base;
    Fr.pool = P.pool + (size_t)f * P.pool_per_frame; Fr.pool_cap = P.pool_per_frame; Fr.pool_head = P.pool_head + f;
    Fr.chg = chg; Fr.tw = (P.w + rg::TILE - 1) / rg::TILE;
    Fr.status = P.status + 4 * f;
    Fr.min_reg_size = P.min_reg_size;
    return Fr;
}

// one workgroup per frame: rounds of transactions until no seed is dirty
__global__ void __launch_bounds__(1024) lsd_rg_fixpoint(RgParams P) {
    extern __shared__ int rg_lds[];
    const int tw = (P.w + rg::TILE - 1) / rg::TILE, th = (P.h + rg::TILE - 1) / rg::TILE, ntile = tw * th;
    int *chg = rg_lds;              // ntile
    int *ctl = rg_lds + ntile;      // [0] dirty count, [1] next dirty count (long transactions), [2] work counter, [3] next dirty count (short ones)
    const int f = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
#if defined(RG_PROFILE)
    const unsigned long long c_start = wall_clock64();
#endif
    const rg::Frame Fr = rg_frame(P, f, chg);
    const int ne = Fr.ne;
    int *dirty_a = P.dirty + 3 * (size_t)P.frame_base[f], *dirty_b = dirty_a + ne, *dirty_s = dirty_b + ne;
    rg::Txn T;
    T.L = P.scratch + ((size_t)f * NT + tid) * RG_LISTS * rg::CAP; T.E = T.L + rg::CAP; T.P1 = T.E + rg::CAP; T.TL = T.P1 + rg::CAP;
    for (int q = tid; q < P.w * P.h; q += NT) Fr.own[q] = rg::FREE;
    for (int i = tid; i < ne; i += NT) { Fr.fp_cnt[i] = 0; Fr.fp_off[i] = 0; Fr.fp_cap[i] = 0; Fr.fp_nt[i] = 0; Fr.execs[i] = 0; Fr.flag[i] = 0; }
    for (int t = tid; t < ntile; t += NT) chg[t] = rg::INF;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; *Fr.pool_head = 0; Fr.status[0] = 0; Fr.status[1] = 0; Fr.status[2] = 0; Fr.status[3] = 0; }
    __syncthreads();
    for (int i = tid; i < ne; i += NT) if (rg::is_initial(Fr, i)) dirty_a[rg::add32(&ctl[0], 1)] = i;
    __syncthreads();
    int *cur = dirty_a, *nxt = dirty_b;
    int rounds = 0;
    for (;;) {
        const int n = ctl[0];
        if (n == 0 || rounds >= P.max_rounds) break;
        rounds++;
        T.phase = rg::PH_IDLE;
        while (T.phase != rg::PH_DONE) rg::step(Fr, T, cur, n, ctl); // the lanes pull transactions until the round's list is empty
        __syncthreads();
        // who runs next round: the long ones first (the round ends with its longest transaction)
        for (int i = tid; i < ne; i += NT)
            if (rg::is_dirty(Fr, i)) { if (Fr.fp_cnt[i] >= 48) nxt[rg::add32(&ctl[1], 1)] = i; else dirty_s[rg::add32(&ctl[3], 1)] = i; }
        __syncthreads();
        const int nl = ctl[1], ns = ctl[3];
        for (int k = tid; k < ns; k += NT) nxt[nl + k] = dirty_s[k];
        for (int t = tid; t < ntile; t += NT) chg[t] = rg::INF;
        __syncthreads();
        if (tid == 0) { ctl[0] = nl + ns; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; }
        int *sw = cur; cur = nxt; nxt = sw;
        __syncthreads();
    }
#if defined(RG_PROFILE)
    for (int k = 0; k < 7; k++) { atomicAdd((unsigned long long *)&P.prof[2 * k], T.prof[k]); atomicAdd((unsigned long long *)&P.prof[2 * k + 1], (unsigned long long)T.profn[k]); }
    { const unsigned long long c1 = wall_clock64(); if (tid == 0) atomicAdd((unsigned long long *)&P.prof[14], c1 - c_start); }
#endif
    atomicAdd(&Fr.status[2], (int)T.n_exec);
    atomicAdd(&Fr.status[3], (int)T.n_steps);
    if (tid == 0) { Fr.status[0] = rounds; if (ctl[0] != 0) Fr.status[1] = 1; }
}

// ---- lines: one wave per candidate region ------------------------------------------------------------------------------------------------
__device__ double rg_log_gamma(double x) { // lsd.cpp:70,124-160
    if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
    const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0;
    for (int n = 0; n < 7; ++n) { a -= log(x + double(n)); b += q[n] * pow(x, double(n)); }
    return a + log(b);
}
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
// rect_nfa (:977-1098): the four corners ordered like the reference's std::sort + selection, rows counted by the lanes of the wave
__device__ double rg_rect_nfa(const rg::Frame &F, const rg::Rect &rec, double LOG_NT, int lane, const double *lgt) {
    const double half_width = rec.width / 2.0, dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    int ox[4], oy[4]; bool taken[4] = {false, false, false, false};
    ox[0] = int(rec.x1 - dyhw); oy[0] = int(rec.y1 + dxhw); ox[1] = int(rec.x2 - dyhw); oy[1] = int(rec.y2 + dxhw);
    ox[2] = int(rec.x2 + dyhw); oy[2] = int(rec.y2 - dxhw); ox[3] = int(rec.x1 + dyhw); oy[3] = int(rec.y1 - dxhw);
    for (int i = 1; i < 4; i++) // std::sort by (x, y): insertion sort gives the same order for distinct keys, ties are equal points
        for (int j = i; j > 0 && (ox[j] < ox[j - 1] || (ox[j] == ox[j - 1] && oy[j] < oy[j - 1])); j--) { int t = ox[j]; ox[j] = ox[j - 1]; ox[j - 1] = t; t = oy[j]; oy[j] = oy[j - 1]; oy[j - 1] = t; }
    int mn = 0, mx = 0;
    for (int i = 1; i < 4; ++i) { if (oy[mn] > oy[i]) mn = i; if (oy[mx] < oy[i]) mx = i; }
    taken[mn] = true;
    int lm = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (lm < 0) lm = i; else if (ox[lm] > ox[i]) lm = i; }
    taken[lm] = true;
    int rm = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (rm < 0) rm = i; else if (ox[rm] < ox[i]) rm = i; }
    taken[rm] = true;
    int tl = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (tl < 0) tl = i; else if (ox[tl] > ox[i]) tl = i; }
    // integer divisions and the tailp->x comparisons are the reference's (:1057-1065); the steps are integers
    const int flstep = (oy[mn] != oy[lm]) ? (ox[mn] - ox[lm]) / (oy[mn] - oy[lm]) : 0;
    const int slstep = (oy[lm] != ox[tl]) ? (ox[lm] - ox[tl]) / (oy[lm] - ox[tl]) : 0;
    const int frstep = (oy[mn] != oy[rm]) ? (ox[mn] - ox[rm]) / (oy[mn] - oy[rm]) : 0;
    const int srstep = (oy[rm] != ox[tl]) ? (ox[rm] - ox[tl]) / (oy[rm] - ox[tl]) : 0;
    // the limits move after every row INSIDE the image (the reference `continue`s past the stepping for the others): by the first step while
    // y < leftmost.y (rightmost.y), by the second from then on -- a closed form per row, so the lanes take rows independently
    int total_pts = 0, alg_pts = 0;
    const int y_lo = oy[mn], y_hi = min(oy[mx], F.h - 1), y_first = max(y_lo, 0);
    for (int y = y_first + lane; y <= y_hi; y += 64) {
        const long r = (long)y - y_first; // rows stepped before this one: y_first .. y-1
        auto adv = [&](int first, int second, int ysw) -> long { // sum over y' in [y_first, y) of (y' >= ysw ? second : first)
            long nf = (long)ysw - y_first; if (nf < 0) nf = 0; if (nf > r) nf = r;
            return nf * first + (r - nf) * second;
        };
        const long lx = max((long)ox[mn] + adv(flstep, slstep, oy[lm]), 0L), rx = min((long)ox[mn] + adv(frstep, srstep, oy[rm]), (long)F.w - 1);
        for (long x = lx; x <= rx; ++x) {
            ++total_pts;
            if (rg::aligned_ang(F.ang[(int)(y * F.w + x)], rec.theta, rec.prec)) ++alg_pts;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { total_pts += __shfl_xor(total_pts, off); alg_pts += __shfl_xor(alg_pts, off); }
    return rg_nfa(total_pts, alg_pts, rec.p, LOG_NT, lgt);
}
__device__ double rg_rect_improve(const rg::Frame &F, rg::Rect &rec, double LOG_NT, int lane, const double *lgt = nullptr) { // :873-975
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rg_rect_nfa(F, rec, LOG_NT, lane, lgt);
    if (log_nfa > LOG_EPS) return log_nfa;
    rg::Rect r = rec;
    for (int n = 0; n < 5; ++n) { r.p /= 2; r.prec = r.p * PI_; const double v = rg_rect_nfa(F, r, LOG_NT, lane, lgt); if (v > log_nfa) { log_nfa = v; rec = r; } }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.width -= delta; const double v = rg_rect_nfa(F, r, LOG_NT, lane, lgt); if (v > log_nfa) { rec = r; log_nfa = v; } }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
        r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; r.width -= delta;
        const double v = rg_rect_nfa(F, r, LOG_NT, lane, lgt); if (v > log_nfa) { rec = r; log_nfa = v; } }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
        r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; r.width -= delta;
        const double v = rg_rect_nfa(F, r, LOG_NT, lane, lgt); if (v > log_nfa) { rec = r; log_nfa = v; } }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.p /= 2; r.prec = r.p * PI_; const double v = rg_rect_nfa(F, r, LOG_NT, lane, lgt); if (v > log_nfa) { rec = r; log_nfa = v; } }
    return log_nfa;
}

// one wave: the region of rank s of frame f from its footprint to a segment (original-image coordinates) or nothing
__device__ bool rg_region_line(const RgParams &P, int f, int s, int lane, float4 &out) {
    const rg::Frame Fr = rg_frame(P, f, nullptr);
    const int *L = Fr.pool + Fr.fp_off[s];
    const int n = Fr.fp_cnt[s];
    const double prec = PI_ * rg::ANG_TH / 180, p = rg::ANG_TH / 180;
    const double LOG_NT = 5 * (log10(double(P.w)) + log10(double(P.h))) / 2 + log10(11.0);
    rg::Rect rec;
    rg::region2rect(Fr, L, n, Fr.reg_angle[s], prec, p, rec); // every lane the same sums, in list order
    const double log_nfa = rg_rect_improve(Fr, rec, LOG_NT, lane);
    if (!(log_nfa > LOG_EPS)) return false;
    rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
    rec.x1 /= LSD_SCALE; rec.y1 /= LSD_SCALE; rec.x2 /= LSD_SCALE; rec.y2 /= LSD_SCALE;
    out = make_float4(float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2));
    return true;
}
// candidate list: (frame, rank) pairs; one wave each.  line[rank] = x1 y1 x2 y2, has[rank] = 1 -- or, compact, line[candidate] / has[candidate]
__global__ void __launch_bounds__(256) lsd_rg_lines(RgParams P, const int2 *cand, int n_cand, float4 *line, uint8_t *has, int compact) {
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= n_cand) return;
    const int f = cand[wv].x, s = cand[wv].y;
    float4 out = make_float4(0, 0, 0, 0);
    const bool ok = rg_region_line(P, f, s, lane, out);
    if (lane == 0) {
        const size_t o = compact ? (size_t)wv : (size_t)P.frame_base[f] + s;
        if (ok) line[o] = out;
        has[o] = ok ? 1 : 0;
    }
}

// per frame: candidates (flag bits 0 and 1) appended to the global list
__global__ void __launch_bounds__(256) lsd_rg_candidates(RgParams P, int2 *cand, int *n_cand, uint8_t *has) {
    const int f = blockIdx.y, base = P.frame_base[f], ne = P.frame_base[f + 1] - base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ne) return;
    has[base + i] = 0;
    if ((P.flag[base + i] & 3) == 3) cand[atomicAdd(n_cand, 1)] = make_int2(f, i); // (a refined region may end below min_reg_size, down to 2 pixels)
}

// ---- the sequential stage: one wave per frame (lsd_rg_seq.h) ------------------------------------------------------------------------------
struct SeqParams {
    int F, w, h;
    const int *caddr; const int *frame_base; const float *cdeg; const float2 *ccs; const double *mod, *ang;
    rgs::Px *pix; int *glist; double *rect; int cand_cap; int *cand_cnt; int *status;
    int min_reg_size;
    unsigned long long *prof;
};
__global__ void __launch_bounds__(256) lsd_rg_fill(rgs::Px *pix, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) reinterpret_cast<float4 *>(pix)[i] = make_float4(rgs::NOTDEF_F, 0.f, 0.f, rgs::NOTDEF_F);
}
// the defined pixels' records from lsd_emit's compact lists
__global__ void __launch_bounds__(256) lsd_rg_scatter(SeqParams P) {
    const int f = blockIdx.y, base = P.frame_base[f], ne = P.frame_base[f + 1] - base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ne) return;
    const int q = P.caddr[base + i] & 0x7fffffff;
    const float d = P.cdeg[base + i];
    const float2 cs = P.ccs[base + i];
    reinterpret_cast<float4 *>(P.pix + (size_t)f * P.w * P.h)[q] = make_float4(d, cs.x, cs.y, d);
}
// A workgroup is a bundle of independent waves, one frame each (no LDS, no barrier).  Sixteen waves fill a CU (4 a SIMD, 128 VGPRs each): the frames
// of a batch then sit on F / 16 CUs and leave the others EMPTY -- cuboid_sweep_score's workgroups need a whole CU (160 KB of LDS, 2 x 240 VGPRs a
// SIMD) and would otherwise wait for a frame's 100 ms to pass.
__global__ void __launch_bounds__(1024) lsd_rg_seq(SeqParams P) {
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (f >= P.F) return;
    const int base = P.frame_base[f];
    rgs::Frame Fr;
    Fr.w = P.w; Fr.h = P.h; Fr.ne = P.frame_base[f + 1] - base;
    Fr.caddr = P.caddr + base; Fr.pix = P.pix + (size_t)f * P.w * P.h; Fr.mod = P.mod + (size_t)f * P.w * P.h;
    Fr.rect = P.rect + (size_t)f * P.cand_cap * 12; Fr.cand_cap = P.cand_cap; Fr.cand_cnt = P.cand_cnt + f;
    Fr.status = P.status + 4 * f; Fr.min_reg_size = P.min_reg_size; Fr.prof = P.prof ? P.prof + 16 * (size_t)f : nullptr;
    rgs::List L;
    L.glob = P.glist + (size_t)f * rgs::CAP; L.ring[0] = 0;
    rgs::run_frame<rgs::Wave>(Fr, L);
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
__global__ void __launch_bounds__(256) lsd_rg_improve(SeqParams P, const int *cand_base, int n_cand, const double *lgt, float4 *line, uint8_t *has) {
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= n_cand) return;
    int lo = 0, hi = P.F; // the frame: cand_base[f] <= wv < cand_base[f + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cand_base[mid] <= wv) lo = mid; else hi = mid; }
    const int f = lo;
    const double *o = P.rect + ((size_t)f * P.cand_cap + (wv - cand_base[f])) * 12;
    rg::Rect rec;
    rec.x1 = o[0]; rec.y1 = o[1]; rec.x2 = o[2]; rec.y2 = o[3]; rec.width = o[4]; rec.x = o[5]; rec.y = o[6]; rec.theta = o[7]; rec.dx = o[8]; rec.dy = o[9]; rec.prec = o[10]; rec.p = o[11];
    rg::Frame Fr = {};
    Fr.w = P.w; Fr.h = P.h; Fr.ang = P.ang + (size_t)f * P.w * P.h;
    const double LOG_NT = 5 * (log10(double(P.w)) + log10(double(P.h))) / 2 + log10(11.0);
    const double log_nfa = rg_rect_improve(Fr, rec, LOG_NT, lane, lgt);
    if (lane == 0) {
        const bool ok = log_nfa > LOG_EPS;
        if (ok) {
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= LSD_SCALE; rec.y1 /= LSD_SCALE; rec.x2 /= LSD_SCALE; rec.y2 /= LSD_SCALE;
            line[wv] = make_float4(float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2));
        }
        has[wv] = ok ? 1 : 0;
    }
}
} // namespace

struct LsdRegions {
    int F = 0, w = 0, h = 0, lanes = 0; size_t cap_def = 0;
    int pool_per_frame = 0;
    rg::u64 *d_own = nullptr;
    int *d_fp_off = nullptr, *d_fp_cnt = nullptr, *d_fp_cap = nullptr, *d_fp_nt = nullptr, *d_pool = nullptr, *d_pool_head = nullptr, *d_dirty = nullptr, *d_scratch = nullptr, *d_status = nullptr,
        *d_frame_base = nullptr, *d_ncand = nullptr;
    unsigned *d_execs = nullptr;
    int2 *d_cand = nullptr;
    uint8_t *d_flag = nullptr, *d_has = nullptr;
    double *d_reg_angle = nullptr;
    float4 *d_line = nullptr;
};

void lsd_regions_destroy(LsdRegions *r) {
    if (!r) return;
    void *ptrs[] = {r->d_own, r->d_fp_off, r->d_fp_cnt, r->d_fp_cap, r->d_fp_nt, r->d_pool, r->d_pool_head, r->d_dirty, r->d_scratch, r->d_status, r->d_frame_base, r->d_ncand, r->d_execs, r->d_cand,
                    r->d_flag, r->d_has, r->d_reg_angle, r->d_line};
    for (void *p : ptrs) if (p) hipFree(p);
    delete r;
}

// Runs the device stage for F frames.  lines[f] = x1 y1 x2 y2 floats in the reference's emission order.  Returns CS_OK, or CS_ERR_CAPACITY when
// a frame overflowed a device buffer or did not converge (the caller then uses the host stage for the batch).
int lsd_regions_run(cs_ctx *ctx, LsdRegions **handle, int F, int w, int h, const double *d_ang, const double *d_mod, const int *d_caddr, const int *frame_base,
                    std::vector<std::vector<float>> &lines, long *stats /* [0] max rounds, [1] executions, [2] candidates, [3] lane steps */) {
    LsdRegions *r = *handle;
    const size_t total = (size_t)frame_base[F];
    int max_ne = 0;
    for (int f = 0; f < F; f++) max_ne = std::max(max_ne, frame_base[f + 1] - frame_base[f]);
    const int pool_need = 24 * max_ne + 262144;
    int lanes = 1024;
    if (const char *e = getenv("CUBESLAM_LSD_LANES")) lanes = std::max(64, std::min(1024, atoi(e) / 64 * 64));
    if (!r || r->F < F || r->w != w || r->h != h || r->cap_def < total || r->pool_per_frame < pool_need || r->lanes != lanes) {
        lsd_regions_destroy(r);
        r = new LsdRegions();
        *handle = r;
        r->F = F; r->w = w; r->h = h; r->lanes = lanes; r->cap_def = total + total / 4 + 4096; r->pool_per_frame = pool_need + pool_need / 4;
        int rc;
#define RA_(call) do { rc = (call); if (rc != CS_OK) return rc; } while (0)
        RA_(cs_dalloc(ctx, &r->d_own, (size_t)F * w * h));
        RA_(cs_dalloc(ctx, &r->d_fp_off, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_fp_cnt, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_fp_cap, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_fp_nt, r->cap_def));
        RA_(cs_dalloc(ctx, &r->d_execs, r->cap_def));
        RA_(cs_dalloc(ctx, &r->d_flag, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_has, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_reg_angle, r->cap_def)); RA_(cs_dalloc(ctx, &r->d_line, r->cap_def));
        RA_(cs_dalloc(ctx, &r->d_pool, (size_t)F * r->pool_per_frame)); RA_(cs_dalloc(ctx, &r->d_pool_head, (size_t)F));
        RA_(cs_dalloc(ctx, &r->d_dirty, 3 * r->cap_def)); RA_(cs_dalloc(ctx, &r->d_scratch, (size_t)F * lanes * RG_LISTS * rg::CAP));
        RA_(cs_dalloc(ctx, &r->d_status, (size_t)F * 4)); RA_(cs_dalloc(ctx, &r->d_frame_base, (size_t)F + 1)); RA_(cs_dalloc(ctx, &r->d_ncand, 1));
        RA_(cs_dalloc(ctx, &r->d_cand, r->cap_def));
#undef RA_
    }
    int rc = cs_h2d(ctx, r->d_frame_base, frame_base, (size_t)F + 1); if (rc) return rc;
    RgParams P;
    P.F = F; P.w = w; P.h = h; P.lanes = lanes; P.caddr = d_caddr; P.frame_base = r->d_frame_base; P.ang = d_ang; P.mod = d_mod;
    P.own = r->d_own; P.fp_off = r->d_fp_off; P.fp_cnt = r->d_fp_cnt; P.fp_cap = r->d_fp_cap; P.fp_nt = r->d_fp_nt; P.execs = r->d_execs; P.flag = r->d_flag; P.reg_angle = r->d_reg_angle;
    P.pool = r->d_pool; P.pool_per_frame = r->pool_per_frame; P.pool_head = r->d_pool_head; P.dirty = r->d_dirty; P.scratch = r->d_scratch; P.status = r->d_status;
    const double LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    P.min_reg_size = int(-LOG_NT / std::log10(rg::ANG_TH / 180));
    P.max_rounds = 1000;
    P.prof = nullptr;
#if defined(RG_PROFILE)
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) hipMalloc((void **)&d_prof, 16 * sizeof(unsigned long long));
    hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), ctx->stream);
    P.prof = d_prof;
#endif
    const int tw = (w + rg::TILE - 1) / rg::TILE, th = (h + rg::TILE - 1) / rg::TILE;
    const size_t lds = sizeof(int) * ((size_t)tw * th + 4);
    if (lds > 150 * 1024) return CS_ERR_CAPACITY;
    static bool attr_set = false;
    if (!attr_set) { CS_HIP(ctx, hipFuncSetAttribute((const void *)lsd_rg_fixpoint, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_set = true; }
    CS_LAUNCH(ctx, "lsd_rg_fixpoint", lsd_rg_fixpoint, dim3(F), dim3(lanes), lds, P);
    CS_HIP(ctx, hipMemsetAsync(r->d_ncand, 0, sizeof(int), ctx->stream));
    CS_LAUNCH(ctx, "lsd_rg_candidates", lsd_rg_candidates, dim3((max_ne + 255) / 256, F), dim3(256), 0, P, r->d_cand, r->d_ncand, r->d_has);
    int n_cand = 0;
    std::vector<int> status((size_t)F * 4);
    rc = cs_d2h(ctx, &n_cand, r->d_ncand, 1); if (rc) return rc;
    rc = cs_d2h(ctx, status.data(), r->d_status, status.size()); if (rc) return rc;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    long max_rounds = 0, execs = 0, steps = 0;
    bool bad = false;
    for (int f = 0; f < F; f++) { max_rounds = std::max<long>(max_rounds, status[4 * f]); execs += status[4 * f + 2]; steps += status[4 * f + 3]; bad = bad || status[4 * f + 1] != 0; }
#if defined(RG_PROFILE)
    { unsigned long long hp[16]; hipMemcpy(hp, P.prof, sizeof hp, hipMemcpyDeviceToHost); const char *nm[7] = {"idle", "old", "seed", "grow", "post", "final", "done"};
      for (int k = 0; k < 7; k++) fprintf(stderr, "[rg prof] %-6s wave-entries %10llu  ms(sum over waves) %10.2f  us/entry %7.2f\n", nm[k], hp[2 * k + 1], hp[2 * k] / 1e5, hp[2 * k + 1] ? hp[2 * k] / 100.0 / hp[2 * k + 1] : 0.0);
      fprintf(stderr, "[rg prof] kernel ms summed over frames %.2f\n", hp[14] / 1e5); }
#endif
    if (stats) { stats[0] = max_rounds; stats[1] = execs; stats[2] = n_cand; stats[3] = steps; }
    if (bad) return CS_ERR_CAPACITY;
    if (n_cand > 0) CS_LAUNCH(ctx, "lsd_rg_lines", lsd_rg_lines, dim3((n_cand + 3) / 4), dim3(256), 0, P, r->d_cand, n_cand, r->d_line, r->d_has, 0);
    std::vector<uint8_t> has(total);
    std::vector<float4> line(total);
    rc = cs_d2h(ctx, has.data(), r->d_has, total); if (rc) return rc;
    rc = cs_d2h(ctx, line.data(), r->d_line, total); if (rc) return rc;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lines.assign((size_t)F, {});
    for (int f = 0; f < F; f++)
        for (int i = frame_base[f]; i < frame_base[f + 1]; i++)
            if (has[i]) { lines[f].push_back(line[i].x); lines[f].push_back(line[i].y); lines[f].push_back(line[i].z); lines[f].push_back(line[i].w); }
    return CS_OK;
}

// ---- host side of the sequential stage ----------------------------------------------------------------------------------------------------
struct LsdSeq {
    int F = 0, w = 0, h = 0; int cand_cap = 0; size_t cap_lines = 0;
    rgs::Px *d_pix = nullptr;
    int *d_glist = nullptr, *d_cand_cnt = nullptr, *d_cand_base = nullptr, *d_status = nullptr, *d_frame_base = nullptr;
    double *d_rect = nullptr, *d_lgt = nullptr;
    uint8_t *d_has = nullptr;
    float4 *d_line = nullptr;
    std::vector<int> h_base, h_status; std::vector<uint8_t> h_has; std::vector<float4> h_line;
};
void lsd_seq_destroy(LsdSeq *r) {
    if (!r) return;
    void *ptrs[] = {r->d_pix, r->d_glist, r->d_cand_cnt, r->d_cand_base, r->d_status, r->d_frame_base, r->d_rect, r->d_lgt, r->d_has, r->d_line};
    for (void *p : ptrs) if (p) hipFree(p);
    delete r;
}
// The region stage of F frames, one wave per frame.  lines[f] = x1 y1 x2 y2 floats in the reference's emission order.  CS_ERR_CAPACITY: a region
// outgrew the wave's list (rgs::CAP pixels) or a frame its rectangle list -- the caller then runs the host stage for the batch.
int lsd_seq_run(cs_ctx *ctx, LsdSeq **handle, int F, int w, int h, const double *d_ang, const double *d_mod, const int *d_caddr, const float *d_cdeg, const float2 *d_ccs, const int *frame_base,
                std::vector<std::vector<float>> &lines, long *stats /* [0] region_grow calls, [1] window fetches, [2] rectangles at rect_improve, [3] regions at the rectangle stage */) {
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
        RA_(cs_dalloc(ctx, &r->d_pix, (size_t)F * w * h)); RA_(cs_dalloc(ctx, &r->d_glist, (size_t)F * rgs::CAP)); RA_(cs_dalloc(ctx, &r->d_rect, (size_t)F * r->cand_cap * 12));
        RA_(cs_dalloc(ctx, &r->d_cand_cnt, (size_t)F)); RA_(cs_dalloc(ctx, &r->d_cand_base, (size_t)F + 1));
        RA_(cs_dalloc(ctx, &r->d_status, (size_t)F * 4)); RA_(cs_dalloc(ctx, &r->d_frame_base, (size_t)F + 1));
        RA_(cs_dalloc(ctx, &r->d_lgt, (size_t)LG_N));
        CS_LAUNCH(ctx, "lsd_rg_lgamma_table", lsd_rg_lgamma_table, dim3(LG_N / 256), dim3(256), 0, r->d_lgt);
    }
    RA_(cs_h2d(ctx, r->d_frame_base, frame_base, (size_t)F + 1));
    SeqParams S;
    S.F = F; S.w = w; S.h = h; S.caddr = d_caddr; S.frame_base = r->d_frame_base; S.cdeg = d_cdeg; S.ccs = d_ccs; S.mod = d_mod; S.ang = d_ang;
    S.pix = r->d_pix; S.glist = r->d_glist; S.rect = r->d_rect; S.cand_cap = r->cand_cap; S.cand_cnt = r->d_cand_cnt; S.status = r->d_status;
    const double LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    S.min_reg_size = int(-LOG_NT / std::log10(rg::ANG_TH / 180));
    S.prof = nullptr;
#if defined(RGS_PROFILE)
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) hipMalloc((void **)&d_prof, 4096 * 16 * sizeof(unsigned long long));
    hipMemsetAsync(d_prof, 0, 4096 * 16 * sizeof(unsigned long long), ctx->stream);
    S.prof = d_prof;
#endif
    const size_t npx = (size_t)F * w * h;
    CS_LAUNCH(ctx, "lsd_rg_fill", lsd_rg_fill, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, r->d_pix, npx);
    CS_LAUNCH(ctx, "lsd_rg_scatter", lsd_rg_scatter, dim3((max_ne + 255) / 256, F), dim3(256), 0, S);
    int wpb = 16; // waves (= frames) per workgroup
    if (const char *e = getenv("CUBESLAM_LSD_SEQ_WPB")) wpb = std::max(1, std::min(16, atoi(e)));
    CS_LAUNCH(ctx, "lsd_rg_seq", lsd_rg_seq, dim3((F + wpb - 1) / wpb), dim3(64 * wpb), 0, S);
    CS_LAUNCH(ctx, "lsd_rg_cand_scan", lsd_rg_cand_scan, dim3(1), dim3(1024), 0, r->d_cand_cnt, F, r->d_cand_base);
    r->h_base.resize((size_t)F + 1); r->h_status.resize((size_t)F * 4);
    RA_(cs_d2h(ctx, r->h_base.data(), r->d_cand_base, (size_t)F + 1));
    RA_(cs_d2h(ctx, r->h_status.data(), r->d_status, (size_t)F * 4));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    CS_LAUNCH(ctx, "lsd_rg_improve", lsd_rg_improve, dim3((n_cand + 3) / 4), dim3(256), 0, S, r->d_cand_base, n_cand, r->d_lgt, r->d_line, r->d_has);
    r->h_has.resize((size_t)n_cand); r->h_line.resize((size_t)n_cand);
    RA_(cs_d2h(ctx, r->h_has.data(), r->d_has, (size_t)n_cand));
    RA_(cs_d2h(ctx, r->h_line.data(), r->d_line, (size_t)n_cand));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
#undef RA_
    for (int f = 0; f < F; f++)
        for (int k = r->h_base[f]; k < r->h_base[f + 1]; k++)
            if (r->h_has[k]) { const float4 v = r->h_line[k]; lines[f].push_back(v.x); lines[f].push_back(v.y); lines[f].push_back(v.z); lines[f].push_back(v.w); }
    return CS_OK;
}
