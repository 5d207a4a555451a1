// lsd.hip -- LSD line segment detector of line_lbd on MI355X (gfx950) + host.
//
// Replaces line_lbd_detect::detect_raw_lines / detect_filter_lines (reference line_lbd/class/line_lbd_allclass.cpp:125-148,
// 200-221) -> LSDDetector::detectImpl (libs/LSDDetector.cpp:153-287) -> LineSegmentDetectorImpl::flsd (libs/lsd.cpp:440-536).
//   lsd_blur_hv               GaussianBlur(7x7, sigma 0.75) u8 -> double, REFLECT_101, symmetric summation order, both passes fused
//   lsd_resize                cv::resize(0.8, 0.8, INTER_LINEAR) on CV_64F with float coefficients (tables from the host)
//   lsd_gradient              ll_angle (:538-585): 2x2 gradient, norm, level-line angle via cv::fastAtan2, NOTDEF below rho
// Host, per frame (OpenMP across frames): the sequential part of the algorithm, seeds in address order (see LsdHost::run) --
// region_grow (each accepted pixel updates the region angle that the next test uses, :665-683), region2rect, refine,
// rect_improve, rect_nfa, nfa -- which has no order-preserving parallel form.  Double precision as in the reference.
#include "common.h"
#include "glibc_sincosf.h"
#include "lsd_regions.h"

#include <cfloat>
#include <chrono>
#include <cmath>
#include <mutex>
#include <omp.h>
#include <vector>

namespace {
constexpr double NOTDEF = -1024.0, PI_ = 3.1415926535897932384626433832795, DEG_TO_RADS = PI_ / 180;
constexpr float NOTDEF_DEG = -1024.0f; // the undefined mark of the float-degree angle map
constexpr int KH = 3; // half kernel: ceil(0.75 * sqrt(6 ln 10)) = 3

__host__ __device__ inline float fast_atan2f_(float y, float x) { // cv::fastAtan2
    const float p1 = 0.9997878412794807f * (float)(180 / PI_), p3 = -0.3258083974640975f * (float)(180 / PI_), p5 = 0.1555786518463281f * (float)(180 / PI_),
                p7 = -0.04432655554792128f * (float)(180 / PI_);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
__device__ __forceinline__ int reflect101d(int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; } return p; }

struct GK { double k[2 * KH + 1]; };

// GaussianBlur(7x7, sigma 0.75) of the u8 frame into doubles, both passes in one row-streaming kernel: one wave per 64-column
// strip and LSD_ROWS output rows, one column per lane.  Every input row is loaded once (8 rows in flight), filtered horizontally
// from an LDS line of bytes, and the last seven horizontal results stay in registers for the vertical pass -- the double
// intermediate image (8 B written + 8 B x 7 read per pixel in the two-kernel version) never exists.  Summation order of both
// passes as in OpenCV's symmetric filters: centre tap first, then k[t] * (left + right) for t = 1..3.
constexpr int LSD_ROWS = 64;
__global__ void __launch_bounds__(64) lsd_blur_hv(const uint8_t *gray, int W, int H, GK g, double *blur) {
    const int strips = (W + 63) / 64;
    const int sx = (blockIdx.x % strips) * 64, y0 = (blockIdx.x / strips) * LSD_ROWS, tid = threadIdx.x;
    if (y0 >= H) return;
    const int rows = min(LSD_ROWS, H - y0);
    __shared__ uint8_t line[64 + 8]; // column sx + c at byte 3 + c
    const uint8_t *img = gray + (long)blockIdx.z * W * H;
    double *out = blur + (long)blockIdx.z * W * H;
    const int xc = reflect101d(sx + tid, W);
    const int xh = tid < 3 ? reflect101d(sx - 3 + tid, W) : reflect101d(sx + 64 + (tid - 3), W); // halo columns, lanes 0..5
    const int x = sx + tid;
    double ring[7] = {0, 0, 0, 0, 0, 0, 0};
    constexpr int G = 8;
    const int total = rows + 6;
    uint8_t cur[G], nxt[G], curh[G], nxth[G];
    auto fetch = [&](int r0, uint8_t (&a)[G], uint8_t (&hh)[G]) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            a[u] = 0; hh[u] = 0;
            if (r0 + u < total) {
                const uint8_t *row = img + (long)reflect101d(y0 + r0 + u - 3, H) * W;
                a[u] = row[xc];
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
                line[3 + tid] = cur[u];
                if (tid < 3) line[tid] = curh[u];
                else if (tid < 6) line[3 + 64 + (tid - 3)] = curh[u];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double pix[7];
#pragma unroll
                for (int t = 0; t < 7; t++) pix[t] = (double)line[tid + t]; // columns x-3 .. x+3
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double h = g.k[KH] * pix[3];
#pragma unroll
                for (int t = 1; t <= KH; t++) h += g.k[KH + t] * (pix[3 - t] + pix[3 + t]);
#pragma unroll
                for (int t = 0; t < 6; t++) ring[t] = ring[t + 1];
                ring[6] = h;
                if (r >= 6 && x < W) {
                    double v = g.k[KH] * ring[3];
#pragma unroll
                    for (int t = 1; t <= KH; t++) v += g.k[KH + t] * (ring[3 - t] + ring[3 + t]);
                    out[(long)(y0 + r - 6) * W + x] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; u++) { cur[u] = nxt[u]; curh[u] = nxth[u]; }
    }
}
__global__ void __launch_bounds__(256) lsd_resize(const double *blur, int W, int H, int w, int h, const int *xofs, const float *ax, const int *yofs, const float *ay,
                                                  double *scaled) {
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= w) return;
    const double *src = blur + (long)blockIdx.z * H * W;
    const int sx0 = xofs[dx], sx1 = min(sx0 + 1, W - 1), sy0 = yofs[dy], sy1 = min(sy0 + 1, H - 1);
    const double r0 = src[(long)sy0 * W + sx0] * ax[dx * 2] + src[(long)sy0 * W + sx1] * ax[dx * 2 + 1];
    const double r1 = src[(long)sy1 * W + sx0] * ax[dx * 2] + src[(long)sy1 * W + sx1] * ax[dx * 2 + 1];
    scaled[((long)blockIdx.z * h + dy) * w + dx] = r0 * ay[dy * 2] + r1 * ay[dy * 2 + 1];
}
// also counts the defined pixels of its 256-pixel row segment: seg_cnt[(frame * h + y) * gridDim.x + blockIdx.x]
// The level-line angle is kept as what ll_angle computes it from: cv::fastAtan2's FLOAT DEGREES (NOTDEF_DEG where undefined).  The map value of the reference is that float times
// DEG_TO_RADS in double (:566), which every reader forms itself -- 4 bytes per pixel instead of 8, and lsd_emit no longer has to divide its way back to the float.
__global__ void __launch_bounds__(256) lsd_gradient(const double *scaled, int w, int h, double threshold, double *modgrad, float *angles, int *seg_cnt) {
    __shared__ int wc[4];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    bool def = false;
    if (x < w) {
        const double *img = scaled + (long)blockIdx.z * w * h;
        const long o = ((long)blockIdx.z * h + y) * w + x;
        if (x >= w - 1 || y >= h - 1) { modgrad[o] = 0; angles[o] = NOTDEF_DEG; } // down / right boundaries undefined (:553-554)
        else {
            const int addr = y * w + x;
            const double DA = img[addr + w + 1] - img[addr], BC = img[addr + 1] - img[addr + w];
            const double gx = DA + BC, gy = DA - BC;
            const double norm = sqrt((gx * gx + gy * gy) / 4);
            modgrad[o] = norm;
            def = !(norm <= threshold);
            angles[o] = def ? fast_atan2f_(float(gx), float(-gy)) : NOTDEF_DEG;
        }
    }
    const unsigned long long m = __ballot(def);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[((long)blockIdx.z * h + y) * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
// exclusive scan of n ints (out has n + 1 entries, out[n] = total) in three coalesced passes: 1024-element blocks, their totals, add
__global__ void __launch_bounds__(1024) lsd_scan_blocks(const int *in, int n, int *out, int *block_tot) {
    __shared__ int ws[16];
    const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = i < n ? in[i] : 0;
    int sc = v;
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(sc, d); if (lane >= d) sc += t; }
    if (lane == 63) ws[wave] = sc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; k++) base += ws[k];
    if (i < n) out[i] = base + sc - v;
    if (threadIdx.x == 1023) block_tot[blockIdx.x] = base + sc;
}
__global__ void __launch_bounds__(1024) lsd_scan_top(int *block_tot, int nblk, int *total_out) { // in place: exclusive scan of the block totals (nblk <= 1024 per pass chunk)
    __shared__ int ws[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int i = b0 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int v = i < nblk ? block_tot[i] : 0;
        int sc = v;
        for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(sc, d); if (lane >= d) sc += t; }
        if (lane == 63) ws[wave] = sc;
        __syncthreads();
        int base = carry;
        for (int k = 0; k < wave; k++) base += ws[k];
        if (i < nblk) block_tot[i] = base + sc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + sc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(1024) lsd_scan_add(int *out, int n, const int *block_base) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_base[blockIdx.x];
}
// ordered compaction of the defined pixels (address order inside a frame): address, level-line angle, gradient norm
// c_deg: the angle as cv::fastAtan2 returned it (float degrees; the level-line angle is exactly double(c_deg) * DEG_TO_RADS, lsd.cpp:566), c_cs: cos / sin
// of float(angle) as region_grow adds them up (:676-677, glibc's cosf / sinf restated) -- computed here once per pixel instead of by the host per visit
// PIX (the device region stage one wave per frame, lsd_rg_seq): the kernel also leaves that stage's 16-byte record of EVERY pixel of its segment -- (angle while free, cos, sin, angle)
// for a defined pixel, (NOTDEF, 0, 0, NOTDEF) for the others -- and the seeds' cos / sin (of the angle as a double, region_grow's start values :651-652) in place of the pixels'
// own: what lsd_rg_fill + lsd_rg_scatter did in two more passes over the frame, from lists this kernel had just written (c_deg is not written then).
template <bool PIX> __global__ void __launch_bounds__(256) lsd_emit(const double *modgrad, const float *angles, int w, int h, const int *seg_base, int *c_addr, float *c_deg, float2 *c_cs, double *c_mod, float *fre /* PIX: the region walk's own copy of the angle map, a float per pixel */) {
    __shared__ int wc[4];
    __shared__ short s_x[256];
    __shared__ float s_a[256];
    const int y = blockIdx.y;
    const long row = ((long)blockIdx.z * h + y) * w;
    {   // phase 1, a lane per pixel: which pixels of the segment are defined; they are packed (in address order) into the workgroup's list, so that the
        // expensive part below -- an IEEE division, eight double alignment tests, the double-precision polynomial of glibc's cosf / sinf -- runs on full
        // waves of defined pixels only (37 % of the pixels of a textured frame)
        const int x = blockIdx.x * 256 + threadIdx.x;
        float a = NOTDEF_DEG;
        if (x < w) a = angles[row + x];
        const bool def = a != NOTDEF_DEG;
        const unsigned long long m = __ballot(def);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) wc[wv] = __popcll(m);
        if (PIX && x < w) fre[row + x] = a;
        __syncthreads();
        if (def) {
            int r = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1));
            for (int i = 0; i < wv; i++) r += wc[i];
            s_x[r] = (short)x; s_a[r] = a;
        }
        __syncthreads();
    }
    const int n_def = wc[0] + wc[1] + wc[2] + wc[3];
    if ((int)threadIdx.x >= n_def) return;
    const int x = s_x[threadIdx.x];
    const float d = s_a[threadIdx.x];
    const double a = double(d) * DEG_TO_RADS; // the map value (:566)
    const long o = row + x;
    const int pos = seg_base[((long)blockIdx.z * h + y) * gridDim.x + blockIdx.x] + threadIdx.x;
    // bit 31 of the address: no neighbour is aligned with this pixel's own angle (angles never change), so as a seed it stays alone -- region_grow's
    // first nine tests use exactly that angle -- and the host marks it used without testing anything (12 k of 25 k seeds on a textured frame are single)
    bool alone = true;
    const double prec = PI_ * 22.5 / 180;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            if ((dx == 0 && dy == 0) || xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            const float bd = angles[((long)blockIdx.z * h + yy) * w + xx];
            if (bd == NOTDEF_DEG) continue;
            const double b = double(bd) * DEG_TO_RADS;
            double n_theta = a - b; // isAligned(neighbour, a, prec) :1138-1154
            if (n_theta < 0) n_theta = -n_theta;
            if (n_theta > (3 * PI_) / 2) { n_theta -= 2 * PI_; if (n_theta < 0) n_theta = -n_theta; }
            if (n_theta <= prec) alone = false;
        }
    c_addr[pos] = (y * w + x) | (alone ? (int)0x80000000 : 0);
    if (PIX) { // (the device walk computes cos / sin of float(angle) at its window fetches; a seed starts its sums with cos / sin of the angle as a double :651-652)
        c_cs[pos] = make_float2(float(cos(a)), float(sin(a)));
    } else {
        const float pc = glibc_sincosf::cosf_(float(a)), ps = glibc_sincosf::sinf_(float(a));
        c_deg[pos] = d;
        if (c_mod) c_mod[pos] = modgrad[o]; // (the host stage's copy of the norms; the device stage reads the dense map)
        c_cs[pos] = make_float2(pc, ps);
    }
}

// ------------------------------------------------------------------------------------------------ host: sequential LSD stages
struct RectH { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

class LsdHost {
  public:
    int w = 0, h = 0;
    // Dense per-pixel map of this thread, one 16-byte record per pixel so that what an accepted pixel contributes sits in the cache line its test
    // just loaded (the stage is bound by the misses of the nine-neighbour walk; most regions are a handful of pixels).  free_deg: the level-line angle in float degrees while the pixel is defined AND unused, NOTDEF_F
    // otherwise -- region_grow's "used == 0 && isAligned" reads one 4-byte value per neighbour; deg: the same without the used marks (rect_nfa
    // counts aligned pixels whatever their use); aux: what an accepted pixel contributes (gradient norm, cos / sin of its angle).
    static constexpr float NOTDEF_F = -1024.0f;
    struct Px { float free_deg, c, s, deg; }; // c, s: cos / sin of float(angle); deg: the angle whatever the use
    std::vector<Px> pix;
    std::vector<double> dmod;         // gradient norm, dense: only read for the pixels of regions that reach the rectangle stage
    std::vector<int> rx, ry;          // region points (structure of arrays)
    std::vector<double> rang;
    double LOG_NT = 0;

    static inline bool aligned_deg(float af, double theta, double prec) { // isAligned lsd.cpp:1138-1154 on a stored angle
        if (af == NOTDEF_F) return false;
        const double a = double(af) * DEG_TO_RADS; // the map value, exactly (:566)
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > (3 * PI_) / 2) { n_theta -= 2 * PI_; if (n_theta < 0) n_theta = -n_theta; }
        return n_theta <= prec;
    }
    static double sdiff(double a, double b) { double d = a - b; while (d <= -PI_) d += 2 * PI_; while (d > PI_) d -= 2 * PI_; return d; }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }

    void grow(int sx, int sy, int &n, double &reg_angle, double prec) { // region_grow :637-688
        n = 1;
        int addr = sx + sy * w;
        reg_angle = double(pix[addr].deg) * DEG_TO_RADS;
        rx[0] = sx; ry[0] = sy; rang[0] = reg_angle;
        float sumdx = 0, sumdy = 0; // cos / sin of the seed angle (:651-652, doubles): only needed once a second pixel joins -- half the seeds stay alone
        bool have_sums = false;
        pix[addr].free_deg = NOTDEF_F;
        for (int i = 0; i < n; ++i) {
            const int px = rx[i], py = ry[i];
            const int x0 = std::max(px - 1, 0), x1 = std::min(px + 1, w - 1), y0 = std::max(py - 1, 0), y1 = std::min(py + 1, h - 1);
            for (int yy = y0; yy <= y1; ++yy) {
                int c = x0 + yy * w;
                for (int xx = x0; xx <= x1; ++xx, ++c) {
                    Px &ax = pix[c];
                    const float af = ax.free_deg;
                    if (aligned_deg(af, reg_angle, prec)) { // defined, unused and aligned
                        ax.free_deg = NOTDEF_F;
                        rx[n] = xx; ry[n] = yy; rang[n] = double(af) * DEG_TO_RADS;
                        ++n;
                        if (!have_sums) { const double a0 = rang[0]; sumdx = float(std::cos(a0)); sumdy = float(std::sin(a0)); have_sums = true; }
                        // its own neighbourhood is read when the list reaches it: ask for the two rows not in cache yet (the walk is bound by these misses)
                        __builtin_prefetch(&pix[c - w - 1 < 0 ? 0 : c - w - 1]); __builtin_prefetch(&pix[c + w + 1 >= w * h ? c : c + w + 1]);
                        sumdx += ax.c; // cos(float(angle)), sin(float(angle)) :676-677, computed by lsd_emit
                        sumdy += ax.s;
                        reg_angle = fast_atan2f_(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
            }
        }
    }
    void to_rect(int n, double reg_angle, double prec, double p, RectH &rec) const { // region2rect :690-746 + get_theta :748-784
        double x = 0, y = 0, sum = 0;
        for (int i = 0; i < n; ++i) { const double wg = dmod[(size_t)ry[i] * w + rx[i]]; x += double(rx[i]) * wg; y += double(ry[i]) * wg; sum += wg; }
        x /= sum; y /= sum;
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (int i = 0; i < n; ++i) { const double dx = (double)rx[i] - x, dy = (double)ry[i] - y, wg = dmod[(size_t)ry[i] * w + rx[i]]; Ixx += dy * dy * wg; Iyy += dx * dx * wg; Ixy -= dx * dy * wg; }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2f_(float(lambda - Ixx), float(Ixy))) : double(fast_atan2f_(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (std::fabs(sdiff(theta, reg_angle)) > prec) theta += PI_;
        const double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (int i = 0; i < n; ++i) {
            const double ddx = double(rx[i]) - x, ddy = double(ry[i]) - y, l = ddx * dx + ddy * dy, ww = -ddx * dy + ddy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (ww > w_max) w_max = ww; else if (ww < w_min) w_min = ww;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }
    bool shrink(int &n, double reg_angle, double prec, double p, RectH &rec, double density, double density_th) { // reduce_region_radius :834-871
        const double xc = double(rx[0]), yc = double(ry[0]);
        auto dsq = [](double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); };
        const double r1 = dsq(xc, yc, rec.x1, rec.y1), r2 = dsq(xc, yc, rec.x2, rec.y2);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (int i = 0; i < n; ++i)
                if (dsq(xc, yc, double(rx[i]), double(ry[i])) > radSq) {
                    { Px &q = pix[rx[i] + ry[i] * w]; q.free_deg = q.deg; }
                    std::swap(rx[i], rx[n - 1]); std::swap(ry[i], ry[n - 1]); std::swap(rang[i], rang[n - 1]);
                    --n; --i;
                }
            if (n < 2) return false;
            to_rect(n, reg_angle, prec, p, rec);
            density = double(n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }
    bool refine(int &n, double reg_angle, double prec, double p, RectH &rec, double density_th) { // :786-832
        double density = double(n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        const double xc = double(rx[0]), yc = double(ry[0]), ang_c = rang[0];
        double sum = 0, s_sum = 0;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            { Px &q = pix[rx[i] + ry[i] * w]; q.free_deg = q.deg; }
            if (dist(xc, yc, rx[i], ry[i]) < rec.width) { const double d = sdiff(rang[i], ang_c); sum += d; s_sum += d * d; ++cnt; }
        }
        const double mean_angle = sum / double(cnt);
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(cnt) + mean_angle * mean_angle);
        grow(rx[0], ry[0], n, reg_angle, tau);
        if (n < 2) return false;
        to_rect(n, reg_angle, prec, p, rec);
        density = double(n) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return shrink(n, reg_angle, prec, p, rec, density, density_th);
        return true;
    }
    static bool deq(double a, double b) {
        if (a == b) return true;
        double ad = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b), am = (aa > bb) ? aa : bb;
        if (am < DBL_MIN) am = DBL_MIN;
        return (ad / am) <= (100.0 * DBL_EPSILON);
    }
    static double lgam(double x) { // log_gamma :70,124-160
        if (x > 15.0) return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), b = 0;
        for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
        return a + std::log(b);
    }
    // log_gamma is only ever asked for integer arguments (pixel counts): the values are computed once per thread by the same
    // function and kept (each costs 14 pow + 16 log calls, and rect_nfa runs ~115 times per frame)
    mutable std::vector<double> lg_memo;
    double lgam_int(int x) const {
        if (x < 0 || x >= (1 << 20)) return lgam(double(x));
        if ((size_t)x >= lg_memo.size()) lg_memo.resize(std::max<size_t>((size_t)x + 1, lg_memo.size() * 2 + 1024), -1.0);
        double &v = lg_memo[x];
        if (v < 0) v = lgam(double(x)); // lgam(x) >= 0 for the integers >= 1 (lgam(1) = lgam(2) = ~1e-10 in this approximation is re-evaluated, harmless)
        return v;
    }
    double nfa(int n, int k, double p) const { // :1100-1136
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * std::log10(p);
        const double p_term = p / (1 - p);
        const double l1 = (double(n) + 1) - lgam_int(k + 1) - lgam_int(n - k + 1) + double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
        double term = std::exp(l1);
        if (deq(term, 0)) { if (k > n * p) return -l1 / M_LN10 - LOG_NT; else return -LOG_NT; }
        double tail = term;
        for (int i = k + 1; i <= n; ++i) {
            const double bt = double(n - i + 1) / double(i), mt = bt * p_term;
            term *= mt;
            tail += term;
            if (bt < 1) { const double err = term * ((1 - std::pow(mt, double(n - i + 1))) / (1 - mt) - 1); if (err < 0.1 * std::fabs(-std::log10(tail) - LOG_NT) * tail) break; }
        }
        return -std::log10(tail) - LOG_NT;
    }
    // Pixel walk of rect_nfa (:977-1098, integer-division and tailp->p.x quirks kept :1057-1065): number of pixels of the rectangle inside
    // the image and, for each of the np tolerances, how many of them are aligned with rec.theta.  The row spans are contiguous, the
    // alignment test is written without branches (same IEEE operations as isAligned), so the compiler vectorises the inner loop;
    // several tolerances share one walk because rect_improve's first loop only halves the tolerance of an unchanged rectangle.
    template <int NP> int rect_count(const RectH &rec, const double *precs, int *algs) const {
        int total = 0, alg[NP];
        for (int k = 0; k < NP; k++) alg[k] = 0;
        const double hw = rec.width / 2.0, dyhw = rec.dy * hw, dxhw = rec.dx * hw;
        int vx[4] = {int(rec.x1 - dyhw), int(rec.x2 - dyhw), int(rec.x2 + dyhw), int(rec.x1 + dyhw)};
        int vy[4] = {int(rec.y1 + dxhw), int(rec.y2 + dxhw), int(rec.y2 - dxhw), int(rec.y1 - dxhw)};
        for (int i = 1; i < 4; i++) // insertion sort by (x, y): the four keys are ordered exactly like std::sort with AsmallerB_XoverY would
            for (int j = i; j > 0 && (vx[j] < vx[j - 1] || (vx[j] == vx[j - 1] && vy[j] < vy[j - 1])); j--) { std::swap(vx[j], vx[j - 1]); std::swap(vy[j], vy[j - 1]); }
        bool taken[4] = {false, false, false, false};
        int mn = 0, mx = 0;
        for (int i = 1; i < 4; ++i) { if (vy[mn] > vy[i]) mn = i; if (vy[mx] < vy[i]) mx = i; }
        taken[mn] = true;
        int lm = -1, rm = -1, tp = -1;
        for (int i = 0; i < 4; ++i) if (!taken[i]) { if (lm < 0) lm = i; else if (vx[lm] > vx[i]) lm = i; }
        taken[lm] = true;
        for (int i = 0; i < 4; ++i) if (!taken[i]) { if (rm < 0) rm = i; else if (vx[rm] < vx[i]) rm = i; }
        taken[rm] = true;
        for (int i = 0; i < 4; ++i) if (!taken[i]) { if (tp < 0) tp = i; else if (vx[tp] > vx[i]) tp = i; }
        const double flstep = (vy[mn] != vy[lm]) ? (vx[mn] - vx[lm]) / (vy[mn] - vy[lm]) : 0;
        const double slstep = (vy[lm] != vx[tp]) ? (vx[lm] - vx[tp]) / (vy[lm] - vx[tp]) : 0;
        const double frstep = (vy[mn] != vy[rm]) ? (vx[mn] - vx[rm]) / (vy[mn] - vy[rm]) : 0;
        const double srstep = (vy[rm] != vx[tp]) ? (vx[rm] - vx[tp]) / (vy[rm] - vx[tp]) : 0;
        double lstep = flstep, rstep = frstep, left_x = vx[mn], right_x = vx[mn];
        const double theta = rec.theta, two_pi = 2 * PI_, wrap = (3 * PI_) / 2;
        for (int y = vy[mn]; y <= vy[mx]; ++y) {
            if (y < 0 || y >= h) continue; // (the reference skips the edge stepping for these rows too)
            const int xa = std::max(int(left_x), 0), xb = std::min(int(right_x), w - 1);
            if (xb >= xa) {
                total += xb - xa + 1;
                const Px *row = pix.data() + (size_t)y * w;
                for (int x = xa; x <= xb; ++x) { // isAligned :1138-1154
                    const float af = row[x].deg;
                    const double a = double(af) * DEG_TO_RADS;
                    double d = std::fabs(theta - a);
                    const double d2 = std::fabs(d - two_pi);
                    d = d > wrap ? d2 : d;
                    const int def = af != NOTDEF_F;
                    for (int k = 0; k < NP; k++) alg[k] += def & (d <= precs[k]);
                }
            }
            if (y >= vy[lm]) lstep = slstep;
            if (y >= vy[rm]) rstep = srstep;
            left_x += lstep; right_x += rstep;
        }
        for (int k = 0; k < NP; k++) algs[k] = alg[k];
        return total;
    }
    double rect_nfa(const RectH &rec) const {
        int alg;
        const int total = rect_count<1>(rec, &rec.prec, &alg);
        return nfa(total, alg, rec.p);
    }
    double improve(RectH &rec) const { // rect_improve :873-975, log_eps = 0
        const double delta = 0.5, d2 = delta / 2.0;
        double best = rect_nfa(rec);
        if (best > 0) return best;
        RectH r = rec;
        { // five halvings of the tolerance: one walk, five counts
            RectH rs[5]; double precs[5]; int algs[5];
            for (int n = 0; n < 5; ++n) { r.p /= 2; r.prec = r.p * PI_; rs[n] = r; precs[n] = r.prec; }
            const int total = rect_count<5>(rec, precs, algs);
            for (int n = 0; n < 5; ++n) { const double v = nfa(total, algs[n], rs[n].p); if (v > best) { best = v; rec = rs[n]; } }
        }
        if (best > 0) return best;
        r = rec;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.width -= delta; double v = rect_nfa(r); if (v > best) { rec = r; best = v; } }
        if (best > 0) return best;
        for (int side = 0; side < 2; side++) {
            r = rec;
            for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
                if (side == 0) { r.x1 += -r.dy * d2; r.y1 += r.dx * d2; r.x2 += -r.dy * d2; r.y2 += r.dx * d2; }
                else { r.x1 -= -r.dy * d2; r.y1 -= r.dx * d2; r.x2 -= -r.dy * d2; r.y2 -= r.dx * d2; }
                r.width -= delta;
                double v = rect_nfa(r); if (v > best) { rec = r; best = v; } }
            if (best > 0) return best;
        }
        r = rec;
        if ((r.width - delta) >= 0.5) { // the width does not change in this loop: again one walk for the five tolerances
            RectH rs[5]; double precs[5]; int algs[5];
            for (int n = 0; n < 5; ++n) { r.p /= 2; r.prec = r.p * PI_; rs[n] = r; precs[n] = r.prec; }
            const int total = rect_count<5>(rec, precs, algs);
            for (int n = 0; n < 5; ++n) { const double v = nfa(total, algs[n], rs[n].p); if (v > best) { best = v; rec = rs[n]; } }
        }
        return best;
    }
    bool timed = false;
    double t_imp = 0, t_sort = 0, t_grow = 0, t_rect = 0; long n_seeds = 0, n_regions = 0, n_pix = 0; // stage timers (ms) of this thread, reported by the caller
    // flsd :464-535 (LSD_REFINE_ADV, scale 0.8): segments as x1 y1 x2 y2 floats.  Input: the frame's defined pixels in address
    // order (the undefined ones are skipped by the reference's seed loop and fail every alignment test, so they never matter).
    void run(int w_, int h_, int ne, const int *e_addr, const float *e_deg, const float2 *e_cs, const double *e_mod, std::vector<float> &lines) {
        const auto tt0 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        w = w_; h = h_;
        const size_t n = (size_t)w * h;
        if (pix.size() != n) { pix.assign(n, Px{NOTDEF_F, 0.f, 0.f, NOTDEF_F}); dmod.assign(n, 0.0); rx.resize(n); ry.resize(n); rang.resize(n); }
        const double prec = PI_ * 22.5 / 180, p = 22.5 / 180;
        // Seed order = address order.  ll_angle links the 1024-bin pseudo-ordering through `next` pointers (:588-634), but flsd walks the
        // `list` vector by index (:477-480) and its entries were appended in raster order: the gradient ordering has no effect on the
        // reference's output (established by running the reference's own lsd.cpp: oracle/_ref, tests/test_ref_pins.py).
        constexpr int PFD = 24; // the scatter is sparse in the dense maps: ask for the lines a few entries ahead
        for (int i = 0; i < ne; i++) {
            if (i + PFD < ne) { const int a = e_addr[i + PFD] & 0x7fffffff; __builtin_prefetch(&pix[a], 1); __builtin_prefetch(&dmod[a], 1); }
            const int q = e_addr[i] & 0x7fffffff;
            pix[q] = Px{e_deg[i], e_cs[i].x, e_cs[i].y, e_deg[i]}; dmod[q] = e_mod[i];
        }
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        const int min_reg_size = int(-LOG_NT / std::log10(p));
        lines.clear();
        if (timed) t_sort += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count();
        for (int i = 0; i < ne; ++i) {
            const int adx = e_addr[i] & 0x7fffffff;
            if (pix[adx].free_deg == NOTDEF_F) continue; // used
            if (e_addr[i] < 0) { pix[adx].free_deg = NOTDEF_F; if (timed) n_seeds++; continue; } // a region of one pixel (flagged by lsd_emit): used, nothing else
            int rn; double reg_angle;
            const auto tg0 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            grow(adx % w, adx / w, rn, reg_angle, prec);
            const auto tg1 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            if (timed) { t_grow += std::chrono::duration<double, std::milli>(tg1 - tg0).count(); n_seeds++; n_pix += rn; }
            if (rn < min_reg_size) continue;
            n_regions++;
            struct RT { bool on; double &acc; std::chrono::steady_clock::time_point t0; ~RT() { if (on) acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } rt{timed, t_rect, tg1};
            RectH rec;
            to_rect(rn, reg_angle, prec, p, rec);
            if (!refine(rn, reg_angle, prec, p, rec, 0.7)) continue;
            const auto ti0 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            const double imp = improve(rec);
            if (timed) t_imp += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
            if (imp <= 0) continue;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= 0.8; rec.y1 /= 0.8; rec.x2 /= 0.8; rec.y2 /= 0.8;
            lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1)); lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
        }
        for (int i = 0; i < ne; i++) { // leave the dense maps clean for the next frame
            if (i + PFD < ne) { const int a = e_addr[i + PFD] & 0x7fffffff; __builtin_prefetch(&pix[a], 1); }
            { Px &q = pix[e_addr[i] & 0x7fffffff]; q.deg = NOTDEF_F; q.free_deg = NOTDEF_F; }
        }
    }
};

static void to_keylines(const std::vector<float> &lines, int W, int H, std::vector<cs_keyline> &out) { // LSDDetector.cpp:75-101,205-263
    out.clear();
    const float thre = 10;
    int cls = -1;
    for (size_t k = 0; k + 3 < lines.size(); k += 4) {
        float e[4] = {lines[k], lines[k + 1], lines[k + 2], lines[k + 3]};
        for (int q = 0; q < 4; q++) { const int lim = (q & 1) ? H : W; if (e[q] < 0) e[q] = 0; if (e[q] >= lim) e[q] = (float)lim - 1.0f; }
        const float os = std::pow((float)1, 0);
        cs_keyline kl;
        memset(&kl, 0, sizeof(kl));
        kl.startPointX = e[0] * os; kl.startPointY = e[1] * os; kl.endPointX = e[2] * os; kl.endPointY = e[3] * os;
        if (((kl.startPointX < thre) && (kl.endPointX < thre)) || ((kl.startPointX > W - thre) && (kl.endPointX > W - thre)) ||
            ((kl.startPointY < thre) && (kl.endPointY < thre)) || ((kl.startPointY > H - thre) && (kl.endPointY > H - thre)))
            continue;
        kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
        kl.lineLength = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
        const int x1 = (int)std::lrint(e[0]), y1 = (int)std::lrint(e[1]), x2 = (int)std::lrint(e[2]), y2 = (int)std::lrint(e[3]);
        kl.numOfPixels = std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1; // cv::LineIterator(...).count, 8-connected
        kl.angle = std::atan2((kl.endPointY - kl.startPointY), (kl.endPointX - kl.startPointX));
        kl.class_id = ++cls; kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / std::max(W, H);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2; kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        out.push_back(kl);
    }
}
} // namespace

struct cs_lsd {
    int W = 0, H = 0, w = 0, h = 0, max_frames = 0, n_frames = 0;
    int region_stage = 0; // cs_lsd_set_region_stage: CS_LSD_REGIONS_AUTO / _HOST / _WAVE_PER_FRAME / _BACKLOG
    GK gk{};
    double threshold = 0;
    size_t pix_bytes = 0; bool scaled_kept = true;           // the arena d_tmp = [d_blur | d_scaled], later [region stage's pixel records (pix_bytes) | LBD blur]
    uint8_t *d_gray = nullptr; double *d_tmp = nullptr, *d_blur = nullptr, *d_scaled = nullptr, *d_mod = nullptr; float *d_ang = nullptr; // d_ang: float degrees (see lsd_gradient)
    int *d_xofs = nullptr, *d_yofs = nullptr; float *d_ax = nullptr, *d_ay = nullptr;
    int nbx = 0;                                            // 256-pixel segments per scaled row
    int *d_seg_cnt = nullptr, *d_seg_base = nullptr;        // per (frame, row, segment) defined-pixel count / exclusive scan
    int *d_blk_tot = nullptr;                               // totals of the 1024-segment scan blocks
    int *d_caddr = nullptr; float *d_cdeg = nullptr; float2 *d_ccs = nullptr; double *d_cmod = nullptr; size_t ccap = 0;   // compacted defined pixels (device): address, angle, cos / sin, norm
    int *h_caddr = nullptr; float *h_cdeg = nullptr; float2 *h_ccs = nullptr; double *h_cmod = nullptr; size_t hcap = 0;   // same, pinned host
    std::vector<int> frame_base;
    std::vector<std::vector<cs_keyline>> keylines;
    // LBD descriptors of the detected lines (optional second half of cs_lsd_run)
    uint8_t *d_lblur = nullptr; uint32_t *d_dxy = nullptr, *dxy_cur = nullptr; // LBD: blurred frames, interleaved Sobel dx | dy << 16 (dxy_cur: where the last batch's map is -- d_dxy or the arena)
    cs_keyline *d_kl = nullptr; int *d_line_frame = nullptr; uint8_t *d_desc = nullptr;
    size_t line_cap = 0;
    std::vector<int> line_off;       // per frame offset into the concatenated line list
    std::vector<uint8_t> h_desc;     // concatenated n x 32
    bool have_desc = false;
    LsdSeq *seq = nullptr;           // device buffers of the region stage (lsd_regions.hip)
    int seq_wpb = 16; // frames per workgroup of the device region stage (cs_lsd_set_shared_gpu)
    bool walk_bg = false; // the region walk on the context's background stream (cs_lsd_set_shared_gpu; CUBESLAM_LSD_WALK_BG=0 keeps it on the context's own)
    void (*gate_wait)(void *) = nullptr; void (*gate_done)(void *) = nullptr; void *gate_arg = nullptr; // the front-end runner's phase gate around the region stage (frontend.hip)
    long rg_stats[5] = {0, 0, 0, 0, 0}; // last batch: 1 = device stage asked for, region_grow calls, rectangles at rect_improve, 1 = fell back to the host stage, window fetches
};

int cs_lbd_batch_maps(cs_ctx *ctx, const uint8_t *d_gray, int W, int H, int F, uint8_t *d_blur, uint32_t *d_dxy);
int cs_lbd_batch_desc(cs_ctx *ctx, const cs_keyline *d_kl, const int *d_line_frame, int n, const uint32_t *d_dxy, int W, int H, uint8_t *d_desc, float *d_f);

static void lsd_free_lines(cs_lsd *l) {
    void *ptrs[] = {l->d_kl, l->d_line_frame, l->d_desc};
    for (void *p : ptrs) if (p) hipFree(p);
    l->d_kl = nullptr; l->d_line_frame = nullptr; l->d_desc = nullptr; l->line_cap = 0;
}

static int lsd_upload(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride) {
    if (!ctx || !l || !gray || n_frames < 1 || n_frames > l->max_frames || stride < l->W) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    l->n_frames = n_frames; l->have_desc = false;
    for (int f = 0; f < n_frames; f++) // every frame is `height` rows of `stride` bytes
        CS_HIP(ctx, hipMemcpy2DAsync(l->d_gray + (size_t)f * l->W * l->H, (size_t)l->W, gray + (size_t)f * stride * l->H, (size_t)stride, (size_t)l->W, (size_t)l->H, hipMemcpyHostToDevice, ctx->stream));
    return CS_OK;
}

static int lsd_run(cs_ctx *ctx, cs_lsd *l, int with_lbd) {
    if (!ctx || !l || l->n_frames < 1) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = l->W, H = l->H, w = l->w, h = l->h, F = l->n_frames;
    l->have_desc = false; l->scaled_kept = true;
    // CUBESLAM_LSD_HOSTPROF: wall clock of a pass's phases as the calling thread sees them (maps + counts back | region stage | KeyLines on the host | LBD)
    static const bool hostprof = getenv("CUBESLAM_LSD_HOSTPROF") != nullptr;
    double tp[5] = {0, 0, 0, 0, 0};
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    tp[0] = now_ms();
    // d_blur and d_scaled (doubles: 2.5 + 1.6 MB per frame) only live from one kernel to the next: they are the two halves of one arena (d_tmp) that the device region stage
    // takes over for its pixel records once lsd_gradient is through, and the LBD blur for its one-kernel life (see below): 4.0 MB per frame that are not allocated twice
    CS_LAUNCH(ctx, "lsd_blur_hv", lsd_blur_hv, dim3(((W + 63) / 64) * ((H + LSD_ROWS - 1) / LSD_ROWS), 1, F), dim3(64), 0, l->d_gray, W, H, l->gk, l->d_blur);
    CS_LAUNCH(ctx, "lsd_resize", lsd_resize, dim3((w + 255) / 256, h, F), dim3(256), 0, l->d_blur, W, H, w, h, l->d_xofs, l->d_ax, l->d_yofs, l->d_ay, l->d_scaled);
    const int nbx = l->nbx, n_seg = F * h * nbx;
    CS_LAUNCH(ctx, "lsd_gradient", lsd_gradient, dim3(nbx, h, F), dim3(256), 0, l->d_scaled, w, h, l->threshold, l->d_mod, l->d_ang, l->d_seg_cnt);
    const int nblk = (n_seg + 1023) / 1024;
    CS_LAUNCH(ctx, "lsd_scan", lsd_scan_blocks, dim3(nblk), dim3(1024), 0, l->d_seg_cnt, n_seg, l->d_seg_base, l->d_blk_tot);
    CS_LAUNCH(ctx, "lsd_scan", lsd_scan_top, dim3(1), dim3(1024), 0, l->d_blk_tot, nblk, l->d_seg_base + n_seg);
    CS_LAUNCH(ctx, "lsd_scan", lsd_scan_add, dim3(nblk), dim3(1024), 0, l->d_seg_base, n_seg, l->d_blk_tot);
    // only the defined pixels (gradient above rho) go to the host: frame bases first, then the compacted (address, angle, norm) lists
    l->frame_base.assign((size_t)F + 1, 0);
    CS_HIP(ctx, hipMemcpy2DAsync(l->frame_base.data(), sizeof(int), l->d_seg_base, sizeof(int) * (size_t)h * nbx, sizeof(int), (size_t)F + 1, hipMemcpyDeviceToHost, ctx->stream));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t total = (size_t)l->frame_base[F];
    tp[1] = now_ms();
    int r;
    if (total > l->ccap) {
        void *old[] = {l->d_caddr, l->d_cdeg, l->d_ccs, l->d_cmod};
        for (void *q : old) if (q) hipFree(q);
        l->d_caddr = nullptr; l->d_cdeg = nullptr; l->d_ccs = nullptr; l->d_cmod = nullptr; l->ccap = 0;
        const size_t cap = total + total / 16 + 4096; // (16 B per defined pixel, 75 M of them in a 1 024-frame batch: a sixteenth of headroom, a larger batch reallocates)
        r = cs_dalloc(ctx, &l->d_caddr, cap); if (r) return r;
        r = cs_dalloc(ctx, &l->d_cdeg, cap); if (r) return r;
        r = cs_dalloc(ctx, &l->d_ccs, cap); if (r) return r;
        l->ccap = cap;
    }
    // region growing / rectangles / NFA (a15): interchangeable stages with byte-identical KeyLines (tests/test_lsd_gpu.py).
    //   seq    one wave per frame (lsd_rg_seq.h): ~110 ms per frame, 36 k frames/s with the chip full of them -- the default from 512 frames on;
    //   wlk    a walker wave with one lane per frame that seeds and grows + rectangle waves for the regions it parks, one workgroup (lsd_rg_wlk.h): 64 frames per
    //          wave slot instead of one, ~450 ms per frame -- for a backlog nobody waits for, by name only (round 4's grp / grp2 / lpf are gone);
    //   host   the OpenMP stage below, one frame per thread -- the default for smaller batches (one frame: 4 ms).
    // cs_lsd_set_region_stage picks one per detector (a caller with a backlog nobody waits for takes wlk: 3.7 x seq's frames/s with the chip full of it); the
    // development override CUBESLAM_LSD_REGIONS = seq | wlk | host comes first (tests, tools).
    std::vector<std::vector<float>> dev_lines;
    bool on_device = false;
    const char *mode = getenv("CUBESLAM_LSD_REGIONS");
    if (!mode || !*mode) mode = l->region_stage == CS_LSD_REGIONS_HOST ? "host" : l->region_stage == CS_LSD_REGIONS_WAVE_PER_FRAME ? "seq" : l->region_stage == CS_LSD_REGIONS_BACKLOG ? "wlk" : (F >= 512 ? "seq" : "host");
    const int grp_p = strcmp(mode, "wlk") == 0 ? 65 : 0; // lsd_rg_wlk: walker waves with one lane per frame + rectangle waves
    const bool use_seq = total > 0 && (strcmp(mode, "seq") == 0 || grp_p);
    // lsd_rg_seq's records are written by the emit kernel itself when they live in the arena (always, unless a caller's frames outgrew it): no fill, no scatter
    const bool pix_by_emit = use_seq && grp_p == 0 && l->pix_bytes >= (size_t)F * w * h * 4 + 64 && !(getenv("CUBESLAM_LSD_EMIT_PIX") && atoi(getenv("CUBESLAM_LSD_EMIT_PIX")) == 0);
    auto emit = [&](bool with_norms) -> int { // the compacted norms (8 B per defined pixel) are the host stage's: the device stage reads the dense map
        if (with_norms && !l->d_cmod) { const int q = cs_dalloc(ctx, &l->d_cmod, l->ccap); if (q) return q; }
        if (pix_by_emit && !with_norms)
            CS_LAUNCH(ctx, "lsd_emit", lsd_emit<true>, dim3(nbx, h, F), dim3(256), 0, l->d_mod, l->d_ang, w, h, l->d_seg_base, l->d_caddr, l->d_cdeg, l->d_ccs, (double *)nullptr, reinterpret_cast<float *>(l->d_tmp));
        else
            CS_LAUNCH(ctx, "lsd_emit", lsd_emit<false>, dim3(nbx, h, F), dim3(256), 0, l->d_mod, l->d_ang, w, h, l->d_seg_base, l->d_caddr, l->d_cdeg, l->d_ccs, with_norms ? l->d_cmod : (double *)nullptr, (float *)nullptr);
        return CS_OK;
    };
    if (total > 0) { r = emit(!use_seq); if (r) return r; }
    const bool late_maps = [&] { const char *e = getenv("CUBESLAM_LBD_MAPS"); return !(e && !strcmp(e, "early")); }();
    auto lbd_maps = [&]() -> int { // the derivative maps only depend on the gray frames
        // device region stage: the Sobel map (4 B per pixel) is made once the stage is through, in the arena it has left -- [dxy | ... | blurred frames behind the pixel records' place];
        // CUBESLAM_LBD_MAPS=early makes it ahead of the stage in a buffer of its own (the phased runner wants every map kernel of a batch on the GPU before its gate)
        uint8_t *lblur = l->d_lblur;
        uint32_t *dxy = l->d_dxy;
        if (use_seq) lblur = reinterpret_cast<uint8_t *>(l->d_tmp) + std::max(l->pix_bytes, (size_t)W * H * l->max_frames * 4); // the blurred frames are the Sobel kernel's input and nothing else; behind the walk's map AND behind the Sobel map that is written at the arena's head while they are read (4 B per pixel)
        else if (!lblur) { const int q = cs_dalloc(ctx, &l->d_lblur, (size_t)W * H * l->max_frames); if (q) return q; lblur = l->d_lblur; }
        if (use_seq && late_maps) dxy = reinterpret_cast<uint32_t *>(l->d_tmp);
        else if (!dxy) { const int q = cs_dalloc(ctx, &l->d_dxy, (size_t)W * H * l->max_frames); if (q) return q; dxy = l->d_dxy; }
        l->dxy_cur = dxy;
        return cs_lbd_batch_maps(ctx, l->d_gray, W, H, F, lblur, dxy);
    };
    bool maps_done = false;
    if (use_seq && with_lbd && !late_maps) { r = lbd_maps(); if (r) return r; maps_done = true; } // ahead of the region stage: every map kernel of the batch is on the GPU before the phase gate
    if (l->gate_wait && !use_seq) l->gate_wait(l->gate_arg); // (phased front-end: the region stage starts when the caller's own GPU work of the phase is done; the device stage waits inside lsd_seq_run, in front of its one long kernel)
    if (use_seq) {
        long st[4] = {0, 0, 0, 0};
        l->scaled_kept = false; // the arena is the region stage's from here on
        r = lsd_seq_run(ctx, &l->seq, F, w, h, l->d_ang, l->d_mod, l->d_caddr, l->d_cdeg, l->d_ccs, l->frame_base.data(), dev_lines, st, l->gate_wait, l->gate_done, l->gate_arg, grp_p, l->seq_wpb,
                        l->d_tmp, l->pix_bytes, pix_by_emit, l->walk_bg);
        tp[2] = now_ms();
        l->rg_stats[0] = 1; l->rg_stats[1] = st[0]; l->rg_stats[2] = st[2]; l->rg_stats[3] = r == CS_OK ? 0 : 1; l->rg_stats[4] = st[1];
        if (r == CS_OK) on_device = true;
        else if (r != CS_ERR_CAPACITY) return r; // a region outgrew the wave's list: the host stage takes the batch
        else { r = emit(true); if (r) return r; }
    } else
        for (int k = 0; k < 5; k++) l->rg_stats[k] = 0;
    if (!on_device && total > l->hcap) {
        if (l->h_caddr) hipHostFree(l->h_caddr); if (l->h_cdeg) hipHostFree(l->h_cdeg); if (l->h_ccs) hipHostFree(l->h_ccs); if (l->h_cmod) hipHostFree(l->h_cmod);
        l->h_caddr = nullptr; l->h_cdeg = nullptr; l->h_ccs = nullptr; l->h_cmod = nullptr; l->hcap = 0;
        const size_t cap = total + total / 4 + 4096;
        if (hipHostMalloc((void **)&l->h_caddr, cap * sizeof(int), hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&l->h_cdeg, cap * sizeof(float), hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&l->h_ccs, cap * sizeof(float2), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&l->h_cmod, cap * sizeof(double), hipHostMallocDefault) != hipSuccess) return CS_ERR_NOMEM;
        l->hcap = cap;
    }
    if (!on_device && total > 0) {
        r = cs_d2h(ctx, l->h_caddr, l->d_caddr, total); if (r) return r;
        r = cs_d2h(ctx, l->h_cdeg, l->d_cdeg, total); if (r) return r;
        r = cs_d2h(ctx, l->h_ccs, l->d_ccs, total); if (r) return r;
        r = cs_d2h(ctx, l->h_cmod, l->d_cmod, total); if (r) return r;
    }
    hipEvent_t ev = ctx->get_event();
    CS_HIP(ctx, hipEventRecord(ev, ctx->stream));
    if (with_lbd && !maps_done && !(use_seq && late_maps && on_device)) { r = lbd_maps(); if (r) return r; maps_done = true; } // they run while the host grows regions
    CS_HIP(ctx, hipEventSynchronize(ev));
    ctx->pool.push_back(ev);
    // one BATCHED host stage at a time per process: two line detectors that alternate batches (bench.py) overlap their GPU phases with
    // the other's region growing instead of splitting the host cores between two OpenMP teams.  A single frame (the drop-in call of one
    // camera frame) is one core's work: it runs on the calling thread, outside the lock and without a team, so that callers on several
    // threads (each with its own context) grow their frames side by side.
    if (!on_device) {
    static std::mutex host_stage;
    std::unique_lock<std::mutex> host_lock(host_stage, std::defer_lock);
    const bool solo = F == 1;
    if (!solo) host_lock.lock();
    const auto t0 = std::chrono::steady_clock::now();
    l->keylines.assign((size_t)F, {});
    std::vector<int> lpt((size_t)F);
    for (int f = 0; f < F; f++) lpt[f] = f;
    std::stable_sort(lpt.begin(), lpt.end(), [&](int a, int b) { return l->frame_base[a + 1] - l->frame_base[a] > l->frame_base[b + 1] - l->frame_base[b]; });
    if (!solo) cs_omp_prepare();
#pragma omp parallel num_threads(std::max(1, std::min(ctx->host_threads, F))) if (!solo)
    {
        LsdHost host;
        host.timed = ctx->timing;
        std::vector<float> lines;
#pragma omp for schedule(dynamic, 1)
        for (int fi = 0; fi < F; fi++) {
            const int f = lpt[fi]; // most defined pixels first: the tail of the dynamic schedule is made of the cheapest frames
            const int b0 = l->frame_base[f];
            host.run(w, h, l->frame_base[f + 1] - b0, l->h_caddr + b0, l->h_cdeg + b0, l->h_ccs + b0, l->h_cmod + b0, lines);
            to_keylines(lines, W, H, l->keylines[f]);
        }
        if (ctx->timing) {
#pragma omp critical
            { // CPU-milliseconds summed over the threads (divide by the thread count for wall time)
                ctx->timings["host_lsd_cpu_sort"].total_ms += host.t_sort; ctx->timings["host_lsd_cpu_grow"].total_ms += host.t_grow; ctx->timings["host_lsd_cpu_rect"].total_ms += host.t_rect; ctx->timings["host_lsd_cpu_improve"].total_ms += host.t_imp; ctx->timings["host_lsd_cpu_improve"].count = 1;
                ctx->timings["host_lsd_n_seeds"].total_ms += (double)host.n_seeds; ctx->timings["host_lsd_n_regions"].total_ms += (double)host.n_regions; ctx->timings["host_lsd_n_pix"].total_ms += (double)host.n_pix;
                ctx->timings["host_lsd_n_def"].total_ms += (double)l->frame_base[F] / omp_get_num_threads();
                for (const char *k : {"host_lsd_cpu_sort", "host_lsd_cpu_grow", "host_lsd_cpu_rect", "host_lsd_n_seeds", "host_lsd_n_regions", "host_lsd_n_pix", "host_lsd_n_def"}) ctx->timings[k].count = 1;
            }
        }
    }
    if (ctx->timing) { auto &rec = ctx->timings["host_lsd_regions"]; rec.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); rec.count++; }
    if (!solo) host_lock.unlock();
    } else {
        l->keylines.assign((size_t)F, {});
        cs_omp_prepare();
#pragma omp parallel for schedule(static) num_threads(std::max(1, std::min(ctx->host_threads, F)))
        for (int f = 0; f < F; f++) to_keylines(dev_lines[f], W, H, l->keylines[f]);
    }
    tp[3] = now_ms();
    l->line_off.assign((size_t)F + 1, 0);
    for (int f = 0; f < F; f++) l->line_off[f + 1] = l->line_off[f] + (int)l->keylines[f].size();
    if (with_lbd) {
        const int nl = l->line_off[F];
        l->h_desc.assign((size_t)nl * 32, 0);
        if (nl > 0) {
            if ((size_t)nl > l->line_cap) {
                lsd_free_lines(l);
                const size_t cap = (size_t)nl + nl / 4 + 256;
                r = cs_dalloc(ctx, &l->d_kl, cap); if (r) return r;
                r = cs_dalloc(ctx, &l->d_line_frame, cap); if (r) return r;
                r = cs_dalloc(ctx, &l->d_desc, cap * 32); if (r) return r;
                l->line_cap = cap;
            }
            std::vector<int> lf((size_t)nl);
            std::vector<cs_keyline> kl((size_t)nl); // one upload for the whole batch (128 per-frame copies cost more than the descriptors)
            for (int f = 0; f < F; f++) {
                std::copy(l->keylines[f].begin(), l->keylines[f].end(), kl.begin() + l->line_off[f]);
                for (int i = l->line_off[f]; i < l->line_off[f + 1]; i++) lf[i] = f;
            }
            r = cs_h2d(ctx, l->d_kl, kl.data(), (size_t)nl); if (r) return r;
            r = cs_h2d(ctx, l->d_line_frame, lf.data(), (size_t)nl); if (r) return r;
            CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // kl, lf are locals
            if (!maps_done) { r = lbd_maps(); if (r) return r; } // (device region stage: its pixel records are dead now, the Sobel map takes their place)
            r = cs_lbd_batch_desc(ctx, l->d_kl, l->d_line_frame, nl, l->dxy_cur, W, H, l->d_desc, nullptr); if (r) return r;
            r = cs_d2h(ctx, l->h_desc.data(), l->d_desc, (size_t)nl * 32); if (r) return r;
        }
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        l->have_desc = true;
    }
    if (hostprof) { tp[4] = now_ms(); fprintf(stderr, "[lsd pass %p] maps+counts %.1f  regions %.1f  keylines %.1f  lbd %.1f  total %.1f ms\n", (void *)l, tp[1] - tp[0], tp[2] ? tp[2] - tp[1] : 0.0, tp[3] - (tp[2] ? tp[2] : tp[1]), tp[4] - tp[3], tp[4] - tp[0]); }
    return CS_OK;
}

extern "C" {

void cs_lsd_destroy(cs_ctx *ctx, cs_lsd *l) {
    if (!l) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {l->d_gray, l->d_tmp, l->d_mod, l->d_ang, l->d_xofs, l->d_yofs, l->d_ax, l->d_ay, l->d_lblur, l->d_dxy};
    for (void *p : ptrs) if (p) hipFree(p);
    lsd_free_lines(l);
    lsd_seq_destroy(l->seq);
    void *more[] = {l->d_seg_cnt, l->d_seg_base, l->d_blk_tot, l->d_caddr, l->d_cdeg, l->d_ccs, l->d_cmod};
    for (void *p : more) if (p) hipFree(p);
    if (l->h_caddr) hipHostFree(l->h_caddr);
    if (l->h_cdeg) hipHostFree(l->h_cdeg);
    if (l->h_ccs) hipHostFree(l->h_ccs);
    if (l->h_cmod) hipHostFree(l->h_cmod);
    delete l;
}

int cs_lsd_create(cs_ctx *ctx, int width, int height, int max_frames, cs_lsd **out) {
    if (!ctx || !out || width < 16 || height < 16 || max_frames < 1) return CS_ERR_BAD_ARG;
    *out = nullptr;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_lsd *l = new (std::nothrow) cs_lsd();
    if (!l) return CS_ERR_NOMEM;
    l->W = width; l->H = height; l->max_frames = max_frames;
    const double SCALE = 0.8, sigma = 0.6 / SCALE; // lsd.cpp:185-187 defaults, flsd :451-457
    if ((int)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))) != KH) { delete l; return CS_ERR_BAD_ARG; }
    { double sum = 0, s2 = -0.5 / (sigma * sigma); for (int i = 0; i < 2 * KH + 1; i++) { double x = i - (2 * KH) * 0.5; l->gk.k[i] = std::exp(s2 * x * x); sum += l->gk.k[i]; } sum = 1. / sum; for (int i = 0; i < 2 * KH + 1; i++) l->gk.k[i] *= sum; }
    l->w = (int)std::lrint(width * SCALE); l->h = (int)std::lrint(height * SCALE);
    l->threshold = 2.0 / std::sin(PI_ * 22.5 / 180); // rho = QUANT / sin(prec)
    const double sx = 1. / SCALE;
    std::vector<int> xofs(l->w), yofs(l->h);
    std::vector<float> ax((size_t)l->w * 2), ay((size_t)l->h * 2);
    auto fl = [](double v) { int i = (int)v; return i - (i > v); };
    for (int dx = 0; dx < l->w; dx++) { float fx = (float)((dx + 0.5) * sx - 0.5); int s = fl(fx); fx -= s; if (s < 0) { fx = 0; s = 0; } if (s >= width - 1) { fx = 0; s = width - 1; } xofs[dx] = s; ax[dx * 2] = 1.f - fx; ax[dx * 2 + 1] = fx; }
    for (int dy = 0; dy < l->h; dy++) { float fy = (float)((dy + 0.5) * sx - 0.5); int s = fl(fy); fy -= s; if (s < 0) { fy = 0; s = 0; } if (s >= height - 1) { fy = 0; s = height - 1; } yofs[dy] = s; ay[dy * 2] = 1.f - fy; ay[dy * 2 + 1] = fy; }
    const size_t N = (size_t)width * height * max_frames, n = (size_t)l->w * l->h * max_frames;
#define A_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_lsd_destroy(ctx, l); return r__; } } while (0)
    // one arena for [blurred | scaled] frames; afterwards [the device region stage's own copy of the angle map (4 B per scaled pixel) | LBD blur (1 B per pixel)]: 4 n + N <= 8 N + 8 n bytes
    l->pix_bytes = (n * 4 + 64 + 255) / 256 * 256;
    A_(cs_dalloc(ctx, &l->d_gray, N)); A_(cs_dalloc(ctx, &l->d_tmp, N + n)); l->d_blur = l->d_tmp; l->d_scaled = l->d_tmp + N;
    A_(cs_dalloc(ctx, &l->d_mod, n)); A_(cs_dalloc(ctx, &l->d_ang, n));
    A_(cs_dalloc(ctx, &l->d_xofs, xofs.size())); A_(cs_dalloc(ctx, &l->d_yofs, yofs.size())); A_(cs_dalloc(ctx, &l->d_ax, ax.size())); A_(cs_dalloc(ctx, &l->d_ay, ay.size()));
    l->nbx = (l->w + 255) / 256;
    A_(cs_dalloc(ctx, &l->d_seg_cnt, (size_t)max_frames * l->h * l->nbx)); A_(cs_dalloc(ctx, &l->d_seg_base, (size_t)max_frames * l->h * l->nbx + 1));
    A_(cs_dalloc(ctx, &l->d_blk_tot, ((size_t)max_frames * l->h * l->nbx + 1023) / 1024 + 1));
    A_(cs_h2d(ctx, l->d_xofs, xofs.data(), xofs.size())); A_(cs_h2d(ctx, l->d_yofs, yofs.data(), yofs.size()));
    A_(cs_h2d(ctx, l->d_ax, ax.data(), ax.size())); A_(cs_h2d(ctx, l->d_ay, ay.data(), ay.size()));
#undef A_
    { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) { cs_lsd_destroy(ctx, l); return CS_ERR_HIP; } }
    *out = l;
    return CS_OK;
}

int cs_lsd_detect(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride, cs_keyline *out, int cap, int *counts) {
    if (!out || !counts || cap < 1) return CS_ERR_BAD_ARG;
    int r = lsd_upload(ctx, l, gray, n_frames, stride); if (r) return r;
    r = lsd_run(ctx, l, 0); if (r) return r;
    int status = CS_OK;
    for (int f = 0; f < n_frames; f++) {
        const int n = (int)l->keylines[f].size();
        counts[f] = std::min(n, cap);
        if (n > cap) status = CS_ERR_CAPACITY;
        memcpy(out + (size_t)f * cap, l->keylines[f].data(), sizeof(cs_keyline) * (size_t)counts[f]);
    }
    return status;
}

int cs_lsd_detect_filter_lines(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride, float length_thres, float *lines, int cap, int *counts) {
    if (!lines || !counts || cap < 1) return CS_ERR_BAD_ARG;
    int r = lsd_upload(ctx, l, gray, n_frames, stride); if (r) return r;
    r = lsd_run(ctx, l, 0); if (r) return r;
    int status = CS_OK;
    for (int f = 0; f < n_frames; f++) { // filter_lines :200-207 + keylines_to_mat :26-36
        int n = 0;
        for (const cs_keyline &k : l->keylines[f])
            if (k.octave == 0 && k.lineLength > length_thres) {
                if (n < cap) { float *o = lines + ((size_t)f * cap + n) * 4; o[0] = k.startPointX * 1.f; o[1] = k.startPointY * 1.f; o[2] = k.endPointX * 1.f; o[3] = k.endPointY * 1.f; }
                n++;
            }
        counts[f] = std::min(n, cap);
        if (n > cap) status = CS_ERR_CAPACITY;
    }
    return status;
}

// filter_lines (:200-207) + keylines_to_mat (:26-36) over the KeyLines of the last cs_lsd_run of the resident frames: what detect_filter_lines returns,
// without a second pass over the images (the chain's hand-over to cs_cuboid_batch_set_lines)
int cs_lsd_read_filter_lines(cs_ctx *ctx, cs_lsd *l, float length_thres, float *lines, int cap, int *counts, int n_frames) {
    if (!ctx || !l || !lines || !counts || cap < 1 || n_frames < 0 || (size_t)n_frames < l->keylines.size()) return CS_ERR_BAD_ARG; // (the caller's arrays hold n_frames frames)
    int status = CS_OK;
    for (size_t f = 0; f < l->keylines.size(); f++) {
        int n = 0;
        for (const cs_keyline &k : l->keylines[f])
            if (k.octave == 0 && k.lineLength > length_thres) {
                if (n < cap) { float *o = lines + (f * cap + n) * 4; o[0] = k.startPointX * 1.f; o[1] = k.startPointY * 1.f; o[2] = k.endPointX * 1.f; o[3] = k.endPointY * 1.f; }
                n++;
            }
        counts[f] = std::min(n, cap);
        if (n > cap) status = CS_ERR_CAPACITY;
    }
    return status;
}

int cs_lsd_upload(cs_ctx *ctx, cs_lsd *l, const uint8_t *gray, int n_frames, int stride) {
    int r = lsd_upload(ctx, l, gray, n_frames, stride); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

// the frames of the next run from DEVICE memory (n_frames x H x W bytes): a copy on the context's stream, nothing waits (the streaming front-end's hand-over)
int cs_lsd_geometry(const cs_lsd *l, int *width, int *height, int *max_frames) { // (library-internal, see cs_orb_geometry)
    if (!l) return CS_ERR_BAD_ARG;
    *width = l->W; *height = l->H; *max_frames = l->max_frames;
    return CS_OK;
}
int cs_lsd_set_frames_device(cs_ctx *ctx, cs_lsd *l, const uint8_t *d_gray, int n_frames) {
    if (!ctx || !l || !d_gray || n_frames < 1 || n_frames > l->max_frames) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    l->n_frames = n_frames; l->have_desc = false;
    CS_HIP(ctx, hipMemcpyAsync(l->d_gray, d_gray, (size_t)n_frames * l->W * l->H, hipMemcpyDeviceToDevice, ctx->stream));
    return CS_OK;
}

int cs_lsd_run(cs_ctx *ctx, cs_lsd *l, int with_lbd) { return lsd_run(ctx, l, with_lbd); }

int cs_lsd_set_region_stage(cs_lsd *l, int stage) {
    if (!l || stage < CS_LSD_REGIONS_AUTO || stage > CS_LSD_REGIONS_BACKLOG) return CS_ERR_BAD_ARG;
    l->region_stage = stage;
    return CS_OK;
}

int cs_lsd_region_stats(cs_ctx *ctx, cs_lsd *l, long out[5]) {
    if (!ctx || !l || !out) return CS_ERR_BAD_ARG;
    for (int i = 0; i < 5; i++) out[i] = l->rg_stats[i];
    return CS_OK;
}

int cs_lsd_read(cs_ctx *ctx, cs_lsd *l, int frame, cs_keyline *out, int cap, int *count, uint8_t *desc) {
    if (!ctx || !l || frame < 0 || frame >= l->n_frames || !count || (int)l->keylines.size() <= frame || (desc && !l->have_desc)) return CS_ERR_BAD_ARG;
    const int n = (int)l->keylines[frame].size();
    *count = n;
    if (!out) return CS_OK; // size query
    if (n > cap) return CS_ERR_CAPACITY;
    memcpy(out, l->keylines[frame].data(), sizeof(cs_keyline) * (size_t)n);
    if (desc) memcpy(desc, l->h_desc.data() + (size_t)l->line_off[frame] * 32, (size_t)n * 32);
    return CS_OK;
}

int cs_lsd_get_maps(cs_ctx *ctx, cs_lsd *l, int frame, double *scaled, double *modgrad, double *angles, int *sw, int *sh) {
    if (!ctx || !l || frame < 0 || frame >= l->n_frames || !sw || !sh) return CS_ERR_BAD_ARG;
    *sw = l->w; *sh = l->h;
    const size_t n = (size_t)l->w * l->h;
    int r;
    if (scaled) {
        if (!l->scaled_kept) { ctx->err = "cs_lsd_get_maps: the scaled frames are gone once the device region stage ran (it takes their buffer over); modgrad and angles are kept"; return CS_ERR_BAD_ARG; }
        r = cs_d2h(ctx, scaled, l->d_scaled + n * frame, n); if (r) return r;
    }
    if (modgrad) { r = cs_d2h(ctx, modgrad, l->d_mod + n * frame, n); if (r) return r; }
    std::vector<float> deg;
    if (angles) { deg.resize(n); r = cs_d2h(ctx, deg.data(), l->d_ang + n * frame, n); if (r) return r; }
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (angles) for (size_t i = 0; i < n; i++) angles[i] = deg[i] == NOTDEF_DEG ? NOTDEF : double(deg[i]) * DEG_TO_RADS; // the reference's map value (:566)
    return CS_OK;
}

} // extern "C"

// internal (frontend.hip): filter_lines (:200-207) + keylines_to_mat (:26-36) over the KeyLines of the last run as cs_cuboid_batch_set_lines wants them
int cs_lsd_filter_lines_packed(cs_lsd *l, float length_thres, std::vector<int> &offsets, std::vector<double> &lines) {
    if (!l) return CS_ERR_BAD_ARG;
    offsets.assign(l->keylines.size() + 1, 0); lines.clear();
    for (size_t f = 0; f < l->keylines.size(); f++) {
        for (const cs_keyline &k : l->keylines[f])
            if (k.octave == 0 && k.lineLength > length_thres) { lines.push_back((double)(k.startPointX * 1.f)); lines.push_back((double)(k.startPointY * 1.f)); lines.push_back((double)(k.endPointX * 1.f)); lines.push_back((double)(k.endPointY * 1.f)); }
        offsets[f + 1] = (int)(lines.size() / 4);
    }
    if (lines.empty()) lines.push_back(0.0); // (a valid pointer for an empty hand-over)
    return CS_OK;
}
// internal (frontend.hip): the runner's phase gate around the region stage; wait() is called before it, done() once lsd_rg_seq has left the GPU
void cs_lsd_set_gate(cs_lsd *l, void (*wait)(void *), void (*done)(void *), void *arg) { l->gate_wait = wait; l->gate_done = done; l->gate_arg = arg; }
// shared: other detectors' walks and the other streams' kernels keep every CU busy anyway (the alternating front-end runner): the region stage spreads over the chip
void cs_lsd_set_shared_gpu(cs_lsd *l, int shared) {
    l->seq_wpb = shared ? 4 : 16;
    l->walk_bg = shared && !(getenv("CUBESLAM_LSD_WALK_BG") && atoi(getenv("CUBESLAM_LSD_WALK_BG")) == 0);
}
