/*
 * oracle/lsd_oracle.cpp -- CPU oracle for the LSD line detector of line_lbd.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PINNED: tests/test_ref_pins.py runs the reference's own lsd.cpp / LSDDetector.cpp, compiled whole from /root/reference
 * (oracle/_ref), on the same images -- KeyLines and the intermediate maps identical bit for bit; the OpenCV primitives under it stay restated.
 * Restated from /root/reference/line_lbd/libs/lsd.cpp
 * (LineSegmentDetectorImpl: flsd :440-536, ll_angle :538-635, region_grow :637-688, region2rect :690-746, get_theta :748-784,
 * refine :786-832, reduce_region_radius :834-871, rect_improve :873-975, rect_nfa :977-1098, nfa :1100-1136, isAligned
 * :1138-1154), libs/LSDDetector.cpp:75-101,153-287 and class/line_lbd_allclass.cpp:26-36,125-148,200-221.
 * OpenCV semantics assumed: GaussianBlur on CV_64F (separable, REFLECT_101, kernel exp(-x^2/2s^2) normalised, symmetric
 * summation k0*c + sum_k k_k*(l+r)), resize INTER_LINEAR on CV_64F with float coefficients, fastAtan2, cvRound,
 * LineIterator::count (8-connected, rounded end points).  Quirks kept: integer division and `tailp->p.x` in rect_nfa.
 */
#include "oracle.h"
#include "cv_prims.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
const double NOTDEF = -1024.0, PI = 3.1415926535897932384626433832795, M_3_2_PI_ = (3 * PI) / 2, M_2__PI_ = 2 * PI, DEG_TO_RADS = PI / 180;

static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static float fastAtan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / PI), p3 = -0.3258083974640975f * (float)(180 / PI), p5 = 0.1555786518463281f * (float)(180 / PI),
                p7 = -0.04432655554792128f * (float)(180 / PI);
    float ax = std::abs(x), ay = std::abs(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
static inline int reflect101(int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; } return p; }

// cv::GaussianBlur on CV_64F: getGaussianKernel(ksize, sigma, CV_64F), separable, rows then columns, symmetric taps summed pairwise
// (SymmRowFilter / SymmColumnFilter), BORDER_REFLECT_101
static void gaussian_blur_f64(const double *src, int W, int H, int ks, double sigma, double *dst) {
    const int hk = ks / 2;
    std::vector<double> k(ks);
    { double sum = 0, scale2X = -0.5 / (sigma * sigma); for (int i = 0; i < ks; i++) { double x = i - (ks - 1) * 0.5; k[i] = std::exp(scale2X * x * x); sum += k[i]; } sum = 1. / sum; for (int i = 0; i < ks; i++) k[i] *= sum; }
    std::vector<double> tmp((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double s = k[hk] * src[(size_t)y * W + x];
            for (int t = 1; t <= hk; t++) s += k[hk + t] * (src[(size_t)y * W + reflect101(x - t, W)] + src[(size_t)y * W + reflect101(x + t, W)]);
            tmp[(size_t)y * W + x] = s;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double s = k[hk] * tmp[(size_t)y * W + x];
            for (int t = 1; t <= hk; t++) s += k[hk + t] * (tmp[(size_t)reflect101(y - t, H) * W + x] + tmp[(size_t)reflect101(y + t, H) * W + x]);
            dst[(size_t)y * W + x] = s;
        }
}
// cv::resize(INTER_LINEAR) on CV_64F: float coefficients, double accumulation (HResizeLinear<double,double,float>, VResizeLinear)
// scale_x, scale_y: 1 / fx, 1 / fy when the caller gave factors (cv::resize keeps them: inv_scale = fx), source / destination size otherwise
static void resize_linear_f64(const double *src, int W, int H, double *dst, int w, int h, double scale_x, double scale_y) {
    std::vector<int> xofs(w), yofs(h);
    std::vector<float> ax((size_t)w * 2), ay((size_t)h * 2);
    for (int dx = 0; dx < w; dx++) { float fx = (float)((dx + 0.5) * scale_x - 0.5); int sx = cvFloor(fx); fx -= sx; if (sx < 0) { fx = 0; sx = 0; } if (sx >= W - 1) { fx = 0; sx = W - 1; } xofs[dx] = sx; ax[dx * 2] = 1.f - fx; ax[dx * 2 + 1] = fx; }
    for (int dy = 0; dy < h; dy++) { float fy = (float)((dy + 0.5) * scale_y - 0.5); int sy = cvFloor(fy); fy -= sy; if (sy < 0) { fy = 0; sy = 0; } if (sy >= H - 1) { fy = 0; sy = H - 1; } yofs[dy] = sy; ay[dy * 2] = 1.f - fy; ay[dy * 2 + 1] = fy; }
    for (int dy = 0; dy < h; dy++) {
        const int sy0 = yofs[dy], sy1 = std::min(sy0 + 1, H - 1);
        for (int dx = 0; dx < w; dx++) {
            const int sx0 = xofs[dx], sx1 = std::min(sx0 + 1, W - 1);
            const double r0 = src[(size_t)sy0 * W + sx0] * ax[dx * 2] + src[(size_t)sy0 * W + sx1] * ax[dx * 2 + 1];
            const double r1 = src[(size_t)sy1 * W + sx0] * ax[dx * 2] + src[(size_t)sy1 * W + sx1] * ax[dx * 2 + 1];
            dst[(size_t)dy * w + dx] = r0 * ay[dy * 2] + r1 * ay[dy * 2 + 1];
        }
    }
}

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };
struct RegionPoint { int x, y; double angle, modgrad; };

struct LSD {
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
    const int N_BINS = 1024;
    int w = 0, h = 0;
    double LOG_NT = 0;
    std::vector<double> scaled, angles, modgrad;
    std::vector<uint8_t> used;
    std::vector<int> order; // pseudo-ordered coordinates (x + y*w)

    void prepare(const uint8_t *gray, int W, int H) { // flsd :440-462: GaussianBlur + resize, then ll_angle
        const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
        const int hk = (int)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        std::vector<double> img((size_t)W * H), blur((size_t)W * H);
        for (size_t i = 0; i < img.size(); i++) img[i] = gray[i]; // img.convertTo(image, CV_64FC1), :421
        gaussian_blur_f64(img.data(), W, H, 1 + 2 * hk, sigma, blur.data());
        w = cvRound(W * SCALE); h = cvRound(H * SCALE); // resize(gaussian_img, scaled_image, Size(), SCALE, SCALE)
        scaled.resize((size_t)w * h);
        resize_linear_f64(blur.data(), W, H, scaled.data(), w, h, 1. / SCALE, 1. / SCALE);
        ll_angle();
    }

    void ll_angle() { // :538-635
        const double prec = PI * ANG_TH / 180, threshold = QUANT / std::sin(prec);
        angles.assign((size_t)w * h, NOTDEF); modgrad.assign((size_t)w * h, 0.0);
        double max_grad = -1;
        for (int y = 0; y < h - 1; ++y)
            for (int addr = y * w, addr_end = addr + w - 1; addr < addr_end; ++addr) {
                double DA = scaled[addr + w + 1] - scaled[addr], BC = scaled[addr + 1] - scaled[addr + w];
                double gx = DA + BC, gy = DA - BC;
                double norm = std::sqrt((gx * gx + gy * gy) / 4);
                modgrad[addr] = norm;
                if (norm <= threshold) angles[addr] = NOTDEF;
                else { angles[addr] = fastAtan2(float(gx), float(-gy)) * DEG_TO_RADS; if (norm > max_grad) max_grad = norm; }
            }
        // Seed order.  ll_angle builds the 1024-bin pseudo-ordering as `next` links between the entries of `list` (:588-634), but flsd then
        // walks `list` BY INDEX (:477-480), and the entries were appended in raster order (:600-617): the seeds are visited in raster order
        // over x < w-1, y < h-1 and the gradient ordering has no effect (the trailing, never-filled entries are (0,0), used by then).
        // Found by running the reference's own lsd.cpp (oracle/_ref, tests/test_ref_pins.py); round 1 had restated the intended ordering.
        (void)max_grad;
        order.clear();
        for (int y = 0; y < h - 1; ++y)
            for (int x = 0; x < w - 1; ++x) order.push_back(x + y * w);
    }

    inline bool isAligned(int address, double theta, double prec) const { // :1138-1154
        if (address < 0) return false;
        const double a = angles[address];
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI_) { n_theta -= M_2__PI_; if (n_theta < 0) n_theta = -n_theta; }
        return n_theta <= prec;
    }
    static double angle_diff_signed(double a, double b) { double diff = a - b; while (diff <= -PI) diff += M_2__PI_; while (diff > PI) diff -= M_2__PI_; return diff; }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)); }

    void region_grow(int sx, int sy, std::vector<RegionPoint> &reg, int &reg_size, double &reg_angle, double prec) { // :637-688
        reg_size = 1;
        int addr = sx + sy * w;
        reg[0] = RegionPoint{sx, sy, angles[addr], modgrad[addr]};
        reg_angle = angles[addr];
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        used[addr] = 1;
        for (int i = 0; i < reg_size; ++i) {
            const int px = reg[i].x, py = reg[i].y;
            int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1), yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy) {
                int c_addr = xx_min + yy * w;
                for (int xx = xx_min; xx <= xx_max; ++xx, ++c_addr)
                    if ((used[c_addr] != 1) && isAligned(c_addr, reg_angle, prec)) {
                        used[c_addr] = 1;
                        const double angle = angles[c_addr];
                        reg[reg_size] = RegionPoint{xx, yy, angle, modgrad[c_addr]};
                        ++reg_size;
                        sumdx += std::cos(float(angle));
                        sumdy += std::sin(float(angle));
                        reg_angle = fastAtan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
            }
        }
    }
    static bool double_equal(double a, double b) { // :104-118
        if (a == b) return true;
        double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b), abs_max = (aa > bb) ? aa : bb;
        if (abs_max < DBL_MIN) abs_max = DBL_MIN;
        return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);
    }
    double get_theta(const std::vector<RegionPoint> &reg, int reg_size, double x, double y, double reg_angle, double prec) const { // :748-784
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (int i = 0; i < reg_size; ++i) {
            const double dx = (double)reg[i].x - x, dy = (double)reg[i].y - y, wgt = reg[i].modgrad;
            Ixx += dy * dy * wgt; Iyy += dx * dx * wgt; Ixy -= dx * dy * wgt;
        }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fastAtan2(float(lambda - Ixx), float(Ixy))) : double(fastAtan2(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (std::fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += PI;
        return theta;
    }
    void region2rect(const std::vector<RegionPoint> &reg, int reg_size, double reg_angle, double prec, double p, Rect &rec) const { // :690-746
        double x = 0, y = 0, sum = 0;
        for (int i = 0; i < reg_size; ++i) { const double wgt = reg[i].modgrad; x += double(reg[i].x) * wgt; y += double(reg[i].y) * wgt; sum += wgt; }
        x /= sum; y /= sum;
        const double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
        const double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (int i = 0; i < reg_size; ++i) {
            const double rdx = double(reg[i].x) - x, rdy = double(reg[i].y) - y;
            const double l = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (ww > w_max) w_max = ww; else if (ww < w_min) w_min = ww;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }
    bool reduce_region_radius(std::vector<RegionPoint> &reg, int &reg_size, double reg_angle, double prec, double p, Rect &rec, double density, double density_th) { // :834-871
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        auto dsq = [](double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); };
        const double r1 = dsq(xc, yc, rec.x1, rec.y1), r2 = dsq(xc, yc, rec.x2, rec.y2);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (int i = 0; i < reg_size; ++i)
                if (dsq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    used[reg[i].x + reg[i].y * w] = 0;
                    std::swap(reg[i], reg[reg_size - 1]);
                    --reg_size; --i;
                }
            if (reg_size < 2) return false;
            region2rect(reg, reg_size, reg_angle, prec, p, rec);
            density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }
    bool refine(std::vector<RegionPoint> &reg, int &reg_size, double reg_angle, double prec, double p, Rect &rec, double density_th) { // :786-832
        double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
        double sum = 0, s_sum = 0;
        int n = 0;
        for (int i = 0; i < reg_size; ++i) {
            used[reg[i].x + reg[i].y * w] = 0;
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) { const double ang_d = angle_diff_signed(reg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
        }
        const double mean_angle = sum / double(n);
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        region_grow(reg[0].x, reg[0].y, reg, reg_size, reg_angle, tau);
        if (reg_size < 2) return false;
        region2rect(reg, reg_size, reg_angle, prec, p, rec);
        density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_region_radius(reg, reg_size, reg_angle, prec, p, rec, density, density_th);
        return true;
    }
    static double log_gamma(double x) { // :70,124-160
        if (x > 15.0) return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), b = 0;
        for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
        return a + std::log(b);
    }
    double nfa(int n, int k, double p) const { // :1100-1136
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * std::log10(p);
        const double p_term = p / (1 - p);
        const double log1term = (double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) + double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
        double term = std::exp(log1term);
        if (double_equal(term, 0)) { if (k > n * p) return -log1term / M_LN10 - LOG_NT; else return -LOG_NT; }
        double bin_tail = term;
        const double tolerance = 0.1;
        for (int i = k + 1; i <= n; ++i) {
            const double bin_term = double(n - i + 1) / double(i), mult_term = bin_term * p_term;
            term *= mult_term;
            bin_tail += term;
            if (bin_term < 1) {
                const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
                if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
            }
        }
        return -std::log10(bin_tail) - LOG_NT;
    }
    double rect_nfa(const Rect &rec) const { // :977-1098
        int total_pts = 0, alg_pts = 0;
        const double half_width = rec.width / 2.0, dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        struct E { int x, y; bool taken; } o[4];
        o[0] = E{int(rec.x1 - dyhw), int(rec.y1 + dxhw), false}; o[1] = E{int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
        o[2] = E{int(rec.x2 + dyhw), int(rec.y2 - dxhw), false}; o[3] = E{int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
        std::sort(o, o + 4, [](const E &a, const E &b) { if (a.x == b.x) return a.y < b.y; return a.x < b.x; });
        E *min_y = &o[0], *max_y = &o[0];
        for (int i = 1; i < 4; ++i) { if (min_y->y > o[i].y) min_y = &o[i]; if (max_y->y < o[i].y) max_y = &o[i]; }
        min_y->taken = true;
        E *leftmost = 0;
        for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!leftmost) leftmost = &o[i]; else if (leftmost->x > o[i].x) leftmost = &o[i]; }
        leftmost->taken = true;
        E *rightmost = 0;
        for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!rightmost) rightmost = &o[i]; else if (rightmost->x < o[i].x) rightmost = &o[i]; }
        rightmost->taken = true;
        E *tailp = 0;
        for (int i = 0; i < 4; ++i) if (!o[i].taken) { if (!tailp) tailp = &o[i]; else if (tailp->x > o[i].x) tailp = &o[i]; }
        tailp->taken = true;
        // integer divisions and the tailp->x comparisons are the reference's (:1057-1065)
        const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
        const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
        const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
        const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
        double lstep = flstep, rstep = frstep, left_x = min_y->x, right_x = min_y->x;
        for (int y = min_y->y; y <= max_y->y; ++y) {
            if (y < 0 || y >= h) continue;
            int adx = y * w + int(left_x);
            for (int x = int(left_x); x <= int(right_x); ++x, ++adx) {
                if (x < 0 || x >= w) continue;
                ++total_pts;
                if (isAligned(adx, rec.theta, rec.prec)) ++alg_pts;
            }
            if (y >= leftmost->y) lstep = slstep;
            if (y >= rightmost->y) rstep = srstep;
            left_x += lstep; right_x += rstep;
        }
        return nfa(total_pts, alg_pts, rec.p);
    }
    double rect_improve(Rect &rec) const { // :873-975
        const double delta = 0.5, delta_2 = delta / 2.0;
        double log_nfa = rect_nfa(rec);
        if (log_nfa > LOG_EPS) return log_nfa;
        Rect r = rec;
        for (int n = 0; n < 5; ++n) { r.p /= 2; r.prec = r.p * PI; double v = rect_nfa(r); if (v > log_nfa) { log_nfa = v; rec = r; } }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.width -= delta; double v = rect_nfa(r); if (v > log_nfa) { rec = r; log_nfa = v; } }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; r.width -= delta;
            double v = rect_nfa(r); if (v > log_nfa) { rec = r; log_nfa = v; } }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; r.width -= delta;
            double v = rect_nfa(r); if (v > log_nfa) { rec = r; log_nfa = v; } }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) if ((r.width - delta) >= 0.5) { r.p /= 2; r.prec = r.p * PI; double v = rect_nfa(r); if (v > log_nfa) { rec = r; log_nfa = v; } }
        return log_nfa;
    }
    void detect(std::vector<float> &lines) { // flsd :464-535 with LSD_REFINE_ADV
        const double prec = PI * ANG_TH / 180, p = ANG_TH / 180;
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        const int min_reg_size = int(-LOG_NT / std::log10(p));
        used.assign((size_t)w * h, 0);
        std::vector<RegionPoint> reg((size_t)w * h);
        for (size_t i = 0; i < order.size(); ++i) {
            const int adx = order[i];
            if (used[adx] == 0 && angles[adx] != NOTDEF) {
                int reg_size; double reg_angle;
                region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
                if (reg_size < min_reg_size) continue;
                Rect rec;
                region2rect(reg, reg_size, reg_angle, prec, p, rec);
                if (!refine(reg, reg_size, reg_angle, prec, p, rec, DENSITY_TH)) continue;
                const double log_nfa = rect_improve(rec);
                if (log_nfa <= LOG_EPS) continue;
                rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
                rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
                lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1)); lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
            }
        }
    }
};

static int keylines_from(const std::vector<float> &lines, int W, int H, std::vector<orc_keyline> &out) { // LSDDetector.cpp:75-101,205-263 (one octave, scale 1)
    const float pre_boundary_thre = 10;
    int class_counter = -1;
    for (size_t k = 0; k + 3 < lines.size(); k += 4) {
        float e[4] = {lines[k], lines[k + 1], lines[k + 2], lines[k + 3]};
        for (int q = 0; q < 4; q++) { // checkLineExtremes
            const int lim = (q & 1) ? H : W;
            if (e[q] < 0) e[q] = 0;
            if (e[q] >= lim) e[q] = (float)lim - 1.0f;
        }
        const float octaveScale = std::pow((float)1, 0);
        orc_keyline kl;
        std::memset(&kl, 0, sizeof(kl));
        kl.startPointX = e[0] * octaveScale; kl.startPointY = e[1] * octaveScale; kl.endPointX = e[2] * octaveScale; kl.endPointY = e[3] * octaveScale;
        if (((kl.startPointX < pre_boundary_thre) && (kl.endPointX < pre_boundary_thre)) || ((kl.startPointX > W - pre_boundary_thre) && (kl.endPointX > W - pre_boundary_thre)) ||
            ((kl.startPointY < pre_boundary_thre) && (kl.endPointY < pre_boundary_thre)) || ((kl.startPointY > H - pre_boundary_thre) && (kl.endPointY > H - pre_boundary_thre)))
            continue;
        kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
        kl.lineLength = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
        { // cv::LineIterator(img, Point2f -> Point (cvRound), ...).count, 8-connected, end points inside the image
            const int x1 = cvRound(e[0]), y1 = cvRound(e[1]), x2 = cvRound(e[2]), y2 = cvRound(e[3]);
            kl.numOfPixels = std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1;
        }
        kl.angle = std::atan2((kl.endPointY - kl.startPointY), (kl.endPointX - kl.startPointX));
        kl.class_id = ++class_counter;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / std::max(W, H);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2; kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        out.push_back(kl);
    }
    return (int)out.size();
}
} // namespace

extern "C" {

int orc_lsd_detect(const uint8_t *gray, int W, int H, orc_keyline *out, int cap) {
    LSD l;
    l.prepare(gray, W, H);
    std::vector<float> lines;
    l.detect(lines);
    std::vector<orc_keyline> kls;
    keylines_from(lines, W, H, kls);
    for (size_t i = 0; i < kls.size() && (int)i < cap; i++) out[i] = kls[i];
    return (int)kls.size();
}
int orc_lsd_detect_filter_lines(const uint8_t *gray, int W, int H, float length_thres, float *lines, int cap) {
    LSD l;
    l.prepare(gray, W, H);
    std::vector<float> raw;
    l.detect(raw);
    std::vector<orc_keyline> kls;
    keylines_from(raw, W, H, kls);
    int n = 0;
    for (const orc_keyline &k : kls) // filter_lines :200-207, keylines_to_mat :26-36 (scale 1)
        if (k.octave == 0 && k.lineLength > length_thres) {
            if (n < cap) { lines[n * 4] = k.startPointX * 1.f; lines[n * 4 + 1] = k.startPointY * 1.f; lines[n * 4 + 2] = k.endPointX * 1.f; lines[n * 4 + 3] = k.endPointY * 1.f; }
            n++;
        }
    return n;
}
int orc_lsd_maps(const uint8_t *gray, int W, int H, int *sw, int *sh, double *scaled, double *modgrad, double *angles, int *order, int *n_order) {
    LSD l;
    l.prepare(gray, W, H);
    if (sw) *sw = l.w;
    if (sh) *sh = l.h;
    const size_t n = (size_t)l.w * l.h;
    if (scaled) std::memcpy(scaled, l.scaled.data(), n * sizeof(double));
    if (modgrad) std::memcpy(modgrad, l.modgrad.data(), n * sizeof(double));
    if (angles) std::memcpy(angles, l.angles.data(), n * sizeof(double));
    if (order) std::memcpy(order, l.order.data(), l.order.size() * sizeof(int));
    if (n_order) *n_order = (int)l.order.size();
    return 0;
}

} // extern "C"

// the OpenCV restatements above, for oracle/ref_shim (cv_prims.h)
namespace orc_cv {
void gaussian_blur_f64(const double *src, int w, int h, int ksize, double sigma, double *dst) { ::gaussian_blur_f64(src, w, h, ksize, sigma, dst); }
void resize_linear_f64(const double *src, int sw, int sh, double *dst, int dw, int dh, double scale_x, double scale_y) { ::resize_linear_f64(src, sw, sh, dst, dw, dh, scale_x, scale_y); }
}
