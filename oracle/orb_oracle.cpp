/*
 * oracle/orb_oracle.cpp -- CPU oracle for ORB_SLAM2::ORBextractor.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PINNED: tests/test_ref_pins.py runs the reference's own ORBextractor.cc, compiled whole from /root/reference
 * (oracle/_ref) against a stand-in for the OpenCV headers, on the same images -- key points, descriptors and pyramid levels identical bit for bit; the
 * OpenCV primitives under it (FAST, resize, GaussianBlur, fastAtan2) stay restated from their published algorithms.  Restated from
 * /root/reference/orb_object_slam/src/ORBextractor.cc and include/ORBextractor.h, plus the OpenCV semantics its call
 * sites reach.  OpenCV version pinned by assumption to the 2.4 / 3.0-3.3 family (the reference's prebuilt examples link
 * libopencv 2.4): cv::FAST (FAST-9/16 + cornerScore + strict 3x3 NMS), cv::resize(INTER_LINEAR) 8-bit fixed point
 * (11-bit coefficients), cv::GaussianBlur 8-bit (float kernel scaled to 8-bit fixed point per pass, (v+2^15)>>16),
 * cv::fastAtan2 (degree polynomial), cvRound (round half to even).
 * Deliberate pins:
 *  O1 DistributeOctTree sorts pair<int, ExtractorNode*> (ORBextractor.cc:685): ties are broken by heap address in the
 *     reference; here by node creation order.
 *  O2 computeOrbDescriptor reads the blurred *clone* (no border) up to 18 px from a keypoint that may sit 16 px from the
 *     level edge (:1078-1083,:104-150): unchecked addressing.  We replicate the address arithmetic (y*cols+x) and clamp
 *     it to the buffer.
 *  O3 (float)cos / (float)sin of the orientation are taken as the correctly rounded float values (evaluated in double
 *     with a fixed polynomial, so that CPU and GPU agree bit for bit).
 */
#include "oracle.h"
#include "cv_prims.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <vector>

namespace {

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19; // ORBextractor.cc:70-72

static const int bit_pattern_31_[256 * 4] = {
#include "orb_pattern.inc"
};

static inline int cvRound(double v) { return (int)std::lrint(v); }  // default rounding mode: half to even
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2 (OpenCV core mathfuncs, float polynomial, degrees)
static float fastAtan2(float y, float x) {
    const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// O3: sin/cos of a float angle (radians, [0, 2pi]) evaluated in double with +,-,*,/ only, rounded to float.
static void sincos_f(float angle, float *s_out, float *c_out) {
    const double x = (double)angle;
    const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HI = 1.57079632679489655800, PIO2_LO = 6.12323399573676603587e-17;
    const double kd = std::floor(x * TWO_OVER_PI + 0.5);
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
    *s_out = (float)ss;
    *c_out = (float)cc;
}

struct KP { float x, y, response; };

// cv::FAST(img, keypoints, threshold, nonmax=true), 16-pixel ring, 9 contiguous (OpenCV features2d/src/fast.cpp FAST_t<16>)
static const int RING[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int cornerScore16(const uint8_t *ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]); a = std::min(a, (int)d[k + 5]); a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]); a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]); b = std::max(b, (int)d[k + 4]); b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]); b = std::max(b, (int)d[k + 7]); b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    threshold = -b0 - 1;
    return threshold;
}

static void FAST16(const uint8_t *img, int step, int cols, int rows, int threshold, std::vector<KP> &keypoints) {
    const int K = 8, N = 16 + K + 1;
    int i, j, k, pixel[25];
    for (k = 0; k < 16; k++) pixel[k] = RING[k][0] + RING[k][1] * step;
    for (; k < 25; k++) pixel[k] = pixel[k - 16];
    keypoints.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (cols < 7 || rows < 7) return;
    std::vector<uint8_t> sbuf((size_t)cols * 3, 0);
    std::vector<int> cbuf((size_t)(cols + 1) * 3, 0);
    uint8_t *buf[3] = {sbuf.data(), sbuf.data() + cols, sbuf.data() + 2 * cols};
    int *cpbuf[3] = {cbuf.data() + 1, cbuf.data() + 1 + (cols + 1), cbuf.data() + 1 + 2 * (cols + 1)};
    for (i = 3; i < rows - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * step + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3];
        std::memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold); break; }
                        } else count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold); break; }
                        } else count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                keypoints.push_back(KP{(float)j, (float)(i - 1), (float)score});
        }
    }
}

// cv::resize(src, dst, INTER_LINEAR) for CV_8UC1 (imgproc/src/imgwarp.cpp: HResizeLinear<uchar,int,short,2048>, VResizeLinear 8u)
static void resize_linear_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha((size_t)dw * 2), ibeta((size_t)dh * 2);
    auto sat_short = [](float v) { int i = cvRound(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = sat_short(fy * 2048);
    }
    std::vector<int> r0(dw), r1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = std::min(sy0 + 1, sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sw, *S1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
            r0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            r1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        for (int dx = 0; dx < dw; dx++)
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

static inline int reflect101(int p, int len) { // BORDER_REFLECT_101
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

// cv::getGaussianKernel(7, 2, CV_32F) scaled to 8-bit fixed point (createSeparableLinearFilter, bits = 8)
static void gauss7_fixed(int k[7]) {
    float cf[7];
    double sigma = 2, scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); k[i] = cvRound(cf[i] * 256.f); }
}
static void gaussian_blur7_u8(const uint8_t *src, int w, int h, uint8_t *dst) {
    int k[7];
    gauss7_fixed(k);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += src[(size_t)y * w + reflect101(x + t, w)] * k[t + 3];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += tmp[(size_t)reflect101(y + t, h) * w + x] * k[t + 3];
            int v = (s + (1 << 15)) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}

struct Node { // ExtractorNode, ORBextractor.h:40-52
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::vector<int> keys; // indices into the candidate array (order preserved)
    bool bNoMore = false;
    long seq = 0;          // creation order (O1)
    std::list<Node>::iterator lit;
};

} // namespace

struct orc_orb {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    float scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<std::vector<uint8_t>> pyr, blur;
    std::vector<int> lw, lh;
    std::vector<std::vector<KP>> cands;
    long node_seq = 0;

    void divide(const Node &n, Node &n1, Node &n2, Node &n3, Node &n4, const std::vector<KP> &K) { // :483-538
        const int halfX = (int)std::ceil(static_cast<float>(n.URx - n.ULx) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(n.BRy - n.ULy) / 2);
        n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy; n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
        n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy; n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
        n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy; n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
        n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy; n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
        for (int id : n.keys) {
            const KP &kp = K[id];
            if (kp.x < n1.URx) { if (kp.y < n1.BRy) n1.keys.push_back(id); else n3.keys.push_back(id); }
            else if (kp.y < n1.BRy) n2.keys.push_back(id);
            else n4.keys.push_back(id);
        }
        if (n1.keys.size() == 1) n1.bNoMore = true;
        if (n2.keys.size() == 1) n2.bNoMore = true;
        if (n3.keys.size() == 1) n3.bNoMore = true;
        if (n4.keys.size() == 1) n4.bNoMore = true;
    }

    std::vector<KP> distribute(const std::vector<KP> &K, int minX, int maxX, int minY, int maxY, int N) { // :540-763
        const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
        std::vector<KP> res;
        if (nIni < 1) return res;
        const float hX = static_cast<float>(maxX - minX) / nIni;
        std::list<Node> lNodes;
        std::vector<Node *> ini(nIni);
        for (int i = 0; i < nIni; i++) {
            Node ni;
            ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
            ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
            ni.BLx = ni.ULx; ni.BLy = maxY - minY;
            ni.BRx = ni.URx; ni.BRy = maxY - minY;
            ni.seq = node_seq++;
            lNodes.push_back(ni);
            ini[i] = &lNodes.back();
        }
        for (size_t i = 0; i < K.size(); i++) {
            int idx = (int)(K[i].x / hX);
            if (idx >= nIni) idx = nIni - 1;
            ini[idx]->keys.push_back((int)i);
        }
        auto lit = lNodes.begin();
        while (lit != lNodes.end()) {
            if (lit->keys.size() == 1) { lit->bNoMore = true; lit++; }
            else if (lit->keys.empty()) lit = lNodes.erase(lit);
            else lit++;
        }
        bool bFinish = false;
        std::vector<std::pair<int, Node *>> vSize;
        auto push_children = [&](Node *c[4], int &nToExpand) {
            for (int q = 0; q < 4; q++)
                if (c[q]->keys.size() > 0) {
                    c[q]->seq = node_seq++;
                    lNodes.push_front(*c[q]);
                    if (c[q]->keys.size() > 1) {
                        nToExpand++;
                        vSize.push_back(std::make_pair((int)c[q]->keys.size(), &lNodes.front()));
                        lNodes.front().lit = lNodes.begin();
                    }
                }
        };
        auto by_size_then_creation = [](const std::pair<int, Node *> &a, const std::pair<int, Node *> &b) {
            if (a.first != b.first) return a.first < b.first;
            return a.second->seq < b.second->seq; // O1
        };
        while (!bFinish) {
            int prevSize = (int)lNodes.size();
            lit = lNodes.begin();
            int nToExpand = 0;
            vSize.clear();
            while (lit != lNodes.end()) {
                if (lit->bNoMore) { lit++; continue; }
                Node n1, n2, n3, n4;
                divide(*lit, n1, n2, n3, n4, K);
                Node *c[4] = {&n1, &n2, &n3, &n4};
                push_children(c, nToExpand);
                lit = lNodes.erase(lit);
            }
            if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            else if (((int)lNodes.size() + nToExpand * 3) > N) {
                while (!bFinish) {
                    prevSize = (int)lNodes.size();
                    std::vector<std::pair<int, Node *>> vPrev = vSize;
                    vSize.clear();
                    std::sort(vPrev.begin(), vPrev.end(), by_size_then_creation);
                    for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                        Node n1, n2, n3, n4;
                        divide(*vPrev[j].second, n1, n2, n3, n4, K);
                        Node *c[4] = {&n1, &n2, &n3, &n4};
                        int dummy = 0;
                        push_children(c, dummy);
                        lNodes.erase(vPrev[j].second->lit);
                        if ((int)lNodes.size() >= N) break;
                    }
                    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
                }
            }
        }
        res.reserve(nfeatures);
        for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
            const std::vector<int> &v = it->keys;
            int best = v[0];
            float maxResponse = K[best].response;
            for (size_t k = 1; k < v.size(); k++)
                if (K[v[k]].response > maxResponse) { best = v[k]; maxResponse = K[v[k]].response; }
            res.push_back(K[best]);
        }
        return res;
    }
};

extern "C" {

orc_orb *orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) { // :412-471
    orc_orb *e = new orc_orb();
    e->nfeatures = nfeatures; e->scaleFactor = scaleFactor; e->nlevels = nlevels; e->iniThFAST = iniThFAST; e->minThFAST = minThFAST;
    e->mvScaleFactor.resize(nlevels); e->mvInvScaleFactor.resize(nlevels);
    e->mvScaleFactor[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * scaleFactor;
    for (int i = 0; i < nlevels; i++) e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
    e->mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cvRound(nDesired);
        sum += e->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    e->umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cvRound(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}
void orc_orb_destroy(orc_orb *e) { delete e; }

int orc_orb_extract(orc_orb *e, const uint8_t *gray, int W, int H, orc_keypoint *out_kps, uint8_t *out_desc, int cap) {
    const int nlevels = e->nlevels;
    e->pyr.assign(nlevels, {}); e->blur.assign(nlevels, {}); e->lw.assign(nlevels, 0); e->lh.assign(nlevels, 0);
    e->cands.assign(nlevels, {});
    // ComputePyramid :1101-1125
    for (int level = 0; level < nlevels; ++level) {
        float scale = e->mvInvScaleFactor[level];
        int w = cvRound((float)W * scale), h = cvRound((float)H * scale);
        e->lw[level] = w; e->lh[level] = h;
        e->pyr[level].resize((size_t)w * h);
        if (level != 0) resize_linear_u8(e->pyr[level - 1].data(), e->lw[level - 1], e->lh[level - 1], e->pyr[level].data(), w, h);
        else std::memcpy(e->pyr[0].data(), gray, (size_t)W * H);
    }
    // ComputeKeyPointsOctTree :766-853
    std::vector<std::vector<orc_keypoint>> all(nlevels);
    const float Wc = 30;
    for (int level = 0; level < nlevels; ++level) {
        const int cols = e->lw[level], rows = e->lh[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
        std::vector<KP> &vToDistributeKeys = e->cands[level];
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / Wc), nRows = (int)(height / Wc);
        if (nCols < 1 || nRows < 1) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        std::vector<KP> vKeysCell;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0; // rowRange/colRange take ints
                const uint8_t *view = e->pyr[level].data() + (size_t)y0 * cols + x0;
                FAST16(view, cols, cw, ch, e->iniThFAST, vKeysCell);
                if (vKeysCell.empty()) FAST16(view, cols, cw, ch, e->minThFAST, vKeysCell);
                for (KP kp : vKeysCell) {
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    vToDistributeKeys.push_back(kp);
                }
            }
        }
        std::vector<KP> sel = e->distribute(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, e->mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
        for (const KP &k : sel) {
            orc_keypoint kp;
            kp.x = k.x + minBorderX; kp.y = k.y + minBorderY; kp.response = k.response;
            kp.octave = level; kp.size = (float)scaledPatchSize; kp.angle = -1; kp.class_id = -1;
            all[level].push_back(kp);
        }
    }
    // computeOrientation / IC_Angle :74-101
    for (int level = 0; level < nlevels; ++level) {
        const int step = e->lw[level];
        for (orc_keypoint &kp : all[level]) {
            int m_01 = 0, m_10 = 0;
            const uint8_t *center = e->pyr[level].data() + (size_t)cvRound(kp.y) * step + cvRound(kp.x);
            for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
            for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
                int v_sum = 0, d = e->umax[v];
                for (int u = -d; u <= d; ++u) {
                    int val_plus = center[u + v * step], val_minus = center[u - v * step];
                    v_sum += (val_plus - val_minus);
                    m_10 += u * (val_plus + val_minus);
                }
                m_01 += v * v_sum;
            }
            kp.angle = fastAtan2((float)m_01, (float)m_10);
        }
    }
    // descriptors :1069-1098
    int n = 0;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    for (int level = 0; level < nlevels; ++level) {
        std::vector<orc_keypoint> &kps = all[level];
        if (kps.empty()) continue;
        const int w = e->lw[level], h = e->lh[level];
        e->blur[level].resize((size_t)w * h);
        gaussian_blur7_u8(e->pyr[level].data(), w, h, e->blur[level].data());
        const uint8_t *img = e->blur[level].data();
        const long last = (long)w * h - 1;
        for (orc_keypoint &kp : kps) {
            if (n >= cap) break;
            float angle = (float)kp.angle * factorPI, a, b;
            sincos_f(angle, &b, &a);
            const long cidx = (long)cvRound(kp.y) * w + cvRound(kp.x);
            const int *pattern = bit_pattern_31_;
            uint8_t *desc = out_desc + (size_t)n * 32;
            auto get = [&](int idx) {
                float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
                long o = cidx + (long)cvRound(px * b + py * a) * w + cvRound(px * a - py * b);
                o = o < 0 ? 0 : (o > last ? last : o); // O2
                return (int)img[o];
            };
            for (int i = 0; i < 32; ++i, pattern += 32) {
                int val = 0;
                for (int t = 0; t < 8; t++) val |= (get(2 * t) < get(2 * t + 1)) << t;
                desc[i] = (uint8_t)val;
            }
            orc_keypoint o = kp;
            if (level != 0) { float scale = e->mvScaleFactor[level]; o.x *= scale; o.y *= scale; }
            out_kps[n++] = o;
        }
    }
    return n;
}

int orc_orb_features_per_level(orc_orb *e, int *out) { for (int i = 0; i < e->nlevels; i++) out[i] = e->mnFeaturesPerLevel[i]; return e->nlevels; }
int orc_orb_level_dims(orc_orb *e, int level, int *w, int *h) { *w = e->lw[level]; *h = e->lh[level]; return 0; }
int orc_orb_get_level(orc_orb *e, int level, int blurred, uint8_t *out) {
    const std::vector<uint8_t> &v = blurred ? e->blur[level] : e->pyr[level];
    if (v.empty()) return -1;
    std::memcpy(out, v.data(), v.size());
    return 0;
}
int orc_orb_get_candidates(orc_orb *e, int level, float *xyr, int cap) {
    int n = (int)e->cands[level].size();
    for (int i = 0; i < n && i < cap; i++) { xyr[i * 3] = e->cands[level][i].x; xyr[i * 3 + 1] = e->cands[level][i].y; xyr[i * 3 + 2] = e->cands[level][i].response; }
    return n;
}
int orc_fast(const uint8_t *img, int stride, int w, int h, int threshold, float *xyr, int cap) {
    std::vector<KP> k;
    FAST16(img, stride, w, h, threshold, k);
    for (size_t i = 0; i < k.size() && (int)i < cap; i++) { xyr[i * 3] = k[i].x; xyr[i * 3 + 1] = k[i].y; xyr[i * 3 + 2] = k[i].response; }
    return (int)k.size();
}
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void orc_sincos_f(float angle_rad, float *s, float *c) { sincos_f(angle_rad, s, c); }

} // extern "C"

// the OpenCV restatements above, for oracle/ref_shim (cv_prims.h)
namespace orc_cv {
void resize_linear_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) { ::resize_linear_u8(src, sw, sh, dst, dw, dh); }
void gaussian_blur7_u8(const uint8_t *src, int w, int h, uint8_t *dst) { ::gaussian_blur7_u8(src, w, h, dst); }
int reflect101(int p, int len) { return ::reflect101(p, len); }
}

