/*
 * oracle/lbd_oracle.cpp -- CPU oracle for the LBD line descriptor of line_lbd.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned: tests/test_ref_pins.py::test_lbd_descriptor_equals_reference compares its descriptors byte for
 * byte with the reference's own compute path (oracle/_ref, cut out of binary_descriptor.cpp at build time); the two OpenCV primitives under it
 * (GaussianBlur, Sobel) stay restated.  Restated from /root/reference/line_lbd/libs/binary_descriptor.cpp
 * (constructor weights :218-260, computeGaussianPyramid :352-370, computeSobel :373-402, binaryConversion :405-416,
 * computeImpl :603-790, computeLBD :1146-1509).  OpenCV semantics assumed: GaussianBlur 5x5 sigma 1 on u8 in 8-bit fixed point
 * (as for ORB, see orb_oracle.cpp), Sobel 3x3 -> CV_16S with BORDER_REFLECT_101.  cos/sin of the float line direction are
 * the correctly rounded float values; sqrt and 1/sqrt are evaluated in float (std::sqrt(float) overloads).
 */
#include "oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

extern "C" void orc_sincos_f(float angle_rad, float *s, float *c);

namespace {
const int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7;
static const int combinations[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
                                        {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};
static inline int reflect101(int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; } return p; }

static void blur5_sobel(const uint8_t *gray, int W, int H, std::vector<uint8_t> &blur, std::vector<int16_t> &dx, std::vector<int16_t> &dy) {
    int k[5];
    { float cf[5]; double sum = 0, s2 = -0.5 / (1.0 * 1.0); for (int i = 0; i < 5; i++) { double x = i - 2.0; cf[i] = (float)std::exp(s2 * x * x); sum += cf[i]; } sum = 1. / sum;
      for (int i = 0; i < 5; i++) { cf[i] = (float)(cf[i] * sum); k[i] = (int)std::lrint(cf[i] * 256.f); } }
    std::vector<int> tmp((size_t)W * H);
    blur.resize((size_t)W * H); dx.resize((size_t)W * H); dy.resize((size_t)W * H);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { int s = 0; for (int t = -2; t <= 2; t++) s += gray[(size_t)y * W + reflect101(x + t, W)] * k[t + 2]; tmp[(size_t)y * W + x] = s; }
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int s = 0; for (int t = -2; t <= 2; t++) s += tmp[(size_t)reflect101(y + t, H) * W + x] * k[t + 2];
        int v = (s + (1 << 15)) >> 16; blur[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    auto P = [&](int x, int y) { return (int)blur[(size_t)reflect101(y, H) * W + reflect101(x, W)]; };
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        dx[(size_t)y * W + x] = (int16_t)((P(x + 1, y - 1) + 2 * P(x + 1, y) + P(x + 1, y + 1)) - (P(x - 1, y - 1) + 2 * P(x - 1, y) + P(x - 1, y + 1)));
        dy[(size_t)y * W + x] = (int16_t)((P(x - 1, y + 1) + 2 * P(x, y + 1) + P(x + 1, y + 1)) - (P(x - 1, y - 1) + 2 * P(x, y - 1) + P(x + 1, y - 1))); }
}
} // namespace

extern "C" {

int orc_lbd_maps(const uint8_t *gray, int W, int H, uint8_t *blur, int16_t *dx, int16_t *dy) {
    std::vector<uint8_t> b; std::vector<int16_t> x, y;
    blur5_sobel(gray, W, H, b, x, y);
    if (blur) std::memcpy(blur, b.data(), b.size());
    if (dx) std::memcpy(dx, x.data(), x.size() * 2);
    if (dy) std::memcpy(dy, y.data(), y.size() * 2);
    return 0;
}

int orc_lbd_compute(const uint8_t *gray, int W, int H, const orc_keyline *kls, int n, uint8_t *desc, float *fdesc) {
    std::vector<uint8_t> blur; std::vector<int16_t> dxImg, dyImg;
    blur5_sobel(gray, W, H, blur, dxImg, dyImg);
    // constructor :218-260 (integer divisions intended)
    float gaussCoefL[WIDTH_OF_BAND * 3], gaussCoefG[NUM_OF_BANDS * WIDTH_OF_BAND];
    { double u = (WIDTH_OF_BAND * 3 - 1) / 2, sigma = (WIDTH_OF_BAND * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
      for (int i = 0; i < WIDTH_OF_BAND * 3; i++) { double dis = i - u; gaussCoefL[i] = (float)std::exp(dis * dis * inv); }
      u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
      for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; i++) { double dis = i - u; gaussCoefG[i] = (float)std::exp(dis * dis * inv); } }
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS), halfHeight = (heightOfLSP - 1) / 2;
    const short realWidth = (short)W, imageWidth = realWidth - 1, imageHeight = (short)(H - 1);
    for (int li = 0; li < n; li++) {
        const orc_keyline &L = kls[li];
        float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0}, pgdL2BandSum[NUM_OF_BANDS] = {0}, ngdL2BandSum[NUM_OF_BANDS] = {0};
        float pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0}, pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
        const short lengthOfLSP = (short)L.numOfPixels, halfWidth = (lengthOfLSP - 1) / 2;
        const float lineMiddlePointX = (float)(0.5 * (L.sPointInOctaveX + L.ePointInOctaveX)), lineMiddlePointY = (float)(0.5 * (L.sPointInOctaveY + L.ePointInOctaveY));
        float dL[2], dO[2];
        orc_sincos_f(L.angle, &dL[1], &dL[0]); // direction = kl.angle
        dO[0] = -dL[1]; dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; hID++) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; wID++) {
                short tempCor = (short)std::round((double)sCorX);
                short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)std::round((double)sCorY);
                short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                const short dx = dxImg[yCor * realWidth + xCor], dy = dyImg[yCor * realWidth + xCor];
                const float gDL = dx * dL[0] + dy * dL[1], gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0]; sCorY += dL[1];
            }
            sCorX0 -= dL[1]; sCorY0 += dL[0];
            float coef = gaussCoefG[hID];
            pgdLRowSum = coef * pgdLRowSum; ngdLRowSum = coef * ngdLRowSum;
            const float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coef * pgdORowSum; ngdORowSum = coef * ngdORowSum;
            const float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
            auto add = [&](short band, float c) {
                pgdLBandSum[band] += c * pgdLRowSum; ngdLBandSum[band] += c * ngdLRowSum;
                pgdL2BandSum[band] += c * c * pgdL2RowSum; ngdL2BandSum[band] += c * c * ngdL2RowSum;
                pgdOBandSum[band] += c * pgdORowSum; ngdOBandSum[band] += c * ngdORowSum;
                pgdO2BandSum[band] += c * c * pgdO2RowSum; ngdO2BandSum[band] += c * c * ngdO2RowSum;
            };
            short bandID = (short)(hID / WIDTH_OF_BAND);
            add(bandID, gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND]);
            bandID--;
            if (bandID >= 0) add(bandID, gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND]);
            bandID = bandID + 2;
            if (bandID < NUM_OF_BANDS) add(bandID, gaussCoefL[hID % WIDTH_OF_BAND]);
        }
        float desVec[NUM_OF_BANDS * 8];
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        for (short bandID = 0; bandID < NUM_OF_BANDS; bandID++) {
            const float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
            const short desID = bandID * 8;
            float temp = pgdLBandSum[bandID] * invN;
            desVec[desID] = temp; desVec[desID + 4] = std::sqrt(pgdL2BandSum[bandID] * invN - temp * temp);
            temp = ngdLBandSum[bandID] * invN;
            desVec[desID + 1] = temp; desVec[desID + 5] = std::sqrt(ngdL2BandSum[bandID] * invN - temp * temp);
            temp = pgdOBandSum[bandID] * invN;
            desVec[desID + 2] = temp; desVec[desID + 6] = std::sqrt(pgdO2BandSum[bandID] * invN - temp * temp);
            temp = ngdOBandSum[bandID] * invN;
            desVec[desID + 3] = temp; desVec[desID + 7] = std::sqrt(ngdO2BandSum[bandID] * invN - temp * temp);
        }
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NUM_OF_BANDS; b++) {
            const float *d = desVec + b * 8;
            tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
            tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
        }
        tempM = 1 / std::sqrt(tempM); tempS = 1 / std::sqrt(tempS);
        for (int b = 0; b < NUM_OF_BANDS; b++) { float *d = desVec + b * 8; for (int q = 0; q < 4; q++) d[q] = d[q] * tempM; for (int q = 4; q < 8; q++) d[q] = d[q] * tempS; }
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) if (desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) temp += desVec[i] * desVec[i];
        temp = 1 / std::sqrt(temp);
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) desVec[i] = desVec[i] * temp;
        if (fdesc) std::memcpy(fdesc + (size_t)li * 72, desVec, sizeof(desVec));
        for (int c = 0; c < 32; c++) { // binaryConversion :405-416
            const float *f1 = &desVec[8 * combinations[c][0]], *f2 = &desVec[8 * combinations[c][1]];
            uint8_t r = 0;
            for (int i = 0; i < 8; i++) if (f1[i] > f2[i]) r += (uint8_t)(1 << i);
            desc[(size_t)li * 32 + c] = r;
        }
    }
    return n;
}

} // extern "C"
