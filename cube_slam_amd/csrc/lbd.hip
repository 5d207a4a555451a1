// lbd.hip -- LBD line descriptor (and its brute-force Hamming matcher) of line_lbd on MI355X (gfx950).
//
// Replaces BinaryDescriptor::compute / computeImpl / computeLBD (reference line_lbd/libs/binary_descriptor.cpp:588-790,
// 1146-1509) for one octave and line_lbd_detect::match_line_descrip (class/line_lbd_allclass.cpp:339-356).
// Built with -ffp-contract=off: the reference accumulates in float with separate multiplies and adds.
//   lbd_blur5     GaussianBlur(5x5, sigma 1) on u8, 8-bit fixed point, LDS tile              (computeGaussianPyramid :352-370)
//   lbd_sobel     Sobel 3x3 -> int16 dx, dy (interleaved), REFLECT_101                              (computeSobel :373-402)
//   lbd_line_desc workgroup (one wave) per line.  Walk: lane = support-region row, numOfPixels samples with rounded + clamped
//                 coordinates, sequential float sums of the positive / negative projections on dL and dO (:1272-1335), gathers
//                 issued eight samples at a time.  Descriptor: the 63 row sums stay in LDS; lane = band accumulator (72 of them,
//                 in-order sums over the <= 21 rows of a band with the local Gaussian weights), mean/std per band, the two
//                 normalisations with the 0.4 clip, 32 band-pair comparisons -> 32 bytes (:1337-1478, :405-416)
#include "common.h"

#include <cmath>
#include <vector>

namespace {
constexpr int NB = 9, WB = 7, HLSP = NB * WB;
__constant__ int c_comb[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
                                  {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};
struct LbdW { float gL[WB * 3]; float gG[HLSP]; int k5[5]; };

__device__ __forceinline__ int refl(int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; } return p; }

// Row-streaming filters (same scheme as orb_blur in orb.hip): ONE WAVE owns a 256-column strip and LBD_ROWS output rows, every
// lane four neighbouring columns; each input row is loaded once (next rows already in flight), goes through an LDS line, the
// vertical window stays in registers, and results leave as dword / 8-byte stores (64-B byte stores per wave are several times
// slower).  Arithmetic identical to the tile versions they replace.
constexpr int LBD_ROWS = 64;
// GaussianBlur(5x5, sigma 1), 8-bit fixed point, BORDER_REFLECT_101
__global__ void __launch_bounds__(64) lbd_blur5(const uint8_t *gray, int W, int H, LbdW wts, uint8_t *blur) {
    const int strips = (W + 255) / 256;
    const int sx = (blockIdx.x % strips) * 256, y0 = (blockIdx.x / strips) * LBD_ROWS, tid = threadIdx.x;
    if (y0 >= H) return;
    const int rows = min(LBD_ROWS, H - y0);
    __shared__ uint32_t line32[(256 + 16) / 4]; // column sx + c at byte 4 + c
    uint8_t *line = reinterpret_cast<uint8_t *>(line32);
    const uint8_t *img = gray + (long)blockIdx.z * W * H;
    uint8_t *out = blur + (long)blockIdx.z * W * H;
    int xc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) xc[c] = refl(sx + 4 * tid + c, W);
    const int xh = tid < 2 ? refl(sx - 2 + tid, W) : refl(sx + 256 + (tid - 2), W); // halo columns, lanes 0..3
    const int x = sx + 4 * tid;
    const bool inner = x + 3 < W; // the lane's four columns exist: one unaligned dword load per row
    int k[5];
#pragma unroll
    for (int t = 0; t < 5; t++) k[t] = wts.k5[t];
    const uint32_t kA = (uint32_t)k[0] | (uint32_t)k[1] << 8 | (uint32_t)k[2] << 16 | (uint32_t)k[3] << 24;
    int ring[4][5];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int t = 0; t < 5; t++) ring[c][t] = 0;
    constexpr int G = 4;
    const int total = rows + 4;
    uint32_t cur[G], nxt[G]; uint8_t curh[G], nxth[G];
    auto fetch = [&](int r0, uint32_t (&a)[G], uint8_t (&hh)[G]) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            a[u] = 0; hh[u] = 0;
            if (r0 + u < total) {
                const uint8_t *row = img + (long)refl(y0 + r0 + u - 2, H) * W;
                a[u] = inner ? load_u32_unaligned(row + x) : ((uint32_t)row[xc[0]] | ((uint32_t)row[xc[1]] << 8) | ((uint32_t)row[xc[2]] << 16) | ((uint32_t)row[xc[3]] << 24));
                if (tid < 4) hh[u] = row[xh];
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
                line32[1 + tid] = cur[u];
                if (tid < 2) line[2 + tid] = curh[u];                   // sx - 2, sx - 1 at bytes 2, 3
                else if (tid < 4) line[4 + 256 + (tid - 2)] = curh[u];  // sx + 256, sx + 257
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t w0 = line32[tid], w1 = line32[tid + 1], w2 = line32[tid + 2];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // column x + c: taps at bytes c + 2 .. c + 6 of the 12-byte window (same dot4 / mul24 scheme as orb_blur)
                uint32_t packed = 0;
                const uint32_t t4[4] = {win4<2>(w0, w1, w2), win4<3>(w0, w1, w2), win4<4>(w0, w1, w2), win4<5>(w0, w1, w2)}; // taps 0..3 of column c
                const uint32_t t5[4] = {win4<6>(w0, w1, w2), win4<7>(w0, w1, w2), win4<8>(w0, w1, w2), win4<8>(w0, w1, w2) >> 8};     // tap 4 in the low byte
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t h = __builtin_amdgcn_udot4(t4[c], kA, __umul24(t5[c] & 255u, (uint32_t)k[4]), false);
#pragma unroll
                    for (int t = 0; t < 4; t++) ring[c][t] = ring[c][t + 1];
                    ring[c][4] = (int)h;
                    uint32_t sum = 1u << 15;
#pragma unroll
                    for (int t = 0; t < 5; t++) sum = __umul24((uint32_t)ring[c][t], (uint32_t)k[t]) + sum;
                    const uint32_t v = sum >> 16;
                    packed |= (v > 255u ? 255u : v) << (8 * c);
                }
                if (r >= 4 && x < W) {
                    uint8_t *dst = out + (long)(y0 + r - 4) * W + x;
                    if (x + 3 < W) *reinterpret_cast<uint32_t *>(dst) = packed;
                    else for (int c = 0; c < 4 && x + c < W; c++) dst[c] = (uint8_t)(packed >> (8 * c));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; u++) { cur[u] = nxt[u]; curh[u] = nxth[u]; }
    }
}
// Sobel 3x3 -> int16 dx, dy, BORDER_REFLECT_101.  One interleaved map (dx low half, dy high half of a dword): the line walk reads
// both derivatives of a sample with a single gather, and its cost is the number of gathered lanes.
__global__ void __launch_bounds__(64) lbd_sobel(const uint8_t *blur, int W, int H, uint32_t *dxyo) {
    const int strips = (W + 255) / 256;
    const int sx = (blockIdx.x % strips) * 256, y0 = (blockIdx.x / strips) * LBD_ROWS, tid = threadIdx.x;
    if (y0 >= H) return;
    const int rows = min(LBD_ROWS, H - y0);
    __shared__ uint32_t line32[(256 + 16) / 4];
    uint8_t *line = reinterpret_cast<uint8_t *>(line32);
    const uint8_t *img = blur + (long)blockIdx.z * W * H;
    uint32_t *oxy = dxyo + (long)blockIdx.z * W * H;
    int xc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) xc[c] = refl(sx + 4 * tid + c, W);
    const int xh = tid < 1 ? refl(sx - 1, W) : refl(sx + 256, W); // lanes 0, 1
    const int x = sx + 4 * tid;
    const bool inner = x + 3 < W; // the lane's four columns exist: one unaligned dword load per row
    int win[3][6]; // rows y-1, y, y+1; columns x-1 .. x+4
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int c = 0; c < 6; c++) win[i][c] = 0;
    constexpr int G = 4;
    const int total = rows + 2;
    uint32_t cur[G], nxt[G]; uint8_t curh[G], nxth[G];
    auto fetch = [&](int r0, uint32_t (&a)[G], uint8_t (&hh)[G]) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            a[u] = 0; hh[u] = 0;
            if (r0 + u < total) {
                const uint8_t *row = img + (long)refl(y0 + r0 + u - 1, H) * W;
                a[u] = inner ? load_u32_unaligned(row + x) : ((uint32_t)row[xc[0]] | ((uint32_t)row[xc[1]] << 8) | ((uint32_t)row[xc[2]] << 16) | ((uint32_t)row[xc[3]] << 24));
                if (tid < 2) hh[u] = row[xh];
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
                line32[1 + tid] = cur[u];
                if (tid == 0) line[3] = curh[u];              // sx - 1
                else if (tid == 1) line[4 + 256] = curh[u];   // sx + 256
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t w0 = line32[tid], w1 = line32[tid + 1], w2 = line32[tid + 2];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int c = 0; c < 6; c++) { win[0][c] = win[1][c]; win[1][c] = win[2][c]; }
                win[2][0] = (w0 >> 24) & 255;                 // span byte 3 = column x - 1
#pragma unroll
                for (int c = 0; c < 4; c++) win[2][1 + c] = (w1 >> (8 * c)) & 255;
                win[2][5] = w2 & 255;                         // column x + 4
                if (r >= 2 && x < W) {
                    short vx[4], vy[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) { // P(x + c + dx, y + dy) = win[1 + dy][1 + c + dx]
                        vx[c] = (short)((win[0][c + 2] + 2 * win[1][c + 2] + win[2][c + 2]) - (win[0][c] + 2 * win[1][c] + win[2][c]));
                        vy[c] = (short)((win[2][c] + 2 * win[2][c + 1] + win[2][c + 2]) - (win[0][c] + 2 * win[0][c + 1] + win[0][c + 2]));
                    }
                    const long o = (long)(y0 + r - 2) * W + x;
                    uint32_t pk[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) pk[c] = (uint32_t)(uint16_t)vx[c] | ((uint32_t)(uint16_t)vy[c] << 16);
                    if (x + 3 < W && (o & 3) == 0) *reinterpret_cast<uint4 *>(oxy + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    else
                        for (int c = 0; c < 4 && x + c < W; c++) oxy[o + c] = pk[c];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; u++) { cur[u] = nxt[u]; curh[u] = nxth[u]; }
    }
}

__device__ __forceinline__ void sincos_fd(float angle, float &so, float &co) { // same evaluation as orb.hip
    const double x = (double)angle;
    const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HI = 1.57079632679489655800, PIO2_LO = 6.12323399573676603587e-17;
    const double kd = floor(x * TWO_OVER_PI + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO, r2 = r * r;
    const double sp = -1.0 / 6.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 5040.0 + r2 * (1.0 / 362880.0 + r2 * (-1.0 / 39916800.0 + r2 * (1.0 / 6227020800.0 + r2 * (-1.0 / 1307674368000.0))))));
    const double cp = -1.0 / 2.0 + r2 * (1.0 / 24.0 + r2 * (-1.0 / 720.0 + r2 * (1.0 / 40320.0 + r2 * (-1.0 / 3628800.0 + r2 * (1.0 / 479001600.0 + r2 * (-1.0 / 87178291200.0 + r2 * (1.0 / 20922789888000.0)))))));
    const double s = r + r * r2 * sp, c = 1.0 + r2 * cp;
    double ss, cc;
    switch (k & 3) { case 0: ss = s; cc = c; break; case 1: ss = c; cc = -s; break; case 2: ss = -s; cc = -c; break; default: ss = -c; cc = s; break; }
    so = (float)ss; co = (float)cc;
}

// block = one line (64 threads: lane = support-region row hID for the walk, then band accumulators / descriptor elements)
__global__ void __launch_bounds__(64) lbd_line_desc(const cs_keyline *kls, const int *line_frame, const uint32_t *dxyImg, int W, int H, LbdW wts, uint8_t *desc, float *fdesc) {
    __shared__ float s_row[HLSP * 4], s_bs[8 * NB], s_d[8 * NB];
    const int li = blockIdx.x, hID = threadIdx.x;
    if (hID < HLSP) {
    const cs_keyline L = kls[li];
    if (line_frame) dxyImg += (long)line_frame[li] * W * H;
    const short realWidth = (short)W, imageWidth = realWidth - 1, imageHeight = (short)(H - 1);
    const short lengthOfLSP = (short)L.numOfPixels, halfWidth = (lengthOfLSP - 1) / 2, halfHeight = (HLSP - 1) / 2;
    const float mx = (float)(0.5 * (L.sPointInOctaveX + L.ePointInOctaveX)), my = (float)(0.5 * (L.sPointInOctaveY + L.ePointInOctaveY));
    float dL0, dL1;
    sincos_fd(L.angle, dL1, dL0);
    const float dO0 = -dL1, dO1 = dL0;
    float sCorX = -dL0 * halfWidth + dL1 * halfHeight + mx, sCorY = -dL1 * halfWidth - dL0 * halfHeight + my;
    for (int r = 0; r < hID; r++) { sCorX -= dL1; sCorY += dL0; } // sCorX0 -= dL[1]; sCorY0 += dL[0] once per previous row (:1337-1338)
    float pL = 0, nL = 0, pO = 0, nO = 0;
    // The walk is a chain of dependent float additions (coordinates and sums, order of the reference), but the gradient samples do
    // not depend on the sums: eight positions are generated, their sixteen gathers issued together, then accumulated in order --
    // one exposed memory latency per eight samples instead of one per sample (this kernel is latency-bound: ~3k short waves).
    // (short)round((double)x) == (short)roundf(x): float -> double is exact and both round half away from zero.
    constexpr int U = 8;
    for (int w0 = 0; w0 < lengthOfLSP; w0 += U) {
        int off[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            short t = (short)roundf(sCorX);
            const short xCor = (t < 0) ? 0 : (t > imageWidth) ? imageWidth : t;
            t = (short)roundf(sCorY);
            const short yCor = (t < 0) ? 0 : (t > imageHeight) ? imageHeight : t;
            off[u] = yCor * realWidth + xCor;
            sCorX += dL0; sCorY += dL1;
        }
        uint32_t gv[U];
#pragma unroll
        for (int u = 0; u < U; u++) gv[u] = dxyImg[off[u]]; // positions past the end are valid pixels, their values unused
#pragma unroll
        for (int u = 0; u < U; u++)
            if (w0 + u < lengthOfLSP) {
                const short dx = (short)(gv[u] & 0xffff), dy = (short)(gv[u] >> 16);
                const float gDL = dx * dL0 + dy * dL1, gDO = dx * dO0 + dy * dO1;
                if (gDL > 0) pL += gDL; else nL -= gDL;
                if (gDO > 0) pO += gDO; else nO -= gDO;
            }
    }
    // ---- descriptor of the line (:1337-1478, :405-416): the 63 row sums never leave the workgroup
    s_row[hID * 4] = pL; s_row[hID * 4 + 1] = nL; s_row[hID * 4 + 2] = pO; s_row[hID * 4 + 3] = nO;
    }
    __syncthreads();
    // 72 band accumulators (quantity q of band b), each the in-order sum over the <= 21 rows that touch band b: lanes 0..63 own
    // accumulator `lane`, lanes 0..7 also own 64 + lane.  Same additions in the same order as the row-major loop of the reference.
    for (int a = threadIdx.x; a < 8 * NB; a += 64) {
        const int q = a / NB, bnd = a % NB;
        float acc = 0;
        const int h0 = max(0, WB * (bnd - 1)), h1 = min(HLSP, WB * (bnd + 2));
        for (int h = h0; h < h1; h++) {
            const int rel = h / WB - bnd; // -1: the row's band + 1 is bnd; 0: own band; +1: the row's band - 1 is bnd
            const float c = wts.gL[h % WB + (rel == 0 ? WB : (rel == 1 ? 2 * WB : 0))];
            const float coef = wts.gG[h];
            const float v = coef * s_row[h * 4 + (q & 1) + ((q >> 2) << 1)]; // q: 0 pL 1 nL 2 pL2 3 nL2 4 pO 5 nO 6 pO2 7 nO2
            if (q & 2) acc += c * c * (v * v); else acc += c * v;
        }
        s_bs[a] = acc;
    }
    __syncthreads();
    const float invN2 = (float)(1.0 / (WB * 2.0)), invN3 = (float)(1.0 / (WB * 3.0));
    for (int i = threadIdx.x; i < NB * 8; i += 64) { // d[b*8 + k]: k 0..3 means of pL nL pO nO, k 4..7 their standard deviations
        const int bnd = i / 8, k = i % 8, k4 = k & 3;
        const float invN = (bnd == 0 || bnd == NB - 1) ? invN2 : invN3;
        const int qm = (k4 & 1) + ((k4 >> 1) << 2); // pL -> 0, nL -> 1, pO -> 4, nO -> 5
        const float t = s_bs[qm * NB + bnd] * invN;
        s_d[i] = (k < 4) ? t : sqrtf(s_bs[(qm + 2) * NB + bnd] * invN - t * t);
    }
    __syncthreads();
    float tM = 0, tS = 0; // every lane repeats the two ordered sums (LDS broadcasts)
    for (int bnd = 0; bnd < NB; bnd++) {
        const float *e = s_d + bnd * 8;
        tM += e[0] * e[0]; tM += e[1] * e[1]; tM += e[2] * e[2]; tM += e[3] * e[3];
        tS += e[4] * e[4]; tS += e[5] * e[5]; tS += e[6] * e[6]; tS += e[7] * e[7];
    }
    tM = 1 / sqrtf(tM); tS = 1 / sqrtf(tS);
    __syncthreads();
    for (int i = threadIdx.x; i < NB * 8; i += 64) { float v = s_d[i] * ((i & 4) ? tS : tM); if (v > 0.4) v = (float)0.4; s_d[i] = v; }
    __syncthreads();
    float t = 0;
    for (int i = 0; i < NB * 8; i++) t += s_d[i] * s_d[i];
    t = 1 / sqrtf(t);
    __syncthreads();
    for (int i = threadIdx.x; i < NB * 8; i += 64) { const float v = s_d[i] * t; s_d[i] = v; if (fdesc) fdesc[(long)li * 72 + i] = v; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const float *f1 = s_d + 8 * c_comb[threadIdx.x][0], *f2 = s_d + 8 * c_comb[threadIdx.x][1];
        int r = 0;
        for (int i = 0; i < 8; i++) if (f1[i] > f2[i]) r += 1 << i;
        desc[(long)li * 32 + threadIdx.x] = (uint8_t)r;
    }
}

static LbdW make_weights() { // BinaryDescriptor constructor :218-260 (integer divisions intended) + getGaussianKernel(5, 1)
    LbdW w;
    double u = (WB * 3 - 1) / 2, sigma = (WB * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < WB * 3; i++) { double dis = i - u; w.gL[i] = (float)std::exp(dis * dis * inv); }
    u = (NB * WB - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
    for (int i = 0; i < HLSP; i++) { double dis = i - u; w.gG[i] = (float)std::exp(dis * dis * inv); }
    float cf[5]; double sum = 0, s2 = -0.5;
    for (int i = 0; i < 5; i++) { double x = i - 2.0; cf[i] = (float)std::exp(s2 * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 5; i++) { cf[i] = (float)(cf[i] * sum); w.k5[i] = (int)std::lrint(cf[i] * 256.f); }
    return w;
}

} // namespace

// batched entry points shared with lsd.hip (frames resident in HBM)
int cs_lbd_batch_maps(cs_ctx *ctx, const uint8_t *d_gray, int W, int H, int F, uint8_t *d_blur, uint32_t *d_dxy) {
    const LbdW w = make_weights();
    const int nblk = ((W + 255) / 256) * ((H + LBD_ROWS - 1) / LBD_ROWS);
    CS_LAUNCH(ctx, "lbd_blur5", lbd_blur5, dim3(nblk, 1, F), dim3(64), 0, d_gray, W, H, w, d_blur);
    CS_LAUNCH(ctx, "lbd_sobel", lbd_sobel, dim3(nblk, 1, F), dim3(64), 0, d_blur, W, H, d_dxy);
    return CS_OK;
}
int cs_lbd_batch_desc(cs_ctx *ctx, const cs_keyline *d_kl, const int *d_line_frame, int n, const uint32_t *d_dxy, int W, int H, uint8_t *d_desc, float *d_f) {
    if (n <= 0) return CS_OK;
    const LbdW w = make_weights();
    CS_LAUNCH(ctx, "lbd_line_desc", lbd_line_desc, dim3(n), dim3(64), 0, d_kl, d_line_frame, d_dxy, W, H, w, d_desc, d_f);
    return CS_OK;
}

namespace {
struct Bufs { uint8_t *gray = nullptr, *blur = nullptr; uint32_t *dxy = nullptr; };
static void free_bufs(Bufs &b) { if (b.gray) hipFree(b.gray); if (b.blur) hipFree(b.blur); if (b.dxy) hipFree(b.dxy); b = Bufs(); }
static int run_maps(cs_ctx *ctx, const uint8_t *gray, int W, int H, int stride, Bufs &b) {
    const size_t N = (size_t)W * H;
    int r = cs_dalloc(ctx, &b.gray, N); if (r) return r;
    r = cs_dalloc(ctx, &b.blur, N); if (r) return r;
    r = cs_dalloc(ctx, &b.dxy, N); if (r) return r;
    CS_HIP(ctx, hipMemcpy2DAsync(b.gray, (size_t)W, gray, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice, ctx->stream));
    return cs_lbd_batch_maps(ctx, b.gray, W, H, 1, b.blur, b.dxy);
}
} // namespace

extern "C" {

int cs_lbd_maps(cs_ctx *ctx, const uint8_t *gray, int width, int height, int stride, uint8_t *blur, int16_t *dx, int16_t *dy) {
    if (!ctx || !gray || width < 8 || height < 8 || stride < width) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    Bufs b;
    int r = run_maps(ctx, gray, width, height, stride, b);
    const size_t N = (size_t)width * height;
    if (!r && blur) r = cs_d2h(ctx, blur, b.blur, N);
    if (!r && (dx || dy)) {
        std::vector<uint32_t> xy(N);
        r = cs_d2h(ctx, xy.data(), b.dxy, N);
        hipStreamSynchronize(ctx->stream);
        if (!r) for (size_t i = 0; i < N; i++) { if (dx) dx[i] = (int16_t)(xy[i] & 0xffff); if (dy) dy[i] = (int16_t)(xy[i] >> 16); }
    }
    hipStreamSynchronize(ctx->stream);
    free_bufs(b);
    return r;
}

int cs_lbd_compute(cs_ctx *ctx, const uint8_t *gray, int width, int height, int stride, const cs_keyline *keylines, int n, uint8_t *desc, float *float_desc) {
    if (!ctx || !gray || width < 8 || height < 8 || width > 32767 || height > 32767 || stride < width || n < 0 || (n && (!keylines || !desc))) return CS_ERR_BAD_ARG;
    if (n == 0) return CS_OK; // the reference prints "keypoint list is empty" and returns (:619-623)
    for (int i = 0; i < n; i++) if (keylines[i].numOfPixels < 0 || keylines[i].numOfPixels > 32767) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    Bufs b;
    int r = run_maps(ctx, gray, width, height, stride, b);
    cs_keyline *d_kl = nullptr; float *d_f = nullptr; uint8_t *d_desc = nullptr;
    if (!r) r = cs_dalloc(ctx, &d_kl, (size_t)n);
    if (!r) r = cs_dalloc(ctx, &d_desc, (size_t)n * 32);
    if (!r && float_desc) r = cs_dalloc(ctx, &d_f, (size_t)n * 72);
    if (!r) r = cs_h2d(ctx, d_kl, keylines, (size_t)n);
    if (!r) {
        r = cs_lbd_batch_desc(ctx, d_kl, nullptr, n, b.dxy, width, height, d_desc, d_f);
        if (!r) r = cs_d2h(ctx, desc, d_desc, (size_t)n * 32);
        if (!r && float_desc) r = cs_d2h(ctx, float_desc, d_f, (size_t)n * 72);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!r && e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; }
    free_bufs(b);
    if (d_kl) hipFree(d_kl); if (d_desc) hipFree(d_desc); if (d_f) hipFree(d_f);
    return r;
}

int cs_lbd_match(cs_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, float dist_thres, int *query_idx, int *train_idx, int *distance, int *n_matches) {
    if (!ctx || !n_matches || nq < 0 || nt < 0) return CS_ERR_BAD_ARG;
    *n_matches = 0;
    if (nq == 0 || nt == 0) return CS_OK;
    std::vector<int> bi(nq), bd(nq), sd(nq);
    int r = cs_hamming_knn2(ctx, q, nq, t, nt, bi.data(), bd.data(), sd.data()); if (r) return r;
    int m = 0;
    for (int i = 0; i < nq; i++) // match_line_descrip :339-356: keep matches[i].distance < matching_dist_thres
        if ((float)bd[i] < dist_thres) { if (query_idx) query_idx[m] = i; if (train_idx) train_idx[m] = bi[i]; if (distance) distance[m] = bd[i]; m++; }
    *n_matches = m;
    return CS_OK;
}

} // extern "C"
