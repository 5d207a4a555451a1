// glibc_sincosf.h -- cosf / sinf with the results of glibc >= 2.28 (sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c, sincosf.h: the
// double-precision polynomial of the Arm optimized routines), for code that runs on the device and has to agree with a host that calls
// cos(float) / sin(float) (region_grow, reference line_lbd/libs/lsd.cpp:676-677, through <math.h>: the float overloads).
//
// glibc's functions are not correctly rounded (0.56 ULP), so float(cos(double(x))) is not a substitute: it differs from cosf on 0.24 % and
// from sinf on 0.54 % of the floats in (0, 2 pi].  This restatement -- same reduction, same polynomials, evaluated in the same order in IEEE
// doubles -- equals glibc 2.35's cosf and sinf on every float of [-6.3, 6.3] (tests/test_sincosf.py walks all 2 x 189 million of them against
// the host's libm in the slow variant, a strided sample by default; with or without FMA contraction: the margins absorb it).
// Domain: |x| < 120 (the fast reduction); the angles of the level-line field are in [0, 2 pi).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define GS_FN __host__ __device__ inline
#else
#define GS_FN inline
#endif

namespace glibc_sincosf {
struct Table { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
GS_FN uint32_t abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
// polynomials on [-pi/4, pi/4]: n even -> sine of x, n odd -> cosine; neg selects the negated cosine (quadrants 2, 3)
GS_FN float poly(double x, double x2, int n, bool neg) {
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = s2 + x2 * s3, x5 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x5 * t1);
    }
    const double sg = neg ? -1.0 : 1.0; // the second table holds the negated coefficients: same magnitudes, exact sign flips
    const double x4 = x2 * x2, t2 = sg * c3 + x2 * (sg * c4), t1 = sg * c0 + x2 * (sg * c1), x6 = x4 * x2, c = t1 + x4 * (sg * c2);
    return (float)(c + x6 * t2);
}
GS_FN double reduce_fast(double x, bool neg, int *np) { // quadrant in bits 24..31 of x * 2/pi * 2^24
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * (neg ? -hpi_inv : hpi_inv);
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * (neg ? -hpi : hpi);
}
GS_FN float cosf_(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return poly(x, x * x, 1, false);
    }
    int n;
    x = reduce_fast(x, false, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly(x * s, x * x, n ^ 1, (n & 2) != 0);
}
GS_FN float sinf_(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return poly(x, x * x, 0, false);
    }
    int n;
    x = reduce_fast(x, false, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly(x * s, x * x, n, (n & 2) != 0);
}
// cosf_(y) and sinf_(y) together and without branches, for 0 <= y < 120 (the level-line angles are in [0, 2 pi]): the fast reduction also covers
// |y| < pi / 4 (there n = 0 and x is unchanged) and the tiny arguments (the polynomials round to 1.0f and to y there); the two polynomials are
// evaluated once and dealt to the two results by the quadrant, the negated coefficient table is the negated result (rounding is symmetric).
// Equal to cosf_ / sinf_ -- and so to glibc -- on every float of [0, 6.3]: tests/cpp/sincosf_check.cpp.
GS_FN void sincosf_pos(float y, float *sn, float *cs) {
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double x0 = y;
    const int n = ((int32_t)(x0 * hpi_inv) + 0x800000) >> 24;
    const double xr = x0 - n * hpi;
    const bool flip = (n & 3) == 1 || (n & 3) == 2, neg = (n & 2) != 0;
    const double x = flip ? -xr : xr, x2 = xr * xr; // (x * s with s = +-1; x * x)
    const double x3 = x * x2, ts = s2 + x2 * s3, x5 = x3 * x2, ss = x + x3 * s1;
    const float A = (float)(ss + x5 * ts); // the sine polynomial
    const double x4 = x2 * x2, t2 = c3 + x2 * c4, t1 = c0 + x2 * c1, x6 = x4 * x2, cc = t1 + x4 * c2;
    const float Bp = (float)(cc + x6 * t2), B = neg ? -Bp : Bp; // the cosine polynomial
    *sn = (n & 1) ? B : A;
    *cs = (n & 1) ? A : B;
}
} // namespace glibc_sincosf
