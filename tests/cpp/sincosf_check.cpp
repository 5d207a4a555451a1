// glibc_sincosf.h against the host's libm: every STRIDE-th float of [2^-20, 6.3], both signs, plus the edges (argv[1] = stride, 1 = all)
#include "../../cube_slam_amd/csrc/glibc_sincosf.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 97;
    float lo = 9.5e-7f, hi = 6.3f;
    uint32_t a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    long n = 0, bad = 0;
    for (uint32_t u = a; u <= b; u += stride) {
        float x; memcpy(&x, &u, 4);
        n += 4;
        bad += cosf(x) != glibc_sincosf::cosf_(x); bad += sinf(x) != glibc_sincosf::sinf_(x);
        bad += cosf(-x) != glibc_sincosf::cosf_(-x); bad += sinf(-x) != glibc_sincosf::sinf_(-x);
        { float sn, cs; glibc_sincosf::sincosf_pos(x, &sn, &cs); n += 2; bad += cosf(x) != cs; bad += sinf(x) != sn; } // the branch-free pair of lsd_rg_wlk.h
    }
    const float edge[] = {0.0f, 1e-30f, 0x1p-12f, 0x1.921FB6p-1f, 0x1.921FB4p-1f, 3.14159274f, 6.28318548f, 6.2831850f};
    for (float x : edge) { n += 2; bad += cosf(x) != glibc_sincosf::cosf_(x); bad += sinf(x) != glibc_sincosf::sinf_(x); float sn, cs; glibc_sincosf::sincosf_pos(x, &sn, &cs); n += 2; bad += cosf(x) != cs; bad += sinf(x) != sn; }
    printf("%ld values, %ld mismatches\n", n, bad);
    return bad != 0;
}
