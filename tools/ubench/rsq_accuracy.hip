#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *d, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    double x = d[i], inv = __builtin_amdgcn_rsq(x);
    r0[i] = inv;
    inv = inv * (1.5 - (0.5 * x) * (inv * inv)); r1[i] = inv;
    inv = inv * (1.5 - (0.5 * x) * (inv * inv)); r2[i] = inv;
}
int main() {
    const int n = 1 << 20; std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-20, 20);
    std::vector<double> h(n), a(n), b(n), c(n); for (auto &x : h) x = std::exp(u(g));
    double *d, *r0, *r1, *r2; hipMalloc(&d, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, r0, r1, r2, n);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) { long double t = 1.0L / sqrtl((long double)h[i]); e0 = fmax(e0, fabsl((a[i] - t) / t)); e1 = fmax(e1, fabsl((b[i] - t) / t)); e2 = fmax(e2, fabsl((c[i] - t) / t)); }
    printf("v_rsq_f64 max rel err %.3g (2^%.1f); after 1 Newton %.3g (2^%.1f); after 2 %.3g (2^%.1f)\n", e0, log2(e0), e1, log2(e1), e2, log2(e2));
}
