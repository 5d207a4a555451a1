// Dependent-chain latencies on one wave / one workgroup (the regime of ba_band_*): f64 FMA, f64 rsq, LDS read round trip,
// f64 MFMA 16x16x4 accumulation chain, s_barrier with 6 waves.  Prints cycles per operation (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out, unsigned long long *cyc, int n) {
    __shared__ int chase[1024];
    __shared__ double sd[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += blockDim.x) { chase[i] = (i * 37 + 11) & 1023; sd[i] = i * 0.5; }
    __syncthreads();
    double a = out[tid], b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) a = __builtin_fma(a, b, c);
    unsigned long long t1 = __builtin_readcyclecounter();
    double r = a;
    for (int i = 0; i < n; i++) r = __builtin_amdgcn_rsq(r + 2.0);
    unsigned long long t2 = __builtin_readcyclecounter();
    int p = tid & 1023;
    for (int i = 0; i < n; i++) p = chase[p];
    unsigned long long t3 = __builtin_readcyclecounter();
    v4d acc = {a, r, 0, 1};
    for (int i = 0; i < n; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    unsigned long long t4 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) __syncthreads();
    unsigned long long t5 = __builtin_readcyclecounter();
    double dsum = 0; int q = tid;
    for (int i = 0; i < n; i++) { dsum += sd[q & 1023]; q = (int)dsum & 1023; } // LDS f64 read + dependent add + cvt
    unsigned long long t6 = __builtin_readcyclecounter();
    double x = a;
    for (int i = 0; i < n; i++) { x = x * b; x = x - c; } // mul + sub dependent (no contraction)
    unsigned long long t7 = __builtin_readcyclecounter();
    out[tid] = a + r + p + acc[0] + acc[1] + acc[2] + acc[3] + dsum + x;
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; cyc[6] = t7 - t6; }
}
int main() {
    double *o; unsigned long long *c, h[8]; hipMalloc(&o, 4096 * 8); hipMemset(o, 0, 4096 * 8); hipMalloc(&c, 64);
    const int n = 2000;
    for (int nt : {64, 384}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, o, c, n); hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, o, c, n);
        hipMemcpy(h, c, 56, hipMemcpyDeviceToHost);
        printf("threads %3d: fma_f64 %.1f  rsq_f64+add %.1f  lds chase %.1f  mfma_f64_16x16x4 chain %.1f  barrier %.1f  lds f64+add+cvt %.1f  mul+sub %.1f cycles\n", nt,
               h[0] / (double)n, h[1] / (double)n, h[2] / (double)n, h[3] / (double)n, h[4] / (double)n, h[5] / (double)n, h[6] / (double)n);
    }
}
