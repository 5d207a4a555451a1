// Issue-rate probe for a few gfx950 VALU instructions (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP> __global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    long l0 = 0;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %1\n v_mul_f64 %3, %3, %1\n v_mul_f64 %4, %4, %1" : "+v"(a0), "+v"(seed), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %1\n v_add_f64 %3, %3, %1\n v_add_f64 %4, %4, %1" : "+v"(a0), "+v"(seed), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (OP == 3) { REP8(asm volatile("v_trunc_f64 %0, %0\n v_trunc_f64 %1, %1\n v_trunc_f64 %2, %2\n v_trunc_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %0, vcc, %4, %1, %0" : "+v"(l0) : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %1, %1, %2, %3\n v_mad_i32_i24 %2, %2, %3, %0\n v_mad_i32_i24 %3, %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 6) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 7) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1\n v_fma_f64 %3, %3, %1, %1\n v_fma_f64 %4, %4, %1, %1" : "+v"(a0), "+v"(seed), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 8) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 9) { REP8(asm volatile("v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %1, %1, %2, %3\n v_med3_i32 %2, %2, %3, %0\n v_med3_i32 %3, %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (OP == 10) { REP8(asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 11) { REP8(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (OP == 12) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 13) { REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 + l0;
}
template <int OP> void run(const char *name, double *d) {
    const int iters = 2000, blocks = 256 * 8; // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0000001);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_simd = (double)iters * 32 * (blocks * 4.0) / (256 * 4);
    printf("%-16s %.2f cycles per wave instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_mul_f64", d); run<1>("v_add_f64", d); run<7>("v_fma_f64", d); run<2>("v_cvt_i32_f64", d); run<3>("v_trunc_f64", d); run<10>("v_floor_f64", d);
    run<11>("v_cvt_f64_i32", d); run<4>("v_mad_u64_u32", d); run<5>("v_mad_i32_i24", d); run<6>("v_mul_lo_u32", d); run<8>("v_add_f32", d); run<9>("v_med3_i32", d);
    run<12>("v_rcp_f64", d); run<13>("v_sqrt_f64", d);
    return 0;
}
