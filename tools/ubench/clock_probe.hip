// clock_probe.hip -- what shader clock does the GPU run at under a light load (few waves per CU, scalar-heavy) and under a full one?
// s_memtime (clock64, shader cycles) against s_memrealtime (wall_clock64, constant 100 MHz) around a chain of dependent instructions.
// hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long iters, int mode, unsigned long long *out) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x * 1e-3f; int s = blockIdx.x;
    for (long i = 0; i < iters; i++) {
        if (mode == 0) { x = x * 1.0001f + 0.5f; x = x * 0.9999f - 0.25f; x = x * 1.0001f + 0.5f; x = x * 0.9999f - 0.25f; }
        else { s = __builtin_amdgcn_readfirstlane(s * 3 + 1); s ^= s >> 3; s = s * 5 + 7; s ^= s << 2; x += (s & 1); }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (x == 12345.f) out[0] = 1;
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16 * 8192);
    for (int mode = 0; mode < 2; mode++)
        for (int cfg = 0; cfg < 4; cfg++) {
            const int blocks = cfg == 0 ? 64 : cfg == 1 ? 64 : cfg == 2 ? 256 : 1024, threads = cfg == 0 ? 64 : 1024;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, 2000000L, mode, d);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("mode %s blocks %4d x %4d threads: %8.2f ms, shader cycles %llu, realtime ticks %llu -> %.3f GHz if realtime = 100 MHz; cycles / event-ms = %.3f GHz\n", mode ? "scalar" : "valu", blocks, threads,
                   ms, h[0], h[1], (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / (ms * 1e6));
        }
    return 0;
}
