// Gather-rate probe: CU-cycles per wave64 global_load_dword as a function of how the 64 lanes share memory granules.
// Addresses are generated arithmetically (no second gather).  Footprint 256 KB (ROI-sized, L2-resident, mostly missing the 32 KB L1).
#include <hip/hip_runtime.h>
#include <cstdio>
// G lanes (adjacent) share one granule of `gran` bytes; inside the granule lanes pick dword (lane % (gran/4)) (or all the same dword if same_dword)
__global__ void __launch_bounds__(256) k(const float *tab, int n_gran, int G, int gran_dwords, int same_dword, int iters, float *out) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int grp = lane / G;
    const int within = same_dword ? 0 : (lane % G) % gran_dwords;
    float acc = 0;
    unsigned st = wave * 977u + 13u + grp * 7919u;
    for (int it = 0; it < iters; it++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            st = st * 1664525u + 1013904223u;
            const unsigned g = (st >> 10) % (unsigned)n_gran;
            v[j] = tab[g * gran_dwords + within];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 256 << 10;
    float *tab; float *out;
    hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes);
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 300;
    struct P { int G, gd, same; const char *name; };
    P ps[] = {{1, 1, 0, "64 distinct dwords (random lines)"}, {2, 32, 0, "2 lanes / 128B line"}, {4, 32, 0, "4 lanes / 128B line"}, {8, 32, 0, "8 lanes / 128B line"},
              {16, 32, 0, "16 lanes / 128B line"}, {2, 16, 0, "2 lanes / 64B"}, {4, 16, 0, "4 lanes / 64B"}, {8, 16, 0, "8 lanes / 64B"}, {16, 16, 0, "16 lanes / 64B"},
              {4, 8, 0, "4 lanes / 32B"}, {8, 8, 0, "8 lanes / 32B"}, {2, 1, 1, "2 lanes same dword"}, {4, 1, 1, "4 lanes same dword"}, {16, 1, 1, "16 lanes same dword"}, {64, 1, 1, "64 lanes same dword"},
              {64, 32, 0, "64 lanes in one 128B line"}};
    for (auto &p : ps) {
        const int n_gran = (int)(bytes / 4 / p.gd);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, n_gran, p.G, p.gd, p.same, 10, out);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, n_gran, p.G, p.gd, p.same, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double loads_per_cu = (double)iters * 8 * blocks * 4 / 256;
        printf("%-36s %6.1f CU-cycles per wave-load (2.1 GHz)\n", p.name, ms * 1e-3 * 2.1e9 / loads_per_cu);
    }
    return 0;
}
