// walk_thrash.hip -- the MEMORY side of the LSD region walks without their instructions: W waves, each on its own 512 x 384 map of 16-byte pixel records (3.1 MB), fetch an
// 8 x 8 window of records at a pseudo-random place every `period` microseconds (lsd_rg_seq: 14.5 k window fetches per frame and 105 ms, one per 7 us), write four 4-byte
// marks into it, and sleep in between.  Run beside `bench.py --no-lines` it tells whether the walks slow the other kernels through the memory system (L2 / TLB reach over
// 13 GB of records) or through the instructions they issue.   usage: walk_thrash [waves 2660] [frames 4096] [seconds 30] [period_us 7]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) thrash(float4 *pix, int frames, long iters, int period_ticks, unsigned long long *sink) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float4 *map = pix + (size_t)(wave % frames) * 512 * 384;
    unsigned s = 12345u + 977u * wave;
    float acc = 0;
    for (long it = 0; it < iters; it++) {
        const unsigned long long t0 = wall_clock64();
        s = s * 1664525u + 1013904223u;
        const int wx = (s >> 8) % 504, wy = (s >> 20) % 376;
        float4 *p = map + (size_t)(wy + (lane >> 3)) * 512 + wx + (lane & 7);
        const float4 v = *p;
        acc += v.x;
        s ^= __float_as_uint(__shfl(acc, 0)) & 0xffu; // the next window depends on what came back
        if (lane < 4) p->x = acc * 1e-30f;
        while (wall_clock64() - t0 < (unsigned long long)period_ticks) __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) atomicAdd(sink, (unsigned long long)acc);
}
int main(int argc, char **argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 2660, frames = argc > 2 ? atoi(argv[2]) : 4096;
    const double seconds = argc > 3 ? atof(argv[3]) : 30, period_us = argc > 4 ? atof(argv[4]) : 7;
    float4 *pix; unsigned long long *sink;
    const size_t n = (size_t)frames * 512 * 384;
    if (hipMalloc(&pix, n * sizeof(float4)) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(pix, 0, n * sizeof(float4)); hipMemset(sink, 0, 8);
    const long iters = (long)(seconds * 1e6 / period_us);
    hipStream_t st; int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);
    hipLaunchKernelGGL(thrash, dim3((waves + 3) / 4), dim3(256), 0, st, pix, frames, iters, (int)(period_us * 100), sink); // wall_clock64 ticks at 100 MHz
    fprintf(stderr, "[walk_thrash] %d waves on %d maps (%.1f GB), a window every %.1f us for %.0f s\n", waves, frames, n * 16 / 1e9, period_us, seconds);
    hipStreamSynchronize(st);
    fprintf(stderr, "[walk_thrash] done\n");
    return 0;
}
