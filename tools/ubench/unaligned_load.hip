#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t __attribute__((aligned(1))) u32u;
__global__ void k(const uint8_t *p, uint32_t *out, int shift) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    out[t] = *reinterpret_cast<const u32u *>(p + shift + 4 * t);
}
int main() {
    std::vector<uint8_t> h(4096 + 16); for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o; hipMalloc(&d, h.size()); hipMalloc(&o, 1024 * 4); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int s = 0; s < 4; s++) {
        hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, d, o, s);
        std::vector<uint32_t> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        for (int t = 0; t < 1024; t++) { uint32_t e = 0; for (int b = 0; b < 4; b++) e |= (uint32_t)h[s + 4 * t + b] << (8 * b); if (e != r[t]) bad++; }
    }
    printf("unaligned dword loads: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
}
