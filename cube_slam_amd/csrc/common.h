// common.h -- context, error handling and per-kernel hipEvent timing shared by all paths of libcubeslam_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/cubeslam_hip.h"

struct cs_timing_rec {
    double total_ms = 0;
    long count = 0;
};

int cs_host_threads(); // ctx.hip
struct cs_ctx;
int cs_comm_allreduce_sum_f64(cs_ctx *ctx, double *device_buf, long n); // ctx.hip: ncclAllReduce(sum, double) in place on the context's stream
void cs_omp_prepare();
// orb.hip: device-resident results of the last cs_orb_run for one frame (keypoints in mvKeys order, 4 x u64 descriptors), for
// consumers inside the library that must not round-trip through the host (match.hip)
struct cs_orb;
int cs_orb_device_frame(const cs_orb *e, int frame, const cs_keypoint **d_kps, const unsigned long long **d_desc, int *n);  // ctx.hip: call before an OpenMP region of a host stage

struct cs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool timing = false;
    int host_threads = 1; // CPUs this process may really use (affinity mask and cgroup CPU quota), see cs_host_threads()
    void *comm = nullptr;  // RCCL communicator of this rank (cs_comm_init, ctx.hip); collectives run on `stream`
    int comm_rank = 0, comm_world = 1;
    std::map<std::string, cs_timing_rec> timings;
    struct pending_ev { std::string name; hipEvent_t a, b; };
    std::vector<pending_ev> pending;
    std::vector<hipEvent_t> pool;
    // Device blocks kept for reuse while `pooling` is on (the per-frame drop-in calls build and drop a whole batch per call: ~35 hipMalloc / hipFree,
    // which the driver serialises across threads).  Sizes are rounded up to a quarter-octave bucket; at most POOL_CAP bytes stay cached.
    bool pooling = false;
    std::map<void *, size_t> pool_live;
    std::multimap<size_t, void *> pool_free;
    size_t pool_cached = 0;
    static size_t pool_bucket(size_t n) {
        size_t b = 4096;
        while (b < n) b <<= 1;                       // next power of two
        const size_t q = b >> 3;                     // ... refined downwards in eighths of it
        while (b - q >= n && b - q > (b >> 1)) b -= q;
        return b;
    }
    hipError_t pool_alloc(void **p, size_t n) {
        const size_t b = pool_bucket(n);
        auto it = pool_free.find(b);
        if (it != pool_free.end()) { *p = it->second; pool_free.erase(it); pool_cached -= b; pool_live[*p] = b; return hipSuccess; }
        hipError_t e = hipMalloc(p, b);
        if (e != hipSuccess && !pool_free.empty()) { (void)hipGetLastError(); pool_drop(); e = hipMalloc(p, b); } // out of memory with blocks of other sizes cached: give those back and ask again
        if (e == hipSuccess) pool_live[*p] = b;
        return e;
    }
    void pool_release(void *p) { // a block of the pool goes back to it (or to the driver when the cache is full); anything else is freed
        auto it = pool_live.find(p);
        if (it == pool_live.end()) { hipFree(p); return; }
        const size_t b = it->second;
        pool_live.erase(it);
        constexpr size_t POOL_CAP = (size_t)1 << 30;
        if (pool_cached + b > POOL_CAP) { hipFree(p); return; }
        pool_free.emplace(b, p); pool_cached += b;
    }
    void pool_drop() { for (auto &kv : pool_free) hipFree(kv.second); pool_free.clear(); pool_cached = 0; }

    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
    void begin(const char *name) {
        if (!timing) return;
        pending_ev p; p.name = name; p.a = get_event(); p.b = get_event();
        hipEventRecord(p.a, stream);
        pending.push_back(p);
    }
    void end() {
        if (!timing) return;
        hipEventRecord(pending.back().b, stream);
    }
    // A second, lowest-priority stream of this context for ONE kind of launch: a kernel that holds its wave slots for ~100 ms and paces itself (the LSD region
    // walk).  Everything else of the context then runs at the context's own priority beside it, ordered against it by two events.
    hipStream_t bg_stream = nullptr; hipEvent_t bg_in = nullptr, bg_out = nullptr;
    hipStream_t aux_stream = nullptr; // a second stream at the context's priority for a branch of a launch chain that reads nothing of the other branch (cs_cuboid_batch_run); created on first use
    hipError_t bg_begin() { // work queued on `stream` so far precedes what is queued on bg_stream from here on
        hipError_t e = hipSuccess;
        if (!bg_stream) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) e = hipStreamCreateWithPriority(&bg_stream, hipStreamNonBlocking, lo);
            else e = hipStreamCreateWithFlags(&bg_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&bg_in, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&bg_out, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        e = hipEventRecord(bg_in, stream);
        return e == hipSuccess ? hipStreamWaitEvent(bg_stream, bg_in, 0) : e;
    }
    hipError_t bg_end() { // ... and `stream` goes on behind it
        const hipError_t e = hipEventRecord(bg_out, bg_stream);
        return e == hipSuccess ? hipStreamWaitEvent(stream, bg_out, 0) : e;
    }
    void flush() {
        if (pending.empty()) return;
        hipStreamSynchronize(stream);
        for (auto &p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { auto &r = timings[p.name]; r.total_ms += ms; r.count++; }
            pool.push_back(p.a); pool.push_back(p.b);
        }
        pending.clear();
    }
};

#define CS_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            char b__[512];                                                                        \
            snprintf(b__, sizeof b__, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            (ctx)->err = b__;                                                                     \
            return CS_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

// Launch helper: named, timed when ctx->timing is on.
#define CS_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                         \
    do {                                                                              \
        (ctx)->begin(name);                                                           \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);   \
        (ctx)->end();                                                                 \
    } while (0)

template <class T> static inline int cs_dalloc(cs_ctx *ctx, T **p, size_t n) {
    if (n == 0) n = 1;
    if (ctx->pooling) CS_HIP(ctx, ctx->pool_alloc((void **)p, n * sizeof(T)));
    else CS_HIP(ctx, hipMalloc((void **)p, n * sizeof(T)));
    return CS_OK;
}
static inline void cs_dfree(cs_ctx *ctx, void *p) { if (!p) return; if (ctx && !ctx->pool_live.empty()) ctx->pool_release(p); else hipFree(p); }
template <class T> static inline int cs_h2d(cs_ctx *ctx, T *d, const T *h, size_t n) {
    if (n == 0) return CS_OK;
    CS_HIP(ctx, hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return CS_OK;
}
template <class T> static inline int cs_d2h(cs_ctx *ctx, T *h, const T *d, size_t n) {
    if (n == 0) return CS_OK;
    CS_HIP(ctx, hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return CS_OK;
}

// ---- device helpers -------------------------------------------------------------------------------------------
// 64-lane inclusive min-scan with DPP (row_shr 1/2/4/8 inside 16-lane rows, then row_bcast:15 / row_bcast:31).
__device__ __forceinline__ int wave_incl_min_scan(int x) {
    const int id = 0x7fffffff;
    int t;
    t = __builtin_amdgcn_update_dpp(id, x, 0x111, 0xf, 0xf, false); x = min(x, t); // row_shr:1
    t = __builtin_amdgcn_update_dpp(id, x, 0x112, 0xf, 0xf, false); x = min(x, t); // row_shr:2
    t = __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xf, false); x = min(x, t); // row_shr:4
    t = __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xf, false); x = min(x, t); // row_shr:8
    t = __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false); x = min(x, t); // row_bcast:15 -> rows 1,3
    t = __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false); x = min(x, t); // row_bcast:31 -> rows 2,3
    return x;
}
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Four neighbouring pixels of a u8 row with one load.  gfx950 runs global memory in unaligned-access mode (a dword load may start at
// any byte; tools/ubench/unaligned_load.hip checks it on the device), and one dword gather costs the texture-address unit a quarter
// of four byte gathers.
typedef uint32_t __attribute__((aligned(1))) cs_u32_unaligned;
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) { return *reinterpret_cast<const cs_u32_unaligned *>(p); }

// four consecutive bytes of the 12-byte window w0 w1 w2 starting at byte START (0..8), as one dword
template <int START> __device__ __forceinline__ uint32_t win4(uint32_t w0, uint32_t w1, uint32_t w2) {
    if (START == 0) return w0;
    if (START < 4) return __builtin_amdgcn_alignbyte(w1, w0, START);
    if (START == 4) return w1;
    if (START < 8) return __builtin_amdgcn_alignbyte(w2, w1, START - 4);
    return w2;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the global-memory counter (s_waitcnt vmcnt(0)), so
// in a loop that stores a result row / column to global memory and prefetches the next one, every barrier would wait for those
// round trips (measured: 5.4 us per column in ba_band_chol, 3.5 us per row in orb_blur).  Only valid where no thread reads,
// through a cached path, global data another thread of the workgroup wrote before the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
