// ctx.hip -- context lifecycle + timing queries of the C-ABI (include/cubeslam_hip.h)
#include "common.h"

#include <dlfcn.h>
#include <omp.h>
#include <sched.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

// CPUs the host stages (ORB quadtree, LSD region growing) may use: the affinity mask, capped by the cgroup CPU quota
// (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- running more threads than the quota only gets them throttled --
// divided by the number of ranks on the node (LOCAL_WORLD_SIZE), and overridable with CUBESLAM_HOST_THREADS.
int cs_host_threads() {
    if (const char *e = getenv("CUBESLAM_HOST_THREADS")) { int v = atoi(e); if (v > 0) return v; }
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    double quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lf", q, &period) == 2 && q[0] != 'm') quota = atof(q);
        fclose(f);
    } else {
        double qv = -1, pv = -1;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &qv) != 1) qv = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &pv) != 1) pv = -1; fclose(g); }
        if (qv > 0 && pv > 0) { quota = qv; period = pv; }
    }
    if (quota > 0 && period > 0) n = std::min(n, std::max(1, (int)std::floor(quota / period)));
    // one process per GPU (torchrun sets LOCAL_WORLD_SIZE): the ranks of a node share the host cores
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) { int w = atoi(e); if (w > 1) n = std::max(1, n / w); }
    return std::max(1, n);
}

// libomp workers spin for KMP_BLOCKTIME (200 ms by default) after a parallel region; under a cgroup CPU quota that idle spinning
// is charged to the quota and throttles the next host stage (measured: 2.1x on the line path).  Workers of teams forked by the
// calling thread go to sleep immediately instead.
void cs_omp_prepare() {
    static thread_local bool done = false;
    if (!done) { kmp_set_blocktime(0); done = true; }
}

// ---- RCCL (xGMI) inside the library.  librccl.so is opened at cs_comm_init, not linked: a single-GPU host needs no RCCL, and in a process
// that already holds a copy (PyTorch ships one) the same soname resolves to that copy instead of a second runtime.
namespace {
typedef struct { char internal[128]; } rccl_unique_id; // ncclUniqueId
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_unique_id, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (h) return true;
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        GetUniqueId = (int (*)(rccl_unique_id *))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(void **, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
        AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
        CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
        GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
    }
} g_rccl;
constexpr int RCCL_DOUBLE = 8, RCCL_SUM = 0; // ncclFloat64, ncclSum (nccl.h enums)
} // namespace

int cs_comm_allreduce_sum_f64(cs_ctx *ctx, double *device_buf, long n) {
    if (!ctx || !ctx->comm || n < 0) return CS_ERR_BAD_ARG;
    if (n == 0) return CS_OK;
    const int rc = g_rccl.AllReduce(device_buf, device_buf, (size_t)n, RCCL_DOUBLE, RCCL_SUM, ctx->comm, ctx->stream);
    if (rc != 0) { ctx->err = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return CS_ERR_HIP; }
    return CS_OK;
}

extern "C" {

int cs_comm_unique_id(void *id128) {
    if (!id128) return CS_ERR_BAD_ARG;
    if (!g_rccl.load()) return CS_ERR_NO_DEVICE;
    rccl_unique_id id;
    if (g_rccl.GetUniqueId(&id) != 0) return CS_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return CS_OK;
}
int cs_comm_init(cs_ctx *ctx, int rank, int world, const void *id128) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return CS_ERR_BAD_ARG;
    if (ctx->comm) return CS_ERR_BAD_ARG;
    if (!g_rccl.load()) { ctx->err = "librccl.so not found"; return CS_ERR_NO_DEVICE; }
    CS_HIP(ctx, hipSetDevice(ctx->device));
    rccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
    if (rc != 0) { ctx->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return CS_ERR_HIP; }
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return CS_OK;
}
int cs_comm_allreduce_f64(cs_ctx *ctx, double *device_buf, long n) { return cs_comm_allreduce_sum_f64(ctx, device_buf, n); }
void cs_comm_destroy(cs_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 1;
}

int cs_host_thread_count(void) { return cs_host_threads(); }

int cs_version(void) { return CS_VERSION; }

int cs_create_with_priority(int device_id, int priority, cs_ctx **out) {
    if (!out) return CS_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CS_ERR_NO_DEVICE; // no CPU fallback, by design
    if (device_id < 0 || device_id >= n) return CS_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return CS_ERR_NO_DEVICE;
    cs_ctx *c = new (std::nothrow) cs_ctx();
    if (!c) return CS_ERR_NOMEM;
    c->device = device_id;
    c->host_threads = cs_host_threads();
    int lo = 0, hi = 0; // numerically lower = higher priority
    hipError_t e = hipSuccess;
    if (priority != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
        e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority > 0 ? hi : lo);
    else
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); delete c; return CS_ERR_NO_DEVICE; }
    *out = c;
    return CS_OK;
}
int cs_create(int device_id, cs_ctx **out) { return cs_create_with_priority(device_id, 0, out); }

void cs_destroy(cs_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    ctx->flush();
    cs_comm_destroy(ctx);
    for (auto e : ctx->pool) hipEventDestroy(e);
    ctx->pool_drop();
    if (ctx->bg_stream) { hipStreamSynchronize(ctx->bg_stream); hipStreamDestroy(ctx->bg_stream); hipEventDestroy(ctx->bg_in); hipEventDestroy(ctx->bg_out); }
    if (ctx->aux_stream) { hipStreamSynchronize(ctx->aux_stream); hipStreamDestroy(ctx->aux_stream); }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *cs_last_error(const cs_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int cs_sync(cs_ctx *ctx) {
    if (!ctx) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_timing_enable(cs_ctx *ctx, int on) {
    if (!ctx) return CS_ERR_BAD_ARG;
    ctx->flush();
    ctx->timing = on != 0;
    return CS_OK;
}
int cs_timing_reset(cs_ctx *ctx) {
    if (!ctx) return CS_ERR_BAD_ARG;
    ctx->flush();
    ctx->timings.clear();
    return CS_OK;
}
int cs_timing_get(cs_ctx *ctx, const char *name, double *total_ms, long *count) {
    if (!ctx || !name) return CS_ERR_BAD_ARG;
    ctx->flush();
    auto it = ctx->timings.find(name);
    if (total_ms) *total_ms = it == ctx->timings.end() ? 0.0 : it->second.total_ms;
    if (count) *count = it == ctx->timings.end() ? 0 : it->second.count;
    return CS_OK;
}

} // extern "C"
