// ctx.hip -- context lifecycle + timing queries of the C-ABI (include/cubeslam_hip.h)
#include "common.h"

extern "C" {

int cs_version(void) { return CS_VERSION; }

int cs_create(int device_id, cs_ctx **out) {
    if (!out) return CS_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CS_ERR_NO_DEVICE; // no CPU fallback, by design
    if (device_id < 0 || device_id >= n) return CS_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return CS_ERR_NO_DEVICE;
    cs_ctx *c = new (std::nothrow) cs_ctx();
    if (!c) return CS_ERR_NOMEM;
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CS_ERR_NO_DEVICE; }
    *out = c;
    return CS_OK;
}

void cs_destroy(cs_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    ctx->flush();
    for (auto e : ctx->pool) hipEventDestroy(e);
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
