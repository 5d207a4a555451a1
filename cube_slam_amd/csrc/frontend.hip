// frontend.hip -- batch front-end runner of the C-ABI (include/cubeslam_hip.h, cs_frontend_*).
//
// One pass of the per-frame path over a batch of frames that is resident in HBM = ORBextractor + detect_3d_cuboid on the caller's
// thread and stream, LSD + LBD on worker threads with their own streams.  The line path has a host stage (LSD region growing,
// lsd.hip) between two GPU phases; two line detectors alternate passes so that the GPU phases of one pass and the ORB / cuboid
// kernels run while the other pass grows regions (the host stages themselves are serialised inside lsd.hip).  At most one pass
// per worker is in flight; cs_frontend_step blocks until the worker it needs is free.
#include "common.h"

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" int cs_lsd_run(cs_ctx *ctx, cs_lsd *l, int with_lbd);
extern "C" int cs_orb_run(cs_ctx *ctx, cs_orb *e);
extern "C" int cs_cuboid_batch_run(cs_ctx *ctx, cs_cuboid_batch *b);

namespace {
struct LineWorker {
    cs_ctx *ctx = nullptr; cs_lsd *lsd = nullptr;
    std::thread th;
    std::mutex m; std::condition_variable cv;
    bool busy = false, have_job = false, quit = false;
    int last_status = CS_OK;
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return have_job || quit; });
            if (quit) return;
            have_job = false;
            lk.unlock();
            const int r = cs_lsd_run(ctx, lsd, 1);
            lk.lock();
            if (r != CS_OK && last_status == CS_OK) last_status = r;
            busy = false;
            cv.notify_all();
        }
    }
    void submit() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
        busy = true; have_job = true;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
        const int r = last_status; last_status = CS_OK;
        return r;
    }
};
} // namespace

struct cs_frontend {
    cs_ctx *ctx = nullptr; cs_orb *orb = nullptr; cs_cuboid_batch *batch = nullptr;
    std::vector<LineWorker *> workers;
    unsigned long step_no = 0;
};

extern "C" {

int cs_frontend_create(cs_ctx *ctx, cs_orb *orb, cs_cuboid_batch *batch, int n_line_workers, cs_ctx *const *line_ctx, cs_lsd *const *lsd, cs_frontend **out) {
    if (!ctx || !out || n_line_workers < 0 || n_line_workers > 4 || (n_line_workers && (!line_ctx || !lsd))) return CS_ERR_BAD_ARG;
    for (int i = 0; i < n_line_workers; i++) if (!line_ctx[i] || !lsd[i] || line_ctx[i] == ctx) return CS_ERR_BAD_ARG; // a worker needs its own context (stream, timing records)
    cs_frontend *fe = new (std::nothrow) cs_frontend();
    if (!fe) return CS_ERR_NOMEM;
    fe->ctx = ctx; fe->orb = orb; fe->batch = batch;
    for (int i = 0; i < n_line_workers; i++) {
        LineWorker *w = new LineWorker();
        w->ctx = line_ctx[i]; w->lsd = lsd[i];
        w->th = std::thread([w] { w->loop(); });
        fe->workers.push_back(w);
    }
    *out = fe;
    return CS_OK;
}

int cs_frontend_step(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    if (!fe->workers.empty()) fe->workers[fe->step_no % fe->workers.size()]->submit();
    fe->step_no++;
    int r = CS_OK;
    if (fe->orb) r = cs_orb_run(fe->ctx, fe->orb);
    if (r == CS_OK && fe->batch) r = cs_cuboid_batch_run(fe->ctx, fe->batch);
    return r;
}

int cs_frontend_drain(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    int r = CS_OK;
    for (LineWorker *w : fe->workers) { const int s = w->wait(); if (r == CS_OK) r = s; }
    return r;
}

void cs_frontend_destroy(cs_frontend *fe) {
    if (!fe) return;
    for (LineWorker *w : fe->workers) {
        w->wait();
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        w->th.join();
        delete w;
    }
    delete fe;
}

} // extern "C"
