// frontend.hip -- batch front-end runner of the C-ABI (include/cubeslam_hip.h, cs_frontend_*).
//
// One pass of the per-frame path over a batch of frames that is resident in HBM = ORBextractor + detect_3d_cuboid on the caller's
// thread and stream, LSD + LBD on worker threads with their own streams.  The line path has a host stage (LSD region growing,
// lsd.hip) between two GPU phases; two line detectors alternate passes so that the GPU phases of one pass and the ORB / cuboid
// kernels run while the other pass grows regions (the host stages themselves are serialised inside lsd.hip).  At most one pass
// per worker is in flight; cs_frontend_step blocks until the worker it needs is free.
//
// Phased mode (cs_frontend_set_phased): the device region stage of LSD (lsd_rg_seq, one wave per frame, 16 frames per CU) and the
// cuboid score kernel (one workgroup owns a CU's LDS) do not share a CU well, so the runner separates them in time.  A super-step =
// one pass per line worker: the workers run their map kernels beside ORB / cuboid of the same passes and stop at a gate in front of
// the region stage; after the last pass of the super-step the caller's stream is idle, the gate opens, the region stages of all
// workers fill the chip together, and cs_frontend_step returns when they have left the GPU (the workers go on with rectangles,
// KeyLines and LBD beside the next super-step).
#include "common.h"

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" int cs_lsd_run(cs_ctx *ctx, cs_lsd *l, int with_lbd);
void cs_lsd_set_gate(cs_lsd *l, void (*wait)(void *), void (*done)(void *), void *arg); // lsd.hip
void cs_lsd_set_shared_gpu(cs_lsd *l, int shared);                                                // lsd.hip
extern "C" int cs_orb_run(cs_ctx *ctx, cs_orb *e);
extern "C" int cs_cuboid_batch_run(cs_ctx *ctx, cs_cuboid_batch *b);
extern "C" int cs_cuboid_batch_set_shared_gpu(cs_cuboid_batch *b, int shared);
extern "C" int cs_cuboid_batch_set_lines(cs_ctx *ctx, cs_cuboid_batch *b, const int *line_offsets, const double *lines);
int cs_lsd_filter_lines_packed(cs_lsd *l, float length_thres, std::vector<int> &offsets, std::vector<double> &lines); // lsd.hip

namespace {
struct Gate { // phase gate of one runner: tickets are pass numbers, the gate is open for every ticket <= target
    std::mutex m; std::condition_variable cv;
    bool phased = false;
    long submitted = 0, target = 0, done = 0;
};
struct LineWorker {
    cs_ctx *ctx = nullptr; cs_lsd *lsd = nullptr;
    Gate *gate = nullptr; long ticket = 0; bool marked = true; // (ticket, marked) belong to the pass in flight, guarded by gate->m
    static void gate_wait(void *arg) {
        LineWorker *w = (LineWorker *)arg; Gate *g = w->gate;
        std::unique_lock<std::mutex> lk(g->m);
        if (w->marked) return; // not a phased pass
        g->cv.wait(lk, [&] { return w->ticket <= g->target; });
    }
    static void gate_done(void *arg) {
        LineWorker *w = (LineWorker *)arg; Gate *g = w->gate;
        std::lock_guard<std::mutex> lk(g->m);
        if (w->marked) return;
        w->marked = true; g->done++;
        g->cv.notify_all();
    }
    std::thread th;
    std::mutex m; std::condition_variable cv;
    std::mutex *any_m = nullptr; std::condition_variable *any_cv = nullptr; // the runner's "a worker has finished" signal
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
            gate_done(this); // a pass that failed, fell back to the host stage or had nothing to grow never reached the device stage's own call
            lk.lock();
            if (r != CS_OK && last_status == CS_OK) last_status = r;
            busy = false;
            cv.notify_all();
            lk.unlock();
            if (any_m) { std::lock_guard<std::mutex> g(*any_m); any_cv->notify_all(); }
        }
    }
    bool is_free() { std::lock_guard<std::mutex> lk(m); return !busy; }
    void submit() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
        {
            std::lock_guard<std::mutex> gl(gate->m);
            if (gate->phased) { ticket = ++gate->submitted; marked = false; } else marked = true;
        }
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
    cs_ctx *cub_ctx = nullptr; // cs_frontend_set_cuboid_ctx: the context (stream) the cuboid batch is enqueued on -- the caller's own unless set
    std::vector<LineWorker *> workers;
    unsigned long step_no = 0;
    Gate gate;
    std::mutex any_m; std::condition_variable any_cv; // a worker has finished a pass
    size_t next_worker = 0;
    unsigned long in_phase = 0; // passes submitted since the gate last opened
    // chained mode (cs_frontend_set_chain): detect_cuboid of a pass is fed the lines detect_filter_lines found in the pass the same worker finished
    // last -- the reference's chain (main_obj.cpp:428-449), pipelined: the line pass of a batch runs W steps ahead of the batch's cuboid pass
    bool chain = false; float chain_thres = 0;
    std::vector<char> worker_ran; std::vector<int> chain_off; std::vector<double> chain_lines;
    int open_gate() { // caller's stream idle -> region stages of every waiting pass -> return when they have left the GPU
        const int r = hipStreamSynchronize(ctx->stream) == hipSuccess ? CS_OK : CS_ERR_HIP; // (the gate opens either way: a waiting pass must not be left behind)
        std::unique_lock<std::mutex> lk(gate.m);
        gate.target = gate.submitted;
        gate.cv.notify_all();
        gate.cv.wait(lk, [&] { return gate.done == gate.target; });
        in_phase = 0;
        return r;
    }
};

extern "C" {

int cs_frontend_create(cs_ctx *ctx, cs_orb *orb, cs_cuboid_batch *batch, int n_line_workers, cs_ctx *const *line_ctx, cs_lsd *const *lsd, cs_frontend **out) {
    if (!ctx || !out || n_line_workers < 0 || n_line_workers > 8 || (n_line_workers && (!line_ctx || !lsd))) return CS_ERR_BAD_ARG;
    for (int i = 0; i < n_line_workers; i++) if (!line_ctx[i] || !lsd[i] || line_ctx[i] == ctx) return CS_ERR_BAD_ARG; // a worker needs its own context (stream, timing records)
    cs_frontend *fe = new (std::nothrow) cs_frontend();
    if (!fe) return CS_ERR_NOMEM;
    fe->ctx = ctx; fe->orb = orb; fe->batch = batch;
    for (int i = 0; i < n_line_workers; i++) {
        LineWorker *w = new LineWorker();
        w->ctx = line_ctx[i]; w->lsd = lsd[i]; w->gate = &fe->gate; w->any_m = &fe->any_m; w->any_cv = &fe->any_cv;
        cs_lsd_set_gate(w->lsd, LineWorker::gate_wait, LineWorker::gate_done, w);
        w->th = std::thread([w] { w->loop(); });
        fe->workers.push_back(w);
    }
    if (fe->batch) cs_cuboid_batch_set_shared_gpu(fe->batch, n_line_workers > 0); // alternating runner: the detectors' region walks hold most CUs all the time
    for (LineWorker *w : fe->workers) cs_lsd_set_shared_gpu(w->lsd, n_line_workers > 1);
    *out = fe;
    return CS_OK;
}

int cs_frontend_step(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    int r = CS_OK;
    if (!fe->workers.empty()) {
        // the pass goes to the first worker that is free, looked for from the one behind the last choice: passes take 110 - 220 ms beside each other, and a strict
        // rotation made the caller wait for a slow one with idle ones beside it (phased passes keep the rotation: their gate counts one pass per detector)
        size_t wi = fe->step_no % fe->workers.size();
        if (!fe->gate.phased) {
            std::unique_lock<std::mutex> lk(fe->any_m);
            for (;;) {
                bool found = false;
                for (size_t k = 0; k < fe->workers.size(); k++) {
                    const size_t c = (fe->next_worker + k) % fe->workers.size();
                    if (fe->workers[c]->is_free()) { wi = c; found = true; break; }
                }
                if (found) break;
                fe->any_cv.wait_for(lk, std::chrono::milliseconds(2));
            }
            fe->next_worker = (wi + 1) % fe->workers.size();
        }
        LineWorker *w = fe->workers[wi];
        if (fe->chain && fe->batch && fe->worker_ran.size() > wi && fe->worker_ran[wi]) { // the pass this worker ran W steps ago: its lines are this step's edges
            r = w->wait();
            if (r == CS_OK) r = cs_lsd_filter_lines_packed(w->lsd, fe->chain_thres, fe->chain_off, fe->chain_lines);
            if (r == CS_OK) r = cs_cuboid_batch_set_lines(fe->cub_ctx ? fe->cub_ctx : fe->ctx, fe->batch, fe->chain_off.data(), fe->chain_lines.data());
            if (r != CS_OK) return r;
        }
        w->submit();
        if (fe->worker_ran.size() <= wi) fe->worker_ran.resize(fe->workers.size(), 0);
        fe->worker_ran[wi] = 1;
    }
    fe->step_no++;
    // the cuboid pass is a chain of launches without a host round trip: on a stream of its own (cs_frontend_set_cuboid_ctx) it is enqueued first and runs beside the ORB
    // pass, whose two read-backs would otherwise wait behind it
    if (fe->batch && fe->cub_ctx) r = cs_cuboid_batch_run(fe->cub_ctx, fe->batch);
    if (r == CS_OK && fe->orb) r = cs_orb_run(fe->ctx, fe->orb);
    if (r == CS_OK && fe->batch && !fe->cub_ctx) r = cs_cuboid_batch_run(fe->ctx, fe->batch);
    if (fe->gate.phased && !fe->workers.empty() && ++fe->in_phase >= fe->workers.size()) { const int g = fe->open_gate(); if (r == CS_OK) r = g; }
    return r;
}

int cs_frontend_set_phased(cs_frontend *fe, int on) {
    if (!fe) return CS_ERR_BAD_ARG;
    int r = cs_frontend_drain(fe); // no pass in flight across the switch
    std::lock_guard<std::mutex> lk(fe->gate.m);
    fe->gate.phased = on != 0;
    if (fe->batch) cs_cuboid_batch_set_shared_gpu(fe->batch, !on && !fe->workers.empty()); // phased: the score kernel never meets a region walk
    for (LineWorker *w : fe->workers) cs_lsd_set_shared_gpu(w->lsd, !on && fe->workers.size() > 1); // phased: sixteen frames per CU, the walks packed
    return r;
}

// on != 0: from now on a step's cuboid pass takes its edge lists from the line pass the step's worker finished last (filter_lines with `length_thres`,
// main_obj.cpp:366) -- with W workers the lines of a batch are ready W steps before its cuboids are asked for, and nothing waits.  The first W steps
// after the switch still run on the lists the batch was created with.  Needs a batch and at least one line worker.
int cs_frontend_set_chain(cs_frontend *fe, int on, float length_thres) {
    if (!fe || (on && (!fe->batch || fe->workers.empty()))) return CS_ERR_BAD_ARG;
    const int r = cs_frontend_drain(fe);
    fe->chain = on != 0; fe->chain_thres = length_thres;
    fe->worker_ran.assign(fe->workers.size(), 0);
    return r;
}

// ORB and the cuboid batch are independent: with a second context the batch's launches go to that context's stream and overlap the ORB pass of the same step (NULL: back
// onto the caller's stream).  cs_frontend_drain then also waits for that stream.  The batch may have been created on any context of the same device.
int cs_frontend_set_cuboid_ctx(cs_frontend *fe, cs_ctx *cuboid_ctx) {
    if (!fe || cuboid_ctx == fe->ctx) return CS_ERR_BAD_ARG;
    for (LineWorker *w : fe->workers) if (w->ctx == cuboid_ctx) return CS_ERR_BAD_ARG;
    if (fe->cub_ctx && hipStreamSynchronize(fe->cub_ctx->stream) != hipSuccess) return CS_ERR_HIP;
    if (hipStreamSynchronize(fe->ctx->stream) != hipSuccess) return CS_ERR_HIP; // no cuboid pass in flight across the switch
    fe->cub_ctx = cuboid_ctx;
    return CS_OK;
}

int cs_frontend_drain(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    int r = CS_OK;
    if (fe->cub_ctx && hipStreamSynchronize(fe->cub_ctx->stream) != hipSuccess) r = CS_ERR_HIP;
    if (fe->in_phase) r = fe->open_gate(); // an incomplete super-step
    for (LineWorker *w : fe->workers) { const int s = w->wait(); if (r == CS_OK) r = s; }
    return r;
}

void cs_frontend_destroy(cs_frontend *fe) {
    if (!fe) return;
    if (fe->in_phase) fe->open_gate();
    for (LineWorker *w : fe->workers) {
        w->wait();
        cs_lsd_set_gate(w->lsd, nullptr, nullptr, nullptr);
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        w->th.join();
        delete w;
    }
    delete fe;
}

} // extern "C"
