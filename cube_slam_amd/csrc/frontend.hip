// frontend.hip -- batch front-end runner of the C-ABI (include/cubeslam_hip.h, cs_frontend_*).
//
// One pass of the per-frame path over a batch of frames that is resident in HBM = ORBextractor + detect_3d_cuboid on the caller's
// thread and stream, LSD + LBD on worker threads with their own streams.  The line path has a host stage (LSD region growing,
// lsd.hip) between two GPU phases; two line detectors alternate passes so that the GPU phases of one pass and the ORB / cuboid
// kernels run while the other pass grows regions (the host stages themselves are serialised inside lsd.hip).  At most one pass
// per worker is in flight; cs_frontend_step blocks until the worker it needs is free.
//
// Passes are numbered: line pass k belongs to step k.  cs_frontend_step(k) makes sure pass k has been started, no more; with a BACKLOG announced
// (cs_frontend_set_backlog: "n more steps will follow on these frames") a worker that finishes a pass takes the next pass of the backlog at once, up to
// 2 W passes ahead of the caller, instead of waiting for the step that asks for it -- the line pipeline fills at the first step and does not drain behind the
// last one.  Chained mode hands pass k's lines to step k + W through a queue of packets keyed by the pass number.
//
// Phased mode (cs_frontend_set_phased): the device region stage of LSD (lsd_rg_seq, one wave per frame, 16 frames per CU) and the
// cuboid score kernel (one workgroup owns a CU's LDS) do not share a CU well, so the runner separates them in time.  A super-step =
// one pass per line worker: the workers run their map kernels beside ORB / cuboid of the same passes and stop at a gate in front of
// the region stage; after the last pass of the super-step the caller's stream is idle, the gate opens, the region stages of all
// workers fill the chip together, and cs_frontend_step returns when they have left the GPU (the workers go on with rectangles,
// KeyLines and LBD beside the next super-step).
#include "common.h"

#include <condition_variable>
#include <map>
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
extern "C" int cs_cuboid_batch_n_frames(const cs_cuboid_batch *b);
extern "C" int cs_cuboid_batch_set_gray_device(cs_ctx *ctx, cs_cuboid_batch *b, const uint8_t *d_gray);
extern "C" int cs_cuboid_batch_set_scene(cs_ctx *ctx, cs_cuboid_batch *b, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines);
extern "C" int cs_orb_set_frames_device(cs_ctx *ctx, cs_orb *e, const uint8_t *d_gray, int n_frames);
extern "C" int cs_orb_geometry(const cs_orb *e, int *width, int *height, int *max_frames);
extern "C" int cs_lsd_geometry(const cs_lsd *l, int *width, int *height, int *max_frames);
extern "C" int cs_cuboid_batch_geometry(const cs_cuboid_batch *b, int *width, int *height, int *n_frames);
extern "C" int cs_lsd_set_frames_device(cs_ctx *ctx, cs_lsd *l, const uint8_t *d_gray, int n_frames);
extern "C" int cs_orb_read_packed_on(cs_orb *e, void *stream, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total);
extern "C" int cs_cuboid_batch_read_on(cs_cuboid_batch *b, void *stream, cs_cuboid *out, int *counts, int *status_out);
int cs_lsd_filter_lines_packed(cs_lsd *l, float length_thres, std::vector<int> &offsets, std::vector<double> &lines); // lsd.hip

struct cs_frontend;
namespace {
struct LinePacket { std::vector<int> off; std::vector<double> lines; int status = CS_OK; }; // detect_filter_lines of one pass, packed per frame (chained mode)
struct Gate { // phase gate of one runner: tickets are pass numbers, the gate is open for every ticket <= target
    std::mutex m; std::condition_variable cv;
    bool phased = false;
    long submitted = 0, target = 0, done = 0;
};
// Streaming source (cs_frontend_stream_*): the frames of step k arrive from the host while step k - 1 computes.  A ring of device slots filled by a copy stream of its own;
// slot k % n feeds ORB and the cuboid batch of step k (the caller's stream) and line pass k (its worker's stream), each by a device-to-device copy behind the slot's
// `uploaded` event; a slot is refilled behind the events its consumers recorded -- and, on the host, once line pass k has enqueued its copy.
struct FrameRing {
    int n_slots = 0, n_frames = 0; size_t bytes = 0;
    hipStream_t copy = nullptr, copy_out = nullptr; // H2D of the frames; D2H of a step's results (cs_frontend_stream_read_async)
    hipEvent_t step_done = nullptr, cub_done = nullptr, results_out = nullptr; bool results_pending = false, status_read = false; int *h_status = nullptr;
    std::vector<uint8_t *> d;
    std::vector<hipEvent_t> uploaded, used_main, used_line;
    std::vector<long> main_gen, line_gen; // the step / pass whose copy out of the slot has been enqueued (-1: none yet)
    struct Scene { bool has = false, has_lines = false; std::vector<double> Twc, boxes, lines; std::vector<int> box_off, line_off; }; // what a step's frames bring besides pixels (cs_frontend_stream_push_scene)
    std::vector<Scene> scene;
    std::mutex m; std::condition_variable cv;
    long pushed = 0, first = 0, first_line = 0; // steps first .. pushed - 1 have frames; line passes from first_line on take theirs from the ring (a cut backlog may have left earlier passes done)
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
    cs_frontend *fe = nullptr; long pass_no = -1; // the runner and the number of the pass in flight (set by whoever starts it, under fe->any_m)
    int take_frames();
    bool after_pass(int r); // hands the pass's lines over (chained mode); true: this worker has taken the next pass of the backlog
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return have_job || quit; });
            if (quit) return;
            have_job = false;
            lk.unlock();
            int r;
            do {
                r = take_frames(); // (streaming source: this pass's frames out of their ring slot)
                if (r == CS_OK) r = cs_lsd_run(ctx, lsd, 1);
                gate_done(this); // a pass that failed, fell back to the host stage or had nothing to grow never reached the device stage's own call
                if (r != CS_OK) { std::lock_guard<std::mutex> g(m); if (last_status == CS_OK) last_status = r; }
            } while (after_pass(r));
            lk.lock();
            busy = false;
            cv.notify_all();
            lk.unlock();
            if (any_m) { std::lock_guard<std::mutex> g(*any_m); any_cv->notify_all(); }
        }
    }
    bool is_free() { std::lock_guard<std::mutex> lk(m); return !busy; }
    void submit(long no) { // `no`: the number of the pass; taken over only when the pass before has left the worker (a phased worker still runs rectangles / LBD of pass k - W when
                           // step k hands over the next one: after_pass of that earlier pass must file ITS packet under ITS number)
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
        pass_no = no;
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
    int streams = 1, hw_queues = 4; // cs_frontend_queues
    FrameRing *ring = nullptr;      // cs_frontend_stream_begin
    Gate gate;
    // guarded by any_m: the step counter, the pass counters and the packets of chained mode.  Line pass k belongs to step k: passes_started > k once it has been
    // handed to a worker; passes_target: passes [0, passes_target) may be started (a step raises it to its own number + 1, cs_frontend_set_backlog further)
    std::mutex any_m; std::condition_variable any_cv; // a worker has finished a pass
    long step_no = 0, passes_started = 0, passes_target = 0;
    size_t next_worker = 0;
    unsigned long in_phase = 0; // passes submitted since the gate last opened
    // chained mode (cs_frontend_set_chain): detect_cuboid of step k is fed the lines detect_filter_lines found in line pass k - W -- the reference's chain
    // (main_obj.cpp:428-449), pipelined: the line pass of a batch runs at least W steps ahead of the batch's cuboid pass
    bool chain = false; float chain_thres = 0; long chain_first = 0; // passes from chain_first on leave a packet
    std::map<long, LinePacket> packets;
    long lead_max() const { return 2 * (long)workers.size(); }
    int open_gate() { // caller's stream idle -> region stages of every waiting pass -> return when they have left the GPU
        const int r = hipStreamSynchronize(ctx->stream) == hipSuccess ? CS_OK : CS_ERR_HIP; // (the gate opens either way: a waiting pass must not be left behind)
        std::unique_lock<std::mutex> lk(gate.m);
        gate.target = gate.submitted;
        gate.cv.notify_all();
        gate.cv.wait(lk, [&] { return gate.done == gate.target; });
        in_phase = 0;
        return r;
    }
    bool may_start() const { return passes_started < passes_target && passes_started < step_no + lead_max() && (!ring || passes_started < ring->pushed); } // any_m held (ring->pushed only grows, and only on the caller's thread)
    void kick_idle() { // any_m held: free workers take the passes that may be started, looked for from the one behind the last choice (a strict rotation made the caller
                       // wait for a slow pass with idle workers beside it: passes take 110 - 220 ms beside each other)
        for (size_t k = 0; k < workers.size() && may_start(); k++) {
            const size_t c = (next_worker + k) % workers.size();
            if (!workers[c]->is_free()) continue;
            workers[c]->submit(passes_started++);
            next_worker = (c + 1) % workers.size();
        }
    }
};

namespace {
int LineWorker::take_frames() {
    if (!fe || !fe->ring) return CS_OK;
    FrameRing *R = fe->ring;
    const int slot = (int)(pass_no % R->n_slots);
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamWaitEvent(ctx->stream, R->uploaded[slot], 0) != hipSuccess) return CS_ERR_HIP;
    int r = cs_lsd_set_frames_device(ctx, lsd, R->d[slot], R->n_frames);
    if (r == CS_OK && hipEventRecord(R->used_line[slot], ctx->stream) != hipSuccess) r = CS_ERR_HIP;
    { std::lock_guard<std::mutex> lk(R->m); R->line_gen[slot] = pass_no; }
    R->cv.notify_all();
    return r;
}
bool LineWorker::after_pass(int r) {
    if (!fe) return false;
    LinePacket pk; bool leave = false;
    if (fe->chain && pass_no >= fe->chain_first) { // (chain / chain_first only change with no pass in flight)
        leave = true; pk.status = r;
        if (r == CS_OK) pk.status = cs_lsd_filter_lines_packed(lsd, fe->chain_thres, pk.off, pk.lines);
    }
    std::lock_guard<std::mutex> lk(fe->any_m);
    if (leave) { fe->packets[pass_no] = std::move(pk); fe->any_cv.notify_all(); }
    if (r != CS_OK || fe->gate.phased || !fe->may_start()) return false;
    pass_no = fe->passes_started++;
    return true;
}
} // namespace

extern "C" {

int cs_frontend_create(cs_ctx *ctx, cs_orb *orb, cs_cuboid_batch *batch, int n_line_workers, cs_ctx *const *line_ctx, cs_lsd *const *lsd, cs_frontend **out) {
    if (!ctx || !out || n_line_workers < 0 || n_line_workers > 8 || (n_line_workers && (!line_ctx || !lsd))) return CS_ERR_BAD_ARG;
    for (int i = 0; i < n_line_workers; i++) if (!line_ctx[i] || !lsd[i] || line_ctx[i] == ctx) return CS_ERR_BAD_ARG; // a worker needs its own context (stream, timing records)
    cs_frontend *fe = new (std::nothrow) cs_frontend();
    if (!fe) return CS_ERR_NOMEM;
    fe->ctx = ctx; fe->orb = orb; fe->batch = batch;
    for (int i = 0; i < n_line_workers; i++) {
        LineWorker *w = new LineWorker();
        w->ctx = line_ctx[i]; w->lsd = lsd[i]; w->gate = &fe->gate; w->any_m = &fe->any_m; w->any_cv = &fe->any_cv; w->fe = fe;
        cs_lsd_set_gate(w->lsd, LineWorker::gate_wait, LineWorker::gate_done, w);
        w->th = std::thread([w] { w->loop(); });
        fe->workers.push_back(w);
    }
    // the runner keeps 1 + 2 W streams busy (the caller's, every worker's and its background stream for the region walk); the HIP runtime maps streams onto
    // GPU_MAX_HW_QUEUES hardware queues (4 unless the variable was set before the runtime started) and serialises the streams that share one: say so once, loudly
    fe->streams = 1 + 2 * n_line_workers;
    { const char *q = getenv("GPU_MAX_HW_QUEUES"); fe->hw_queues = q && atoi(q) > 0 ? atoi(q) : 4; }
    if (fe->hw_queues < fe->streams) {
        static bool said = false;
        if (!said) { said = true; fprintf(stderr, "[cubeslam-hip] cs_frontend_create: %d streams on %d hardware queues (GPU_MAX_HW_QUEUES%s): streams that share a queue run one after the other -- a 100 ms region walk then "
                                          "stalls another stream's kernels (up to -25 %% measured).  Export GPU_MAX_HW_QUEUES=16 before the HIP runtime starts (INTEGRATION.md).\n", fe->streams, fe->hw_queues, getenv("GPU_MAX_HW_QUEUES") ? "" : " unset: the runtime's default"); }
    }
    if (fe->batch) cs_cuboid_batch_set_shared_gpu(fe->batch, n_line_workers > 0); // alternating runner: the detectors' region walks hold most CUs all the time
    for (LineWorker *w : fe->workers) cs_lsd_set_shared_gpu(w->lsd, n_line_workers > 1);
    *out = fe;
    return CS_OK;
}

int cs_frontend_queues(const cs_frontend *fe, int *streams, int *hw_queues) {
    if (!fe) return CS_ERR_BAD_ARG;
    if (streams) *streams = fe->streams;
    if (hw_queues) *hw_queues = fe->hw_queues;
    return fe->hw_queues >= fe->streams ? 1 : 0;
}

int cs_frontend_step(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    int r = CS_OK;
    if (fe->ring) { // streaming source: this step's frames must have been pushed (its line pass cannot start without them)
        std::lock_guard<std::mutex> lk(fe->any_m);
        if (fe->step_no >= fe->ring->pushed) { fe->ctx->err = "cs_frontend_step: the streaming source holds no frames for this step (cs_frontend_stream_push first)"; return CS_ERR_BAD_ARG; }
    }
    if (!fe->workers.empty()) {
        const long W = (long)fe->workers.size();
        LinePacket pk; bool have_pk = false;
        {
            std::unique_lock<std::mutex> lk(fe->any_m);
            const long k = fe->step_no;
            if (fe->passes_target < k + 1) fe->passes_target = k + 1;
            if (fe->gate.phased) { // one pass per detector and super-step, in rotation (the gate counts them)
                if (fe->passes_started <= k) { LineWorker *w = fe->workers[(size_t)(k % W)]; const long no = fe->passes_started++; lk.unlock(); w->submit(no); lk.lock(); fe->in_phase++; }
            } else {
                fe->kick_idle();
                while (fe->passes_started <= k) { fe->any_cv.wait_for(lk, std::chrono::milliseconds(2)); fe->kick_idle(); } // pass k has been started: the line path is at most W passes behind
            }
            if (fe->chain && fe->batch && k - W >= fe->chain_first) { // the lines of pass k - W are this step's edges (the first W steps after the switch run on the lists the batch holds)
                fe->any_cv.wait(lk, [&] { return fe->packets.count(k - W) != 0; });
                pk = std::move(fe->packets[k - W]); fe->packets.erase(k - W); have_pk = true;
            }
            fe->step_no = k + 1;
        }
        if (have_pk) {
            r = pk.status;
            if (r == CS_OK && (long)pk.off.size() != (long)cs_cuboid_batch_n_frames(fe->batch) + 1) r = CS_ERR_BAD_ARG; // the detectors and the batch must hold the same number of frames: set_lines reads one offset per frame of the batch
            if (r == CS_OK) r = cs_cuboid_batch_set_lines(fe->cub_ctx ? fe->cub_ctx : fe->ctx, fe->batch, pk.off.data(), pk.lines.data());
            if (r != CS_OK) return r;
        }
    } else {
        std::lock_guard<std::mutex> lk(fe->any_m);
        fe->step_no++;
    }
    if (fe->ring && r == CS_OK) { // streaming source: this step's frames out of their ring slot
        FrameRing *R = fe->ring;
        long k; { std::lock_guard<std::mutex> lk(fe->any_m); k = fe->step_no - 1; }
        const int slot = (int)(k % R->n_slots);
        cs_ctx *cc = fe->cub_ctx ? fe->cub_ctx : fe->ctx;
        if (hipStreamWaitEvent(fe->ctx->stream, R->uploaded[slot], 0) != hipSuccess) return CS_ERR_HIP;
        if (R->results_pending) { // the copies of the step before read what this step overwrites
            if (hipStreamWaitEvent(fe->ctx->stream, R->results_out, 0) != hipSuccess || (cc != fe->ctx && hipStreamWaitEvent(cc->stream, R->results_out, 0) != hipSuccess)) return CS_ERR_HIP;
        }
        if (fe->orb) r = cs_orb_set_frames_device(fe->ctx, fe->orb, R->d[slot], R->n_frames);
        if (r == CS_OK && fe->batch) {
            if (cc != fe->ctx && hipStreamWaitEvent(cc->stream, R->uploaded[slot], 0) != hipSuccess) return CS_ERR_HIP;
            r = cs_cuboid_batch_set_gray_device(cc, fe->batch, R->d[slot]);
            if (r == CS_OK && R->scene[(size_t)slot].has) { // the step's own boxes, poses (and edge lists): the plan is rebuilt and uploaded behind the last run on cc's stream
                const FrameRing::Scene &S = R->scene[(size_t)slot];
                r = cs_cuboid_batch_set_scene(cc, fe->batch, S.Twc.data(), S.box_off.data(), S.boxes.data(), S.has_lines ? S.line_off.data() : nullptr, S.has_lines ? S.lines.data() : nullptr);
            }
            if (r == CS_OK && cc != fe->ctx) { hipEvent_t e = R->used_main[slot]; if (hipEventRecord(e, cc->stream) != hipSuccess || hipStreamWaitEvent(fe->ctx->stream, e, 0) != hipSuccess) return CS_ERR_HIP; }
        }
        if (r == CS_OK && hipEventRecord(R->used_main[slot], fe->ctx->stream) != hipSuccess) r = CS_ERR_HIP;
        { std::lock_guard<std::mutex> lk(R->m); R->main_gen[slot] = k; }
        if (r != CS_OK) return r;
    }
    // the cuboid pass is a chain of launches without a host round trip: on a stream of its own (cs_frontend_set_cuboid_ctx) it is enqueued first and runs beside the ORB
    // pass, whose two read-backs would otherwise wait behind it
    if (fe->batch && fe->cub_ctx) r = cs_cuboid_batch_run(fe->cub_ctx, fe->batch);
    if (r == CS_OK && fe->orb) r = cs_orb_run(fe->ctx, fe->orb);
    if (r == CS_OK && fe->batch && !fe->cub_ctx) r = cs_cuboid_batch_run(fe->ctx, fe->batch);
    if (fe->gate.phased && !fe->workers.empty() && fe->in_phase >= fe->workers.size()) { const int g = fe->open_gate(); if (r == CS_OK) r = g; }
    return r;
}

// n more steps will follow on the frames the detectors hold: the line passes of those steps may be started as soon as a worker is free (at most 2 W passes ahead of the
// caller's step) instead of one per step.  Work is neither added nor dropped -- step k still needs line pass k, and cs_frontend_drain still waits for every pass that has
// been started; a backlog that is not followed by its steps leaves passes that later steps find done.  No effect on phased passes.
int cs_frontend_set_backlog(cs_frontend *fe, int n_steps) {
    if (!fe || n_steps < 0) return CS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(fe->any_m);
    if (fe->passes_target < fe->step_no + n_steps) fe->passes_target = fe->step_no + n_steps;
    if (!fe->gate.phased && !fe->workers.empty()) fe->kick_idle();
    return CS_OK;
}

// ---- streaming source
static void ring_free(FrameRing *R) {
    if (!R) return;
    if (R->copy) { hipStreamSynchronize(R->copy); hipStreamDestroy(R->copy); }
    if (R->copy_out) { hipStreamSynchronize(R->copy_out); hipStreamDestroy(R->copy_out); }
    for (hipEvent_t e : {R->step_done, R->cub_done, R->results_out}) if (e) hipEventDestroy(e);
    if (R->h_status) hipHostFree(R->h_status);
    for (uint8_t *p : R->d) if (p) hipFree(p);
    for (auto *v : {&R->uploaded, &R->used_main, &R->used_line}) for (hipEvent_t e : *v) if (e) hipEventDestroy(e);
    delete R;
}
int cs_frontend_stream_begin(cs_frontend *fe, int n_frames, int width, int height, int n_slots) {
    if (!fe || n_frames < 1 || width < 1 || height < 1 || n_slots < 2 || n_slots > 8 || fe->gate.phased) return CS_ERR_BAD_ARG;
    // a slot is handed to the attached objects as n_frames x height x width bytes: it must be exactly what each of them copies out of it
    {
        int w = 0, h = 0, n = 0;
        if (fe->orb && (cs_orb_geometry(fe->orb, &w, &h, &n) != CS_OK || w != width || h != height || n_frames > n)) { fe->ctx->err = "cs_frontend_stream_begin: the ring's geometry is not the extractor's"; return CS_ERR_BAD_ARG; }
        if (fe->batch && (cs_cuboid_batch_geometry(fe->batch, &w, &h, &n) != CS_OK || w != width || h != height || n_frames != n)) { fe->ctx->err = "cs_frontend_stream_begin: the ring's geometry is not the cuboid batch's"; return CS_ERR_BAD_ARG; }
        for (LineWorker *lw : fe->workers)
            if (cs_lsd_geometry(lw->lsd, &w, &h, &n) != CS_OK || w != width || h != height || n_frames > n) { fe->ctx->err = "cs_frontend_stream_begin: the ring's geometry is not the line detectors'"; return CS_ERR_BAD_ARG; }
    }
    int r = cs_frontend_drain(fe);
    if (r != CS_OK) return r;
    if (hipSetDevice(fe->ctx->device) != hipSuccess) return CS_ERR_HIP;
    ring_free(fe->ring); fe->ring = nullptr;
    FrameRing *R = new FrameRing();
    R->n_slots = n_slots; R->n_frames = n_frames; R->bytes = (size_t)n_frames * width * height;
    R->d.assign((size_t)n_slots, nullptr); R->uploaded.assign((size_t)n_slots, nullptr); R->used_main.assign((size_t)n_slots, nullptr); R->used_line.assign((size_t)n_slots, nullptr);
    R->main_gen.assign((size_t)n_slots, -1); R->line_gen.assign((size_t)n_slots, -1); R->scene.assign((size_t)n_slots, FrameRing::Scene());
    bool ok = hipStreamCreateWithFlags(&R->copy, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&R->copy_out, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&R->step_done, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&R->cub_done, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&R->results_out, hipEventDisableTiming) == hipSuccess && hipHostMalloc((void **)&R->h_status, sizeof(int), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; i < n_slots && ok; i++)
        ok = hipMalloc((void **)&R->d[(size_t)i], R->bytes) == hipSuccess && hipEventCreateWithFlags(&R->uploaded[(size_t)i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&R->used_main[(size_t)i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&R->used_line[(size_t)i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { ring_free(R); return CS_ERR_NOMEM; }
    *R->h_status = 0;
    std::lock_guard<std::mutex> lk(fe->any_m);
    R->pushed = R->first = fe->step_no; // step numbers go on: the next step takes slot step_no % n_slots
    R->first_line = std::max(fe->passes_started, fe->step_no); // (passes a cut backlog left done ran on the resident frames and never touch a slot)
    fe->ring = R;
    return CS_OK;
}
// The frames of the next step that has none yet (n_frames x height x width bytes; pinned host memory makes the call asynchronous): an H2D copy on the ring's own
// stream behind everything that still reads the slot.  Blocks only while the line pass that last used the slot has not taken its frames yet.
static int stream_push(cs_frontend *fe, const uint8_t *gray, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines);
int cs_frontend_stream_push(cs_frontend *fe, const uint8_t *gray) { return stream_push(fe, gray, nullptr, nullptr, nullptr, nullptr, nullptr); }
// ... and with what detect_cuboid takes beside the pixels (detect_3d_cuboid.h:62-63): the frames' camera poses (n_frames x 16), their 2-D boxes (box_offsets[n_frames + 1],
// rows of 5) and their edge lists (line_offsets NULL: the batch's lists stay).  The step that takes the
// slot rebuilds the cuboid batch's plan for them (cs_cuboid_batch_set_scene).
int cs_frontend_stream_push_scene(cs_frontend *fe, const uint8_t *gray, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines) {
    if (!fe || !fe->batch || !Twc || !box_offsets) return CS_ERR_BAD_ARG;
    return stream_push(fe, gray, Twc, box_offsets, boxes, line_offsets, lines);
}
static int stream_push(cs_frontend *fe, const uint8_t *gray, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines) {
    if (!fe || !fe->ring || !gray) return CS_ERR_BAD_ARG;
    FrameRing *R = fe->ring;
    if (hipSetDevice(fe->ctx->device) != hipSuccess) return CS_ERR_HIP;
    const long k = R->pushed; const int slot = (int)(k % R->n_slots); const long prev = k - R->n_slots; // the step that used the slot before
    {
        std::unique_lock<std::mutex> lk(fe->any_m);
        if (k >= fe->step_no + R->n_slots) { fe->ctx->err = "cs_frontend_stream_push: every slot holds frames of a step that has not run"; return CS_ERR_CAPACITY; }
    }
    if (prev >= R->first) {
        if (!fe->workers.empty() && prev >= R->first_line) { std::unique_lock<std::mutex> lk(R->m); R->cv.wait(lk, [&] { return R->line_gen[(size_t)slot] >= prev; }); if (hipStreamWaitEvent(R->copy, R->used_line[(size_t)slot], 0) != hipSuccess) return CS_ERR_HIP; }
        if (hipStreamWaitEvent(R->copy, R->used_main[(size_t)slot], 0) != hipSuccess) return CS_ERR_HIP; // (step prev has run: k < step_no + n_slots)
    }
    if (hipMemcpyAsync(R->d[(size_t)slot], gray, R->bytes, hipMemcpyHostToDevice, R->copy) != hipSuccess || hipEventRecord(R->uploaded[(size_t)slot], R->copy) != hipSuccess) return CS_ERR_HIP;
    { // (the slot's scene is host data of the runner until its step has handed it to the batch; the step that used the slot before has run: k < step_no + n_slots)
        FrameRing::Scene &S = R->scene[(size_t)slot];
        S.has = Twc != nullptr; S.has_lines = S.has && line_offsets != nullptr;
        if (S.has) {
            const int F = R->n_frames, nb = box_offsets[F];
            S.Twc.assign(Twc, Twc + (size_t)F * 16); S.box_off.assign(box_offsets, box_offsets + F + 1); S.boxes.assign(boxes, boxes + (size_t)nb * 5);
            if (S.has_lines) { const int nl = line_offsets[F]; S.line_off.assign(line_offsets, line_offsets + F + 1); S.lines.assign(lines, lines + (size_t)nl * 4); }
        }
    }
    { std::lock_guard<std::mutex> lk(fe->any_m); R->pushed = k + 1; if (!fe->gate.phased && !fe->workers.empty()) fe->kick_idle(); }
    return CS_OK;
}
// The results of the step that has just been enqueued -- ORB key points / descriptors packed like cs_orb_read_packed, the cuboids like cs_cuboid_batch_read (either may be
// NULL) -- copied to the caller's (pinned) buffers on a copy stream of the ring, behind the step's kernels: the call returns at once, the NEXT step's kernels wait on the
// device for the copies before they overwrite what is being read, and cs_frontend_stream_read_wait blocks the host until the copies of the last call have arrived.
int cs_frontend_stream_read_async(cs_frontend *fe, cs_keypoint *kps, uint8_t *desc, long cap_total, int *first, long *total, cs_cuboid *cuboids, int *counts) {
    if (!fe || !fe->ring) return CS_ERR_BAD_ARG;
    FrameRing *R = fe->ring;
    if (hipSetDevice(fe->ctx->device) != hipSuccess) return CS_ERR_HIP;
    if (hipEventRecord(R->step_done, fe->ctx->stream) != hipSuccess || hipStreamWaitEvent(R->copy_out, R->step_done, 0) != hipSuccess) return CS_ERR_HIP;
    if (fe->cub_ctx && (hipEventRecord(R->cub_done, fe->cub_ctx->stream) != hipSuccess || hipStreamWaitEvent(R->copy_out, R->cub_done, 0) != hipSuccess)) return CS_ERR_HIP;
    int r = CS_OK;
    if (fe->orb && kps && desc) { if (!first || !total) return CS_ERR_BAD_ARG; r = cs_orb_read_packed_on(fe->orb, R->copy_out, kps, desc, cap_total, first, total); }
    R->status_read = r == CS_OK && fe->batch && cuboids && counts; // (the status word is only written by the cuboid copy)
    if (R->status_read) r = cs_cuboid_batch_read_on(fe->batch, R->copy_out, cuboids, counts, R->h_status);
    if (hipEventRecord(R->results_out, R->copy_out) != hipSuccess) return CS_ERR_HIP;
    R->results_pending = true;
    return r;
}
int cs_frontend_stream_read_wait(cs_frontend *fe) {
    if (!fe || !fe->ring) return CS_ERR_BAD_ARG;
    FrameRing *R = fe->ring;
    if (!R->results_pending) return CS_OK;
    if (hipEventSynchronize(R->results_out) != hipSuccess) return CS_ERR_HIP;
    R->results_pending = false;
    if (R->status_read && *R->h_status != 0) { fe->ctx->err = "more than CS_MAX_ROI_LINES lines inside one box"; const int st = *R->h_status; *R->h_status = 0; return st; }
    return CS_OK;
}
int cs_frontend_stream_end(cs_frontend *fe) {
    if (!fe) return CS_ERR_BAD_ARG;
    if (fe->ring && fe->ring->results_pending) hipEventSynchronize(fe->ring->results_out);
    const int r = cs_frontend_drain(fe);
    if (fe->ring) { hipStreamSynchronize(fe->ctx->stream); if (fe->cub_ctx) hipStreamSynchronize(fe->cub_ctx->stream); }
    std::lock_guard<std::mutex> lk(fe->any_m);
    ring_free(fe->ring); fe->ring = nullptr;
    return r;
}

int cs_frontend_set_phased(cs_frontend *fe, int on) {
    if (!fe || (on && fe->ring)) return CS_ERR_BAD_ARG; // (the streaming source hands a pass its frames when it starts: not with gated passes)
    int r = cs_frontend_drain(fe); // no pass in flight across the switch
    std::lock_guard<std::mutex> lk(fe->gate.m);
    fe->gate.phased = on != 0;
    if (fe->batch) cs_cuboid_batch_set_shared_gpu(fe->batch, !on && !fe->workers.empty()); // phased: the score kernel never meets a region walk
    for (LineWorker *w : fe->workers) cs_lsd_set_shared_gpu(w->lsd, !on && fe->workers.size() > 1); // phased: sixteen frames per CU, the walks packed
    return r;
}

// on != 0: from now on the cuboid pass of step k takes its edge lists from line pass k - W (filter_lines with `length_thres`, main_obj.cpp:366) -- a pass
// that was started at least W steps earlier, so in the steady state nothing waits.  The first W steps after the switch still run on the lists the batch
// holds.  Needs a batch and at least one line worker.
int cs_frontend_set_chain(cs_frontend *fe, int on, float length_thres) {
    if (!fe || (on && (!fe->batch || fe->workers.empty()))) return CS_ERR_BAD_ARG;
    const int r = cs_frontend_drain(fe);
    std::lock_guard<std::mutex> lk(fe->any_m);
    fe->chain = on != 0; fe->chain_thres = length_thres;
    fe->chain_first = std::max(fe->passes_started, fe->step_no); // (passes a cut backlog left done have no packet)
    fe->packets.clear();
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
    { std::lock_guard<std::mutex> lk(fe->any_m); fe->passes_target = fe->passes_started; } // what is left of a backlog is not started behind the caller's back
    if (fe->in_phase) r = fe->open_gate(); // an incomplete super-step
    for (LineWorker *w : fe->workers) { const int s = w->wait(); if (r == CS_OK) r = s; }
    return r;
}

void cs_frontend_destroy(cs_frontend *fe) {
    if (!fe) return;
    { std::lock_guard<std::mutex> lk(fe->any_m); fe->passes_target = fe->passes_started; }
    if (fe->in_phase) fe->open_gate();
    for (LineWorker *w : fe->workers) {
        w->wait();
        cs_lsd_set_gate(w->lsd, nullptr, nullptr, nullptr);
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        w->th.join();
        delete w;
    }
    ring_free(fe->ring);
    delete fe;
}

} // extern "C"
