#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on N MI355X GPUs of one node (contract in the task statement).

Workload at N=1 (BASELINE.json configs[1]): synthetic 640x480 frames -- three drawn cuboids over a band-limited texture, so that ORB, LSD and
Canny see real content (asserted: >= 800 key points and >= 100 line segments per frame) -- with 3 boxes each, `--frames` frames resident in
HBM per GPU; one step = one pass of the front-end over that batch: ORBextractor (1000 features, 8 levels, FAST 20/7) + LSD line detection
with LBD descriptors (line_lbd defaults) + detect_3d_cuboid with a 180-yaw x 3-VP proposal sweep (yaw step 0.5 deg over +-45 deg; Canny +
distance transform + line merge + VP support + sweep / score + selection).  N>1: every rank owns its own block of frames (no data-path
collective, weak scaling).  The same JSON line carries, measured in the same run on rank 0 at N=1:
  "ba"   the second half of BASELINE's metric (LM iterations/s of the object BA at 1000 key frames),
  "c3"   config 3: the 1241x376 stream, 2000 ORB features + LSD/LBD + frame-to-frame SearchByProjection, with the matcher's own roofline,
  "c4"   config 4's per-GPU share: 64 frames x 8 boxes through the cuboid path,
  "chained"  the reference's chain: detect_cuboid fed the lines detect_filter_lines found in the same step,
  "pcie_inclusive"  the drop-in calls frame by frame, host buffers in and out (one caller thread, and sixteen),
  "cpu_baseline" / "cpu_baseline_mt"  the CPU port (oracle, built -march=native on this box) on 1 thread / frame-parallel on the host cores.
  "cpu_baseline_ref"  the reference's own text (oracle/_ref/libref.so, prebuilt from /root/reference by oracle/Makefile.ref), 1 thread, beside the port.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# The runner keeps up to ten HIP streams busy (the caller's, four line workers' and their background streams for the region walk); the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and serialises the streams that share one -- a 100 ms region walk then blocks another stream's kernels.  Must be set
# before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (~6.3 TB/s achievable)
BG_TEXTURE = 0.5


def make_frames(n_frames, n_boxes, seed0, bg_texture=BG_TEXTURE):
    """n_frames scenes with exactly n_boxes boxes, seeds seed0, seed0 + 1, ... in order (the ones with fewer boxes skipped); drawn on the host's threads."""
    from concurrent.futures import ThreadPoolExecutor

    from cube_slam_amd import synth
    scenes = []
    seed = seed0
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        while len(scenes) < n_frames:
            chunk = max(16, n_frames - len(scenes))
            for s in ex.map(lambda sd: synth.cuboid_scene(sd, n_boxes=n_boxes, bg_texture=bg_texture), range(seed, seed + chunk)):
                if len(s["boxes"]) == n_boxes and len(scenes) < n_frames:
                    scenes.append(s)
            seed += chunk
    return scenes


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def native_oracle():
    """The oracle rebuilt -O3 -march=native for THIS box's host CPU (the in-tree liboracle.so is generic because it travels)."""
    from oracle import pyoracle as po
    out = "/tmp/liboracle_native.so"
    try:
        srcs = sorted(f for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith("_oracle.cpp"))
        subprocess.check_call(["g++", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC", "-shared", "-w", "-o", out] + srcs +
                              ["-lpthread"], cwd=os.path.join(ROOT, "oracle"), timeout=600)
        import ctypes
        h = ctypes.CDLL(out)
        h.orc_box_edge_sum_dists.restype = ctypes.c_double
        h.orc_box_edge_angle_error.restype = ctypes.c_double
        po._LIB = h  # pyoracle.lib() returns the cached handle
        return "-O3 -march=native"
    except Exception as e:  # keep the generic build
        return "-O3 (generic build: %s)" % type(e).__name__


def _cpu_frame(po, s, o, ext, with_lines):
    if ext is not None:
        ext(s["gray"])
    if with_lines:
        po.lbd_compute(s["gray"], po.lsd_detect(s["gray"]))
    po.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=o)


def cpu_baseline(scenes, yaw_step, flags, budget_s=10.0, with_orb=True, nfeat=1000, with_lines=True, threads=1):
    """The CPU port on a bounded sample of the same frames: 1 thread like the reference, or frame-parallel on `threads` host cores."""
    from oracle import pyoracle as po
    o = po.cuboid_opts(yaw_step_deg=yaw_step)
    _cpu_frame(po, scenes[0], o, po.ORBextractor(nfeat, 1.2, 8, 20, 7) if with_orb else None, with_lines)
    if threads <= 1:
        ext = po.ORBextractor(nfeat, 1.2, 8, 20, 7) if with_orb else None
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s:
            _cpu_frame(po, scenes[n % len(scenes)], o, ext, with_lines)
            n += 1
        dt = time.perf_counter() - t0
    else:
        from concurrent.futures import ThreadPoolExecutor  # the ctypes calls release the GIL
        deadline = time.perf_counter() + budget_s

        def worker(tid):
            ext = po.ORBextractor(nfeat, 1.2, 8, 20, 7) if with_orb else None
            k = 0
            while time.perf_counter() < deadline:
                _cpu_frame(po, scenes[(tid + k * threads) % len(scenes)], o, ext, with_lines)
                k += 1
            return k
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            n = sum(ex.map(worker, range(threads)))
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port", "build": flags,
            "sample": "%d frames of the same workload in %.1f s, oracle/{orb,lsd,lbd,cuboid}_oracle.cpp, %d thread%s" % (n, dt, threads, "" if threads == 1 else "s (frame-parallel)")}


def cpu_baseline_ref(scenes, yaw_step, budget_s=8.0, with_orb=True, nfeat=1000, with_lines=True):
    """The same bounded sample through the REFERENCE'S OWN TEXT: oracle/_ref/libref.so (oracle/Makefile.ref: ORBextractor.cc, lsd.cpp, LSDDetector.cpp whole; detect_cuboid and
    BinaryDescriptor's compute path cut out of the reference at build time) on the OpenCV / Eigen stand-ins of oracle/ref_shim, 1 thread.  Built -O3 without -march=native (the
    library is prebuilt where /root/reference exists and travels).  None when the library is not there."""
    import ctypes as C
    from oracle import pyoracle as po
    so = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    if not os.path.exists(so):
        return None
    ref = C.CDLL(so)
    o = po.cuboid_opts(yaw_step_deg=yaw_step)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    cap = nfeat * 2 + 64
    kps, desc = np.zeros(cap, po.KEYPOINT_DTYPE), np.zeros((cap, 32), np.uint8)
    kl, ldesc = np.zeros(20000, po.KEYLINE_DTYPE), np.zeros((20000, 32), np.uint8)

    def frame(s):
        gray = np.ascontiguousarray(s["gray"], np.uint8)
        H, W = gray.shape
        if with_orb:
            levels, dims = np.zeros(4 * W * H + 64, np.uint8), np.zeros(16, np.int32)
            ref.ref_orb_extract(nfeat, C.c_float(1.2), 8, 20, 7, vp(gray), W, H, vp(kps), vp(desc), cap, vp(levels), vp(dims))
        if with_lines:
            n = ref.ref_lsd_keylines(vp(gray), W, H, vp(kl), len(kl))
            ref.ref_lbd_compute(vp(gray), W, H, vp(kl), n, vp(ldesc))
        K, Twc = np.ascontiguousarray(s["K"], np.float64), np.ascontiguousarray(s["Twc"], np.float64)
        boxes, lines = np.ascontiguousarray(s["boxes"], np.float64).reshape(-1, 5), np.ascontiguousarray(s["lines"], np.float64).reshape(-1, 4)
        out, cnt = np.zeros((len(boxes), o.max_cuboid_num), po.CUBOID_DTYPE), np.zeros(len(boxes), np.int32)
        if ref.ref_detect_cuboid(vp(gray), W, H, vp(K), vp(Twc), vp(boxes), len(boxes), vp(lines), len(lines), C.byref(o), vp(out), vp(cnt)) != 0:
            raise RuntimeError("ref_detect_cuboid failed")
    frame(scenes[0])
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        frame(scenes[n % len(scenes)])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "reference", "build": "-O3 (prebuilt, generic)",
            "sample": "%d frames of the same workload in %.1f s through oracle/_ref/libref.so: the reference's ORBextractor.cc, lsd.cpp, LSDDetector.cpp, BinaryDescriptor compute path and "
                      "detect_cuboid text on the OpenCV / Eigen stand-ins of oracle/ref_shim, 1 thread" % (n, dt)}


def streamed_bench(fe, ctx, side_ctxs, orb, batch, lsds, scenes, steps, warmup, torch):
    """The front-end with frames that ARRIVE: every step takes 1 024 new frames from pinned host memory through the runner's ring (cs_frontend_stream_*: H2D on a copy stream of
    its own while the step before computes, device copies into ORB, the cuboid batch and the step's line pass) WITH what detect_cuboid takes beside the pixels -- the frames'
    2-D boxes, camera poses and edge lists (cs_frontend_stream_push_scene): the cuboid batch's plan is rebuilt for every step inside the clock -- and its results go back to
    the host inside the clock: ORB key points + descriptors and the cuboids on a second copy stream behind the step's kernels (cs_frontend_stream_read_async), KeyLines + LBD
    descriptors by the line worker at the end of its pass (they are host data when the pass completes; the drain inside the clock waits for the last pass).  No two steps
    hold the same scenes: every step's frames are a draw (without replacement) from a pool of twice as many distinct scenes as a step holds -- other pixels, other geometry,
    other proposal counts, another unit plan per step."""
    from cube_slam_amd.cuboid import CuboidBatch
    F = len(scenes)
    H, W = scenes[0]["gray"].shape
    n_sets = warmup + steps
    n_boxes_per = len(scenes[0]["boxes"])
    pool = list(scenes) + make_frames(F, n_boxes_per, seed0=7000000)  # as many scenes again that nothing has seen yet
    rng = np.random.default_rng(11)
    host = torch.empty((n_sets, F, H, W), dtype=torch.uint8, pin_memory=True)
    sets = host.numpy()
    packs, box_area = [], []
    for k in range(n_sets):
        pick = rng.permutation(len(pool))[:F]
        for f, i in enumerate(pick):
            sets[k, f] = pool[i]["gray"]
        packs.append(CuboidBatch.pack_scene(np.stack([pool[i]["Twc"] for i in pick]), [pool[i]["boxes"] for i in pick], [pool[i]["lines"] for i in pick]))
        box_area.append(float(sum((pool[i]["boxes"][:, 2] * pool[i]["boxes"][:, 3]).sum() for i in pick)))
    from cube_slam_amd.cuboid import CUBOID_DTYPE
    pin = lambda nbytes: torch.empty((nbytes,), dtype=torch.uint8, pin_memory=True).numpy()  # noqa: E731
    bufs = []
    for _ in range(2):  # two sets of result buffers: step k's copies land while step k - 1's are the caller's
        bufs.append({"kps": pin(F * orb.cap * 28).view(orb_keypoint_dtype()) if orb is not None else None, "desc": pin(F * orb.cap * 32).reshape(-1, 32) if orb is not None else None,
                     "cub": pin(batch.n_boxes * batch.max_cuboid_num * CUBOID_DTYPE.itemsize).view(CUBOID_DTYPE).reshape(batch.n_boxes, batch.max_cuboid_num),
                     "cnt": pin(4 * batch.n_boxes).view(np.int32)})

    def barrier():
        fe.stream_read_wait()
        fe.drain(); ctx.sync()
        for c in side_ctxs:
            c.sync()
        torch.cuda.synchronize()

    d2h = [0]

    def one(k):
        if k + 1 < n_sets:
            fe.stream_push_scene(sets[k + 1], packs[k + 1])
        fe.step()
        fe.stream_read_wait()  # the copies of step k - 1 (long done: its buffers are the caller's now)
        b = bufs[k & 1]
        _, total = fe.stream_read_async(b["kps"], b["desc"], b["cub"], b["cnt"])
        d2h[0] += total * (28 + 32) + b["cub"].nbytes + b["cnt"].nbytes
    fe.stream_begin(F, W, H, 3)
    fe.stream_push_scene(sets[0], packs[0])
    for k in range(warmup):
        one(k)
    barrier()
    d2h[0] = 0
    t0 = time.perf_counter()
    for k in range(warmup, n_sets):
        one(k)
    barrier()
    dt = time.perf_counter() - t0
    n_lines = sum(len(lsds[0].read(f, with_desc=False)) for f in range(F)) if lsds else 0
    line_bytes = n_lines * (72 + 32)  # one pass's KeyLines + descriptors (cs_keyline is 72 B)
    fe.stream_end()
    h2d = steps * F * H * W
    return {"metric": "frames/s front-end with 1 024 NEW frames per step: H2D, ORB + cuboid + line pass, results D2H, all inside the clock", "value": F * steps / dt, "unit": "frames/s",
            "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "h2d_GBps": h2d / dt / 1e9, "d2h_GBps": (d2h[0] + steps * line_bytes) / dt / 1e9, "h2d_MB_per_step": F * H * W / 1e6, "d2h_MB_per_step": (d2h[0] / steps + line_bytes) / 1e6,
            "ring_slots": 3, "distinct_scene_sets": n_sets, "scene_pool": len(pool), "box_area_per_step_Mpx": [round(min(box_area) / 1e6, 2), round(max(box_area) / 1e6, 2)],
            "what": "cs_frontend_stream_push_scene of step k + 1 before cs_frontend_step of step k (pinned memory, a copy stream of its own): every step brings its own frames WITH their boxes, "
                    "poses and edge lists (a draw of 1 024 from a pool of 2 048 distinct scenes; the cuboid plan is rebuilt per step inside the clock); ORB + cuboid results copied to pinned host buffers on a second copy stream behind each step's kernels (cs_frontend_stream_read_async; the next step's kernels wait for "
                    "the copies on the device), line results by the workers at the end of each pass"}


def orb_keypoint_dtype():
    from cube_slam_amd.orb import KEYPOINT_DTYPE
    return KEYPOINT_DTYPE


# ---------------------------------------------------------------------------------------------------------------- HBM traffic (PMC)
def measure_traffic(kernel, script, script_args):
    """HBM-side bytes per launch of `kernel` from rocprofv3 --pmc passes of a small script that runs the same workload (tools/pmc_run.py: the
    cuboid path on the bench batch; tools/pmc_ba.py: the object BA), collected in their own runs (no tracing).  TCC_EA0 requests by size
    (MI355X_MICROARCH.md 'HBM': memory-side requests of the L2s, Infinity-Cache hits included; sizes counted explicitly instead of
    FETCH_SIZE's flat 64 B).  None when rocprofv3 is not available."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("CUBESLAM_BENCH_NO_TRAFFIC"):  # (a tracer around the bench: no nested profiler)
        return None
    res = {}
    try:
        for tag, ctrs in (("rd", "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"), ("wr", "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum")):
            d = tempfile.mkdtemp(prefix="pmc_" + tag, dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(["rocprofv3", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "-o", "res", "--", sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in script_args],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            acc = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                    if name.split("(")[0].split("<")[0] == kernel:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            res.update({k: sum(v) / len(v) for k, v in acc.items() if v})
            shutil.rmtree(d, ignore_errors=True)
        n, n32, n64, n128 = (res.get("TCC_EA0_RDREQ_sum", 0), res.get("TCC_EA0_RDREQ_32B_sum", 0), res.get("TCC_EA0_RDREQ_64B_sum", 0), res.get("TCC_EA0_RDREQ_128B_sum", 0))
        if n <= 0:
            return None
        rd = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(0.0, n - n32 - n64 - n128)
        w, w64 = res.get("TCC_EA0_WRREQ_sum", 0), res.get("TCC_EA0_WRREQ_64B_sum", 0)
        wr = 64 * w64 + 32 * max(0.0, w - w64)
        return {"bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "source": "rocprofv3 --pmc TCC_EA0_{RDREQ,WRREQ}* in this run (separate passes, tools/%s)" % script}
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------- object BA
def ba_bench(ctx, rank, world, iters, with_cpu, with_traffic=False):
    """LM iterations/s of the object BA at 1000 keyframes / 100k points / 500 cuboids."""
    from cube_slam_amd import synth
    from cube_slam_amd.ba import BundleAdjuster
    d = synth.ba_problem(20260923, n_kf=1000, n_points=100000, n_cuboids=500)
    allreduce = None  # world > 1: the library's own RCCL communicator (cs_comm_init in main) all-reduces the reduced camera system
    ba = BundleAdjuster(d, ctx=ctx, rank=rank, world=world, allreduce=allreduce)
    ba.optimize(1)  # warm-up (also pages the kernels in)
    ba.close()
    ba = BundleAdjuster(d, ctx=ctx, rank=rank, world=world, allreduce=allreduce)
    ctx.timing(False)
    ctx.sync()
    t0 = time.perf_counter()
    st = ba.optimize(iters)  # the reported rate: no per-launch event pairs (two hipEventRecord per launch cost ~0.1 ms per iteration)
    ctx.sync()
    dt = time.perf_counter() - t0
    ba.close()
    ba = BundleAdjuster(d, ctx=ctx, rank=rank, world=world, allreduce=allreduce)  # same graph again, instrumented, for the per-kernel breakdown
    ctx.timing(True); ctx.timing_reset()
    ba.optimize(iters)
    ctx.sync()
    names = ("ba_err_obs", "ba_build_abc", "ba_lin_lm", "ba_lin_pose", "ba_num_cols", "ba_lin_pose_edges", "ba_lm_dinv", "ba_schur_bd", "ba_schur_slots", "ba_schur_b", "ba_chol_factor",
             "ba_chol_solve", "ba_cub_inv", "ba_band_assemble", "ba_band_rhs", "ba_band_chol", "ba_band_twist_factor", "ba_band_mid", "ba_band_twist_back", "ba_cr_assemble", "ba_cr_eliminate",
             "ba_cr_back", "ba_cub_back", "ba_backsub", "ba_update", "ba_allreduce")
    kern = {}
    for nme in names:
        ms, n = ctx.timing_get(nme)
        if n:
            kern[nme] = round(1e3 * ms / n, 2)
    ctx.timing(False)
    O, Lm, C = len(d["obs_cam"]), len(d["points"]), len(d["cam_pose"])
    # algorithmic bytes of the Schur kernel per launch: read Hpl 144*O + Dinv 72*L, write 288*nnzb
    out = {"metric": "BA LM iterations/s @1k keyframes (100k points, 500 cuboids, %d observations)" % O, "value": st["iterations"] / dt,
           "unit": "iterations/s", "lm_trials_per_s": st["lm_trials"] / dt, "iterations": st["iterations"], "lm_trials": st["lm_trials"],
           "chi2_init": st["chi2_init"], "chi2_final": st["chi2_final"], "ms_per_iteration": 1e3 * dt / max(st["iterations"], 1), "kernels_us": kern}
    if "ba_schur_slots" in kern:
        alg = 144.0 * O / world + 72.0 * Lm / world + 288.0 * 5 * C
        ach = alg / (kern["ba_schur_slots"] * 1e-6) / 1e9
        tr = measure_traffic("ba_schur_slots", "pmc_ba.py", []) if with_traffic else None
        out["roofline"] = {"bound": "hbm", "kernel": "ba_schur_slots", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": None if tr is None else tr["bytes"], "traffic_detail": tr, "algorithmic_bytes_per_launch": alg}
    if with_cpu:
        from oracle import pyoracle as po
        t0 = time.perf_counter()
        _, _, _, rst = po.ba_optimize(d, 4)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": rst["iterations"] / dtc, "unit": "iterations/s", "cores": 1, "kind": "port",
                               "sample": "%d LM iterations of the same graph in %.1f s, oracle/ba_oracle.cpp, 1 thread" % (rst["iterations"], dtc)}
    ba.close()
    return out


# ---------------------------------------------------------------------------------------------------------------- config 3
def c3_bench(ctx, frames, steps, with_cpu, with_traffic=False, with_small_window=0, window_match=True):
    """BASELINE config 3: 1241x376 stream (1/f texture moving 3 px per frame), 2000 ORB features + LSD/LBD lines per frame and the
    tracking thread's frame-to-frame ORBmatcher::SearchByProjection (th = 15) from the extractor's device buffers.  The stream is cut into windows of `frames` frames
    (from 512 on the line detector's region stage runs on the device); window_match: the searches of a window's pairs as ONE cs_match_by_projection_stream call instead of
    five launches per frame issued from Python."""
    from cube_slam_amd import synth
    from cube_slam_amd.lsd import line_lbd_detect
    from cube_slam_amd.matcher import ORBmatcher, ORBmatcherStream
    from cube_slam_amd.orb import ORBextractor
    W, H = 1241, 376
    fx, fy, cx, cy = 721.5377, 721.5377, 609.5593, 172.854
    imgs = synth.texture_stream(77, W, H, frames, step=3)
    orb = ORBextractor(2000, 1.2, 8, 20, 7, W, H, max_frames=frames, ctx=ctx)
    orb.upload(imgs)
    # the stream through the library's runner (cs_frontend_*): ORB on this thread's context, the line path of the same window on two worker contexts that alternate
    # passes -- a window's host region stage (frames < 512: the OpenMP threads grow the regions) runs beside ORB + matching of the next window
    from cube_slam_amd import _lib
    from cube_slam_amd.frontend import Frontend
    lctx = [_lib.Context(0) for _ in range(int(os.environ.get("BENCH_C3_WORKERS", "4")) if frames >= 512 else 2)]  # (from 512 frames per window the region stage runs on the device: four detectors in flight, like the headline.  Six give 7.9 k against 7.2 k frames/s here -- and the blocks that run BEHIND this one in the same process then lose 10 % (chained 24.5 -> 22.3 k, streamed 23.3 -> 20.0 k, two runs each): left at four)
    lsds = [line_lbd_detect(W, H, max_frames=frames, ctx=c) for c in lctx]
    for d_ in lsds:
        d_.upload(imgs)
    lsd = lsds[0]
    fe = Frontend(ctx, orb=orb, batch=None, line_detectors=lsds)
    m = ORBmatcher(0.9, True, ctx=ctx, max_queries=4096)
    mstream = ORBmatcherStream(True, ctx=ctx) if window_match else None
    K4 = np.array([fx, fy, cx, cy], np.float32)
    bounds = (0.0, float(W), 0.0, float(H))
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    Tcw = np.eye(4, dtype=np.float32)[:3]

    def one_pass():
        fe.step()  # ORB of this window here, its line pass (LSD + LBD) on a worker
        n_q = n_m = 0
        if mstream is not None:  # the tracking thread's searches of the whole window: key points of frame f-1 projected into frame f (known 3 px shift), every pair in one call
            kall, _, first = orb.read_packed()  # key points + descriptors of every frame, two copies for the window (the map points' positions are host data in the reference too)
            pk = kall[:first[frames - 1]]
            z = np.full(len(pk), 10.0, np.float32)
            wp = np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32)
            ones = np.ones(len(pk), np.uint8)
            _, nm = mstream.search(orb, 0, frames - 1, K4, None, bounds, wp, ones, ones, np.broadcast_to(Tcw, (frames - 1, 3, 4)), fx, fy, cx, cy, sf, 15.0, int(first[frames] - first[1]))
            return len(pk), int(nm.sum())
        per = orb.read()  # key points + descriptors of every frame (the map points' descriptors are host data in the reference too)
        for f in range(1, frames):  # the tracking thread: key points of frame f-1 projected into frame f (known 3 px shift)
            m.set_frame_from_orb(orb, f, K4, None, bounds, read_keys=False)
            pk, pd = per[f - 1]
            z = np.full(len(pk), 10.0, np.float32)
            wp = np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32)
            ones = np.ones(len(pk), np.uint8)
            tm, nm = m.SearchByProjectionFrame(wp, ones, ones, pd, pk["octave"], pk["angle"], Tcw, fx, fy, cx, cy, sf, 15.0)
            n_q += len(pk); n_m += nm
        return n_q, n_m

    def sync_all():
        fe.drain(); ctx.sync()
        for c in lctx:
            c.sync()

    one_pass(); one_pass()
    sync_all()
    for c in [ctx] + lctx:
        c.timing(True); c.timing_reset()
    t0 = time.perf_counter()
    fe.set_backlog(steps)
    for _ in range(steps):
        n_q, n_m = one_pass()
    sync_all()
    dt = time.perf_counter() - t0
    kern = {}
    for nme in ("orb_resize", "orb_fast_score", "orb_cells", "orb_quadtree", "orb_blur", "orb_angle", "orb_desc", "host_lsd_regions", "lsd_blur_hv", "lsd_resize", "lsd_gradient",
                "lbd_blur5", "lbd_sobel", "lbd_line_desc", "match_undistort", "match_grid", "match_project", "match_candidates", "match_resolve"):
        parts = [c.timing_get(nme) for c in [ctx] + lctx]
        ms, n = sum(p_[0] for p_ in parts), sum(p_[1] for p_ in parts)
        if n:
            kern[nme] = round(1e3 * ms / n, 2)
    cand_ms, cand_n = ctx.timing_get("match_candidates")
    for c in [ctx] + lctx:
        c.timing(False)
    n_kp = sum(len(k) for k, _ in orb.read())
    n_lines = sum(len(lsd.read(f, with_desc=False)) for f in range(frames))
    cstat = m.last_candidate_stats() if mstream is None else mstream.last_counts()
    out = {"metric": "frames/s, 1241x376 stream: ORB 2000 + LSD/LBD + frame-to-frame SearchByProjection", "value": frames * steps / dt, "unit": "frames/s",
           "ms_per_frame": 1e3 * dt / (frames * steps), "frames": frames, "keypoints_per_frame": n_kp / frames, "keylines_per_frame": n_lines / frames,
           "queries_per_pass": n_q, "matches_per_pass": n_m, "kernels_us": kern,
           "region_stage": ("device: one wave per frame (lsd_rg_seq)" if lsd.region_stats()["device"] else "host: %d OpenMP threads" % _lib.lib().cs_host_thread_count()),
           "matching": ("one cs_match_by_projection_stream call per window: frame post-processing + SearchByProjection of its %d pairs in a handful of launches" % (frames - 1)) if mstream is not None
                       else "per frame from Python: cs_matcher_set_frame_from_orb + cs_match_by_projection_frame (five launches -- undistort, grid, project, candidates, resolve -- and one host round trip a frame)",
           "runner": "cs_frontend: ORB + the tracking thread's matching on the caller's context, the line path of the same window on %d worker contexts (a window's region stage "
                     "beside ORB + matching of the next one); %d windows of %d frames timed, drained inside the clock" % (len(lctx), steps, frames)}
    if cand_n and cstat is not None:
        # SURVEY 8d "Hamming window match, per query: 32 + 32 c + 8 B" with c = candidates the window enumerated
        alg = 40.0 * cstat["queries"] + 32.0 * cstat["candidates"]
        us = 1e3 * cand_ms / cand_n
        ach = alg / (us * 1e-6) / 1e9
        tr = (measure_traffic("match_candidates", "pmc_c3_match.py", [frames]) if mstream is not None else measure_traffic("match_candidates", "pmc_c3.py", [8])) if with_traffic else None  # (the window's search alone on the same frames: count + fill launches averaged, like avg_kernel_us)
        out["roofline"] = {"bound": "hbm", "kernel": "match_candidates", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None if tr is None else tr["bytes"],
                           "traffic_detail": tr, "avg_kernel_us": us, "algorithmic_bytes_per_launch": alg, "queries_per_launch": cstat["queries"], "candidates_per_launch": cstat["candidates"],
                           "note": ("a window's queries in ONE launch (a wave counts, takes a slice of the arena from a cursor and fills it); the claims run in match_resolve, one wave per pair" if mstream is not None else "one launch per frame (2000 queries): launch-latency-bound at this size")}
    # The kernel at the size TrackLocalMap gives it: ORBmatcher::SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:50-142) with a local map of 10 000 points in view
    # of one frame -- the key points of five frames of the stream stand in for the map points (projections, descriptors, predicted levels) -- one launch, no frame loop.
    try:
        per = orb.read()
        take = [per[f] for f in range(min(5, frames))]
        pk = np.concatenate([k for k, _ in take]); pd = np.concatenate([dd for _, dd in take])
        rng = np.random.default_rng(5)
        proj = np.stack([pk["x"] + rng.normal(0, 1.0, len(pk)), pk["y"] + rng.normal(0, 1.0, len(pk))], axis=1).astype(np.float32)
        ones = np.ones(len(pk), np.uint8)
        mm = ORBmatcher(0.8, True, ctx=ctx, max_queries=16384)
        mm.set_frame_from_orb(orb, min(5, frames - 1), K4, None, bounds)
        vc = np.full(len(pk), 0.999, np.float32)
        mm.SearchByProjectionLocalMap(proj, vc, pk["octave"], ones, ones, pd, sf, 3.0)
        ctx.sync(); ctx.timing(True); ctx.timing_reset()
        reps = 20
        for _ in range(reps):
            _, n_lm = mm.SearchByProjectionLocalMap(proj, vc, pk["octave"], ones, ones, pd, sf, 3.0)
        ctx.sync()
        lm_ms, lm_n = ctx.timing_get("match_candidates")
        ctx.timing(False)
        st_lm = mm.last_candidate_stats()
        if lm_n and st_lm is not None:
            alg_lm = 40.0 * st_lm["queries"] + 32.0 * st_lm["candidates"]
            us_lm = 1e3 * lm_ms / lm_n
            out["local_map"] = {"what": "SearchByProjection(F, 10^4 map points, th 3): one launch", "map_points": int(len(pk)), "matches": int(n_lm), "queries_per_launch": st_lm["queries"],
                                "candidates_per_launch": st_lm["candidates"], "match_candidates_us": us_lm, "algorithmic_bytes_per_launch": alg_lm,
                                "achieved_GBps": alg_lm / (us_lm * 1e-6) / 1e9, "frac": alg_lm / (us_lm * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "note": "five times the queries of the per-frame search for a fifth more time: the launch is latency-bound at every size the tracking thread produces"}
        mm.close()
    except Exception as e:  # the stream measurement above stands on its own
        out["local_map"] = {"error": str(e)[:200]}
    if with_cpu:
        from oracle import pyoracle as po
        ext = po.ORBextractor(2000, 1.2, 8, 20, 7)
        t0 = time.perf_counter(); n = 0; prev = None
        while time.perf_counter() - t0 < 8.0:
            g = imgs[n % frames]
            k, dsc = ext(g)
            po.lbd_compute(g, po.lsd_detect(g))
            if prev is not None and n % frames:
                pk, pd = prev
                F2 = po.make_frame(k, dsc, bounds)
                z = np.full(len(pk), 10.0, np.float32)
                wp = np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32)
                ones = np.ones(len(pk), np.uint8)
                po.search_by_projection_frame(F2, wp, ones, ones, pd, pk["octave"], pk["angle"], Tcw, fx, fy, cx, cy, sf, 15.0)
            prev = (k, dsc); n += 1
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dtc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "%d frames of the same stream in %.1f s, 1 thread" % (n, dtc)}
        # frame-parallel on the host cores (extraction + lines; the matcher needs the previous frame and stays out: 3 % of a frame)
        from concurrent.futures import ThreadPoolExecutor
        from cube_slam_amd import _lib
        cores = _lib.lib().cs_host_thread_count()
        deadline = time.perf_counter() + 6.0

        def worker(tid):
            e2 = po.ORBextractor(2000, 1.2, 8, 20, 7)
            k = 0
            while time.perf_counter() < deadline:
                g = imgs[(tid + k * cores) % frames]
                e2(g)
                po.lbd_compute(g, po.lsd_detect(g))
                k += 1
            return k
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            nmt = sum(ex.map(worker, range(cores)))
        dtm = time.perf_counter() - t0
        out["cpu_baseline_mt"] = {"value": nmt / dtm, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "%d frames of the same stream in %.1f s, %d threads (frame-parallel ORB + LSD/LBD, no matching)" % (nmt, dtm, cores)}
    m.close()
    if mstream is not None:
        mstream.close()
    fe.close()
    if with_small_window:
        # the same stream in windows of two frames per host thread with the per-frame calls: the latency-oriented form (a window's region stage on the host cores), round 4's c3
        for o_ in (orb, *lsds):
            o_.close()
        bw = c3_bench(ctx, with_small_window, 6, with_cpu=False, window_match=False)
        out["small_window"] = {k: bw[k] for k in ("value", "unit", "ms_per_frame", "frames", "keypoints_per_frame", "keylines_per_frame", "region_stage", "matching", "runner", "kernels_us")}
    return out


def c4_bench(ctx, frames, boxes, yaw_step, steps, with_cpu, with_traffic=False):
    """BASELINE config 4, one GPU's share: `frames` frames x `boxes` boxes through the cuboid path (512 frames shard as 64 per GPU)."""
    from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
    scenes = make_frames(frames, boxes, seed0=500000)
    det = detect_3d_cuboid(ctx)
    det.set_calibration(scenes[0]["K"])
    det.yaw_step_deg = yaw_step
    b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
    b.run(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.timing(True); ctx.timing_reset()  # the same batch again with event pairs: the score kernel's own duration
    for _ in range(steps):
        b.run()
    ctx.sync()
    k_ms, k_n = ctx.timing_get("cuboid_sweep_score")
    f_ms, f_n = ctx.timing_get("cuboid_sweep_filter")
    ctx.timing(False)
    st = b.stats()
    n_out = sum(len(g) for g in b.read())
    b.close()
    out = {"metric": "frames/s, detect_3d_cuboid on %d frames x %d boxes per GPU" % (frames, boxes), "value": frames * steps / dt, "unit": "frames/s", "boxes_per_s": frames * boxes * steps / dt,
           "ms_per_batch": 1e3 * dt / steps, "valid_proposals_per_batch": st["n_valid"], "roi_pixels_per_batch": st["roi_pixels"], "hypotheses_per_batch": st["n_hypotheses"], "cuboids_out": n_out}
    if k_n:
        us = 1e3 * k_ms / k_n
        alg = 4.0 * st["roi_pixels"] + 200.0 * st["n_valid"]
        tr = measure_traffic("cuboid_sweep_score", "pmc_run.py", [frames, boxes, yaw_step, BG_TEXTURE, 500000]) if with_traffic else None
        out["roofline"] = {"bound": "hbm", "kernel": "cuboid_sweep_score", "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "traffic": None if tr is None else tr["bytes"], "traffic_detail": tr, "frac_by_traffic": None if tr is None else tr["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_kernel_us": us, "filter_kernel_us": 1e3 * f_ms / max(f_n, 1), "algorithmic_bytes_per_launch": alg,
                           "algorithmic_bytes_formula": "4*A + 200*n_valid (SURVEY 8d, corner construction fused)",
                           "note": "512 units on 256 CUs: two units per persistent workgroup, the kernel is a third of the way into its steady state (see the N=1 line's roofline for 3072 units)"}
    if with_cpu:
        from oracle import pyoracle as po
        o = po.cuboid_opts(yaw_step_deg=yaw_step)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 6.0:
            sc = scenes[n % len(scenes)]
            po.detect_cuboid(sc["gray"], sc["K"], sc["Twc"], sc["boxes"], sc["lines"], opts=o)
            n += 1
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dtc, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "%d frames x %d boxes of the same batch in %.1f s, oracle/cuboid_oracle.cpp, 1 thread" % (n, boxes, dtc)}
    return out


def chained_bench(ctx, fe, lsds, batch, scenes, frames, steps, barrier, backlog=True):
    """The reference's chain (main_obj.cpp:428-449) with the frames resident, PIPELINED by the runner (cs_frontend_set_chain): every step's detect_cuboid is
    fed the lines detect_filter_lines found for the same frames in line pass k - W (k: the step, W: the line workers) -- a pass started at least W steps
    earlier, so in the steady state no step waits for a line pass; the hand-over goes through the host (KeyLines are assembled there) and costs the step one synchronisation
    of the ORB / cuboid stream.  Same runner, same detectors and batch as the headline; only the edge lists change."""
    for d_ in lsds:
        d_.line_length_thres = 15.0  # main_obj.cpp:366
    fe.set_chain(True, 15.0)
    for _ in range(2 * len(lsds)):  # every worker's lines have reached the batch
        fe.step()
    barrier()
    t0 = time.perf_counter()
    if backlog:
        fe.set_backlog(steps)
    for _ in range(steps):
        fe.step()
    barrier()
    dt = time.perf_counter() - t0
    n_out = sum(len(g) for g in batch.read())
    lines = lsds[0].read_filter_lines(frames)
    fe.set_chain(False)
    batch.set_lines([s["lines"] for s in scenes])  # back to the decoupled edge lists
    return {"value": frames * steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "frames": frames, "steps": steps, "lines_per_frame_handed_over": sum(len(l) for l in lines) / frames,
            "cuboids_out": n_out, "pipeline_depth_steps": len(lsds),
            "note": "ORB + LSD/LBD + detect_cuboid fed detect_filter_lines' output (length > 15) of the same frames, the line pass running %d steps ahead of the cuboid pass" % len(lsds)}


def pcie_inclusive(ctx, scenes, yaw_step, nfeat, n=24, local_rank=0):
    """The drop-in calls, one frame at a time with host buffers in and out (H2D + plan + kernels + D2H): what a ROS node sees per frame.  One caller
    thread, every call synchronous like the reference's; then the same calls from T caller threads at once, each with its own context (= stream) and
    its own extractor / detector objects -- the reference system calls these entry points from several threads (tracking, object thread), a
    multi-camera node or a bag replayer from as many as it likes.  The per-call breakdown says where a frame's time goes: the LSD region stage
    (lsd.cpp:637-1136) is a greedy sequence over the frame's seeds and runs on ONE host core below 512 frames per batch."""
    from concurrent.futures import ThreadPoolExecutor

    from cube_slam_amd import _lib
    from cube_slam_amd.cuboid import detect_3d_cuboid
    from cube_slam_amd.lsd import line_lbd_detect
    from cube_slam_amd.orb import ORBextractor

    def objects(c):
        det = detect_3d_cuboid(c)
        det.set_calibration(scenes[0]["K"])
        det.yaw_step_deg = yaw_step
        return det, ORBextractor(nfeat, 1.2, 8, 20, 7, 640, 480, ctx=c), line_lbd_detect(640, 480, ctx=c)

    det, ext, ll = objects(ctx)
    calls = {"ORBextractor::operator()": 0.0, "detect_raw_lines": 0.0, "get_line_descriptors": 0.0, "detect_cuboid": 0.0}

    def one(s, det=det, ext=ext, ll=ll, acc=None):
        t0 = time.perf_counter()
        ext(s["gray"])
        t1 = time.perf_counter()
        kl = ll.detect_raw_lines(s["gray"])
        t2 = time.perf_counter()
        ll.get_line_descriptors(s["gray"], kl)
        t3 = time.perf_counter()
        det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"])
        t4 = time.perf_counter()
        if acc is not None:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[k] += v
    one(scenes[0])
    t0 = time.perf_counter()
    for i in range(n):
        one(scenes[i % len(scenes)], acc=calls)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "ms_per_frame": 1e3 * dt / n, "callers": 1,
           "ms_per_call": {k: round(1e3 * v / n, 3) for k, v in calls.items()},
           "sample": "%d frames, ORBextractor::operator() + detect_raw_lines + LBD + detect_cuboid per frame, host in / host out, one caller thread" % n}
    T = min(16, _lib.lib().cs_host_thread_count())
    if T > 1:
        ctxs = [_lib.Context(local_rank) for _ in range(T)]
        objs = [objects(c) for c in ctxs]
        for o in objs:
            one(scenes[0], *o)
        per = max(8, n // 2)

        def worker(t):
            for i in range(per):
                one(scenes[(t + i * T) % len(scenes)], *objs[t])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(worker, range(T)))
        dtm = time.perf_counter() - t0
        out["multi_caller"] = {"value": T * per / dtm, "unit": "frames/s", "callers": T,
                               "sample": "%d caller threads x %d frames, each thread its own context and objects, every call synchronous" % (T, per)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1024, help="frames resident per GPU = frames per step (the line detector's region stage runs on the device from 512 on)")
    ap.add_argument("--line-workers", type=int, default=4, help="line detectors that alternate steps on their own streams (1-4)")
    ap.add_argument("--phased", type=int, default=0, help="0 (default): the alternating runner -- the detectors' region walks hold most CUs all the time and every other kernel runs beside them, the edge-scoring kernel in "
                    "the launch shape that fits into a CU's leftovers; 1: the phased runner (the region stages of --line-workers steps run together with the ORB / cuboid stream idle)")
    ap.add_argument("--cuboid-stream", type=int, default=0, help="1: the cuboid batch on a stream of its own beside the ORB pass of the same step (cs_frontend_set_cuboid_ctx); 0 (default): behind it on the caller's stream -- measured equal (20.77 k against 20.71 k frames/s: the kernels of both stretch by what they overlap), and the score kernel keeps more of the GPU to itself")
    ap.add_argument("--backlog", type=int, default=1, help="1 (default): the runner is told how many steps follow (cs_frontend_set_backlog) -- the line passes of those steps start as soon as a worker is free, "
                    "at most 2 W ahead of the ORB / cuboid pass of their step, so the line pipeline is full from the first timed step and does not drain behind the last one; 0: one line pass is started per step")
    ap.add_argument("--sweep", action="store_true", help="(development) after the timed region, time the same steps again under other runner settings: `runner_variants`")
    ap.add_argument("--boxes", type=int, default=3)
    ap.add_argument("--yaw-step", type=float, default=0.5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-orb", action="store_true")
    ap.add_argument("--no-lines", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c3 / c4 / pcie_inclusive / traffic blocks")
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--orb-features", type=int, default=1000)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # `python bench.py --gpus N` launches its own ranks (the driver uses torch.distributed.run directly)
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                   "--master-port", str(29500 + os.getpid() % 2000)] + sys.argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cube_slam_amd import _lib
    from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid

    ctx = _lib.Context(local_rank, priority=int(os.environ.get("BENCH_PRIO_MAIN", "1")))  # ORB + cuboid: the path a tracking thread waits for
    if world > 1:  # RCCL inside the library: rank 0's ncclUniqueId travels through the process group that the timing barrier uses anyway
        from cube_slam_amd import shard
        shard.comm_init_from_process_group(ctx, rank, world, device="cuda")  # (the same call tests/test_rccl_gpu.py's two-rank worker makes)
    scenes = make_frames(args.frames, args.boxes, seed0=1000 + 100000 * rank)
    hbm_marks = [("start", torch.cuda.mem_get_info()[0])]   # free bytes after each engine is resident: the front-end's working set by path
    det = detect_3d_cuboid(ctx)
    det.set_calibration(scenes[0]["K"])
    det.yaw_step_deg = args.yaw_step
    batch = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]),
                        [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())

    hbm_marks.append(("cuboid", torch.cuda.mem_get_info()[0]))
    orb = None
    if not args.no_orb:
        from cube_slam_amd.orb import ORBextractor
        orb = ORBextractor(args.orb_features, 1.2, 8, 20, 7, 640, 480, max_frames=args.frames, ctx=ctx)
        orb.upload(np.stack([s["gray"] for s in scenes]))
    hbm_marks.append(("orb", torch.cuda.mem_get_info()[0]))

    lsd = None
    if not args.no_lines:
        from cube_slam_amd.lsd import line_lbd_detect
        args.phased = 1 if args.phased and args.frames >= 512 and os.environ.get("CUBESLAM_LSD_REGIONS", "seq") == "seq" else 0
        # The library's front-end runner (cs_frontend_*, csrc/frontend.hip) runs ORB + cuboid on this thread and the line path on worker
        # threads with their own contexts (= HIP streams).  The detectors alternate steps.  From 512 frames per step on, region growing runs on
        # the device, one wave per frame for ~110 ms (lsd_regions.hip), sixteen frames to a CU.  Default: the ALTERNATING runner with four
        # detectors -- their walks (4 x 64 CUs, scalar-unit-bound) hold the chip all the time and every other kernel of the step runs in the
        # wave slot and the vector-ALU time they leave; the edge-scoring kernel, which otherwise needs whole CUs, takes its 256-thread shape for
        # that (cs_cuboid_batch_set_shared_gpu).  It is the faster runner (20.0 k against 16.7 k frames/s); the PHASED runner (--phased 1: four
        # detectors queue their map kernels beside ORB / cuboid of four steps, then the four region stages run together with the ORB / cuboid
        # stream idle) keeps the score kernel away from the walks and is timed beside it (`phased_runner`).  Below 512 frames the 16 host
        # threads grow the regions and the GPU phases of one step overlap the host stage of the neighbouring one.
        ctx_lines = [_lib.Context(local_rank, priority=int(os.environ.get("BENCH_PRIO_LINES", "1"))) for _ in range(max(1, min(8, args.line_workers)))]  # the line detectors' short kernels at the caller's priority; their region walks go to the contexts' lowest-priority background streams (cs_ctx::bg_begin)
        lsds = [line_lbd_detect(640, 480, max_frames=args.frames, ctx=c) for c in ctx_lines]
        for d_ in lsds:
            d_.upload(np.stack([s["gray"] for s in scenes]))
        lsd = lsds[0]
    from cube_slam_amd.frontend import Frontend
    fe = Frontend(ctx, orb=orb, batch=batch, line_detectors=lsds if lsd is not None else (), phased=bool(args.phased) and lsd is not None)

    ctx_cub = None
    if args.cuboid_stream and orb is not None:
        ctx_cub = _lib.Context(local_rank, priority=int(os.environ.get("BENCH_PRIO_MAIN", "1")))
        fe.set_cuboid_ctx(ctx_cub)
    side_ctxs = (ctx_lines if lsd is not None else []) + ([ctx_cub] if ctx_cub is not None else [])

    def barrier():
        fe.drain()
        ctx.sync()
        if ctx_cub is not None:
            ctx_cub.sync()
        if lsd is not None:
            for c in ctx_lines:
                c.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    hbm_marks.append(("lines_created", torch.cuda.mem_get_info()[0]))
    if args.backlog and args.warmup:
        fe.set_backlog(args.warmup)
    for _ in range(args.warmup):
        fe.step()
    barrier()
    hbm_marks.append(("after_warmup", torch.cuda.mem_get_info()[0]))
    for c in [ctx] + side_ctxs:
        c.timing(True)
        c.timing_reset()
    t0 = time.perf_counter()
    if args.backlog:
        fe.set_backlog(args.steps)  # inside the timed region: every line pass of these steps starts after t0 and is waited for by the barrier below
    for _ in range(args.steps):
        fe.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernels = {}
    for name in ("host_orb_quadtree", "orb_resize", "orb_fast_score", "orb_cells", "orb_scan", "orb_quadtree", "orb_blur", "orb_angle", "orb_desc", "host_lsd_regions", "lsd_blur_hv", "lsd_resize",
                 "lsd_gradient", "lsd_emit", "lsd_rg_fill", "lsd_rg_scatter", "lsd_rg_seq", "lsd_rg_wlk", "lsd_rg_improve", "lbd_blur5", "lbd_sobel", "lbd_line_desc", "cuboid_frame_prep", "cuboid_unit_lines", "cuboid_canny_nms", "cuboid_canny_cc_local", "cuboid_canny_cc_border", "cuboid_canny_cc", "cuboid_dt",
                 "cuboid_vp", "cuboid_sweep_filter", "cuboid_sweep_score", "cuboid_select"):
        ms, n = ctx.timing_get(name)
        if n == 0 and side_ctxs:
            parts = [c.timing_get(name) for c in side_ctxs]
            ms, n = sum(p_[0] for p_ in parts), sum(p_[1] for p_ in parts)
        kernels[name] = {"avg_us": 1e3 * ms / max(n, 1), "launches": n}
    ctx.timing(False)
    for c in side_ctxs:
        c.timing(False)
    variants = None
    if args.sweep and rank == 0 and world == 1 and lsd is not None:
        variants = []
        sweep_ctx = _lib.Context(local_rank, priority=int(os.environ.get("BENCH_PRIO_MAIN", "1")))
        for name, bl, cub in (("backlog", 1, 0), ("step_by_step", 0, 0), ("backlog+cuboid_stream", 1, 1), ("step_by_step+cuboid_stream", 0, 1), ("backlog", 1, 0)):
            fe.set_cuboid_ctx(sweep_ctx if cub else None)

            def bar2():
                barrier(); sweep_ctx.sync()
            bar2()
            t0 = time.perf_counter()
            if bl:
                fe.set_backlog(args.steps)
            for _ in range(args.steps):
                fe.step()
            bar2()
            dtv = time.perf_counter() - t0
            variants.append({"runner": name, "value": args.frames * args.steps / dtv, "ms_per_step": 1e3 * dtv / args.steps})
        fe.set_cuboid_ctx(ctx_cub)
    # the phased runner on the same objects, shortly: its throughput and the score kernel's time in ITS timed region (the kernel never meets a region walk there)
    phased_alt = None
    if lsd is not None and not args.phased and args.frames >= 512 and os.environ.get("CUBESLAM_LSD_REGIONS", "seq") == "seq" and len(ctx_lines) >= 2 and rank == 0 and world == 1:
        fe.set_phased(True)
        for _ in range(len(ctx_lines)):
            fe.step()
        barrier()
        tctx = ctx_cub if ctx_cub is not None else ctx  # (the context the cuboid batch's launches are timed on)
        tctx.timing(True); tctx.timing_reset()
        n_ph = 4 * len(ctx_lines)
        t0 = time.perf_counter()
        for _ in range(n_ph):
            fe.step()
        barrier()
        dt_ph = time.perf_counter() - t0
        ph_ms, ph_n = tctx.timing_get("cuboid_sweep_score")
        tctx.timing(False)
        phased_alt = {"value": args.frames * n_ph / dt_ph, "unit": "frames/s", "ms_per_step": 1e3 * dt_ph / n_ph, "steps": n_ph, "score_kernel_us_in_run": 1e3 * ph_ms / max(ph_n, 1)}
        fe.set_phased(False)
    # the same kernel without kernels of the other paths sharing the GPU: in the launch shape of the timed region, and in the 512-thread shape it has when it owns the CUs
    iso = {}
    for shared in ((True, False) if (lsd is not None and not args.phased) else (False,)):
        batch.set_shared_gpu(shared)
        batch.run(); ctx.sync()
        ctx.timing(True); ctx.timing_reset()
        for _ in range(10):
            batch.run()
        ctx.sync()
        iso[shared] = ctx.timing_get("cuboid_sweep_score")
        ctx.timing(False)
    iso_ms, iso_n = iso[False]
    batch.set_shared_gpu(lsd is not None and not args.phased)
    st = batch.stats()
    ss = batch.score_stats()
    got = batch.read()
    assert sum(len(g) for g in got) > 0
    n_kp = sum(len(k) for k, _ in orb.read()) if orb is not None else 0
    n_lines = sum(len(lsd.read(f, with_desc=False)) for f in range(args.frames)) if lsd is not None else 0
    if orb is not None and BG_TEXTURE > 0 and args.orb_features >= 1000:
        assert n_kp >= 800 * args.frames, "the frames are too empty for the ORB quota: %d key points per frame" % (n_kp // args.frames)
    if lsd is not None and BG_TEXTURE > 0:
        assert n_lines >= 100 * args.frames, "the frames are too empty for LSD: %d segments per frame" % (n_lines // args.frames)

    solo = rank == 0 and world == 1
    extra = {}
    if solo and not args.no_extra:
        tr = measure_traffic("cuboid_sweep_score", "pmc_run.py", [args.frames, args.boxes, args.yaw_step, BG_TEXTURE])
        if not args.no_cpu:
            native_oracle()  # the c3 / c4 CPU legs use the -march=native build too
        if os.environ.get("BENCH_SKIP_C3"):  # (development: the blocks behind c3 without it in front of them)
            extra["c3"] = {"skipped": True}
        else:
            extra["c3"] = c3_bench(ctx, 512, 8, with_cpu=not args.no_cpu, with_traffic=True, with_small_window=2 * _lib.lib().cs_host_thread_count())  # windows of 512 frames: the region stage on the device, a window's searches in one call; small_window: two frames per host thread, the per-frame calls (round 4's form)
        extra["c4"] = c4_bench(ctx, 64, 8, args.yaw_step, 10, with_cpu=not args.no_cpu, with_traffic=True)
        if lsd is not None:
            extra["chained"] = chained_bench(ctx, fe, lsds, batch, scenes, args.frames, args.steps, barrier, backlog=bool(args.backlog))
            try:
                extra["streamed"] = streamed_bench(fe, ctx, side_ctxs, orb, batch, lsds, scenes, args.steps, 3, torch)
            except Exception as e:  # (the block stands beside the headline; a failure must not cost the line)
                extra["streamed"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # the drop-in calls belong to another kind of process than the batch runner: measured in one, with the runtime's default hardware queues (tools/pcie_probe.py)
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pcie_probe.py"), "default", str(args.yaw_step), str(args.orb_features)], capture_output=True, text=True, timeout=300, check=True)
            extra["pcie_inclusive"] = json.loads(pr.stdout.strip().splitlines()[-1])
        except Exception:
            extra["pcie_inclusive"] = pcie_inclusive(ctx, scenes, args.yaw_step, args.orb_features, local_rank=local_rank)
            extra["pcie_inclusive"]["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)") + " (in this process)"
    else:
        tr = None
    ba_out = None
    if not args.no_ba:
        try:
            ba_out = ba_bench(ctx, rank, world, args.ba_iters, with_cpu=(solo and not args.no_cpu), with_traffic=solo and not args.no_extra)
        except Exception as e:  # (the BA block stands beside the headline: with world > 1 its in-library RCCL all-reduce runs for the first time on the driver's node -- a failure there must not cost the front-end's line)
            ba_out = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if rank == 0:
        total_frames = args.frames * world * args.steps
        # Algorithmic bytes of one cuboid_sweep_score launch, SURVEY 8d's figure for the edge-scoring kernel with corner construction fused into it
        # (K3 + K4): every distance-map ROI read once as the float map the distance transform writes (4*A) + 200 B per surviving proposal (the 9 + 16
        # doubles the reference stores per valid proposal, box_proposal_detail.cpp:450-456).  The kernel reads exactly that float map (it encodes it
        # into LDS itself) and no corner ever touches HBM; what it physically moves per proposal is 4 B of list in and 16 B of errors out, so
        # `traffic` (PMC) sits below the credited bytes.  `accountings` also gives the fraction by the bytes it must move.
        alg_bytes = 4.0 * st["roi_pixels"] + 200.0 * st["n_valid"]
        k_us = kernels["cuboid_sweep_score"]["avg_us"]
        achieved = alg_bytes / (k_us * 1e-6) / 1e9 if k_us > 0 else 0.0
        iso_us = 1e3 * iso_ms / max(iso_n, 1)
        out = {
            "metric": "frames/sec front-end (%s%scuboid: Canny+DT+sweep+score+select) @640x480" % ("ORB extract + " if orb is not None else "", "LSD+LBD lines + " if lsd is not None else ""),
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "front-end per frame: ORB extract + LSD/LBD lines + detect_3d_cuboid: 640x480 frames (3 drawn cuboids over a 1/f texture, amplitude %.2f) x %d boxes, "
                                   "180-yaw x 3-VP sweep (yaw step %.2f deg), %d frames resident per GPU; detect_cuboid is fed the scene's own edge list (cuboid edges + 40 clutter "
                                   "segments), not this step's LSD output (decoupled, SURVEY 8d C2)" % (BG_TEXTURE, args.boxes, args.yaw_step, args.frames),
                       "lines": None if lsd is None else {"keylines_per_step": n_lines, "keylines_per_frame": n_lines / args.frames, "descriptor": "LBD 32 B", "detectors_in_flight": len(ctx_lines), "runner": "phased" if args.phased else "alternating",
                                                                "region_stage": ("device: one wave per frame (lsd_rg_seq)" if lsd.region_stats()["device"] else "host: %d OpenMP threads" % _lib.lib().cs_host_thread_count())},
                       "orb": None if orb is None else {"nfeatures": args.orb_features, "levels": 8, "keypoints_per_step": n_kp, "keypoints_per_frame": n_kp / args.frames},
                       "frames_per_gpu": args.frames, "boxes_per_frame": args.boxes,
                       "hypotheses_per_step": st["n_hypotheses"], "valid_proposals_per_step": st["n_valid"],
                       "roi_pixels_per_step": st["roi_pixels"], "score_kernels": ss, "parallelism": "frames sharded, no collective"},
            "roofline": {"bound": "hbm", "kernel": "cuboid_sweep_score", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if tr is None else tr["bytes"], "traffic_detail": tr, "avg_kernel_us": k_us,
                         "frac_by_traffic": None if tr is None or k_us <= 0 else tr["bytes"] / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,  # the bytes that crossed HBM (PMC) over the same time
                         "peak_note": "peak = 8000 GB/s, the spec figure of MI355X_MICROARCH.md; a device copy reaches 6290 GB/s on this part: frac x 1.27 is the fraction of that",
                         "limiter": "vector-ALU issue, not HBM: one launch is 144.1 M vector + 14.5 M scalar wave-instructions (profiles/r05_pmc_sweep_score.txt; per 64-proposal task ~2 180: f64 sample addressing "
                                    "and decode ~1 380, six f64 atan2 ~340, corner construction ~400), 235 us at one vector instruction per SIMD per 4 cycles -- `bound` stays \"hbm\" because SURVEY 8d prices this kernel in bytes",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_formula": "4*A + 200*n_valid (SURVEY 8d, fused K3+K4: float distance-map ROI read once + the reference's 25 doubles per valid proposal)",
                         "accountings": {
                             "must_move_4A_plus_20_n_valid": {"bytes": 4.0 * st["roi_pixels"] + 20.0 * st["n_valid"],
                                                              "frac": (4.0 * st["roi_pixels"] + 20.0 * st["n_valid"]) / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS if k_us > 0 else None,
                                                              "note": "what this kernel has to move: the float map once, 4 B of proposal list in and 2 error doubles out per valid proposal"},
                             "edge_scoring_stage_filter_plus_score": {"us": k_us + kernels["cuboid_sweep_filter"]["avg_us"],
                                                                      "frac": alg_bytes / ((k_us + kernels["cuboid_sweep_filter"]["avg_us"]) * 1e-6) / 1e9 / HBM_PEAK_GBS if k_us > 0 else None,
                                                                      "note": "the same bytes over cuboid_sweep_filter (reject tests of all hypotheses) + cuboid_sweep_score: everything between the distance transform and the selection"}},
                         "isolated": {"avg_kernel_us": iso_us, "frac": (alg_bytes / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if iso_n else None, "threads_per_workgroup": 512,
                                      "note": "cuboid path alone on the GPU, the kernel in the shape it has when it owns the CUs (512 threads, 2 waves per SIMD)"},
                         "isolated_shared_shape": None if True not in iso else {"avg_kernel_us": 1e3 * iso[True][0] / max(iso[True][1], 1), "threads_per_workgroup": 256,
                                                                                  "frac": alg_bytes / (1e3 * iso[True][0] / max(iso[True][1], 1) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                                                  "note": "alone on the GPU in the shape of the timed region: 256 threads of at most 128 registers, which fit beside a CU's sixteen lsd_rg_seq waves"},
                         "phased_runner": None if phased_alt is None else dict(phased_alt, frac_in_run=alg_bytes / (phased_alt["score_kernel_us_in_run"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                                               note="the other runner on the same objects: the region stages of the detectors run together with the ORB / cuboid stream idle, the score kernel "
                                                                                    "(512-thread shape) only meets the detectors' map / rectangle / LBD kernels"),
                         "in_run_note": (None if lsd is None or not lsd.region_stats()["device"] else
                                         ("phased runner: lsd_rg_seq (one wave per frame, sixteen per CU) of the %d detectors runs between the super-steps with this stream idle, so this kernel never "
                                          "shares a CU with it; in the timed region it shares the chip with the detectors' map / rectangle / LBD kernels on their own streams (HBM-bound, many "
                                          "workgroups): `frac` is the in-run figure, `isolated.frac` the kernel's own" % len(ctx_lines)) if args.phased else
                                         ("alternating runner (the faster one): every line detector in flight holds frames / 16 CUs for its whole lsd_rg_seq (one wave per frame, sixteen per CU) -- "
                                          "%d of 256 CUs with %d detectors x %d frames -- and this kernel runs BESIDE those walks in its 256-thread shape (one wave per SIMD in the slot the walks leave, "
                                          "the LDS they do not use), sharing every SIMD with four walking waves: `frac` is that in-run figure; `isolated` is the kernel alone in its own shape, "
                                          "`phased_runner` what the runner that keeps it away from the walks measures"
                                          % (min(256, len(ctx_lines) * ((args.frames + 15) // 16)), len(ctx_lines), args.frames)))},
            "kernels_us": {k: round(v["avg_us"], 2) for k, v in kernels.items()},
            "host_threads": _lib.lib().cs_host_thread_count(),
            "hbm_in_use_gb": round((lambda fr_to: (fr_to[1] - fr_to[0]) / 1e9)(torch.cuda.mem_get_info()), 1),  # everything resident for the run (all blocks of this line)
            "hbm_by_path_gb": {b[0]: round((a[1] - b[1]) / 1e9, 2) for a, b in zip(hbm_marks[:-1], hbm_marks[1:])},  # the front-end's own working set, engine by engine (after_warmup: buffers sized at the first run)
        }
        out.update(extra)
        if variants is not None:
            out["runner_variants"] = variants
        if not args.no_cpu and world == 1:
            flags = native_oracle()
            cores = _lib.lib().cs_host_thread_count()
            out["cpu_baseline"] = cpu_baseline(scenes[:16], args.yaw_step, flags, with_orb=orb is not None, nfeat=args.orb_features, with_lines=lsd is not None)
            out["cpu_baseline"]["host_cores_available"] = cores
            out["cpu_baseline_mt"] = cpu_baseline(scenes[:max(16, cores)], args.yaw_step, flags, budget_s=8.0, with_orb=orb is not None, nfeat=args.orb_features,
                                                  with_lines=lsd is not None, threads=cores)
            try:
                rb = cpu_baseline_ref(scenes[:16], args.yaw_step, with_orb=orb is not None, nfeat=args.orb_features, with_lines=lsd is not None)
            except Exception as e:  # (a checker library that does not load must not cost the line)
                rb = {"error": "%s: %s" % (type(e).__name__, e)}
            if rb is not None:
                out["cpu_baseline_ref"] = rb
        if ba_out is not None:
            out["ba"] = ba_out
        print(json.dumps(out))
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
