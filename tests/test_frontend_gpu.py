"""GPU test of the batch front-end runner (cs_frontend_*): same results as the individual calls, repeatable, workers drained."""
import numpy as np
import pytest

from cube_slam_amd import _lib, synth
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid
from cube_slam_amd.frontend import Frontend
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor

pytestmark = pytest.mark.gpu


def test_frontend_equals_separate_calls(ctx, oracle):
    scenes = [synth.cuboid_scene(500 + i, n_boxes=3) for i in range(6)]
    gray = np.stack([s["gray"] for s in scenes])
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    batch = CuboidBatch(ctx, gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
    orb = ORBextractor(500, 1.2, 8, 20, 7, 640, 480, max_frames=len(scenes), ctx=ctx); orb.upload(gray)
    lctx = [_lib.Context(0), _lib.Context(0)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(gray)
    fe = Frontend(ctx, orb=orb, batch=batch, line_detectors=lsds)
    for _ in range(5):  # odd count: both workers have run, the last pass sits in worker 0
        fe.step()
    fe.drain()
    cub = batch.read()
    kps = orb.read()
    for f, s in enumerate(scenes):
        rk, rd = oracle.ORBextractor(500, 1.2, 8, 20, 7)(s["gray"])
        assert kps[f][0].tobytes() == rk.tobytes() and np.array_equal(kps[f][1], rd)
        ref_kl = oracle.lsd_detect(s["gray"])
        for d in lsds:
            kl, desc = d.read(f)
            assert kl.tobytes() == ref_kl.tobytes() and np.array_equal(desc, oracle.lbd_compute(s["gray"], ref_kl))
    ref, _ = oracle.detect_cuboid(scenes[2]["gray"], scenes[2]["K"], scenes[2]["Twc"], scenes[2]["boxes"], scenes[2]["lines"], opts=oracle.cuboid_opts())
    off = sum(len(s["boxes"]) for s in scenes[:2])
    for k, r in enumerate(ref):
        assert len(cub[off + k]) == len(r) and np.array_equal(cub[off + k]["box_corners_2d"], r["box_corners_2d"])
    # the cuboid batch on a stream of its own beside the ORB pass (cs_frontend_set_cuboid_ctx): the same cuboids, the same key points
    cctx = _lib.Context(0)
    fe.set_cuboid_ctx(cctx)
    for _ in range(3):
        fe.step()
    fe.drain(); ctx.sync()
    cub2, kps2 = batch.read(), orb.read()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, cub2)) and all(a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) for a, b in zip(kps, kps2))
    with pytest.raises(Exception):
        fe.set_cuboid_ctx(lctx[0])  # (a line worker's context is taken)
    fe.set_cuboid_ctx(None)
    fe.close()
    with pytest.raises(Exception):
        Frontend(ctx, orb=orb, batch=batch, line_detectors=[line_lbd_detect(640, 480, ctx=ctx)])  # a worker may not share the caller's context


def test_frontend_phased_passes(ctx, oracle, monkeypatch):
    """Phased runner (cs_frontend_set_phased): the detectors wait in front of the device region stage until one pass per detector is
    submitted; incomplete super-steps are released by drain; lines and descriptors equal the oracle's either way."""
    monkeypatch.setenv("CUBESLAM_LSD_REGIONS", "seq")  # the device stage at a test-sized batch
    scenes = [synth.cuboid_scene(900 + i, n_boxes=2) for i in range(4)]
    gray = np.stack([s["gray"] for s in scenes])
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    batch = CuboidBatch(ctx, gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
    orb = ORBextractor(500, 1.2, 8, 20, 7, 640, 480, max_frames=len(scenes), ctx=ctx); orb.upload(gray)
    lctx = [_lib.Context(0) for _ in range(3)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(gray)
    fe = Frontend(ctx, orb=orb, batch=batch, line_detectors=lsds, phased=True)
    for _ in range(7):  # two full super-steps and one pass of a third
        fe.step()
    fe.drain()
    fe.drain()
    fe.step()  # a single pass after a drain
    fe.drain()
    fe.set_phased(False)
    fe.step()
    fe.drain()
    for d in lsds:
        assert d.region_stats()["device"] == 1
    for f, s in enumerate(scenes):
        ref_kl = oracle.lsd_detect(s["gray"])
        for d in lsds:
            kl, desc = d.read(f)
            assert kl.tobytes() == ref_kl.tobytes() and np.array_equal(desc, oracle.lbd_compute(s["gray"], ref_kl))
    fe.close()
    # a detector that was under a runner runs on its own afterwards
    lsds[0].run(); kl, _ = lsds[0].read(0)
    assert kl.tobytes() == oracle.lsd_detect(scenes[0]["gray"]).tobytes()


def test_frontend_chain_is_the_reference_chain_pipelined(ctx, oracle):
    """cs_frontend_set_chain: a step's cuboid pass is fed the lines detect_filter_lines found in the pass its line worker finished last (the reference's
    chain, object_slam/src/main_obj.cpp:428-449, with the line pass running W steps ahead).  After the first W steps every frame's cuboids equal the
    sequential chain (line pass, hand-over, cuboid pass) on the same frames, and the oracle's chain on a frame."""
    scenes = [synth.cuboid_scene(700 + i, n_boxes=2, bg_texture=0.5) for i in range(5)]
    gray = np.stack([s["gray"] for s in scenes])
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    args = (gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes])
    batch = CuboidBatch(ctx, *args, [s["lines"] for s in scenes], det.opts())
    batch.run()
    decoupled = batch.read()
    lctx = [_lib.Context(0) for _ in range(3)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(gray)
        d.line_length_thres = 15.0
    # the sequential chain
    lsds[0].run(False)
    lines = lsds[0].read_filter_lines(len(scenes))
    seq = CuboidBatch(ctx, *args, [np.asarray(l, np.float64) for l in lines], det.opts())
    seq.run()
    want = seq.read()
    seq.close()
    fe = Frontend(ctx, orb=None, batch=batch, line_detectors=lsds)
    fe.set_chain(True, 15.0)
    for k in range(3):  # the workers' first passes: the batch still runs on the lists it was created with
        fe.step()
    ctx.sync()
    first = batch.read()
    assert all(len(a) == len(b) and (len(a) == 0 or np.array_equal(a["box_corners_2d"], b["box_corners_2d"])) for a, b in zip(first, decoupled))
    for k in range(4):
        fe.step()
    fe.drain(); ctx.sync()
    got = batch.read()
    differs = 0
    for a, b, c in zip(got, want, decoupled):
        assert len(a) == len(b) and (len(a) == 0 or (np.array_equal(a["box_corners_2d"], b["box_corners_2d"]) and np.array_equal(a["normalized_error"], b["normalized_error"])))
        differs += len(a) != len(c) or (len(a) > 0 and not np.array_equal(a["box_corners_2d"], c["box_corners_2d"]))
    assert differs > 0, "the handed-over lines change some cuboid"
    s = scenes[1]
    ref, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], np.asarray(oracle.lsd_detect_filter_lines(s["gray"], 15.0), np.float64), opts=oracle.cuboid_opts())
    off = sum(len(x["boxes"]) for x in scenes[:1])
    for k, r in enumerate(ref):
        assert len(got[off + k]) == len(r) and np.array_equal(got[off + k]["box_corners_2d"], r["box_corners_2d"])
    fe.set_chain(False)
    fe.close()


def test_frontend_backlog_runs_the_same_passes_ahead_of_the_caller(ctx, oracle):
    """cs_frontend_set_backlog: the line passes of the announced steps start as soon as a worker is free (up to 2 W ahead of the caller) -- the same number of
    passes, the same results; a backlog cut by a drain leaves passes that the following steps find done; the chain under a backlog is the chain."""
    scenes = [synth.cuboid_scene(700 + i, n_boxes=2, bg_texture=0.5) for i in range(5)]
    gray = np.stack([s["gray"] for s in scenes])
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    args = (gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes])
    batch = CuboidBatch(ctx, *args, [s["lines"] for s in scenes], det.opts())
    orb = ORBextractor(500, 1.2, 8, 20, 7, 640, 480, max_frames=len(scenes), ctx=ctx); orb.upload(gray)
    lctx = [_lib.Context(0) for _ in range(3)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(gray)
        d.line_length_thres = 15.0
    fe = Frontend(ctx, orb=orb, batch=batch, line_detectors=lsds)

    def passes():  # line passes run so far: launches of a kernel every pass makes once
        return sum(c.timing_get("lsd_gradient")[1] for c in lctx)
    for c in lctx:
        c.timing(True); c.timing_reset()
    fe.set_backlog(7)
    for _ in range(7):
        fe.step()
    fe.drain(); ctx.sync()
    assert passes() == 7
    cub, kps = batch.read(), orb.read()
    for f, s in enumerate(scenes):
        rk, rd = oracle.ORBextractor(500, 1.2, 8, 20, 7)(s["gray"])
        assert kps[f][0].tobytes() == rk.tobytes() and np.array_equal(kps[f][1], rd)
        ref_kl = oracle.lsd_detect(s["gray"])
        for d in lsds:
            kl, desc = d.read(f)
            assert kl.tobytes() == ref_kl.tobytes() and np.array_equal(desc, oracle.lbd_compute(s["gray"], ref_kl))
    # a backlog of 6, cut after 2 steps: the workers have started at most 2 + 2 W passes and at least the two asked for; the steps that follow start what is missing, no more
    fe.set_backlog(6)
    fe.step(); fe.step()
    fe.drain()
    ahead = passes() - 9
    assert 0 <= ahead <= 6
    for _ in range(6):
        fe.step()
    fe.drain(); ctx.sync()
    assert passes() == 15
    assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, batch.read()))
    for c in lctx:
        c.timing(False)
    # the chain under a backlog: step k is fed the lines of pass k - W
    lsds[0].run(False)
    lines = lsds[0].read_filter_lines(len(scenes))
    seq = CuboidBatch(ctx, *args, [np.asarray(l, np.float64) for l in lines], det.opts())
    seq.run()
    want = seq.read()
    seq.close()
    fe.set_chain(True, 15.0)
    fe.set_backlog(8)
    for k in range(3):
        fe.step()
    ctx.sync()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, batch.read()))  # the first W steps: the lists the batch holds
    for k in range(5):
        fe.step()
    fe.drain(); ctx.sync()
    got = batch.read()
    assert all(len(a) == len(b) and (len(a) == 0 or (np.array_equal(a["box_corners_2d"], b["box_corners_2d"]) and np.array_equal(a["normalized_error"], b["normalized_error"]))) for a, b in zip(got, want))
    assert any(a.tobytes() != b.tobytes() for a, b in zip(got, cub)), "the handed-over lines change some cuboid"
    fe.set_chain(False)
    fe.close()


def test_streaming_source_equals_resident_frames(ctx, oracle):
    """cs_frontend_stream_*: every step takes NEW pixels from the host through the ring (H2D on a copy stream, device copies into ORB, the cuboid batch and the step's line
    pass).  Five steps over three distinct pixel sets: ORB key points / descriptors, KeyLines / LBD descriptors of every pass and the cuboids equal objects that were
    created on those pixels; a step without pushed frames and a push beyond the ring are refused."""
    scenes = [synth.cuboid_scene(700 + i, n_boxes=2) for i in range(4)]
    base = np.stack([s["gray"] for s in scenes])
    rng = np.random.default_rng(5)
    sets = [np.ascontiguousarray(base ^ rng.integers(0, 4, base.shape, dtype=np.uint8)) for _ in range(3)]  # the same geometry, other pixels
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    mk = lambda g: CuboidBatch(ctx, g, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())  # noqa: E731
    batch = mk(base)
    orb = ORBextractor(500, 1.2, 8, 20, 7, 640, 480, max_frames=len(scenes), ctx=ctx); orb.upload(base)
    lctx = [_lib.Context(0), _lib.Context(0)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(base)
    fe = Frontend(ctx, orb=orb, batch=batch, line_detectors=lsds)
    fe.step(); fe.drain()  # (a resident step first: the ring takes over at step 1)
    fe.stream_begin(len(scenes), 640, 480, n_slots=2)
    with pytest.raises(Exception):
        fe.step()  # nothing pushed
    order = [0, 1, 2, 1, 0]
    fe.stream_push(sets[order[0]])
    for k, which in enumerate(order):
        if k + 1 < len(order):
            fe.stream_push(sets[order[k + 1]])  # the next step's frames while this one runs
        if k == 0:
            with pytest.raises(Exception):
                fe.stream_push(sets[0])  # both slots hold frames of steps that have not run
        fe.step()
        if k % 2 == 0:  # the step's results through the ring's own read-back (pinned buffers are the caller's business; any host memory works) ...
            from cube_slam_amd.cuboid import CUBOID_DTYPE
            from cube_slam_amd.orb import KEYPOINT_DTYPE
            a_k, a_d = np.zeros(len(scenes) * orb.cap, KEYPOINT_DTYPE), np.zeros((len(scenes) * orb.cap, 32), np.uint8)
            a_c, a_n = np.zeros((batch.n_boxes, batch.max_cuboid_num), CUBOID_DTYPE), np.zeros(batch.n_boxes, np.int32)
            first, total = fe.stream_read_async(a_k, a_d, a_c, a_n)
            fe.stream_read_wait()
            kps = [(a_k[first[f]:first[f + 1]], a_d[first[f]:first[f + 1]]) for f in range(len(scenes))]
            cub = [a_c[i, :a_n[i]] for i in range(batch.n_boxes)]
            assert total == first[-1]
        else:           # ... or the objects' own reads
            ctx.sync()
            kps, cub = orb.read(), batch.read()
        want_b = mk(sets[which]); want_b.run(); want_c = want_b.read(); want_b.close()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, want_c)), k
        for f in range(len(scenes)):
            rk, rd = oracle.ORBextractor(500, 1.2, 8, 20, 7)(sets[which][f])
            assert kps[f][0].tobytes() == rk.tobytes() and np.array_equal(kps[f][1], rd), (k, f)
    fe.drain()
    # the last two passes sit in the two detectors: pass 4 (set order[4]) and pass 3 (set order[3]) -- a pass goes to the first free worker, so look at the pixels it holds
    seen = set()
    for d in lsds:
        got = [d.read(f) for f in range(len(scenes))]
        hit = [w for w in (order[3], order[4]) if got[0][0].tobytes() == oracle.lsd_detect(sets[w][0]).tobytes()]
        assert hit, "a detector holds lines of no streamed set"
        w = hit[0]; seen.add(w)
        for f in range(len(scenes)):
            ref_kl = oracle.lsd_detect(sets[w][f])
            assert got[f][0].tobytes() == ref_kl.tobytes() and np.array_equal(got[f][1], oracle.lbd_compute(sets[w][f], ref_kl))
    assert seen == {order[3], order[4]}
    fe.stream_end()
    fe.step(); fe.drain()  # resident again (the objects keep the last streamed pixels)
    fe.close(); batch.close(); orb.close()
    for d in lsds:
        d.close()


def test_streaming_source_carries_boxes_poses_and_lines(ctx, oracle):
    """cs_frontend_stream_push_scene / cs_cuboid_batch_set_scene: every step brings its own frames WITH their 2-D boxes, camera poses and edge lists -- other geometry, other box
    counts (the third set has three boxes in two of its frames: the arenas grow), other proposal counts -- and the cuboids of every step equal a fresh batch created on that
    step's scenes (and, for one frame per step, the oracle's detect_cuboid).  A scene whose box leaves the image is refused and leaves the batch as it was."""
    F = 3
    sets = []
    for k, nb in enumerate((2, 1, 3, 2)):
        sc = [synth.cuboid_scene(900 + 10 * k + i, n_boxes=nb if i else max(1, nb - 1), bg_texture=0.25 * k) for i in range(F)]
        sets.append(sc)
    K = sets[0][0]["K"]
    det = detect_3d_cuboid(ctx); det.set_calibration(K)
    gray_of = lambda sc: np.ascontiguousarray(np.stack([s["gray"] for s in sc]))  # noqa: E731
    mk = lambda sc: CuboidBatch(ctx, gray_of(sc), K, np.stack([s["Twc"] for s in sc]), [s["boxes"] for s in sc], [s["lines"] for s in sc], det.opts())  # noqa: E731
    pack = lambda sc: CuboidBatch.pack_scene(np.stack([s["Twc"] for s in sc]), [s["boxes"] for s in sc], [s["lines"] for s in sc])  # noqa: E731
    batch = mk(sets[0])
    fe = Frontend(ctx, orb=None, batch=batch, line_detectors=[])
    fe.stream_begin(F, 640, 480, n_slots=2)
    order = [1, 2, 0, 3, 2]
    grays = [gray_of(sc) for sc in sets]
    fe.stream_push_scene(grays[order[0]], pack(sets[order[0]]))
    from cube_slam_amd.cuboid import CUBOID_DTYPE
    for k, which in enumerate(order):
        if k + 1 < len(order):
            fe.stream_push_scene(grays[order[k + 1]], pack(sets[order[k + 1]]))
        fe.step()
        nb = sum(len(s["boxes"]) for s in sets[which])
        if k % 2 == 0:
            a_c, a_n = np.zeros((nb, batch.max_cuboid_num), CUBOID_DTYPE), np.zeros(nb, np.int32)
            fe.stream_read_async(None, None, a_c, a_n)
            fe.stream_read_wait()
            cub = [a_c[i, :a_n[i]] for i in range(nb)]
        else:
            ctx.sync()
            batch.n_boxes = nb
            cub = batch.read()
        want_b = mk(sets[which]); want_b.run(); want_c = want_b.read(); want_b.close()
        assert len(cub) == len(want_c) == nb and sum(len(c) for c in want_c) >= 1
        assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, want_c)), (k, which)
        s0 = sets[which][0]
        ref, _ = oracle.detect_cuboid(s0["gray"], K, s0["Twc"], s0["boxes"], s0["lines"])
        for g, r in zip(cub[:len(ref)], ref):
            assert len(g) == len(r)
            for name in ("pos", "scale", "rotY", "edge_distance_error", "edge_angle_error"):
                assert np.allclose(g[name], r[name], rtol=1e-5, atol=1e-9), (k, name)
    fe.stream_end()
    # directly on the batch: a box outside the image is refused, the batch still holds the last scene
    bad = [dict(s) for s in sets[0]]
    bad[1] = dict(bad[1], boxes=np.array([[700.0, 100.0, 50.0, 50.0, 0.9]]))
    ctx.sync()
    with pytest.raises(Exception):
        batch.set_scene(np.stack([s["Twc"] for s in bad]), [s["boxes"] for s in bad], [s["lines"] for s in bad])
    batch.run(); ctx.sync()
    batch.n_boxes = sum(len(s["boxes"]) for s in sets[order[-1]])
    again = batch.read()
    want_b = mk(sets[order[-1]]); want_b.run(); want_c = want_b.read(); want_b.close()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(again, want_c))
    # ... and the edge lists may stay (line_offsets NULL): other boxes over the same frames' lines
    sc = sets[order[-1]]
    alt = [dict(s, boxes=s["boxes"][:1]) for s in sc]
    batch.set_scene(np.stack([s["Twc"] for s in alt]), [s["boxes"] for s in alt], None)
    batch.run(); ctx.sync()
    got = batch.read()
    want_b = mk(alt); want_b.run(); want_c = want_b.read(); want_b.close()
    assert len(got) == len(want_c) and all(a.tobytes() == b.tobytes() for a, b in zip(got, want_c))
    fe.close(); batch.close()


def test_phased_passes_with_the_chain(ctx, oracle, monkeypatch):
    """Phased runner + chained hand-over together (a phased worker still runs rectangles / LBD of pass k - W when step k hands it the next pass; its packet must be
    filed under ITS number): ten steps complete, and the cuboids from the chained lines equal a batch that was given detect_filter_lines' lists directly."""
    monkeypatch.setenv("CUBESLAM_LSD_REGIONS", "seq")
    scenes = [synth.cuboid_scene(800 + i, n_boxes=2) for i in range(4)]
    gray = np.stack([s["gray"] for s in scenes])
    det = detect_3d_cuboid(ctx); det.set_calibration(scenes[0]["K"])
    batch = CuboidBatch(ctx, gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
    lctx = [_lib.Context(0), _lib.Context(0)]
    lsds = [line_lbd_detect(640, 480, max_frames=len(scenes), ctx=c) for c in lctx]
    for d in lsds:
        d.upload(gray)
    fe = Frontend(ctx, orb=None, batch=batch, line_detectors=lsds, phased=True)
    fe.set_chain(True, 15.0)
    import threading
    done = threading.Event()

    def run():
        for _ in range(10):
            fe.step()
        fe.drain()
        done.set()
    th = threading.Thread(target=run, daemon=True); th.start()
    assert done.wait(120), "phased + chained steps did not complete (the gate never opened)"
    cub = batch.read()
    lines = [lsds[0].filter_lines(f, 15.0) if hasattr(lsds[0], "filter_lines") else None for f in range(len(scenes))]
    if all(l is not None for l in lines):
        ref_b = CuboidBatch(ctx, gray, scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes], lines, det.opts())
        ref_b.run(); ref = ref_b.read(); ref_b.close()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(cub, ref))
    fe.close(); batch.close()
    for d in lsds:
        d.close()
