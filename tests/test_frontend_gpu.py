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
