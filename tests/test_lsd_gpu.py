"""GPU parity: LSD line detector (device gradient maps bit-exact, segments identical) vs the CPU oracle."""
import os

import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.lsd import line_lbd_detect

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_maps_and_keylines_bit_exact(ctx, oracle):
    imgs = [np.load(os.path.join(GOLD, "orb_cabinet.npz"))["gray"], synth.cuboid_scene(3)["gray"], synth.texture_image(8, 640, 480)]
    det = line_lbd_detect(640, 480, max_frames=3, ctx=ctx)
    got = det.detect_raw_lines(np.stack(imgs))
    for f, img in enumerate(imgs):
        sc, mg, an = det.maps(f)
        rsc, rmg, ran, _ = oracle.lsd_maps(img)
        assert np.array_equal(sc, rsc), "scaled image"
        inner = (slice(0, -1), slice(0, -1))
        assert np.array_equal(mg[inner], rmg[inner]) and np.array_equal(an, ran), "gradient norm / level-line angle"
        ref = oracle.lsd_detect(img)
        assert len(got[f]) == len(ref) and len(ref) > 10
        assert got[f].tobytes() == ref.tobytes()
    det.line_length_thres = 15.0  # object_slam/src/main_obj.cpp:366
    fl = det.detect_filter_lines(imgs[0])
    assert np.array_equal(fl, oracle.lsd_detect_filter_lines(imgs[0], 15.0)) and len(fl) > 50
    det.close()


def test_other_sizes_and_flat(ctx, oracle):
    det = line_lbd_detect(1241, 376, ctx=ctx)
    img = synth.cuboid_scene(5, W=1241, H=376)["gray"]
    assert det.detect_raw_lines(img).tobytes() == oracle.lsd_detect(img).tobytes()
    assert len(det.detect_raw_lines(np.full((376, 1241), 90, np.uint8))) == 0
    det.close()


@pytest.mark.parametrize("stage,shape", [("seq", ""), ("wlk", "1,8,1"), ("wlk", "1,8,2"), ("wlk", "2,8,1"), ("wlk", "8,16,1")])
def test_device_region_stage_equals_host_stage(ctx, oracle, monkeypatch, stage, shape):
    """Region growing / rectangles / NFA on the device (lsd_regions.hip: the reference's sequence walked by one wave per frame, lsd_rg_seq.h; by walker
    waves with one lane per frame + rectangle waves, lsd_rg_wlk.h, in several workgroup shapes: walkers, waves, accepted pixels per iteration) against
    the host stage and the oracle: KeyLines byte for byte.  The batch is small, so the stage is asked for; large batches take `seq` by themselves."""
    if shape:
        monkeypatch.setenv("CUBESLAM_LSD_WLK", shape)
    imgs = [np.load(os.path.join(GOLD, "orb_cabinet.npz"))["gray"], synth.cuboid_scene(7, n_boxes=3, bg_texture=0.5)["gray"], synth.texture_image(8, 640, 480)]
    imgs += [synth.cuboid_scene(60 + i, n_boxes=3, bg_texture=0.125 * i)["gray"] for i in range(8)]
    det = line_lbd_detect(640, 480, max_frames=len(imgs), ctx=ctx)
    host = det.detect_raw_lines(np.stack(imgs))
    assert not det.region_stats()["device"]
    monkeypatch.setenv("CUBESLAM_LSD_REGIONS", stage)
    dev = det.detect_raw_lines(np.stack(imgs))
    st = det.region_stats()
    assert st["device"] and not st["host_fallback"] and st["candidates"] > 100 and st["fetches"] > 1000
    for f, img in enumerate(imgs):
        assert dev[f].tobytes() == host[f].tobytes()
        if f < 5:
            assert dev[f].tobytes() == oracle.lsd_detect(img).tobytes()
    det.close()


@pytest.mark.parametrize("stage", ["seq", "wlk"])
def test_device_region_stage_other_size_and_capacity_fallback(ctx, oracle, monkeypatch, stage):
    """The device stage on KITTI-sized frames, and its way out: a region larger than the wave's list (here cut to 64 pixels) hands the batch to the
    host stage -- same KeyLines, and the statistics say so."""
    monkeypatch.setenv("CUBESLAM_LSD_REGIONS", stage)
    imgs = [synth.cuboid_scene(5 + i, W=1241, H=376, bg_texture=0.5 * i)["gray"] for i in range(2)]
    det = line_lbd_detect(1241, 376, max_frames=2, ctx=ctx)
    got = det.detect_raw_lines(np.stack(imgs))
    st = det.region_stats()
    assert st["device"] and not st["host_fallback"]
    want = [oracle.lsd_detect(im) for im in imgs]
    for f in range(2):
        assert got[f].tobytes() == want[f].tobytes() and len(want[f]) > 10
    monkeypatch.setenv("CUBESLAM_LSD_SEQ_CAP", "64")
    got = det.detect_raw_lines(np.stack(imgs))
    st = det.region_stats()
    assert st["device"] and st["host_fallback"]
    for f in range(2):
        assert got[f].tobytes() == want[f].tobytes()
    det.close()


def test_region_stage_by_the_api(ctx, oracle, monkeypatch):
    """cs_lsd_set_region_stage: the caller's choice per detector (the backlog stage has no other way in), same KeyLines from every stage."""
    monkeypatch.delenv("CUBESLAM_LSD_REGIONS", raising=False)
    imgs = [synth.cuboid_scene(90 + i, n_boxes=3, bg_texture=0.25 * i)["gray"] for i in range(4)]
    det = line_lbd_detect(640, 480, max_frames=len(imgs), ctx=ctx)
    want = [oracle.lsd_detect(im) for im in imgs]
    for stage, device in (("auto", False), ("backlog", True), ("wave_per_frame", True), ("host", False)):
        det.set_region_stage(stage)
        got = det.detect_raw_lines(np.stack(imgs))
        assert det.region_stats()["device"] == device, stage
        for f in range(len(imgs)):
            assert got[f].tobytes() == want[f].tobytes(), (stage, f)
    from cube_slam_amd._lib import lib
    assert lib().cs_lsd_set_region_stage(det._l, 9) != 0  # not a stage
    det.close()


def test_large_batches_take_the_device_region_stage(ctx, oracle, monkeypatch):
    """512 frames (16 distinct ones, repeated) through the resident-batch form: the device stage is the default there, all frames give their own lines."""
    monkeypatch.delenv("CUBESLAM_LSD_REGIONS", raising=False)
    base = [synth.cuboid_scene(40 + i, n_boxes=3, bg_texture=0.5)["gray"] for i in range(16)]
    F = 512
    det = line_lbd_detect(640, 480, max_frames=F, ctx=ctx)
    det.upload(np.stack([base[i % 16] for i in range(F)]))
    det.run(with_lbd=True)
    st = det.region_stats()
    assert st["device"] and not st["host_fallback"]
    want = [oracle.lsd_detect(b) for b in base[:4]]
    for f in (0, 1, 2, 3, 16, 17, 258, 511):
        kl, desc = det.read(f)
        assert kl.tobytes() == want[f % 16].tobytes() if f % 16 < 4 else len(kl) > 50
        if f >= 16:
            k0, d0 = det.read(f % 16)
            assert kl.tobytes() == k0.tobytes() and desc.tobytes() == d0.tobytes()
    det.close()
