"""GPU parity: HIP ORBextractor vs the CPU oracle, bit-exact (keypoints, angles, descriptors), through the C-ABI."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.orb import ORBextractor

pytestmark = pytest.mark.gpu


def _check_frame(oracle, ext, ora, img, f):
    rk, rd = ora(img)
    for l in range(ext.nlevels):
        lv = ora.level(l)
        assert np.array_equal(ext.level(f, l), lv), "pyramid level %d" % l
        assert np.array_equal(ext.candidates(f, l), ora.candidates(l)), "FAST candidates level %d" % l
        bl = ora.level(l, blurred=True)
        if bl is not None:
            assert np.array_equal(ext.level(f, l, blurred=True), bl), "blurred level %d" % l
    return rk, rd


@pytest.mark.parametrize("W,H,nfeat,kind", [(640, 480, 1000, "texture"), (1241, 376, 2000, "texture"), (640, 480, 500, "scene"), (752, 480, 1200, "texture")])
def test_extract_bit_exact(ctx, oracle, W, H, nfeat, kind):
    imgs = []
    for i in range(2):
        if kind == "texture":
            imgs.append(synth.texture_image(40 + i, W, H, shift=3 * i))
        else:
            imgs.append(synth.cuboid_scene(50 + i, W=W, H=H)["gray"])
    ext = ORBextractor(nfeat, 1.2, 8, 20, 7, W, H, max_frames=2, ctx=ctx)
    ora = oracle.ORBextractor(nfeat, 1.2, 8, 20, 7)
    assert np.array_equal(ext.features_per_level(), ora.features_per_level())
    got = ext.extract_batch(np.stack(imgs))
    for f, img in enumerate(imgs):
        rk, rd = _check_frame(oracle, ext, ora, img, f)
        gk, gd = got[f]
        assert len(gk) == len(rk) and len(gk) > 0
        assert gk.tobytes() == rk.tobytes(), "keypoints (x, y, size, angle, response, octave) bit-exact"
        assert np.array_equal(gd, rd), "descriptors bit-exact"
    ext.close()


def test_flat_image_and_thresholds(ctx, oracle):
    W, H = 640, 480
    flat = np.full((H, W), 77, np.uint8)
    ext = ORBextractor(1000, 1.2, 8, 20, 7, W, H, ctx=ctx)
    k, d = ext(flat)
    assert len(k) == 0 and d.shape == (0, 32)
    # low-contrast texture: most cells fall back to minThFAST
    img = (128 + (synth.texture_image(3, W, H).astype(np.int32) - 128) // 6).astype(np.uint8)
    ora = oracle.ORBextractor(1000, 1.2, 8, 20, 7)
    rk, rd = ora(img)
    gk, gd = ext(img)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    ext.close()
    # other pyramid parameters
    ext2 = ORBextractor(800, 1.5, 4, 30, 10, W, H, ctx=ctx)
    ora2 = oracle.ORBextractor(800, 1.5, 4, 30, 10)
    img2 = synth.texture_image(9, W, H)
    rk, rd = ora2(img2)
    gk, gd = ext2(img2)
    assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
    ext2.close()


def test_device_and_host_quadtree_agree(ctx, monkeypatch):
    """DistributeOctTree on the device (default) and on the host select the same keypoints in the same order."""
    from cube_slam_amd import synth
    from cube_slam_amd.orb import ORBextractor
    imgs = np.stack([synth.texture_image(31 + i, 640, 480, shift=5 * i) for i in range(3)] + [synth.cuboid_scene(9)["gray"]])
    res = {}
    for mode in ("host", "device"):
        if mode == "host":
            monkeypatch.setenv("CUBESLAM_ORB_QUADTREE", "host")
        else:
            monkeypatch.delenv("CUBESLAM_ORB_QUADTREE", raising=False)
        for nfeat in (300, 2000):
            e = ORBextractor(nfeat, 1.2, 8, 20, 7, 640, 480, max_frames=len(imgs), ctx=ctx)
            e.upload(imgs); e.run()
            res[(mode, nfeat)] = e.read()
    for nfeat in (300, 2000):
        for (ka, da), (kb, db) in zip(res[("host", nfeat)], res[("device", nfeat)]):
            assert len(ka) == len(kb) and len(ka) > 50
            assert ka.tobytes() == kb.tobytes() and np.array_equal(da, db)


def test_packed_read_and_device_frames_equal_the_per_frame_paths(ctx):
    """cs_orb_read_packed (two copies for the whole batch) returns what cs_orb_read returns frame by frame, and frames handed over from DEVICE memory
    (cs_orb_set_frames_device, the streaming front-end's hand-over) give the key points of the same frames uploaded from the host."""
    import ctypes as C
    import torch
    from cube_slam_amd._lib import check, lib
    W, H = 640, 480
    imgs = np.stack([synth.texture_image(60 + i, W, H, shift=2 * i) for i in range(3)])
    ext = ORBextractor(800, 1.2, 8, 20, 7, W, H, max_frames=3, ctx=ctx)
    want = ext.extract_batch(imgs)
    kps, desc, first = ext.read_packed()
    assert first[0] == 0 and first[-1] == len(kps) == sum(len(k) for k, _ in want)
    for f in range(3):
        assert kps[first[f]:first[f + 1]].tobytes() == want[f][0].tobytes() and np.array_equal(desc[first[f]:first[f + 1]], want[f][1])
    other = np.ascontiguousarray(imgs[::-1])  # the same frames in another order, from device memory
    d = torch.from_numpy(other).to("cuda:0")
    torch.cuda.synchronize()
    check(ctx.ptr, lib().cs_orb_set_frames_device(ctx.ptr, ext._e, C.c_void_p(d.data_ptr()), 3), "cs_orb_set_frames_device")
    ext.run()
    got = ext.read()
    for f in range(3):
        assert got[f][0].tobytes() == want[2 - f][0].tobytes() and np.array_equal(got[f][1], want[2 - f][1])
    ext.close()
