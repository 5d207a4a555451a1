"""GPU parity: LBD line descriptors (blur / Sobel maps, 72-float vectors, 32-byte binary strings) and the line matcher vs the oracle."""
import os

import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.lsd import line_lbd_detect

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _images():
    return [np.load(os.path.join(GOLD, "orb_cabinet.npz"))["gray"], synth.cuboid_scene(3)["gray"], synth.texture_image(8, 640, 480)]


def test_maps_bit_exact(ctx, oracle):
    det = line_lbd_detect(640, 480, ctx=ctx)
    for img in _images():
        for a, b in zip(det.lbd_maps(img), oracle.lbd_maps(img)):
            assert np.array_equal(a, b)
    det.close()


def test_descriptors_bit_exact(ctx, oracle):
    det = line_lbd_detect(640, 480, ctx=ctx)
    for img in _images():
        kl = det.detect_raw_lines(img)
        assert len(kl) > 10
        desc, fd = det.get_line_descriptors(img, kl, want_float=True)
        rdesc, rfd = oracle.lbd_compute(img, kl, want_float=True)
        assert np.array_equal(fd.view(np.uint32), rfd.view(np.uint32)), "72-float LBD vectors"
        assert np.array_equal(desc, rdesc)
        assert np.all(np.abs(np.linalg.norm(fd.astype(np.float64), axis=1) - 1) < 1e-5) and fd.max() <= 0.4 / 0.4  # unit vectors
    det.line_length_thres = 15.0
    kl2, d2 = det.detect_descrip_lines(_images()[0])
    assert len(kl2) == len(d2) and np.all(kl2["lineLength"] > 15.0)
    assert len(det.get_line_descriptors(_images()[0], kl2[:0])) == 0
    det.close()


def test_lines_near_border_and_other_size(ctx, oracle):
    det = line_lbd_detect(1241, 376, ctx=ctx)
    img = synth.cuboid_scene(5, W=1241, H=376)["gray"]
    kl = det.detect_raw_lines(img)
    # push a few lines against the image border: the support region is clamped to the image (computeLBD :1283-1290)
    kl = kl.copy()
    kl["sPointInOctaveX"][:5] = 0.0; kl["ePointInOctaveY"][5:10] = 375.9
    assert np.array_equal(det.get_line_descriptors(img, kl), oracle.lbd_compute(img, kl))
    det.close()


def test_match_line_descrip(ctx, oracle):
    det = line_lbd_detect(640, 480, ctx=ctx)
    a = synth.cuboid_scene(3)["gray"]
    b = np.roll(a, 3, axis=1)  # small image shift: most lines re-found with close descriptors
    ka, da = det.detect_descrip_lines(a); kb, db = det.detect_descrip_lines(b)
    qi, ti, d = det.match_line_descrip(da, db, 25.0)
    # brute-force 1-NN with first-index ties
    dist = np.unpackbits(da[:, None, :] ^ db[None, :, :], axis=2).sum(2)
    best = dist.argmin(1); bd = dist.min(1); keep = bd < 25
    assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, best[keep]) and np.array_equal(d, bd[keep])
    assert len(qi) > 5
    assert len(det.match_line_descrip(da, db[:0])[0]) == 0
    det.close()


def test_resident_batch_lines_and_descriptors(ctx, oracle):
    imgs = _images()
    det = line_lbd_detect(640, 480, max_frames=3, ctx=ctx)
    det.upload(np.stack(imgs))
    det.run(with_lbd=True)
    det.run(with_lbd=True)  # re-running on the resident frames gives the same answer
    for f, img in enumerate(imgs):
        kl, desc = det.read(f)
        ref = oracle.lsd_detect(img)
        assert kl.tobytes() == ref.tobytes()
        assert np.array_equal(desc, oracle.lbd_compute(img, ref))
    det.close()
