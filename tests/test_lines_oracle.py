"""CPU known-answer / property tests of the LSD + LBD oracles (oracle/lsd_oracle.cpp, oracle/lbd_oracle.cpp)."""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rect_image(W=320, H=240, x0=60, y0=50, x1=250, y1=180):
    img = np.full((H, W), 40, np.uint8)
    img[y0:y1, x0:x1] = 200
    return img


def _slanted(W=320, H=240, ang=0.3, hu=90, hv=50):
    yy, xx = np.mgrid[0:H, 0:W]
    c, s = np.cos(ang), np.sin(ang)
    u = (xx - W / 2) * c + (yy - H / 2) * s; v = -(xx - W / 2) * s + (yy - H / 2) * c
    return np.where((np.abs(u) < hu) & (np.abs(v) < hv), 200, 40).astype(np.uint8)


def test_lsd_finds_the_sides_of_a_slanted_rectangle(oracle):
    ang, hu, hv = 0.3, 90, 50
    kl = oracle.lsd_detect(_slanted(320, 240, ang, hu, hv))
    assert 3 <= len(kl) <= 6
    c, s = np.cos(ang), np.sin(ang)
    sides = set()
    for k in kl[kl["lineLength"] > 60]:
        P = np.array([[k["startPointX"], k["startPointY"]], [k["endPointX"], k["endPointY"]]], np.float64) - [160, 120]
        u = P[:, 0] * c + P[:, 1] * s; v = -P[:, 0] * s + P[:, 1] * c
        on_long = np.all(np.abs(np.abs(v) - hv) < 2.0) and np.sign(v[0]) == np.sign(v[1])
        on_short = np.all(np.abs(np.abs(u) - hu) < 2.0) and np.sign(u[0]) == np.sign(u[1])
        assert on_long != on_short, "segment lies on one side of the rectangle"
        sides.add(("v", np.sign(v[0])) if on_long else ("u", np.sign(u[0])))
        assert np.isclose(k["lineLength"], np.hypot(k["ePointInOctaveX"] - k["sPointInOctaveX"], k["ePointInOctaveY"] - k["sPointInOctaveY"]), rtol=1e-6)
        assert np.isclose(k["angle"], np.arctan2(k["ePointInOctaveY"] - k["sPointInOctaveY"], k["ePointInOctaveX"] - k["sPointInOctaveX"]), atol=1e-6)
        cheb = max(abs(k["ePointInOctaveX"] - k["sPointInOctaveX"]), abs(k["ePointInOctaveY"] - k["sPointInOctaveY"]))
        assert k["octave"] == 0 and abs(k["numOfPixels"] - (cheb + 1)) <= 1.0, "LineIterator (8-connected) pixel count"
    assert len(sides) >= 3 and ("v", 1.0) in sides and ("v", -1.0) in sides
    assert list(kl["class_id"]) == list(range(len(kl)))


def test_lsd_axis_aligned_quirk_is_kept(oracle):
    """rect_nfa's scanline (lsd.cpp:976-1089: int/int steps, `tailp->p.x` in the second-step slopes) degenerates to one pixel
    column for an exactly horizontal rectangle, so the horizontal sides of an axis-aligned box are rejected by the NFA test while
    the vertical sides are kept.  The restatement reproduces it."""
    kl = oracle.lsd_detect(_rect_image())
    assert len(kl) == 2
    assert np.all(np.abs(kl["startPointX"] - kl["endPointX"]) < 0.01) and np.all(kl["lineLength"] > 120)


def test_lsd_flat_noise_and_filter(oracle):
    assert len(oracle.lsd_detect(np.full((120, 160), 77, np.uint8))) == 0
    rng = np.random.default_rng(3)
    noise = np.clip(128 + rng.normal(0, 2, (240, 320)), 0, 255).astype(np.uint8)
    assert len(oracle.lsd_detect(noise)) <= 2, "NFA control: (almost) nothing in weak noise"
    img = _rect_image()
    kl = oracle.lsd_detect(img)
    for thr in (15.0, 50.0, 150.0):
        fl = oracle.lsd_detect_filter_lines(img, thr)
        keep = kl[(kl["octave"] == 0) & (kl["lineLength"] > thr)]
        assert fl.shape == (len(keep), 4)
        assert np.array_equal(fl, np.stack([keep["startPointX"], keep["startPointY"], keep["endPointX"], keep["endPointY"]], axis=1))


def test_lsd_gradient_maps_known_answer(oracle):
    """ll_angle (lsd.cpp:538-586) on a vertical step: level-line angle is +-pi/2 along the step, NOTDEF (-1024) elsewhere."""
    img = np.full((100, 100), 10, np.uint8); img[:, 50:] = 240
    sc, mg, an, order = oracle.lsd_maps(img)
    assert sc.shape == (80, 80)
    col = np.nonzero(an[40] != -1024.0)[0]
    assert len(col) >= 1 and abs(col.mean() - 39.5) < 1.5
    assert np.allclose(np.abs(an[40, col]), np.pi / 2, atol=1e-6)
    assert np.all(an[:, :30] == -1024.0) and np.all(mg[:, :30] < 1e-9)


def test_lbd_maps_on_a_ramp(oracle):
    """GaussianBlur of a linear ramp is the ramp (the 8-bit fixed-point kernel {14,63,103,63,14}/256 has gain 257^2/65536, which
    rounds back to the input below 64); Sobel dx = 8 * slope, dy = 0 (away from the border)."""
    x = np.arange(160, dtype=np.int32)
    img = np.broadcast_to((20 + x).astype(np.uint8), (120, 160)).copy()
    b, dx, dy = oracle.lbd_maps(img)
    assert np.array_equal(b[:, 3:40], img[:, 3:40])
    assert np.all((b.astype(int) - img)[:, 3:-3] >= 0) and np.all((b.astype(int) - img)[:, 3:-3] <= 2)
    assert np.all(dx[2:-2, 4:38] == 8) and np.all(dy[2:-2, 4:-4] == 0)


def test_lbd_descriptor_properties(oracle):
    img = _rect_image()
    kl = oracle.lsd_detect(img)
    desc, fd = oracle.lbd_compute(img, kl, want_float=True)
    assert desc.shape == (len(kl), 32) and fd.shape == (len(kl), 72)
    assert np.allclose(np.linalg.norm(fd.astype(np.float64), axis=1), 1.0, atol=1e-5)
    comb = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (2, 3), (2, 4), (2, 5), (2, 6), (2, 7),
            (2, 8), (3, 4), (3, 5), (3, 6), (3, 7), (3, 8), (4, 5), (4, 6), (4, 7), (4, 8), (5, 6), (5, 7), (5, 8), (6, 7), (6, 8), (7, 8)]
    for i in range(len(kl)):
        for c, (a, b) in enumerate(comb):
            bits = sum(1 << k for k in range(8) if fd[i, 8 * a + k] > fd[i, 8 * b + k])
            assert desc[i, c] == bits
    # the same scene shifted by whole pixels gives (nearly) the same descriptors for the same lines
    sh = np.roll(np.roll(img, 7, axis=1), 5, axis=0)
    kl2 = kl.copy()
    for f in ("startPointX", "endPointX", "sPointInOctaveX", "ePointInOctaveX"):
        kl2[f] += 7
    for f in ("startPointY", "endPointY", "sPointInOctaveY", "ePointInOctaveY"):
        kl2[f] += 5
    d2 = oracle.lbd_compute(sh, kl2)
    ham = np.unpackbits(desc ^ d2, axis=1).sum(1)
    assert ham.max() <= 8


def test_golden_lines_cabinet(oracle):
    """Regression pin: KeyLines + LBD descriptors of line_lbd/data/cabinet.png (decoded fixture), see tests/golden/make_golden.py."""
    g = np.load(os.path.join(GOLD, "orb_cabinet.npz"))["gray"]
    ref = np.load(os.path.join(GOLD, "lines_cabinet.npz"))
    kl = oracle.lsd_detect(g)
    desc = oracle.lbd_compute(g, kl)
    assert len(kl) == int(ref["n_lines"])
    assert hashlib.sha256(kl.tobytes()).hexdigest() == str(ref["keylines_sha256"])
    assert np.array_equal(desc, ref["desc"])
    assert np.array_equal(oracle.lsd_detect_filter_lines(g, 15.0), ref["filter15"])


def test_maps_against_scipy_restatements(oracle):
    """Independent restatements of the image operators under the line path: the LBD Sobel maps (cv::Sobel ksize 3 on the 5x5 sigma-1 blurred
    image, BORDER_REFLECT_101, binary_descriptor.cpp:620-640) equal scipy's correlation bit for bit and the blur stays within ~2 grey levels
    of a float Gaussian; LSD's gradient (lsd.cpp ll_angle: 2x2 differences, norm / 2) follows its definition to round-off on the scaled image."""
    from scipy.ndimage import correlate, gaussian_filter
    from cube_slam_amd import synth
    g = synth.texture_image(9, 200, 160)
    b, dx, dy = oracle.lbd_maps(g)
    assert np.abs(b.astype(float) - gaussian_filter(g.astype(np.float64), sigma=1.0, truncate=2.0, mode="mirror")).max() < 2.6
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    assert np.array_equal(correlate(b.astype(np.int32), kx, mode="mirror"), dx.astype(np.int32))
    assert np.array_equal(correlate(b.astype(np.int32), kx.T, mode="mirror"), dy.astype(np.int32))
    sc, mg, an, order = oracle.lsd_maps(g)
    assert sc.shape == (128, 160), "0.8 x the image (lsd.cpp scale)"
    A, B, Cc, D = sc[:-1, :-1], sc[:-1, 1:], sc[1:, :-1], sc[1:, 1:]
    gx = (B + D) - (A + Cc); gy = (Cc + D) - (A + B)
    assert np.allclose(mg[:-1, :-1], np.sqrt((gx * gx + gy * gy) / 4.0), rtol=0, atol=1e-11)
