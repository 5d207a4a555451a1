"""CPU tests: ORB extractor / matcher oracle against independent restatements and golden fixtures."""
import hashlib
import os

import numpy as np

from cube_slam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _fast_numpy(img, t):
    """FAST-9/16 with score = (largest threshold keeping the corner) and strict 3x3 NMS, straight from the definition."""
    h, w = img.shape
    im = img.astype(np.int32)
    S = np.zeros((h, w), np.int32)
    ring = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # 16 x (h-6) x (w-6)
    d = im[3:h - 3, 3:w - 3][None] - ring
    best = np.full(d.shape[1:], -999)
    for k in range(16):
        arc = np.stack([d[(k + j) % 16] for j in range(9)])
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    S[3:h - 3, 3:w - 3] = np.maximum(best, 0)
    score = np.where(S > t, S - 1, 0)
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if S[y, x] > t:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if (s > nb).all():
                    out.append((x, y, s))
    return np.array(out, np.float32).reshape(-1, 3)


def test_fast_matches_definition(oracle):
    rng = np.random.default_rng(0)
    img = synth.texture_image(5, 96, 64)
    img[20:40, 30:60] = 220  # a bright rectangle: strong corners
    for t in (7, 20, 40):
        got = oracle.fast(img, t)
        ref = _fast_numpy(img, t)
        assert np.array_equal(got, ref), t
    assert len(oracle.fast(np.full((30, 30), 9, np.uint8), 5)) == 0
    assert len(oracle.fast(rng.integers(0, 255, (6, 40)).astype(np.uint8), 5)) == 0  # smaller than the 7x7 support


def test_fast_atan2_and_sincos(oracle):
    assert abs(oracle.fast_atan2(1, 1) - 45) < 0.02 and oracle.fast_atan2(0, 1) == 0.0
    assert abs(oracle.fast_atan2(1, -1) - 135) < 0.02 and abs(oracle.fast_atan2(-1, -1) - 225) < 0.02 and abs(oracle.fast_atan2(-1, 1) - 315) < 0.02
    for a in np.linspace(0, 2 * np.pi, 721).astype(np.float32):
        s, c = oracle.sincos_f(a)
        assert s == np.float32(np.sin(np.float64(a))) and c == np.float32(np.cos(np.float64(a)))


def test_pyramid_and_blur_properties(oracle):
    e = oracle.ORBextractor(500, 1.2, 8, 20, 7)
    img = np.full((120, 160), 100, np.uint8)
    e(img)
    for l in range(8):
        lv = e.level(l)
        assert lv.shape == (int(np.rint(np.float32(120) * np.float32(1) / np.float32(1.2) ** l)), int(np.rint(np.float32(160) / np.float32(1.2) ** l))) or l > 0
        assert (lv == 100).all()  # bilinear fixed point keeps constants
    assert list(e.features_per_level()) == [109, 90, 75, 63, 52, 44, 36, 31]
    tex = synth.texture_image(2, 320, 240)
    k, d = e(tex)
    assert 400 <= len(k) <= 520
    bl = e.level(0, blurred=True)
    # 8-bit separable kernel {18,34,49,55,49,34,18}/256 applied twice sums to (257/256)^2: a flat 100 becomes 101
    flat = oracle.ORBextractor(50, 1.2, 2, 20, 7)
    flat(np.pad(np.full((100, 100), 100, np.uint8), 0))
    assert bl.shape == tex.shape and abs(int(bl.mean()) - int(tex.mean())) <= 2
    assert (k["octave"][:-1] <= k["octave"][1:]).all()  # level-major output order
    assert ((k["x"] / 1.2 ** k["octave"] >= 15.9) & (k["y"] / 1.2 ** k["octave"] >= 15.9)).all()  # 16 px border per level


def test_matcher_primitives(oracle):
    rng = np.random.default_rng(4)
    a, b = rng.integers(0, 256, (50, 32), dtype=np.uint8), rng.integers(0, 256, (70, 32), dtype=np.uint8)
    D = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert oracle.descriptor_distance(a[3], b[5]) == D[3, 5]
    bi, bd, sd = oracle.hamming_knn2(a, b)
    assert np.array_equal(bi, D.argmin(1)) and np.array_equal(bd, D.min(1))
    assert np.array_equal(sd, np.sort(D, axis=1)[:, 1])
    # GetFeaturesInArea: same set as a brute-force filter, ordered by (cell x, cell y, index)
    n = 500
    keys = np.zeros(n, oracle.KEYPOINT_DTYPE)
    keys["x"], keys["y"], keys["octave"] = rng.uniform(0, 640, n), rng.uniform(0, 480, n), rng.integers(0, 8, n)
    F = oracle.make_frame(keys, rng.integers(0, 256, (n, 32), dtype=np.uint8), (0, 640, 0, 480))
    got = oracle.get_features_in_area(F, 300, 200, 80, 2, 4)
    sel = [i for i in range(n) if abs(keys["x"][i] - np.float32(300)) < 80 and abs(keys["y"][i] - np.float32(200)) < 80 and 2 <= keys["octave"][i] <= 4]
    assert sorted(got.tolist()) == sel
    cx = np.round((keys["x"] - 0) * np.float32(64 / 640)).astype(int); cy = np.round((keys["y"] - 0) * np.float32(48 / 480)).astype(int)
    order = sorted(got.tolist(), key=lambda i: (cx[i], cy[i], i))
    assert got.tolist() == order


def test_golden_orb_cabinet(oracle):
    """Regression pin on the reference's own sample image (line_lbd/data/cabinet.png, decoded once, tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "orb_cabinet.npz"))
    k, d = oracle.ORBextractor(1000, 1.2, 8, 20, 7)(g["gray"])
    assert len(k) == int(g["n"])
    assert hashlib.sha256(k.tobytes()).hexdigest() == str(g["kp_sha"]) and hashlib.sha256(d.tobytes()).hexdigest() == str(g["desc_sha"])
    assert np.array_equal(k[:16], g["kp_head"]) and np.array_equal(d[:16], g["desc_head"])


def test_bit_pattern_is_the_references_table():
    """The 256 x 4 BRIEF test-point table is data the reference embeds (ORBextractor.cc:152-410); both the oracle and the HIP kernel include a
    generated copy (tools/gen_orb_pattern.py).  Where the reference tree is mounted (the build container), check both copies number by
    number against the source; everywhere, that the two copies are the same file."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = open(os.path.join(root, "oracle", "orb_pattern.inc")).read()
    b = open(os.path.join(root, "cube_slam_amd", "csrc", "orb_pattern.inc")).read()
    assert a == b
    nums = [int(x) for x in re.findall(r"-?\d+", re.sub(r"//.*", "", a))]
    assert len(nums) == 1024 and max(abs(v) for v in nums) <= 15, "points inside the 31 x 31 patch"
    ref = "/root/reference/orb_object_slam/src/ORBextractor.cc"
    if os.path.exists(ref):
        src = open(ref).read()
        i = src.index("static int bit_pattern_31_")
        body = src[src.index("{", i) + 1: src.index("};", i)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        body = re.sub(r"//.*", "", body)
        assert [int(x) for x in re.findall(r"-?\d+", body)] == nums


def test_pyramid_and_blur_against_float_restatements(oracle):
    """Independent float restatements of the two image operators ORBextractor::ComputePyramid / operator() lean on (ORBextractor.cc:1101-1130,
    :1069): cv::resize INTER_LINEAR (pixel centres: src = (dst + 0.5) * scale - 0.5, border replicated) from one level to the next, and
    GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 (scipy's mirror mode).  The oracle's fixed-point arithmetic must stay within a grey level or
    two of them everywhere."""
    from scipy.ndimage import gaussian_filter
    e = oracle.ORBextractor(500, 1.2, 4, 20, 7)
    tex = synth.texture_image(5, 320, 240)
    e(tex)

    def resize_float(src, h, w):
        sy, sx = src.shape[0] / h, src.shape[1] / w
        fy = np.clip((np.arange(h) + 0.5) * sy - 0.5, 0, src.shape[0] - 1); fx = np.clip((np.arange(w) + 0.5) * sx - 0.5, 0, src.shape[1] - 1)
        y0 = np.floor(fy).astype(int); x0 = np.floor(fx).astype(int)
        y1 = np.minimum(y0 + 1, src.shape[0] - 1); x1 = np.minimum(x0 + 1, src.shape[1] - 1)
        wy = (fy - y0)[:, None]; wx = (fx - x0)[None, :]
        s = src.astype(np.float64)
        return (s[y0][:, x0] * (1 - wx) + s[y0][:, x1] * wx) * (1 - wy) + (s[y1][:, x0] * (1 - wx) + s[y1][:, x1] * wx) * wy
    for l in range(1, 4):
        prev, cur = e.level(l - 1), e.level(l)
        ref = resize_float(prev, *cur.shape)
        assert np.abs(cur.astype(np.float64) - ref).max() <= 1.0 + 1e-9, l
    bl = e.level(0, blurred=True).astype(np.float64)
    ref = gaussian_filter(tex.astype(np.float64), sigma=2.0, truncate=1.5, mode="mirror")  # radius 3: the 7-tap kernel
    assert np.abs(bl - ref).max() <= 2.5 and abs((bl - ref).mean()) < 1.2


def test_orientation_and_descriptor_against_numpy_restatement(oracle):
    """IC_Angle (ORBextractor.cc:74-104: intensity centroid over the circular patch, rows bounded by umax) and computeOrbDescriptor
    (:107-149: the 256 test pairs rotated by the keypoint angle in float, cvRound, compared on the blurred level) restated in numpy for the
    level-0 keypoints of a textured image: angles agree to fastAtan2's 0.3 degrees, descriptors bit for bit."""
    import re
    e = oracle.ORBextractor(500, 1.2, 4, 20, 7)
    tex = synth.texture_image(7, 320, 240)
    k, d = e(tex)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = np.array([int(x) for x in re.findall(r"-?\d+", re.sub(r"//.*", "", open(os.path.join(root, "oracle", "orb_pattern.inc")).read()))], np.float32).reshape(512, 2)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]  # the table the constructor builds for HALF_PATCH_SIZE 15 (:433-452)
    img = e.level(0).astype(np.int64); bl = e.level(0, blurred=True)
    sel = np.nonzero(k["octave"] == 0)[0]
    assert len(sel) > 100
    for i in sel:
        x, y = int(round(float(k["x"][i]))), int(round(float(k["y"][i])))
        m10 = m01 = 0
        for v in range(-15, 16):
            u = umax[abs(v)]
            row = img[y + v, x - u:x + u + 1]
            m10 += int((np.arange(-u, u + 1) * row).sum()); m01 += int(v * row.sum())
        ang = np.degrees(np.arctan2(float(m01), float(m10))) % 360.0
        assert abs((ang - float(k["angle"][i]) + 180) % 360 - 180) < 0.35
        a32 = np.float32(k["angle"][i]) * np.float32(np.pi / 180.0)
        a = np.float32(np.cos(np.float64(a32))); b = np.float32(np.sin(np.float64(a32)))
        rx = np.rint((pat[:, 0] * a).astype(np.float32) - (pat[:, 1] * b).astype(np.float32)).astype(int)  # cvRound = round half to even
        ry = np.rint((pat[:, 0] * b).astype(np.float32) + (pat[:, 1] * a).astype(np.float32)).astype(int)
        vals = bl[y + ry, x + rx].astype(int)
        bits = (vals[0::2] < vals[1::2]).astype(np.uint8)
        assert np.array_equal(np.packbits(bits.reshape(32, 8)[:, ::-1], axis=1).reshape(32), d[i])
