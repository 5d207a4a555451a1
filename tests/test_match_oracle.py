"""CPU known-answer tests of the matcher oracle (oracle/match_oracle.cpp): Fuse search and SearchForTriangulation rules on hand-built frames."""
import numpy as np

from cube_slam_amd.orb import KEYPOINT_DTYPE

BOUNDS = (0.0, 640.0, 0.0, 480.0)
SF = np.float32(1.2) ** np.arange(8, dtype=np.float32)


def _keys(xy, octave=0, angle=0.0):
    k = np.zeros(len(xy), KEYPOINT_DTYPE)
    k["x"] = [p[0] for p in xy]; k["y"] = [p[1] for p in xy]
    k["octave"] = octave; k["angle"] = angle; k["size"] = 31
    return k


def _desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def _flip(d, nbits):
    d = d.copy()
    for b in range(nbits):
        d[b // 8] ^= np.uint8(1 << (b % 8))
    return d


def test_fuse_rules(oracle):
    rng = np.random.default_rng(0)
    keys = _keys([(100, 100), (101, 100), (100.5, 100), (300, 300)], octave=np.array([0, 0, 2, 0]))
    base = _desc(rng, 1)[0]
    desc = np.stack([_flip(base, 5), _flip(base, 5), base, _desc(rng, 1)[0]])  # kp 2 matches exactly but sits two levels up
    F = oracle.make_frame(keys, desc, BOUNDS)
    inv_s2 = (1.0 / (SF * SF)).astype(np.float32)
    mono = np.full(4, -1.0, np.float32)
    uv = np.array([[100.2, 100.0]], np.float32)
    bi, bd, n = oracle.fuse(F, mono, inv_s2, uv, [50.0], [0], [1], base[None], SF, 3.0)
    assert (bi[0], bd[0], n) == (0, 5, 1), "level filter drops kp 2; kps 0 and 1 tie at 5 bits, the first in GetFeaturesInArea order wins"
    bi, bd, n = oracle.fuse(F, mono, inv_s2, uv, [50.0], [2], [1], base[None], SF, 3.0)
    assert (bi[0], bd[0], n) == (2, 0, 1), "predicted level 2 accepts levels 1..2"
    far = np.array([[102.6, 100.0]], np.float32)  # ex^2 = 6.76 / 2.56 > 5.99 for kp 0/1 at level 0; radius th * 1.0 = 3 still covers them
    bi, bd, n = oracle.fuse(F, mono, inv_s2, far, [50.0], [0], [1], base[None], SF, 3.0)
    assert (bi[0], n) == (1, 1), "kp 0 fails the chi-square test (2.6^2 > 5.99), kp 1 (1.6^2) passes"
    stereo = np.array([40.0, -1, -1, -1], np.float32)
    bi, bd, n = oracle.fuse(F, stereo, inv_s2, uv, [49.0], [0], [1], base[None], SF, 3.0)
    bi2, _, _ = oracle.fuse(F, stereo, inv_s2, uv, [42.0], [0], [1], base[None], SF, 3.0)
    assert bi[0] == 1 and bi2[0] == 0, "er = 9 rejects the stereo keypoint (81 > 7.8), er = 2 keeps it (4.04 < 7.8)"
    bi, bd, n = oracle.fuse(F, mono, inv_s2, uv, [50.0], [0], [0], base[None], SF, 3.0)
    assert (bi[0], bd[0], n) == (-1, 256, 0), "invalid map points are skipped"
    bi, bd, n = oracle.fuse(F, mono, inv_s2, uv, [50.0], [0], [1], _flip(base, 70)[None], SF, 3.0)
    assert bd[0] == 65 and n == 0, "a best distance above TH_LOW = 50 is reported but not counted as fused"


def test_triangulation_rules(oracle):
    rng = np.random.default_rng(1)
    k1 = _keys([(100, 100)])
    base = _desc(rng, 1)[0]
    # KF2: same-node candidates on the epipolar line y = 100 (pure x translation), one off the line, one in another node
    k2 = _keys([(110, 100), (120, 100), (130, 100), (140, 130), (150, 100)])
    d2 = np.stack([_flip(base, 10), _flip(base, 4), _flip(base, 4), base, base])
    node1 = [7]; node2 = [7, 7, 7, 7, 8]
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # l = x1' F12 = (0, -1, y1): distance to the line y = y1
    sig2 = (SF * SF).astype(np.float32)
    F1 = oracle.make_frame(k1, base[None], BOUNDS); F2 = oracle.make_frame(k2, d2, BOUNDS)
    no = np.zeros(5, np.uint8); mono1 = [-1.0]; mono2 = np.full(5, -1.0, np.float32)
    m, n = oracle.search_for_triangulation(F1, node1, [0], mono1, F2, node2, no, mono2, F12, -1e4, 100.0, SF, sig2, False, False)
    assert (m[0], n) == (2, 1), "kp 3 (exact) is 30 px off the epipolar line, kp 4 is in another node; 1 and 2 tie at 4 bits and the LAST one stays"
    skip2 = np.array([0, 0, 1, 0, 0], np.uint8)
    m, _ = oracle.search_for_triangulation(F1, node1, [0], mono1, F2, node2, skip2, mono2, F12, -1e4, 100.0, SF, sig2, False, False)
    assert m[0] == 1, "features that already have a map point are skipped"
    m, _ = oracle.search_for_triangulation(F1, node1, [0], mono1, F2, node2, no, mono2, F12, 121.0, 100.0, SF, sig2, False, False)
    assert m[0] == 0, "mono-mono pairs closer than 10 px (sqrt(100 * scale)) to the epipole are rejected: kps 1 and 2"
    st2 = np.array([-1, 100.0, -1, -1, -1], np.float32)
    m, _ = oracle.search_for_triangulation(F1, node1, [0], mono1, F2, node2, no, st2, F12, 121.0, 100.0, SF, sig2, False, False)
    assert m[0] == 1, "a stereo keypoint is exempt from the epipole test"
    m, n = oracle.search_for_triangulation(F1, node1, [0], mono1, F2, node2, no, st2, F12, -1e4, 100.0, SF, sig2, True, False)
    assert (m[0], n) == (-1, 0), "bOnlyStereo needs both keypoints stereo"
    m, n = oracle.search_for_triangulation(F1, [-1], [0], mono1, F2, node2, no, mono2, F12, -1e4, 100.0, SF, sig2, False, False)
    assert n == 0
    k2hi = k2.copy(); k2hi["octave"][3] = 7  # sigma2 = 1.2^14 = 12.8: 30^2 = 900 > 3.84 * 12.8 still rejects; 6 px off passes at level 7
    k2hi["y"][3] = 106
    F2h = oracle.make_frame(k2hi, d2, BOUNDS)
    m, _ = oracle.search_for_triangulation(F1, node1, [0], mono1, F2h, node2, no, mono2, F12, -1e4, 100.0, SF, sig2, False, False)
    assert m[0] == 3, "36 < 3.84 * 12.84: coarse-level keypoints tolerate a larger epipolar distance, and the exact descriptor wins"


def test_undistort_points_properties(oracle):
    """cv::undistortPoints (classic five-iteration form): identity without distortion, inverse of the Brown model, image bounds."""
    K4 = (517.3, 516.5, 318.6, 255.3)  # TUM1.yaml-like
    D = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], axis=1).astype(np.float32)
    assert np.array_equal(oracle.undistort_points(pts, K4, None), pts)
    assert np.array_equal(oracle.undistort_points(pts, K4, (0.0, 0.5, 0.1, 0.1, 0.0)), pts), "only dist[0] decides (Frame.cc:548)"
    und = oracle.undistort_points(pts, K4, D).astype(np.float64)
    fx, fy, cx, cy = K4
    x, y = (und[:, 0] - cx) / fx, (und[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
    xd = x * rad + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * rad + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    back = np.stack([xd * fx + cx, yd * fy + cy], axis=1)
    near = np.hypot(pts[:, 0] - cx, pts[:, 1] - cy) < 250
    assert np.abs(back - pts)[near].max() < 0.05, "distorting the undistorted points returns the input (five iterations: a few 1e-2 px)"
    assert np.abs(und - pts).max() > 3, "and the correction is not a no-op"
    b = oracle.image_bounds(640, 480, K4, D)
    c = oracle.undistort_points([[0, 0], [640, 0], [0, 480], [640, 480]], K4, D)
    assert b[0] == min(c[0, 0], c[2, 0]) and b[3] == max(c[2, 1], c[3, 1])
    assert np.array_equal(oracle.image_bounds(640, 480, K4, None), np.array([0, 640, 0, 480], np.float32))
    from cube_slam_amd.matcher import frame_image_bounds
    assert np.array_equal(frame_image_bounds(640, 480, K4, D), b) and np.array_equal(frame_image_bounds(640, 480, K4, None), oracle.image_bounds(640, 480, K4, None))


def test_search_by_bow_claim_order(oracle):
    rng = np.random.default_rng(3)
    base = _desc(rng, 1)[0]
    kf = _keys([(10, 10), (20, 20), (30, 30)]); dkf = np.stack([base, base, _flip(base, 3)])
    fr = _keys([(11, 10), (21, 20), (31, 30), (41, 40)]); dfr = np.stack([_flip(base, 2), _flip(base, 8), _flip(base, 30), _desc(rng, 1)[0]])
    KF = oracle.make_frame(kf, dkf, BOUNDS); F = oracle.make_frame(fr, dfr, BOUNDS)
    m, n = oracle.search_by_bow(KF, [5, 5, 5], [0, 0, 0], F, [5, 5, 5, 6], None, 0.9, False)
    # kf0: best f0 (2) vs second f1 (8): 2 < 0.9*8 -> claims f0.  kf1 (same descriptor): f0 taken, best f1 (8) vs f2 (30) -> claims f1.
    # kf2 (3 bits off base): f0, f1 taken, only f2 at distance ~27..33 with second best 256 -> passes the ratio test if <= TH_LOW
    assert list(m[:2]) == [0, 1] and m[3] == -1 and n == int((m >= 0).sum())
    m2, n2 = oracle.search_by_bow(KF, [5, 5, 5], [0, 1, 0], F, [5, 5, 5, 6], None, 0.9, False)
    assert m2[0] == 0 and m2[1] != 1, "a key-frame feature without a usable map point does not claim anything"
    m3, n3 = oracle.search_by_bow(KF, [5, 5, 5], [0, 0, 0], F, [5, 5, 5, 6], None, 0.2, False)
    assert m3[0] == -1, "2 < 0.2 * 8 fails: the ratio test uses the second best of the UNCLAIMED candidates"


def test_search_by_bow_key_frames(oracle):
    """ORBmatcher::SearchByBoW(KeyFrame, KeyFrame) (:544-677): claims on KF2, the strict TH_LOW test, the orientation filter over KF1 indices."""
    rng = np.random.default_rng(4)
    base = _desc(rng, 1)[0]
    k1 = _keys([(10, 10), (20, 20), (30, 30), (40, 40)]); d1 = np.stack([base, base, _flip(base, 3), _desc(rng, 1)[0]])
    k2 = _keys([(11, 10), (21, 20), (31, 30), (41, 40)]); d2 = np.stack([_flip(base, 2), _flip(base, 8), _flip(base, 30), _desc(rng, 1)[0]])
    K1 = oracle.make_frame(k1, d1, BOUNDS); K2 = oracle.make_frame(k2, d2, BOUNDS)
    z = [0, 0, 0, 0]
    m, n = oracle.search_by_bow_kf(K1, [5, 5, 5, 7], z, K2, [5, 5, 5, 6], z, 0.9, False)
    assert list(m[:2]) == [0, 1] and m[3] == -1 and n == int((m >= 0).sum()), "first claims the best, the twin moves on; nodes 7 / 6 never meet"
    m2, _ = oracle.search_by_bow_kf(K1, [5, 5, 5, 7], z, K2, [5, 5, 5, 6], [1, 0, 0, 0], 0.9, False)
    assert m2[0] == 1, "a KF2 feature without a map point is no candidate"
    # exactly TH_LOW = 50 bits apart: accepted by the frame overload (<=, :268), rejected by this one (<, :625)
    far = _flip(base, 50)
    Ka = oracle.make_frame(_keys([(10, 10)]), np.stack([base]), BOUNDS); Kb = oracle.make_frame(_keys([(10, 10)]), np.stack([far]), BOUNDS)
    assert oracle.search_by_bow_kf(Ka, [1], [0], Kb, [1], [0], 0.9, False)[1] == 0
    assert oracle.search_by_bow(Ka, [1], [0], Kb, [1], None, 0.9, False)[1] == 1
