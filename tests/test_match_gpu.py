"""GPU parity: ORBmatcher searches vs the CPU oracle -- Hamming distances, candidate order and match indices bit-exact."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.matcher import ORBmatcher, hamming_knn2

pytestmark = pytest.mark.gpu

W, H = 1241, 376
FX, FY, CX, CY = 721.5377, 721.5377, 609.5593, 172.854
BOUNDS = (0.0, float(W), 0.0, float(H))
SF = np.float32(1.2) ** np.arange(8, dtype=np.float32)


@pytest.fixture(scope="module")
def frames(oracle):
    e = oracle.ORBextractor(2000, 1.2, 8, 20, 7)
    out = []
    for i in range(2):
        k, d = e(synth.texture_image(77, W, H, shift=4 * i))
        out.append((k, d))
    return out


def test_knn2_and_distance(ctx, oracle, frames):
    (k1, d1), (k2, d2) = frames
    bi, bd, sd = hamming_knn2(ctx, d1, d2)
    ri, rd, rs = oracle.hamming_knn2(d1, d2)
    assert np.array_equal(bi, ri) and np.array_equal(bd, rd) and np.array_equal(sd, rs)
    assert bd[0] == oracle.descriptor_distance(d1[0], d2[bi[0]])
    assert (bd <= 60).mean() > 0.5  # the two frames are the same texture shifted by 4 px


def test_features_in_area(ctx, oracle, frames):
    k2, d2 = frames[1]
    m = ORBmatcher(ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    rng = np.random.default_rng(1)
    for _ in range(40):
        x, y, r = rng.uniform(-50, W + 50), rng.uniform(-50, H + 50), rng.uniform(1, 120)
        lv = int(rng.integers(-1, 8))
        a = m.GetFeaturesInArea(x, y, r, lv - 1, lv + 1)
        b = oracle.get_features_in_area(F2, x, y, r, lv - 1, lv + 1)
        assert np.array_equal(a, b)
    assert len(m.GetFeaturesInArea(600, 180, 5000)) == len(k2)
    m.close()


@pytest.mark.parametrize("th", [15.0, 30.0, 90.0])  # (90: dozens of candidates per query -- lists longer than the resolve kernel keeps in registers, most batches with competing claims)
def test_search_by_projection_frame(ctx, oracle, frames, th):
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(2)
    n = len(k1)
    z = rng.uniform(4, 40, n).astype(np.float32)
    wp = np.stack([(k1["x"] - CX) / FX * z, (k1["y"] - CY) / FY * z, z], axis=1).astype(np.float32)  # last frame at identity
    Tcw = np.eye(4, dtype=np.float32)[:3]
    Tcw[0, 3] = 0.02  # tiny motion; the image shift does the rest
    wp[:, 0] += (-4.0 / FX) * z  # the texture moved 4 px to the left
    valid = (rng.uniform(size=n) < 0.85).astype(np.uint8)
    blocks = (rng.uniform(size=n) < (0.5 if th > 50 else 0.9)).astype(np.uint8)  # (a claim by a map point without observations is overwritten by a later query)
    m = ORBmatcher(0.9, True, ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    got, ng = m.SearchByProjectionFrame(wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th)
    ref, nr = oracle.search_by_projection_frame(F2, wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th)
    assert np.array_equal(got, ref) and ng == nr
    assert ng > 300
    # current-frame keypoints that must not be matched: a map point from before the call, or KeysStatic == false (ORBmatcher.cc:1451-1457)
    blocked = (rng.uniform(size=len(k2)) < 0.3).astype(np.uint8)
    got_b, ng_b = m.SearchByProjectionFrame(wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th, train_blocked=blocked)
    ref_b, nr_b = oracle.search_by_projection_frame(F2, wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th, train_blocked=blocked)
    assert np.array_equal(got_b, ref_b) and ng_b == nr_b
    assert not np.any(got_b[blocked != 0] >= 0) and 100 < ng_b < ng and not np.array_equal(got_b, got)
    m.close()


def test_search_local_map(ctx, oracle, frames):
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(3)
    n = len(k1)
    proj = np.stack([k1["x"] - 4 + rng.normal(0, 1, n), k1["y"] + rng.normal(0, 1, n)], axis=1).astype(np.float32)
    view_cos = rng.uniform(0.99, 1.0, n).astype(np.float32)
    in_view = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    blocks = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    tblocked = (rng.uniform(size=len(k2)) < 0.2).astype(np.uint8)
    m = ORBmatcher(0.8, True, ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    for th in (1.0, 3.0):
        got, ng = m.SearchByProjectionLocalMap(proj, view_cos, k1["octave"], in_view, blocks, d1, SF, th, tblocked)
        ref, nr = oracle.search_local_map(F2, proj, view_cos, k1["octave"], in_view, blocks, d1, SF, th, 0.8, tblocked)
        assert np.array_equal(got, ref) and ng == nr and ng > 100
    m.close()


def test_search_for_initialization(ctx, oracle, frames):
    (k1, d1), (k2, d2) = frames
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    m = ORBmatcher(0.9, True, ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F1 = oracle.make_frame(k1, d1, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    g12, gprev, ng = m.SearchForInitialization(k1, d1, prev, 100)
    r12, rprev, nr = oracle.search_for_initialization(F1, F2, prev, 100, 0.9, True)
    assert np.array_equal(g12, r12) and np.array_equal(gprev, rprev) and ng == nr and ng > 50
    # empty inputs
    e12, _, en = m.SearchForInitialization(k1[:0], d1[:0], prev[:0], 100)
    assert len(e12) == 0 and en == 0
    m.close()


def test_fuse_search(ctx, oracle, frames):
    """ORBmatcher::Fuse search part: per map point the best keypoint after the level / chi-square tests, mono and stereo keypoints mixed."""
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(4)
    n = len(k1)
    uv = np.stack([k1["x"] - 4 + rng.normal(0, 0.8, n), k1["y"] + rng.normal(0, 0.8, n)], axis=1).astype(np.float32)
    z = rng.uniform(4, 40, n).astype(np.float32)
    bf = np.float32(386.1448)
    ur = (uv[:, 0] - bf / z).astype(np.float32)
    pred = np.clip(k1["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32)  # the matching keypoint is at level pred or pred - 1
    valid = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    u_right2 = np.where(rng.uniform(size=len(k2)) < 0.5, k2["x"] - bf / rng.uniform(4, 40, len(k2)), -1.0).astype(np.float32)
    inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
    static = (rng.uniform(size=len(k2)) < 0.95).astype(np.uint8)
    m = ORBmatcher(ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    for th, ks in ((3.0, None), (5.0, static)):
        gi, gd, ng = m.Fuse(u_right2, inv_sigma2, uv, ur, pred, valid, d1, SF, th, ks)
        ri, rd, nr = oracle.fuse(F2, u_right2, inv_sigma2, uv, ur, pred, valid, d1, SF, th, ks)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd) and ng == nr
        assert ng > 150 and (gi[valid == 0] == -1).all()
    m.close()


def test_search_for_triangulation(ctx, oracle, frames):
    """ORBmatcher::SearchForTriangulation: same-node candidates, epipole / epipolar-line tests, the reference's last-minimum tie rule,
    rotation histogram.  The vocabulary node of a feature is emulated by a coarse image cell (a few hundred features per node)."""
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(6)
    node = lambda k, dx: ((np.clip(k["x"] + dx, 0, W - 1) // 120).astype(np.int32) * 4 + (k["y"] // 100).astype(np.int32))
    node1, node2 = node(k1, 0.0), node(k2, 4.0)
    node1[rng.uniform(size=len(k1)) < 0.05] = -1
    skip1 = (rng.uniform(size=len(k1)) < 0.3).astype(np.uint8); skip2 = (rng.uniform(size=len(k2)) < 0.3).astype(np.uint8)
    bf = 386.1448
    ur1 = np.where(rng.uniform(size=len(k1)) < 0.3, k1["x"] - bf / rng.uniform(4, 40, len(k1)), -1.0).astype(np.float32)
    ur2 = np.where(rng.uniform(size=len(k2)) < 0.3, k2["x"] - bf / rng.uniform(4, 40, len(k2)), -1.0).astype(np.float32)
    # pure sideways translation between the key frames: F12 = [t]_x in normalised coordinates -> horizontal epipolar lines, epipole at infinity
    Kinv = np.linalg.inv(np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1]]))
    tx = np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
    F12 = (Kinv.T @ tx @ Kinv).astype(np.float32)
    sigma2 = (SF * SF).astype(np.float32)
    F1o = oracle.make_frame(k1, d1, BOUNDS); F2o = oracle.make_frame(k2, d2, BOUNDS)
    for only_stereo, (ex, ey) in ((False, (-5000.0, 170.0)), (False, (600.0, 170.0)), (True, (-5000.0, 170.0))):
        m = ORBmatcher(0.6, True, ctx=ctx)
        g12, ng = m.SearchForTriangulation(k1, d1, node1, skip1, ur1, k2, d2, node2, skip2, ur2, F12, ex, ey, SF, sigma2, only_stereo)
        r12, nr = oracle.search_for_triangulation(F1o, node1, skip1, ur1, F2o, node2, skip2, ur2, F12, ex, ey, SF, sigma2, only_stereo, True)
        assert np.array_equal(g12, r12) and ng == nr
        assert ng > (20 if only_stereo else 150)
        assert (g12[skip1 == 1] == -1).all() and (g12[node1 < 0] == -1).all()
        m.close()
    # duplicate descriptors in KF2 inside one node: the reference keeps the last one that reaches the minimum
    k2b = np.concatenate([k2, k2[:50]]); d2b = np.concatenate([d2, d2[:50]])
    node2b = np.concatenate([node2, node2[:50]]); skip2b = np.concatenate([skip2, np.zeros(50, np.uint8)]); ur2b = np.concatenate([ur2, ur2[:50]])
    F2b = oracle.make_frame(k2b, d2b, BOUNDS)
    m = ORBmatcher(0.6, False, ctx=ctx)
    g12, ng = m.SearchForTriangulation(k1, d1, node1, skip1, ur1, k2b, d2b, node2b, skip2b, ur2b, F12, -5000.0, 170.0, SF, sigma2, False)
    r12, nr = oracle.search_for_triangulation(F1o, node1, skip1, ur1, F2b, node2b, skip2b, ur2b, F12, -5000.0, 170.0, SF, sigma2, False, False)
    assert np.array_equal(g12, r12) and ng == nr and (g12 >= len(k2)).sum() > 5
    m.close()


def test_frame_postprocessing_on_device(ctx, oracle):
    """ORB extractor -> matcher without a host round trip: UndistortKeyPoints + AssignFeaturesToGrid on the device.
    mvKeysUn bit-exact vs the oracle's cv::undistortPoints restatement, searches identical to the host-fed matcher."""
    from cube_slam_amd.orb import ORBextractor
    from cube_slam_amd.matcher import frame_image_bounds
    Wt, Ht = 640, 480
    imgs = np.stack([synth.texture_image(5 + i, Wt, Ht, shift=3 * i) for i in range(2)])
    orb = ORBextractor(1000, 1.2, 8, 20, 7, Wt, Ht, max_frames=2, ctx=ctx)
    res = orb.extract_batch(imgs)
    K4 = (517.3, 516.5, 318.6, 255.3)
    rng = np.random.default_rng(9)
    for dist in (None, (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)):
        for f in range(2):
            keys, desc = res[f]
            m = ORBmatcher(ctx=ctx)
            keysUn, bounds = m.set_frame_from_orb(orb, f, K4, dist, width=Wt, height=Ht)
            exp_xy = oracle.undistort_points(np.stack([keys["x"], keys["y"]], axis=1), K4, dist)
            assert len(keysUn) == len(keys) and np.array_equal(keysUn["x"], exp_xy[:, 0]) and np.array_equal(keysUn["y"], exp_xy[:, 1])
            for fld in ("angle", "octave", "response", "size"):
                assert np.array_equal(keysUn[fld], keys[fld])
            assert np.array_equal(np.array(bounds, np.float32), oracle.image_bounds(Wt, Ht, K4, dist))
            if dist is not None:
                assert np.abs(keysUn["x"] - keys["x"]).max() > 1.0
            ref = ORBmatcher(ctx=ctx)
            ref.set_frame(keysUn, desc, bounds)
            Fo = oracle.make_frame(keysUn, desc, bounds)
            for _ in range(15):
                x, y, r = rng.uniform(-20, Wt + 20), rng.uniform(-20, Ht + 20), rng.uniform(2, 90)
                a = m.GetFeaturesInArea(x, y, r); b = ref.GetFeaturesInArea(x, y, r)
                assert np.array_equal(a, b) and np.array_equal(a, oracle.get_features_in_area(Fo, x, y, r))
            bi, bd, _ = hamming_knn2(ctx, desc[:50], desc)
            assert (bd == 0).all()
            m.close(); ref.close()
    orb.close()


def test_search_by_projection_over_a_window_equals_the_per_frame_calls(ctx, oracle):
    """cs_match_by_projection_stream: frame post-processing + SearchByProjection(CurrentFrame, LastFrame) of every pair of a window of the stream the extractor holds, as a
    handful of launches -- train_match and nmatches of every pair equal the per-frame calls' (set_frame_from_orb + SearchByProjectionFrame) and the oracle's, with and
    without distortion, with dropped queries, `blocks` cleared for some and the map points' own descriptors given explicitly."""
    from cube_slam_amd.orb import ORBextractor
    from cube_slam_amd.matcher import ORBmatcherStream, frame_image_bounds
    Wt, Ht, NF = 640, 480, 6
    imgs = synth.texture_stream(31, Wt, Ht, NF, step=3)
    orb = ORBextractor(1000, 1.2, 8, 20, 7, Wt, Ht, max_frames=NF, ctx=ctx)
    res = orb.extract_batch(imgs)
    fx, fy, cx, cy = 517.3, 516.5, 318.6, 255.3
    K4 = (fx, fy, cx, cy)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    rng = np.random.default_rng(3)
    ms = ORBmatcherStream(True, ctx=ctx)
    for dist in (None, (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)):
        bounds = tuple(float(b) for b in frame_image_bounds(Wt, Ht, np.array(K4, np.float32), None if dist is None else np.array(dist, np.float32)))
        for f0, n_pairs, own_desc in ((0, NF - 1, True), (2, 2, False)):
            wps, vas, bls, Ts, mds = [], [], [], [], []
            for p in range(n_pairs):
                pk, pd = res[f0 + p]
                z = rng.uniform(5.0, 20.0, len(pk)).astype(np.float32)
                wps.append(np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32))
                vas.append((rng.uniform(size=len(pk)) < 0.9).astype(np.uint8)); bls.append((rng.uniform(size=len(pk)) < 0.8).astype(np.uint8))
                T = np.eye(4, dtype=np.float32)[:3].copy(); T[0, 3] = 0.01 * p
                Ts.append(T)
                mds.append(pd if own_desc else np.roll(pd, 1, axis=0))  # (not own_desc: some other descriptor per map point)
            n_train = sum(len(res[f0 + p + 1][0]) for p in range(n_pairs))
            tm, nm = ms.search(orb, f0, n_pairs, K4, dist, bounds, np.concatenate(wps), np.concatenate(vas), np.concatenate(bls), np.stack(Ts), fx, fy, cx, cy, sf, 15.0, n_train,
                               mp_desc=None if own_desc else np.concatenate(mds))
            off = 0
            m = ORBmatcher(0.9, True, ctx=ctx, max_queries=4096)
            for p in range(n_pairs):
                pk, pd = res[f0 + p]
                keysUn, _ = m.set_frame_from_orb(orb, f0 + p + 1, K4, dist, bounds)
                want, nw = m.SearchByProjectionFrame(wps[p], vas[p], bls[p], mds[p], pk["octave"], pk["angle"], Ts[p], fx, fy, cx, cy, sf, 15.0)
                got = tm[off:off + len(keysUn)]
                assert nm[p] == nw and np.array_equal(got, want), (dist is not None, f0, p, nm[p], nw)
                Fo = oracle.make_frame(keysUn, res[f0 + p + 1][1], bounds)
                otm, onm = oracle.search_by_projection_frame(Fo, wps[p], vas[p], bls[p], mds[p], pk["octave"], pk["angle"], Ts[p], fx, fy, cx, cy, sf, 15.0)
                assert onm == nw and np.array_equal(otm, want)
                assert nw > 100
                off += len(keysUn)
            assert off == n_train
            st = ms.last_counts()
            assert st["queries"] == sum(len(w) for w in wps) and st["candidates"] > st["queries"]
            m.close()
    ms.close()
    # a matcher whose FIRST window is small and whose second is the large one (arrays that shared one capacity), through an arena that starts too small
    # for the second window's candidates (the cursor reports the need and the search runs again); and counts that are not the extractor's are refused
    ms = ORBmatcherStream(True, ctx=ctx)
    bounds = (0.0, float(Wt), 0.0, float(Ht))
    for f0, n_pairs, th in ((4, 1, 15.0), (0, NF - 1, 60.0)):
        wps, ones, Ts = [], [], []
        for p in range(n_pairs):
            pk, _ = res[f0 + p]
            z = np.full(len(pk), 10.0, np.float32)
            wps.append(np.stack([(pk["x"] - 3.0 - cx) / fx * z, (pk["y"] - cy) / fy * z, z], axis=1).astype(np.float32)); ones.append(np.ones(len(pk), np.uint8))
            Ts.append(np.eye(4, dtype=np.float32)[:3])
        n_train = sum(len(res[f0 + p + 1][0]) for p in range(n_pairs))
        tm, nm = ms.search(orb, f0, n_pairs, K4, None, bounds, np.concatenate(wps), np.concatenate(ones), np.concatenate(ones), np.stack(Ts), fx, fy, cx, cy, sf, th, n_train)
        off = 0
        for p in range(n_pairs):
            pk, pd = res[f0 + p]
            ck, cd = res[f0 + p + 1]
            Fo = oracle.make_frame(ck, cd, bounds)
            otm, onm = oracle.search_by_projection_frame(Fo, wps[p], ones[p], ones[p], pd, pk["octave"], pk["angle"], Ts[p], fx, fy, cx, cy, sf, th)
            assert onm == nm[p] and np.array_equal(otm, tm[off:off + len(ck)]) and onm > 100
            off += len(ck)
        with pytest.raises(Exception):
            ms.search(orb, f0, n_pairs, K4, None, bounds, np.concatenate(wps), np.concatenate(ones), np.concatenate(ones), np.stack(Ts), fx, fy, cx, cy, sf, th, n_train + 1)
    ms.close(); orb.close()


def test_search_by_bow(ctx, oracle, frames):
    """ORBmatcher::SearchByBoW(KeyFrame, Frame): same-node candidates, greedy claims in (node, index) order, ratio test, rotation histogram."""
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(8)
    node = lambda k, dx: ((np.clip(k["x"] + dx, 0, W - 1) // 60).astype(np.int32) * 8 + (k["y"] // 50).astype(np.int32))
    node1, node2 = node(k1, 0.0), node(k2, 4.0)
    node2[rng.uniform(size=len(k2)) < 0.03] = -1
    skip1 = (rng.uniform(size=len(k1)) < 0.25).astype(np.uint8)
    skip2 = (rng.uniform(size=len(k2)) < 0.05).astype(np.uint8)
    KFo = oracle.make_frame(k1, d1, BOUNDS); Fo = oracle.make_frame(k2, d2, BOUNDS)
    for ratio, ori, sf in ((0.75, True, None), (0.9, False, skip2), (0.6, True, skip2)):
        m = ORBmatcher(ratio, ori, ctx=ctx)
        g, ng = m.SearchByBoW(k1, d1, node1, skip1, k2, d2, node2, sf)
        r, nr = oracle.search_by_bow(KFo, node1, skip1, Fo, node2, sf, ratio, ori)
        assert np.array_equal(g, r) and ng == nr and ng > 100
        assert (skip1[g[g >= 0]] == 0).all()
        m.close()
    # two key-frame features with the same descriptor in one node: the first claims the best frame feature, the second must move on
    k1b = np.concatenate([k1[:300], k1[:300]]); d1b = np.concatenate([d1[:300], d1[:300]]); n1b = np.concatenate([node1[:300], node1[:300]])
    KFb = oracle.make_frame(k1b, d1b, BOUNDS)
    m = ORBmatcher(0.95, False, ctx=ctx)
    g, ng = m.SearchByBoW(k1b, d1b, n1b, np.zeros(600, np.uint8), k2, d2, node2, None)
    r, nr = oracle.search_by_bow(KFb, n1b, np.zeros(600, np.uint8), Fo, node2, None, 0.95, False)
    assert np.array_equal(g, r) and ng == nr and (g >= 300).sum() > 0 and len(set(g[g >= 0])) == (g >= 0).sum()
    m.close()


def test_search_by_bow_key_frames(ctx, oracle, frames):
    """ORBmatcher::SearchByBoW(KeyFrame, KeyFrame) (:544-677): matches12 indexed by KF1, claims on KF2, strict TH_LOW."""
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(9)
    node = lambda k, dx: ((np.clip(k["x"] + dx, 0, W - 1) // 60).astype(np.int32) * 8 + (k["y"] // 50).astype(np.int32))
    node1, node2 = node(k1, 0.0), node(k2, 4.0)
    node1[rng.uniform(size=len(k1)) < 0.03] = -1
    skip1 = (rng.uniform(size=len(k1)) < 0.2).astype(np.uint8)
    skip2 = (rng.uniform(size=len(k2)) < 0.2).astype(np.uint8)
    K1 = oracle.make_frame(k1, d1, BOUNDS); K2 = oracle.make_frame(k2, d2, BOUNDS)
    for ratio, ori in ((0.75, True), (0.9, False), (0.6, True)):
        m = ORBmatcher(ratio, ori, ctx=ctx)
        g, ng = m.SearchByBoWKeyFrames(k1, d1, node1, skip1, k2, d2, node2, skip2)
        r, nr = oracle.search_by_bow_kf(K1, node1, skip1, K2, node2, skip2, ratio, ori)
        assert np.array_equal(g, r) and ng == nr == int((g >= 0).sum()) and ng > 80
        assert (skip1[g >= 0] == 0).all() and (skip2[g[g >= 0]] == 0).all() and len(set(g[g >= 0])) == ng
        m.close()
    m = ORBmatcher(0.9, True, ctx=ctx)
    g, ng = m.SearchByBoWKeyFrames(k1[:0], d1[:0], node1[:0], skip1[:0], k2, d2, node2, skip2)
    assert len(g) == 0 and ng == 0
    m.close()


def test_matcher_edge_cases(ctx, oracle, frames):
    """Empty and degenerate inputs of the device searches: no queries, no valid query, a frame without key points, a candidate arena that is too small (reported, not
    overrun), every query competing for the same few key points (one batch settles over many rounds)."""
    (k1, d1), (k2, d2) = frames
    n = len(k1)
    Tcw = np.eye(4, dtype=np.float32)[:3]
    z = np.full(n, 10.0, np.float32)
    wp = np.stack([(k1["x"] - 4.0 - CX) / FX * z, (k1["y"] - CY) / FY * z, z], axis=1).astype(np.float32)
    ones = np.ones(n, np.uint8)
    m = ORBmatcher(0.9, True, ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    tm, nm = m.SearchByProjectionFrame(wp[:0], ones[:0], ones[:0], d1[:0], k1["octave"][:0], k1["angle"][:0], Tcw, FX, FY, CX, CY, SF, 15.0)
    assert nm == 0 and (tm == -1).all() and len(tm) == len(k2)
    tm, nm = m.SearchByProjectionFrame(wp, np.zeros(n, np.uint8), ones, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, 15.0)
    assert nm == 0 and (tm == -1).all()
    # all queries at ONE place with the same descriptor: the first query with observations takes the best key point, the next takes the second best, ...
    wp1 = np.repeat(wp[:1], 200, axis=0); d_same = np.repeat(d1[:1], 200, axis=0); oc = np.repeat(k1["octave"][:1], 200); an = np.repeat(k1["angle"][:1], 200)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    for bl in (np.ones(200, np.uint8), (np.arange(200) % 3 != 0).astype(np.uint8)):
        got, ng = m.SearchByProjectionFrame(wp1, np.ones(200, np.uint8), bl, d_same, oc, an, Tcw, FX, FY, CX, CY, SF, 40.0)
        ref, nr = oracle.search_by_projection_frame(F2, wp1, np.ones(200, np.uint8), bl, d_same, oc, an, Tcw, FX, FY, CX, CY, SF, 40.0)
        assert np.array_equal(got, ref) and ng == nr
    m.close()
    # an arena of 64 candidates for a search that enumerates thousands: CS_ERR_CAPACITY, and the matcher still works afterwards with a smaller search
    small = ORBmatcher(0.9, True, ctx=ctx, max_candidates=64)
    small.set_frame(k2, d2, BOUNDS)
    with pytest.raises(Exception):
        small.SearchByProjectionFrame(wp, ones, ones, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, 15.0)
    got, ng = small.SearchByProjectionFrame(wp[:5], ones[:5], ones[:5], d1[:5], k1["octave"][:5], k1["angle"][:5], Tcw, FX, FY, CX, CY, SF, 15.0)
    ref, nr = oracle.search_by_projection_frame(F2, wp[:5], ones[:5], ones[:5], d1[:5], k1["octave"][:5], k1["angle"][:5], Tcw, FX, FY, CX, CY, SF, 15.0)
    assert np.array_equal(got, ref) and ng == nr
    small.close()
    # a frame without key points
    e = ORBmatcher(0.9, True, ctx=ctx)
    e.set_frame(k2[:0], d2[:0], BOUNDS)
    tm, nm = e.SearchByProjectionFrame(wp, ones, ones, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, 15.0)
    assert nm == 0 and len(tm) == 0
    e.close()
