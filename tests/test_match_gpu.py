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


@pytest.mark.parametrize("th", [15.0, 30.0])
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
    blocks = (rng.uniform(size=n) < 0.9).astype(np.uint8)
    m = ORBmatcher(0.9, True, ctx=ctx)
    m.set_frame(k2, d2, BOUNDS)
    F2 = oracle.make_frame(k2, d2, BOUNDS)
    got, ng = m.SearchByProjectionFrame(wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th)
    ref, nr = oracle.search_by_projection_frame(F2, wp, valid, blocks, d1, k1["octave"], k1["angle"], Tcw, FX, FY, CX, CY, SF, th)
    assert np.array_equal(got, ref) and ng == nr
    assert ng > 300
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
