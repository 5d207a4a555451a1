"""The drop-in adapters, run through the REFERENCE'S class interfaces on the MI355X (oracle/_ref/libadapters.so = adapters/*.cc compiled
against the reference's headers, linked with libcubeslam_hip.so), next to the reference's own translation units (oracle/_ref/libref.so):
ORB_SLAM2::ORBextractor::operator() and line_lbd_detect::detect_raw_lines / detect_filter_lines give the same bytes."""
import ctypes as C
import os

import numpy as np
import pytest

from cube_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs(ctx, oracle):
    a, r = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("libadapters.so", "libref.so"))
    if not (os.path.exists(a) and os.path.exists(r)):
        pytest.skip("oracle/_ref/*.so are built where /root/reference exists and travel with the snapshot")
    return C.CDLL(a), C.CDLL(r)


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_orbextractor_adapter_equals_reference_class(libs, oracle):
    adp, ref = libs
    for gray in (synth.cuboid_scene(synth.SEED, n_boxes=3, bg_texture=0.5)["gray"], synth.texture_image(5, 1241, 376)):
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        nf, nl = 1000, 8
        cap = 2 * nf + 64
        out = []
        for lib, fn in ((adp, "adp_orb_extract"), (ref, "ref_orb_extract")):
            kps = np.zeros(cap, oracle.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
            levels = np.zeros(4 * W * H, np.uint8); dims = np.zeros(2 * nl, np.int32)
            args = [nf, C.c_float(1.2), nl, 20, 7, _dp(gray), W, H, _dp(kps), _dp(desc), cap, _dp(levels), _dp(dims)]
            if fn.startswith("adp"):
                tables = np.zeros(4 * nl, np.float32)
                args.append(_dp(tables))
            n = getattr(lib, fn)(*args)
            out.append((n, kps[:n].tobytes(), desc[:n].copy(), levels.copy(), dims.copy()))
        (na, ka, da, la, dima), (nr, kr, dr, lr, dimr) = out
        assert na == nr > 500 and ka == kr and np.array_equal(da, dr)
        assert np.array_equal(dima, dimr) and np.array_equal(la, lr)  # mvImagePyramid
        ext = oracle.ORBextractor(nf, 1.2, nl, 20, 7)
        assert np.allclose(tables[:nl], [1.2 ** i for i in range(nl)], rtol=1e-6)


def test_line_lbd_detect_adapter_equals_reference_class(libs, oracle):
    adp, ref = libs
    for gray in (synth.cuboid_scene(synth.SEED, n_boxes=3, bg_texture=0.5)["gray"], synth.texture_image(9, 200, 150)):
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        cap = 20000
        kl_a = np.zeros(cap, oracle.KEYLINE_DTYPE); kl_r = np.zeros(cap, oracle.KEYLINE_DTYPE)
        filt = np.zeros((cap, 4), np.float32); nfilt = C.c_int()
        na = adp.adp_lsd_keylines(_dp(gray), W, H, C.c_float(15.0), _dp(kl_a), cap, _dp(filt), C.byref(nfilt))
        nr = ref.ref_lsd_keylines(_dp(gray), W, H, _dp(kl_r), cap)
        assert na == nr > 20 and kl_a[:na].tobytes() == kl_r[:nr].tobytes()
        want = oracle.lsd_detect_filter_lines(gray, 15.0)
        assert nfilt.value == len(want) and np.array_equal(filt[:nfilt.value], want)
