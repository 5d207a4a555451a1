"""Pins of the oracle against the REFERENCE'S OWN CODE: oracle/_ref/libref.so is built (oracle/Makefile.ref) from the reference's
translation units where they lie under /root/reference -- orb_object_slam/src/ORBextractor.cc, line_lbd/libs/lsd.cpp,
line_lbd/libs/LSDDetector.cpp, ... -- against a stand-in for the OpenCV headers (oracle/ref_shim/).  Every function the reference itself
wrote is therefore the real thing; only the OpenCV primitives underneath (resize, GaussianBlur, FAST, ...) are the oracle's restatements
(SURVEY.md Appendix B).  The oracle's restatement must reproduce the reference bit for bit on the same inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cube_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")


@pytest.fixture(scope="module")
def ref(oracle):
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref.so is built from /root/reference, which is not present here")
    return C.CDLL(REF_SO)


def _images():
    out = [("cuboid_scene", synth.cuboid_scene(synth.SEED, n_boxes=3)["gray"]), ("texture_640x480", synth.texture_image(3, 640, 480)),
           ("texture_1241x376", synth.texture_image(5, 1241, 376)), ("texture_200x150", synth.texture_image(9, 200, 150))]
    g = np.load(os.path.join(ROOT, "tests", "golden", "cuboid_ref_0000.npz"))
    if "gray" in g.files:
        out.append(("cabinet_0000", g["gray"]))
    return out


@pytest.mark.parametrize("nfeatures,nlevels", [(1000, 8), (2000, 8), (300, 4)])
def test_orb_extractor_equals_reference(ref, oracle, nfeatures, nlevels):
    """ORBextractor::ORBextractor, ComputePyramid, ComputeKeyPointsOctTree, DistributeOctTree / DivideNode, IC_Angle, computeOrbDescriptor,
    operator() (ORBextractor.cc:74-150, 412-471, 483-763, 766-853, 1036-1125): key points (28-byte records), descriptors, pyramid levels."""
    for name, gray in _images():
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        cap = nfeatures * 2 + 64
        kps = np.zeros(cap, oracle.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        levels = np.zeros(4 * W * H + 64, np.uint8)
        dims = np.zeros(2 * nlevels, np.int32)
        n = ref.ref_orb_extract(nfeatures, C.c_float(1.2), nlevels, 20, 7, gray.ctypes.data_as(C.c_void_p), W, H, kps.ctypes.data_as(C.c_void_p),
                                desc.ctypes.data_as(C.c_void_p), cap, levels.ctypes.data_as(C.c_void_p), dims.ctypes.data_as(C.c_void_p))
        ext = oracle.ORBextractor(nfeatures, 1.2, nlevels, 20, 7)
        okp, odesc = ext(gray)
        assert n == len(okp) and n > 0, (name, n, len(okp))
        assert kps[:n].tobytes() == okp.tobytes(), name
        assert np.array_equal(desc[:n], odesc), name
        off = 0
        for l in range(nlevels):
            lv = ext.level(l)
            assert (dims[2 * l], dims[2 * l + 1]) == (lv.shape[1], lv.shape[0])
            assert np.array_equal(levels[off:off + lv.size].reshape(lv.shape), lv), (name, l)
            off += lv.size
        fpl = np.zeros(nlevels, np.int32)
        ref.ref_orb_features_per_level(nfeatures, C.c_float(1.2), nlevels, fpl.ctypes.data_as(C.c_void_p))
        assert np.array_equal(fpl, ext.features_per_level())


def test_lsd_equals_reference(ref, oracle):
    """LineSegmentDetectorImpl::detect / flsd / ll_angle / region_grow / region2rect / refine / rect_improve / rect_nfa / nfa
    (line_lbd/libs/lsd.cpp:414-1155) and LSDDetector::detectImpl's KeyLine fill (LSDDetector.cpp:153-263): KeyLines byte for byte.  This also
    settles the overload question of lsd.cpp:680-681 (`cos(float)` inside namespace cv) empirically."""
    total = 0
    for name, gray in _images():
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        cap = 20000
        kl = np.zeros(cap, oracle.KEYLINE_DTYPE)
        n = ref.ref_lsd_keylines(gray.ctypes.data_as(C.c_void_p), W, H, kl.ctypes.data_as(C.c_void_p), cap)
        okl = oracle.lsd_detect(gray)
        assert n == len(okl), (name, n, len(okl))
        assert kl[:n].tobytes() == okl.tobytes(), name
        seg = np.zeros((cap, 7), np.float64)
        m = ref.ref_lsd_segments(gray.ctypes.data_as(C.c_void_p), W, H, seg.ctypes.data_as(C.c_void_p), cap)
        assert m >= n  # detectImpl drops segments along the image border
        total += n
    assert total > 300
