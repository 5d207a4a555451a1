"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares
(no compute calls without a GPU), struct layouts match the Python mirrors, and there is no CPU fallback."""
import numpy as np
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from cube_slam_amd import _lib
    lib = _lib.lib()
    names = _declared_functions()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cube_slam_amd import _lib
    with pytest.raises(_lib.CubeSlamError):
        _lib.Context(0)


def test_struct_layouts():
    import numpy as np
    from cube_slam_amd import cuboid
    from oracle import pyoracle
    assert cuboid.CUBOID_DTYPE.itemsize == 416 == pyoracle.CUBOID_DTYPE.itemsize
    assert C.sizeof(cuboid.CuboidOpts) == 64
    o = cuboid.CuboidOpts()
    from cube_slam_amd import _lib
    _lib.lib().cs_cuboid_default_opts(C.byref(o))
    assert (o.consider_config_1, o.consider_config_2, o.max_cuboid_num, o.yaw_step_deg, o.canny_low, o.canny_high) == (1, 1, 1, 6.0, 80, 200)
    assert np.dtype(cuboid.CUBOID_DTYPE).fields["box_corners_3d_world"][1] == 136


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under cube_slam_amd/ or include/ may reference it."""
    bad = []
    for pat in ("cube_slam_amd/**/*.py", "cube_slam_amd/**/*.hip", "cube_slam_amd/**/*.h", "cube_slam_amd/**/*.cpp", "include/*.h"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            if re.search(r"#\s*include[^\n]*oracle|^\s*(from|import)\s+oracle|liboracle|orc_[a-z]+\(", open(f).read(), flags=re.M):
                bad.append(f)
    assert not bad, bad


def test_short_edge_threshold_without_sqrt():
    """corners_build (cuboid_sweep_filter) tests `dist < 20` as `dx*dx + dy*dy < pred(400)`: equivalent for correctly rounded sqrt."""
    import struct
    import numpy as np
    T = struct.unpack("<d", struct.pack("<Q", 0x4078ffffffffffff))[0]
    assert T == np.nextafter(400.0, 0.0)
    below = T
    for _ in range(2000):
        below = np.nextafter(below, 0.0)
        assert np.sqrt(below) < 20.0
    above = T
    for _ in range(2000):
        assert not (np.sqrt(above) < 20.0)
        above = np.nextafter(above, 1e9)
    rng = np.random.default_rng(0)
    d2 = np.concatenate([rng.uniform(0, 800, 200000), 400.0 + rng.normal(0, 1e-12, 200000)])
    assert np.array_equal(np.sqrt(d2) < 20.0, d2 < T)


def test_chamfer_code_number_theory():
    """cuboid_sweep_score (cuboid.hip) keeps a chamfer value t = i*62587 + j*89738 (t * 2^-16 px) as the 16-bit code i | j << 8 for
    cuboid_sweep_score and recovers (i, j) from t with one float FMA and a 978-entry table.  The facts its encoder relies on: the 256
    residues j*89738 mod 62587 fall into distinct 64-wide buckets (they are >= 97 apart), floor(t / 62587) comes out right from
    float(t) * float(1/62587) + 0.0005 for every representable pair, i = floor(t/62587) - floor(j*89738/62587), the table entry
    (j << 8) - floor(j*89738/62587) fits 16 bits, d < 244 px bounds i <= 255 and j <= 178, and the decode fma(j, 89738 * 2^-16,
    i * 62587 * 2^-16) in float equals float(t) * 2^-16."""
    a, b = 62587, 89738
    res = [(j * b) % a for j in range(256)]
    assert len({r >> 6 for r in res}) == 256 and max(r >> 6 for r in res) < (a + 63) // 64
    srt = sorted(res)
    assert min(y - x for x, y in zip(srt, srt[1:])) >= 97 and srt[1] >= 97 and a - srt[-1] >= 97
    I, J = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    t = I * a + J * b
    ok = t < (1 << 24)  # the codes the encoder accepts: d < 256 px
    tf = t.astype(np.float32)
    qn = np.floor(tf.astype(np.float64) * np.float64(np.float32(1.0 / 62587.0)) + np.float64(np.float32(0.0005)))  # one rounding at most below the true FMA
    qn32 = np.floor((tf * np.float32(1.0 / 62587.0)).astype(np.float32) + np.float32(0.0005))
    assert np.array_equal(qn[ok], (t // a)[ok]) and np.array_equal(qn32[ok], (t // a)[ok])
    assert np.array_equal((t // a - (J * b) // a)[ok], I[ok]) and ((J * b) // a).max() < 1 << 16
    assert 255 * a + 255 * b >= 1 << 24, "the escape code (255, 255) decodes above every valid t"
    lut = (np.arange(256) << 8) - (np.arange(256) * b) // a
    assert lut.min() >= 0 and lut.max() < 1 << 16
    assert np.array_equal((t // a + lut[J])[ok], (I | (J << 8))[ok]), "code = qn + lut[bucket of the residue]"
    near = t < 244 * 65536  # what the encoder accepts (farther pixels become the escape code)
    assert I[near].max() <= 255 and J[near].max() <= 178
    assert np.array_equal((I.astype(np.float32) * np.float32(a / 65536.0)).astype(np.float64), I * (a / 65536.0)), "float(i) * HV * 2^-16 is exact"
    dec = J * (b / 65536.0) + I * (a / 65536.0)  # the fused multiply-add: exact in double, rounded to float once
    assert np.array_equal(dec.astype(np.float32)[near], (tf * np.float32(1.0 / 65536.0))[near])
    assert np.array_equal((tf * np.float32(1.0 / 65536.0))[ok].astype(np.float64), t[ok] / 65536.0), "decode: float(t) * 2^-16 is exact"
