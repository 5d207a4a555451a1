"""GPU: the dependency-free C++ host mirrors (cube_slam_amd/host/orb_slam_mirrors.hpp) compiled with g++ against the C-ABI library and run
on one frame; their results must be byte-identical to the Python mirrors' (which the other tests pin against the oracle)."""
import os
import subprocess

import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.lsd import line_lbd_detect
from cube_slam_amd.orb import ORBextractor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_mirrors_match_python(ctx, tmp_path):
    W, H = 640, 480
    img = synth.cuboid_scene(77)["gray"]
    raw = tmp_path / "frame.raw"
    raw.write_bytes(img.tobytes())
    exe = tmp_path / "host_mirrors"
    lib_dir = os.path.join(ROOT, "cube_slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, os.path.join(ROOT, "tests", "cpp", "host_mirrors.cpp"), "-o", str(exe), "-L", lib_dir, "-lcubeslam_hip",
                           "-Wl,-rpath," + lib_dir])
    out = subprocess.check_output([str(exe), str(raw), str(W), str(H)], timeout=300).decode().splitlines()
    tok = {ln.split()[0]: ln.split()[1:] for ln in out}
    kp, desc = ORBextractor(500, 1.2, 8, 20, 7, W, H, ctx=ctx)(img)
    assert int(tok["orb"][0]) == len(kp) > 100
    assert int(tok["orb"][1], 16) == _fnv(kp.tobytes()) and int(tok["orb"][2], 16) == _fnv(desc.tobytes())
    assert tok["levels"][0] == "8" and abs(float(tok["levels"][2]) - 1.2) < 1e-6
    det = line_lbd_detect(W, H, ctx=ctx)
    kl = det.detect_raw_lines(img)
    lm = det.detect_filter_lines(img)
    ld = det.get_line_descriptors(img, kl)
    assert int(tok["lines"][0]) == len(kl) > 5 and int(tok["lines"][1], 16) == _fnv(kl.tobytes())
    assert int(tok["lines"][3]) == len(lm) and int(tok["lines"][4], 16) == _fnv(np.ascontiguousarray(lm, np.float32).tobytes())
    assert int(tok["lines"][6], 16) == _fnv(np.ascontiguousarray(ld, np.uint8).tobytes())
