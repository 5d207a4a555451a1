"""CPU checks of the device region stage of the line detector (cube_slam_amd/csrc/lsd_regions.hip):
  * the transaction source the kernel compiles (lsd_rg_txn.h), run on the host with 256 interleaved lanes, reaches the owner map and the
    line candidates of the oracle's sequential algorithm (tools/lsd_sim/txn_sim.cpp);
  * the cosf / sinf restatement the device uses equals the host's libm (glibc_sincosf.h)."""
import os
import subprocess

import numpy as np

from cube_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(src, out, extra=()):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-ffp-contract=off", "-fno-builtin", "-o", out, src, *extra], cwd=ROOT)


def test_sincosf_restatement_equals_libm(tmp_path):
    exe = str(tmp_path / "sincosf_check")
    _build("tests/cpp/sincosf_check.cpp", exe)
    out = subprocess.run([exe, os.environ.get("SINCOSF_STRIDE", "97")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_interleaved_transactions_reach_the_sequential_result(tmp_path):
    exe = str(tmp_path / "txn_sim")
    _build("tools/lsd_sim/txn_sim.cpp", exe)
    for seed, tex, lanes in ((11, 0.5, 256), (12, 0.0, 1024)):
        raw = str(tmp_path / ("f%d.raw" % seed))
        synth.cuboid_scene(seed, n_boxes=3, bg_texture=tex)["gray"].astype(np.uint8).tofile(raw)
        out = subprocess.run([exe, raw, "640", "480", str(lanes)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout[-600:]
        assert "owner map wrong 0" in out.stdout and "EQUAL" in out.stdout
