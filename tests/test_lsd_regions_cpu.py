"""CPU checks of the device region stage of the line detector (cube_slam_amd/csrc/lsd_regions.hip):
  * the source the kernel compiles (lsd_rg_seq.h: one wave per frame), run on the host with the 64 lanes as loops, leaves the `used` map and
    hands over the rectangles of the oracle's sequential algorithm, bit for bit and in the same order (tools/lsd_sim/seq_sim.cpp);
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


def test_wave_per_frame_stage_equals_the_sequential_algorithm(tmp_path):
    exe = str(tmp_path / "seq_sim")
    _build("tools/lsd_sim/seq_sim.cpp", exe)
    exe4 = str(tmp_path / "seq_sim4")  # the list's register ring cut to 4 entries: the growth reads the list from memory
    _build("tools/lsd_sim/seq_sim.cpp", exe4, ["-DRGS_RING=4"])
    for seed, tex in ((11, 0.5), (12, 0.0), (13, 1.0), (14, 0.25), (3, 1.0)):
        raw = str(tmp_path / ("f%d.raw" % seed))
        synth.cuboid_scene(seed, n_boxes=3, bg_texture=tex)["gray"].astype(np.uint8).tofile(raw)
        for e in (exe, exe4) if seed in (11, 3) else (exe,):
            out = subprocess.run([e, raw, "640", "480"], capture_output=True, text=True)
            assert out.returncode == 0, out.stdout[-600:]
            assert "used map wrong 0" in out.stdout and "EQUAL" in out.stdout
    raw = str(tmp_path / "tex.raw")  # dense texture: the most seeds and refinements
    synth.texture_image(8, 640, 480).astype(np.uint8).tofile(raw)
    out = subprocess.run([exe, raw, "640", "480"], capture_output=True, text=True)
    assert out.returncode == 0 and "EQUAL" in out.stdout, out.stdout[-600:]
