"""CPU checks of the device region stage of the line detector (cube_slam_amd/csrc/lsd_regions.hip):
  * the sources the kernels compile (lsd_rg_seq.h: one wave per frame, the default of large batches; lsd_rg_grp.h: eight or four frames per wave;
    lsd_rg_lpf.h: one lane per frame), run on the host with the 64 lanes as loops, leave the `used` map and hand over the rectangles of the oracle's
    sequential algorithm, bit for bit and in the same order (tools/lsd_sim/seq_sim.cpp, grp_sim.cpp, lpf_sim.cpp);
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


def test_frames_per_wave_stage_equals_the_sequential_algorithm(tmp_path):
    """lsd_rg_grp.h on the host: a wave of eight (P = 1) / four (P = 2) different frames, every one of them held to the oracle's sequence; also with a
    list capacity that the regions outgrow (the frame must say so and stop)."""
    exes = {}
    for p in (1, 2):
        exes[p] = str(tmp_path / ("grp_sim%d" % p))
        _build("tools/lsd_sim/grp_sim.cpp", exes[p], ["-DGRP_P=%d" % p])
    raws = []
    for seed, tex in ((11, 0.5), (12, 0.0), (13, 1.0), (14, 0.25), (3, 1.0), (21, 0.5), (22, 0.75)):
        raws.append(str(tmp_path / ("g%d.raw" % seed)))
        synth.cuboid_scene(seed, n_boxes=3, bg_texture=tex)["gray"].astype(np.uint8).tofile(raws[-1])
    raws.append(str(tmp_path / "gtex.raw"))  # dense texture: the most seeds and refinements
    synth.texture_image(8, 640, 480).astype(np.uint8).tofile(raws[-1])
    out = subprocess.run([exes[1], "640", "480", *raws], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.count("EQUAL") == 8 and "DIFFERENT" not in out.stdout, out.stdout[-1200:]
    out = subprocess.run([exes[2], "640", "480", raws[0], raws[2], raws[4], raws[7]], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.count("EQUAL") == 4 and "DIFFERENT" not in out.stdout, out.stdout[-1200:]
    out = subprocess.run([exes[1], "640", "480", raws[0], raws[1]], capture_output=True, text=True, env=dict(os.environ, GRP_CAP="64"))
    assert out.returncode == 2 and out.stdout.count("fail 1") == 2, out.stdout[-600:]


def test_lane_per_frame_stage_equals_the_sequential_algorithm(tmp_path):
    """lsd_rg_lpf.h on the host: a wave whose lanes walk eleven different frames (flat to densely textured), every one of them held to the oracle's
    sequence; also with a list capacity that the regions outgrow."""
    exe = str(tmp_path / "lpf_sim")
    _build("tools/lsd_sim/lpf_sim.cpp", exe)
    raws = []
    for seed, tex in ((11, 0.5), (12, 0.0), (13, 1.0), (14, 0.25), (3, 1.0), (21, 0.5), (22, 0.75), (31, 0.6), (32, 0.4), (33, 0.9)):
        raws.append(str(tmp_path / ("l%d.raw" % seed)))
        synth.cuboid_scene(seed, n_boxes=3, bg_texture=tex)["gray"].astype(np.uint8).tofile(raws[-1])
    raws.append(str(tmp_path / "ltex.raw"))
    synth.texture_image(8, 640, 480).astype(np.uint8).tofile(raws[-1])
    out = subprocess.run([exe, "640", "480", *raws], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.count("EQUAL") == len(raws) and "DIFFERENT" not in out.stdout, out.stdout[-1500:]
    out = subprocess.run([exe, "640", "480", raws[0], raws[1]], capture_output=True, text=True, env=dict(os.environ, GRP_CAP="64"))
    assert out.returncode == 2 and out.stdout.count("fail 1") == 2, out.stdout[-600:]
