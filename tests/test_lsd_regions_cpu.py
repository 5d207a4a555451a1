"""CPU checks of the device region stage of the line detector (cube_slam_amd/csrc/lsd_regions.hip):
  * the sources the kernels compile (lsd_rg_seq.h: one wave per frame, the default of large batches; lsd_rg_wlk.h: walker waves with one lane per
    frame + rectangle waves), run on the host with the 64 lanes as loops, leave the `used` map and hand over the rectangles of the oracle's
    sequential algorithm, bit for bit and in the same order (tools/lsd_sim/seq_sim.cpp, wlk_sim.cpp);
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


def test_walker_and_rectangle_waves_equal_the_sequential_algorithm(tmp_path):
    """lsd_rg_wlk.h on the host: a walker wave whose lanes walk eleven different frames (flat to densely textured) and park their regions with the
    rectangle stage (region2rect, refine's statistics, reduce_region_radius as a compaction in the order of the reference's swaps), every frame held to
    the oracle's sequence -- with one and with two accepted pixels per iteration; also with a list capacity that the regions outgrow."""
    raws = []
    for seed, tex in ((11, 0.5), (12, 0.0), (13, 1.0), (14, 0.25), (3, 1.0), (21, 0.5), (22, 0.75), (31, 0.6), (32, 0.4), (33, 0.9)):
        raws.append(str(tmp_path / ("l%d.raw" % seed)))
        synth.cuboid_scene(seed, n_boxes=3, bg_texture=tex)["gray"].astype(np.uint8).tofile(raws[-1])
    raws.append(str(tmp_path / "ltex.raw"))
    synth.texture_image(8, 640, 480).astype(np.uint8).tofile(raws[-1])
    for acc in (1, 2):
        exe = str(tmp_path / ("wlk_sim%d" % acc))
        _build("tools/lsd_sim/wlk_sim.cpp", exe, ["-DWLK_ACC=%d" % acc])
        out = subprocess.run([exe, "640", "480", *raws], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.count("EQUAL") == len(raws) and "DIFFERENT" not in out.stdout, out.stdout[-1500:]
    out = subprocess.run([exe, "640", "480", raws[0], raws[1]], capture_output=True, text=True, env=dict(os.environ, GRP_CAP="64"))
    assert out.returncode == 2 and out.stdout.count("fail 1") == 2, out.stdout[-600:]


def test_corner_selection_on_values_equals_the_indexed_form():
    """lsd_rg_improve's RectSpan::set picks rect_nfa's lowest / leftmost / rightmost / last corner from a sorting network and the positions the sorted order implies
    (lsd_regions.hip); the reference sorts with std::sort and selects through `taken` flags and strict comparisons (lsd.cpp:1010-1050, restated in LsdHost::rect_count).
    Both forms on every placement of four corners on a 4 x 4 grid -- ties in x, in y and in both included: the values the row walk uses are the same."""
    import itertools

    def indexed(pts):  # LsdHost::rect_count, lsd.hip
        v = list(pts)
        for i in range(1, 4):  # insertion sort by (x, y)
            j = i
            while j > 0 and (v[j][0] < v[j - 1][0] or (v[j][0] == v[j - 1][0] and v[j][1] < v[j - 1][1])):
                v[j], v[j - 1] = v[j - 1], v[j]
                j -= 1
        vx, vy = [p[0] for p in v], [p[1] for p in v]
        taken = [False] * 4
        mn = mx = 0
        for i in range(1, 4):
            if vy[mn] > vy[i]:
                mn = i
            if vy[mx] < vy[i]:
                mx = i
        taken[mn] = True
        lm = rm = tp = -1
        for i in range(4):
            if not taken[i]:
                if lm < 0 or vx[lm] > vx[i]:
                    lm = i
        taken[lm] = True
        for i in range(4):
            if not taken[i]:
                if rm < 0 or vx[rm] < vx[i]:
                    rm = i
        taken[rm] = True
        for i in range(4):
            if not taken[i]:
                if tp < 0 or vx[tp] > vx[i]:
                    tp = i
        return vx[mn], vy[mn], vx[lm], vy[lm], vx[rm], vy[rm], vx[tp], vy[mx]

    def on_values(pts):  # RectSpan::set, lsd_regions.hip
        (x0, y0), (x1, y1), (x2, y2), (x3, y3) = pts

        def cswap(a, b):
            return (b, a) if (a[0] > b[0] or (a[0] == b[0] and a[1] > b[1])) else (a, b)
        p0, p1 = cswap((x0, y0), (x1, y1)); p2, p3 = cswap((x2, y2), (x3, y3))
        p0, p2 = cswap(p0, p2); p1, p3 = cswap(p1, p3); p1, p2 = cswap(p1, p2)
        s = [p0, p1, p2, p3]
        mnI, mnX, mnY = 0, s[0][0], s[0][1]
        for i in (1, 2, 3):
            if mnY > s[i][1]:
                mnI, mnX, mnY = i, s[i][0], s[i][1]
        mxY = max(p[1] for p in s)
        lm = s[1] if mnI == 0 else s[0]
        i_ = s[2] if mnI <= 1 else s[1]
        j_ = s[2] if mnI == 3 else s[3]
        rj = i_[0] < j_[0]
        rm, tl = (j_, i_) if rj else (i_, j_)
        return mnX, mnY, lm[0], lm[1], rm[0], rm[1], tl[0], mxY

    n = 0
    for c in itertools.product(range(4), repeat=8):
        pts = [(c[0], c[1]), (c[2], c[3]), (c[4], c[5]), (c[6], c[7])]
        assert indexed(pts) == on_values(pts), pts
        n += 1
    assert n == 4 ** 8


def test_texture_stream_is_texture_image_frame_by_frame():
    import numpy as np
    from cube_slam_amd import synth
    st = synth.texture_stream(77, 320, 96, 6, step=3)
    for i in range(6):
        assert np.array_equal(st[i], synth.texture_image(77, 320, 96, shift=3 * i))
