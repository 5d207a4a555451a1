"""GPU tests at the sizes BASELINE.json quotes (configs 4 and 5), through properties that do not need the oracle to run the whole
workload: batch independence and run-to-run determinism for batched cuboid detection, oracle agreement on a slice; monotone,
oracle-matching chi2 for the 2000-keyframe object BA."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.ba import BundleAdjuster
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid

pytestmark = pytest.mark.gpu


def _batch(ctx, det, scenes):
    K = scenes[0]["K"]
    b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), K, np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes],
                    [s["lines"] for s in scenes], det.opts())
    b.run()
    out = b.read()
    b.close()
    return out


def test_config4_batched_cuboids_8_boxes(ctx, oracle):
    """config 4 shape: many frames x up to 8 boxes, 180-yaw sweep.  64 frames here (one GPU's share of 512)."""
    scenes = []
    seed = 7000
    while len(scenes) < 64:
        s = synth.cuboid_scene(seed, n_boxes=8)
        seed += 1
        if len(s["boxes"]) >= 5:
            scenes.append(s)
    det = detect_3d_cuboid(ctx)
    det.set_calibration(scenes[0]["K"])
    det.yaw_step_deg = 0.5
    full = _batch(ctx, det, scenes)
    again = _batch(ctx, det, scenes)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(full, again)), "deterministic"
    # batch independence: a frame gives the same cuboids whatever else is in the batch, and in any position
    sub = _batch(ctx, det, [scenes[41], scenes[3]])
    nb = [len(s["boxes"]) for s in scenes]
    off = np.concatenate([[0], np.cumsum(nb)])
    for j, f in enumerate((41, 3)):
        o2 = 0 if j == 0 else nb[41]
        for k in range(nb[f]):
            assert full[off[f] + k].tobytes() == sub[o2 + k].tobytes()
    # oracle agreement on six frames spread over the big batch
    oo = oracle.cuboid_opts(yaw_step_deg=0.5)
    for f in (0, 9, 17, 30, 48, 63):
        ref, _ = oracle.detect_cuboid(scenes[f]["gray"], scenes[f]["K"], scenes[f]["Twc"], scenes[f]["boxes"], scenes[f]["lines"], opts=oo)
        for k, r in enumerate(ref):
            g = full[off[f] + k]
            assert len(g) == len(r), (f, k)
            for name in g.dtype.names:
                if name == "box_corners_2d":
                    assert np.array_equal(g[name], r[name]), (f, k)
                else:
                    assert np.allclose(g[name], r[name], rtol=1e-5, atol=1e-9), (f, k, name)
    assert sum(len(c) for c in full) > 64


def test_bench_sized_batches(ctx, oracle):
    """The sizes bench.py runs at: 1 024 frames resident in the extractor and in the cuboid batch (16 different scenes, repeated).  Every copy of a
    scene gives the bytes of its first copy wherever it sits in the batch, and the first copies equal the oracle."""
    from cube_slam_amd.orb import ORBextractor
    F, D = 1024, 16
    base = [synth.cuboid_scene(8100 + i, n_boxes=3, bg_texture=0.5) for i in range(D)]
    scenes = [base[i % D] for i in range(F)]
    det = detect_3d_cuboid(ctx)
    det.set_calibration(base[0]["K"])
    cub = _batch(ctx, det, scenes)
    nb = [len(s["boxes"]) for s in scenes]
    off = np.concatenate([[0], np.cumsum(nb)])
    for f in range(D, F):
        for k in range(nb[f]):
            assert cub[off[f] + k].tobytes() == cub[off[f % D] + k].tobytes(), (f, k)
    for f in (0, 7):
        ref, _ = oracle.detect_cuboid(base[f]["gray"], base[f]["K"], base[f]["Twc"], base[f]["boxes"], base[f]["lines"], opts=oracle.cuboid_opts())
        for k, r in enumerate(ref):
            g = cub[off[f] + k]
            assert len(g) == len(r) and np.array_equal(g["box_corners_2d"], r["box_corners_2d"])
            assert np.allclose(g["pos"], r["pos"], rtol=1e-5, atol=1e-9) and np.allclose(g["edge_distance_error"], r["edge_distance_error"], rtol=1e-5)
    orb = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_frames=F, ctx=ctx)
    orb.upload(np.stack([s["gray"] for s in scenes]))
    orb.run()
    per = orb.read()
    for f in range(D, F, 37):
        assert per[f][0].tobytes() == per[f % D][0].tobytes() and per[f][1].tobytes() == per[f % D][1].tobytes(), f
    ext = oracle.ORBextractor(1000, 1.2, 8, 20, 7)
    for f in (0, 7):
        k, d = ext(base[f]["gray"])
        assert per[f][0].tobytes() == k.tobytes() and np.array_equal(per[f][1], d)
    orb.close()


def test_config5_object_ba_2000_keyframes(ctx, oracle):
    d = synth.ba_problem(20260923, n_kf=2000, n_points=100000, n_cuboids=500)
    ba = BundleAdjuster(d, ctx=ctx)
    chi0, _, _, _ = ba.errors()
    st = ba.optimize(3)
    cam, pts, cub = ba.read()
    ba.close()
    tr = [st["chi2_init"]] + st["chi2_trace"]
    assert abs(chi0 - st["chi2_init"]) <= 1e-9 * chi0
    assert all(b <= a for a, b in zip(tr, tr[1:])) and tr[-1] < 0.9 * tr[0]
    _, _, _, rst = oracle.ba_optimize(d, 3)
    assert rst["iterations"] == st["iterations"] and rst["lm_trials"] == st["lm_trials"]
    assert np.allclose(st["chi2_trace"], rst["chi2_trace"], rtol=1e-5)
    assert abs(st["chi2_final"] - rst["chi2_final"]) <= 1e-5 * rst["chi2_final"]
    assert np.allclose(np.linalg.norm(cam[:, 3:], axis=1), 1.0, atol=1e-12)
