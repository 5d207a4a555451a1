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
    # oracle agreement on one frame of the big batch
    oo = oracle.cuboid_opts(yaw_step_deg=0.5)
    ref, _ = oracle.detect_cuboid(scenes[17]["gray"], scenes[17]["K"], scenes[17]["Twc"], scenes[17]["boxes"], scenes[17]["lines"], opts=oo)
    for k, r in enumerate(ref):
        g = full[off[17] + k]
        assert len(g) == len(r)
        for name in g.dtype.names:
            if name == "box_corners_2d":
                assert np.array_equal(g[name], r[name])
            else:
                assert np.allclose(g[name], r[name], rtol=1e-5, atol=1e-9), name
    assert sum(len(c) for c in full) > 64


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
