"""GPU parity: HIP detect_3d_cuboid path vs the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.cuboid import CuboidBatch, detect_3d_cuboid

pytestmark = pytest.mark.gpu

REL = 1e-5  # BASELINE.json north_star: cuboid scores within 1e-5 relative


def _oracle_opts(po, det):
    return po.cuboid_opts(consider_config_1=int(det.consider_config_1), consider_config_2=int(det.consider_config_2),
                          whether_sample_cam_roll_pitch=int(det.whether_sample_cam_roll_pitch),
                          whether_sample_bbox_height=int(det.whether_sample_bbox_height), max_cuboid_num=det.max_cuboid_num,
                          nominal_skew_ratio=det.nominal_skew_ratio, max_cut_skew=det.max_cut_skew,
                          yaw_range_deg=det.yaw_range_deg, yaw_step_deg=det.yaw_step_deg)


def _cmp_cuboids(got, ref):
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert len(g) == len(r)
        for name in g.dtype.names:
            a, b = g[name], r[name]
            if name == "box_corners_2d":
                assert np.array_equal(a, b), name
            else:
                assert np.allclose(a, b, rtol=REL, atol=1e-9), (name, a, b)


@pytest.mark.parametrize("seed,kw", [
    (1, {}),
    (2, {"yaw_step_deg": 0.5}),
    (3, {"whether_sample_cam_roll_pitch": True, "max_cuboid_num": 3}),
    (4, {"whether_sample_bbox_height": True, "max_cuboid_num": 5, "nominal_skew_ratio": 2.0}),
    (5, {"consider_config_2": False}),
    (6, {"consider_config_1": False, "yaw_step_deg": 2.0}),
])
def test_batch_stages_match_oracle(ctx, oracle, seed, kw):
    det = detect_3d_cuboid(ctx)
    for k, v in kw.items():
        setattr(det, k, v)
    scenes = [synth.cuboid_scene(100 * seed + i, n_boxes=3) for i in range(2)]
    K = scenes[0]["K"]
    det.set_calibration(K)
    b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), K, np.stack([s["Twc"] for s in scenes]),
                    [s["boxes"] for s in scenes], [s["lines"] for s in scenes], det.opts())
    b.run()
    got = b.read()
    oo = _oracle_opts(oracle, det)
    ref, u = [], 0
    for s in scenes:
        r, dbg = oracle.detect_cuboid(s["gray"], K, s["Twc"], s["boxes"], s["lines"], opts=oo, debug=True)
        ref += r
        seg, row0 = 0, 0
        for bi in range(len(s["boxes"])):
            n_hs = b.unit(u)["n_hs"]
            for hs in range(n_hs):
                info = b.unit(u)
                assert info["hs"] == hs
                x, y, w, h = info["roi"]
                assert np.array_equal(info["edges"], oracle.canny_roi(s["gray"], x, y, w, h)), "canny"
                assert np.array_equal(info["dist"], oracle.canny_dt_roi(s["gray"], x, y, w, h)), "distance transform"
                n = int(dbg["row_count"][seg])
                rows_ref = dbg["rows"][row0:row0 + n]
                assert info["n_valid"] == n, (info["n_valid"], n)
                assert np.array_equal(info["rows"][:, [0, 1, 3, 6]], rows_ref[:, [0, 1, 3, 6]])
                assert np.allclose(info["rows"], rows_ref, rtol=REL, atol=1e-9)
                row0 += n
                seg += 1
                u += 1
    _cmp_cuboids(got, ref)
    b.close()


def test_single_frame_dropin_matches_oracle(ctx, oracle):
    s = synth.cuboid_scene(synth.SEED, n_boxes=3)
    det = detect_3d_cuboid(ctx)
    det.set_calibration(s["K"])
    det.max_cuboid_num = 2
    got = det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"])
    ref, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=_oracle_opts(oracle, det))
    _cmp_cuboids(got, ref)
    # BGR input goes through the cvtColor kernel
    bgr = np.stack([s["gray"]] * 3, axis=-1)
    got2 = det.detect_cuboid(bgr, s["Twc"], s["boxes"], s["lines"])
    gray2 = oracle.bgr2gray(bgr)
    ref2, _ = oracle.detect_cuboid(gray2, s["K"], s["Twc"], s["boxes"], s["lines"], opts=_oracle_opts(oracle, det))
    _cmp_cuboids(got2, ref2)


def test_edge_cases(ctx, oracle):
    s = synth.cuboid_scene(7, n_boxes=2)
    det = detect_3d_cuboid(ctx)
    det.set_calibration(s["K"])
    # no boxes
    assert det.detect_cuboid(s["gray"], s["Twc"], np.zeros((0, 5)), s["lines"]) == []
    # no lines at all: every VP gets the not-found penalty
    got = det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], np.zeros((0, 4)))
    ref, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], np.zeros((0, 4)), opts=_oracle_opts(oracle, det))
    _cmp_cuboids(got, ref)
    # a thin box (width < 10 -> linespace<int> step 0 guard, 1001 identical top samples) and a box at the image border
    boxes = np.array([[300, 100, 8, 150, 0.5], [0, 0, 200, 300, 0.5], [440, 150, 199, 329, 0.5]], np.float64)
    got = det.detect_cuboid(s["gray"], s["Twc"], boxes, s["lines"])
    ref, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], boxes, s["lines"], opts=_oracle_opts(oracle, det))
    _cmp_cuboids(got, ref)


@pytest.mark.parametrize("W,H", [(640, 480), (1241, 376)])
def test_distance_transform_variants_agree(ctx, oracle, monkeypatch, W, H):
    """The wave-per-ROI distance transform (default, ROI width <= 1280) and the workgroup-per-ROI one (CUBESLAM_DT=block) are the same
    integer recurrence: identical maps, also for a wide ROI that needs more columns per lane, and both equal to the oracle."""
    det = detect_3d_cuboid(ctx)
    s = synth.cuboid_scene(42, n_boxes=3, W=W, H=H)
    boxes = np.array(s["boxes"], np.float64)
    boxes = np.concatenate([boxes, [[5, 5, W - 30, H - 60, 0.5]]])  # one ROI almost as wide as the image
    det.set_calibration(s["K"])
    maps = {}
    for mode in ("wave", "block"):
        if mode == "block":
            monkeypatch.setenv("CUBESLAM_DT", "block")
        b = CuboidBatch(ctx, s["gray"][None], s["K"], s["Twc"][None], [boxes], [s["lines"]], det.opts())
        b.run()
        maps[mode] = [(b.unit(u)["roi"], b.unit(u)["dist"].copy()) for u in range(len(boxes))]
        b.close()
    monkeypatch.delenv("CUBESLAM_DT")
    for (roi, a), (_, c) in zip(maps["wave"], maps["block"]):
        x, y, w, h = roi
        assert np.array_equal(a, c) and np.array_equal(a, oracle.canny_dt_roi(s["gray"], x, y, w, h))
    assert max(r[0][2] for r in maps["wave"]) > 560


def test_score_paths_agree(ctx, oracle, monkeypatch):
    """cuboid_sweep_score keeps a unit's chamfer map in LDS as exact 16-bit (i, j) codes (it encodes the float map itself) and hands units -- or
    slices of units, when there are fewer units than CUs -- to persistent workgroups.  Whatever the number of workgroups, slices per unit and
    threads per workgroup (512: shared corner products; 768 / 1024: the lean body), the cuboids are byte-identical, and they equal the
    oracle's -- including a box whose ROI does not fit one CU's LDS (the head of its code map is resident, samples past it read the float map
    in global memory) and a flat image whose distance map has no codes at all (its units sample the float map only)."""
    det = detect_3d_cuboid(ctx)
    det.yaw_step_deg = 2.0
    scenes = [synth.cuboid_scene(70 + i, n_boxes=3) for i in range(3)]
    flat = dict(scenes[0]); flat["gray"] = np.full_like(scenes[0]["gray"], 128)
    scenes.append(flat)
    boxes = [np.array(s["boxes"], np.float64) for s in scenes]
    boxes[1] = np.concatenate([boxes[1], [[20, 20, 560, 400, 0.5]]])  # 600 x 440 ROI: 264 000 pixels do not fit
    det.set_calibration(scenes[0]["K"])
    settings = [{}, {"CUBESLAM_SCORE_SEGMENTS": "1"}, {"CUBESLAM_SCORE_SEGMENTS": "7", "CUBESLAM_SCORE_SLICES": "1"}, {"CUBESLAM_SCORE_SEGMENTS": "1000", "CUBESLAM_SCORE_SLICES": "5"},
                {"CUBESLAM_SCORE_THREADS": "1024"}, {"CUBESLAM_SCORE_THREADS": "768", "CUBESLAM_SCORE_SLICES": "64"}]
    out = []
    for env in settings:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = CuboidBatch(ctx, np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), boxes, [s["lines"] for s in scenes], det.opts())
        b.run()
        out.append(b.read())
        if not env:
            ss = b.score_stats()
            assert ss["float_units"] == 3 and ss["code_units"] == len(np.concatenate(boxes)) - 3  # the flat frame's three boxes have no codes
        b.close()
        for k in env:
            monkeypatch.delenv(k)
    for other in out[1:]:
        for g, l in zip(out[0], other):
            assert len(g) == len(l) and np.array_equal(np.asarray(g).view(np.uint8), np.asarray(l).view(np.uint8))
    ref = []
    oo = _oracle_opts(oracle, det)
    for s, bx in zip(scenes, boxes):
        r, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], bx, s["lines"], opts=oo)
        ref += r
    _cmp_cuboids(out[0], ref)


def test_set_lines_equals_a_fresh_batch(ctx):
    """cs_cuboid_batch_set_lines: new edge lists for resident frames (longer and shorter than the ones the batch was created with) give the
    cuboids of a batch created with them."""
    det = detect_3d_cuboid(ctx)
    det.yaw_step_deg = 3.0
    scenes = [synth.cuboid_scene(300 + i, n_boxes=3) for i in range(3)]
    det.set_calibration(scenes[0]["K"])
    args = (np.stack([s["gray"] for s in scenes]), scenes[0]["K"], np.stack([s["Twc"] for s in scenes]), [s["boxes"] for s in scenes])
    rng = np.random.default_rng(5)
    variants = [[s["lines"] for s in scenes], [s["lines"][::3] for s in scenes], [np.concatenate([s["lines"], rng.uniform(0, 470, (300, 4))]) for s in scenes]]
    b = CuboidBatch(ctx, *args, variants[1], det.opts())
    for v in (variants[0], variants[2], variants[1]):
        b.set_lines(v)
        b.run()
        got = b.read()
        f = CuboidBatch(ctx, *args, v, det.opts())
        f.run()
        ref = f.read()
        f.close()
        for a, c in zip(got, ref):
            assert len(a) == len(c) and np.array_equal(np.asarray(a).view(np.uint8), np.asarray(c).view(np.uint8))
    b.close()


@pytest.mark.parametrize("seed,kw", [(1, {}), (19, {}), (36, {"max_cuboid_num": 3}), (7, {"whether_sample_bbox_height": True})])
def test_per_frame_call_chains_the_boxes_like_the_reference(ctx, oracle, seed, kw):
    """cs_cuboid_detect with camera roll / pitch sampling and several boxes: box b + 1 starts its yaw samples from the camera pose box b left behind (the pose of the
    last proposal it turned into a cuboid, box_proposal_detail.cpp:126 after :481-487 -- the configuration of object_slam/src/main_obj.cpp:442).  Against the oracle in
    its stateful mode, which is held to the reference's own text (tests/test_ref_pins.py::test_detect_cuboid_equals_reference[rollpitch])."""
    det = detect_3d_cuboid(ctx)
    det.whether_sample_cam_roll_pitch = True
    for k, v in kw.items():
        setattr(det, k, v)
    s = synth.cuboid_scene(seed, n_boxes=3, bg_texture=0.5 if seed == 9 else 0.0)
    assert len(s["boxes"]) >= 2
    det.set_calibration(s["K"])
    got = det.detect_cuboid(s["gray"], s["Twc"], s["boxes"], s["lines"])
    oo = _oracle_opts(oracle, det)
    oo.stateful_cam_pose = 1
    ref, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=oo)
    _cmp_cuboids(got, ref)
    assert sum(len(g) for g in got) >= 3
    if seed in (1, 19, 36):   # frames on which the chain changes a later box: the stateless variant (every box from the raw pose, what a batch computes) gives something else
        oo.stateful_cam_pose = 0
        other, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=oo)
        assert any(len(a) != len(b) or any(not np.array_equal(a[f], b[f]) for f in a.dtype.names) for a, b in zip(ref, other))
