"""Object association after detect_cuboid: known answers of the oracle restatement (oracle/pyoracle.py) and the
host-side C-ABI implementation of Tracking::AssociateCuboids against it (no GPU involved: that loop is serial host code by design)."""
import numpy as np

from cube_slam_amd import objects


def test_bbox_overlap_ratio_known_answers(oracle):
    assert oracle.bbox_overlap_ratio((0, 0, 10, 10), (5, 5, 10, 10)) == np.float32(25) / np.float32(175)
    assert oracle.bbox_overlap_ratio((0, 0, 10, 10), (10, 0, 10, 10)) == 0, "touching rectangles do not overlap"
    assert oracle.bbox_overlap_ratio((0, 0, 10, 10), (0, 0, 10, 10)) == 1
    assert np.isnan(oracle.bbox_overlap_ratio((0, 0, 0, 0), (0, 0, 0, 0))), "0/0 like the reference; NaN > 0.15 is false"


def test_keypoint_association_rules(oracle):
    boxes = [(10, 10, 100, 100), (60, 60, 100, 100), (300, 50, 80, 80), (305, 55, 60, 60), (500, 300, 50, 50)]
    # 0-1 overlap: IoU = 2500/17500 = 0.143 < 0.15 -> both stay; 2-3: 3600/6400 > 0.15 -> both flagged
    kp = np.array([[20, 20], [70, 70], [109.5, 20], [110.5, 20], [320, 70], [510, 310], [549.4, 349.4], [549.6, 310], [9.5, 10], [9.4, 10]], np.float32)
    a, inany, ov = oracle.associate_keypoints(kp, boxes)
    assert list(ov) == [0, 0, 1, 1, 0]
    #            in 0   in 0&1  x rounds to 110 (half to even: out)  111 -> out  overlapped boxes  box 4   549 in   550 out   9.5 -> 10 in   9 out
    assert list(a) == [0, -1, -1, -1, -1, 4, 4, -1, 0, -1]
    a2, inany2, _ = oracle.associate_keypoints(kp, boxes, enable_ground_height_scale=True)
    assert list(a2) == list(a) and list(inany2) == [1, 1, 0, 0, 1, 1, 1, 0, 1, 0], "inany also counts boxes flagged as overlapped"
    # order dependence of the flags: box 1 overlaps 0 and 2; once 0-1 are flagged, 1 is not tested against 2 any more
    chain = [(0, 0, 100, 100), (40, 0, 100, 100), (80, 0, 100, 100)]
    _, _, ovc = oracle.associate_keypoints(np.zeros((0, 2), np.float32), chain)
    assert list(ovc) == [1, 1, 0]


def _scene(rng, n_points=400, n_land=4, n_cand=6):
    votes = [dict() for _ in range(n_points)]
    land_pts = [rng.choice(n_points, 60, replace=False) for _ in range(n_land)]
    for o, pts in enumerate(land_pts):
        for p in pts:
            votes[int(p)][100 + o] = int(rng.integers(1, 4))
    cands = []
    for i in range(n_cand):
        if i % 3 == 2:
            pts = rng.choice(n_points, 30, replace=False)  # mostly unseen points: a new landmark
        else:
            base = land_pts[i % n_land]
            pts = np.concatenate([rng.choice(base, 25, replace=False), rng.choice(n_points, 10, replace=False)])
        cands.append(sorted(set(int(p) for p in pts)))
    return votes, cands


def test_associate_cuboids_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    for trial in range(20):
        votes, cands = _scene(rng)
        cand_id = [200 + i for i in range(len(cands))]
        land = [100, 101, 102, 103]; bad = [0, trial % 2, 0, 0]
        v1 = [dict(d) for d in votes]; v2 = [dict(d) for d in votes]
        bo1 = np.full(len(votes), -1, np.int32); mv1 = np.zeros(len(votes), np.int32)
        for p, d in enumerate(votes):
            for o, c in d.items():
                if c > mv1[p]:
                    bo1[p] = o; mv1[p] = c
        bo2, mv2 = bo1.copy(), mv1.copy()
        ra, rc = oracle.associate_cuboids(cand_id, cands, land, bad, v1, 10, bo1, mv1)
        ga, gc = objects.associate_cuboids(cand_id, cands, land, bad, v2, 10, bo2, mv2)
        assert np.array_equal(ra, ga) and np.array_equal(rc, gc)
        assert v1 == v2 and np.array_equal(bo1, bo2) and np.array_equal(mv1, mv2)
        assert rc.sum() >= 1 and (rc == 0).sum() >= 1, "the sample exercises both branches"


def test_new_landmark_is_visible_to_the_next_candidate(oracle):
    """A candidate that becomes a landmark joins LocalObjectsLandmarks before the next candidate is examined (Tracking.cc:1891-1893) and
    its points carry its vote (SetAsLandmark): an identical second candidate merges into it instead of creating another landmark."""
    votes = [dict() for _ in range(50)]
    pts = list(range(30))
    for f in (oracle.associate_cuboids, objects.associate_cuboids):
        v = [dict(d) for d in votes]
        a, c = f([7, 8], [pts, pts], [], [], v, 10)
        assert list(a) == [7, 7] and list(c) == [1, 0]
        assert all(v[p] == {7: 2} for p in pts)
    # exactly `thres` shared points is not enough (strictly greater)
    v = [dict(d) for d in votes]
    a, c = objects.associate_cuboids([7, 8], [list(range(10)), list(range(10))], [], [], v, 10)
    assert list(a) == [7, 8] and list(c) == [1, 1]
