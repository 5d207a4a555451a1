"""GPU parity: batched keypoint -> cuboid association (cs_associate_keypoints) vs the oracle restatement of Tracking::DetectCuboid :1717-1775."""
import numpy as np
import pytest

from cube_slam_amd import objects

pytestmark = pytest.mark.gpu


def _frame(rng, n_kp, n_box, W=640, H=480):
    boxes = np.stack([rng.integers(0, W - 60, n_box), rng.integers(0, H - 60, n_box), rng.integers(20, 260, n_box), rng.integers(20, 260, n_box)], axis=1).astype(np.int32)
    kp = np.stack([rng.uniform(0, W, n_kp), rng.uniform(0, H, n_kp)], axis=1).astype(np.float32)
    # exercise the rounding rule: some keypoints exactly on x.5 next to box borders
    for b in boxes[: min(n_box, 4) if n_kp >= 2 else 0]:
        i = rng.integers(0, n_kp)
        kp[i] = (b[0] - 0.5, b[1] + 1); kp[(i + 1) % n_kp] = (b[0] + b[2] - 0.5, b[1] + b[3] - 0.5)
    return kp, boxes


@pytest.mark.parametrize("mode", [False, True])
def test_keypoint_association_matches_oracle(ctx, oracle, mode):
    rng = np.random.default_rng(11)
    kps, bxs = [], []
    for f in range(12):
        kp, bx = _frame(rng, int(rng.integers(0, 1500)) if f else 0, int(rng.integers(0, 9)))
        kps.append(kp); bxs.append(bx)
    kps.append(np.zeros((0, 2), np.float32)); bxs.append(np.zeros((0, 4), np.int32))
    got = objects.associate_keypoints(ctx, kps, bxs, mode)
    n_assoc = 0
    for (a, inany, ov), kp, bx in zip(got, kps, bxs):
        ra, ri, ro = oracle.associate_keypoints(kp, bx, mode)
        assert np.array_equal(a, ra) and np.array_equal(ov, ro)
        if mode:
            assert np.array_equal(inany, ri)
        n_assoc += int((ra >= 0).sum())
    assert n_assoc > 100


def test_full_batch_size(ctx, oracle):
    """128 keyframes x 2000 keypoints x 8 boxes (the benchmark's batch): every keypoint either unassociated or inside exactly the box it names."""
    rng = np.random.default_rng(5)
    kps, bxs = zip(*[_frame(rng, 2000, 8) for _ in range(128)])
    got = objects.associate_keypoints(ctx, list(kps), list(bxs))
    for (a, _, ov), kp, bx in zip(got, kps, bxs):
        px, py = np.rint(kp[:, 0]).astype(int), np.rint(kp[:, 1]).astype(int)
        inside = (bx[None, :, 0] <= px[:, None]) & (px[:, None] < bx[None, :, 0] + bx[None, :, 2]) & (bx[None, :, 1] <= py[:, None]) & (py[:, None] < bx[None, :, 1] + bx[None, :, 3])
        inside &= (ov == 0)[None, :]
        cnt = inside.sum(1)
        assert np.array_equal(a >= 0, cnt == 1)
        sel = a >= 0
        assert inside[np.nonzero(sel)[0], a[sel]].all()
    ra, _, ro = oracle.associate_keypoints(kps[0], bxs[0])
    assert np.array_equal(got[0][0], ra) and np.array_equal(got[0][2], ro)
