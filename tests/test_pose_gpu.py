"""GPU parity: batched Optimizer::PoseOptimization (one workgroup per frame) vs the CPU oracle."""
import numpy as np
import pytest

from cube_slam_amd import synth
from cube_slam_amd.optimizer import PoseOptimization

pytestmark = pytest.mark.gpu


def test_batch_matches_oracle(ctx, oracle):
    frames = [synth.pose_frame(10 + i, n=n, outlier_frac=of, stereo_frac=sf) for i, (n, of, sf) in enumerate(
        [(800, 0.15, 0.0), (2000, 0.3, 0.0), (300, 0.05, 0.5), (64, 0.2, 1.0), (9, 0.0, 0.0), (2, 0.0, 0.0), (0, 0.0, 0.0), (1200, 0.5, 0.2)])]
    got = PoseOptimization(frames, ctx=ctx)
    for fr, (pose, flags, ninl) in zip(frames, got):
        rp, rf, rn = oracle.pose_optimization(fr["Xw"], fr["obs"], fr["inv_sigma2"], fr["intr"], fr["pose"])
        assert np.allclose(pose, rp, rtol=0, atol=1e-8), "pose (BASELINE tolerance 1e-5 relative on residuals; observed ~1e-12)"
        assert ninl == rn
        # a chi2 sitting on the threshold within round-off may flip: allow at most one flag per thousand
        assert (flags != rf).sum() <= max(0, len(rf) // 1000)


def test_many_frames(ctx, oracle):
    frames = [synth.pose_frame(100 + i, n=500 + 37 * (i % 7), outlier_frac=0.1 + 0.05 * (i % 4)) for i in range(64)]
    got = PoseOptimization(frames, ctx=ctx)
    for i in (0, 13, 63):
        fr = frames[i]
        rp, rf, rn = oracle.pose_optimization(fr["Xw"], fr["obs"], fr["inv_sigma2"], fr["intr"], fr["pose"])
        assert np.allclose(got[i][0], rp, rtol=0, atol=1e-8) and got[i][2] == rn
    assert all(np.linalg.norm(g[0][:3] - f["pose_true"][:3]) < 0.1 for g, f in zip(got, frames))
