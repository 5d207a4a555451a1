"""9-dof g2o::cuboid math of object_slam: CPU known answers for the oracle, GPU parity."""
import numpy as np
import pytest

from cube_slam_amd import synth


def _q(axis, a):
    v = np.zeros(4); v[axis] = np.sin(a / 2); v[3] = np.cos(a / 2)
    return v


def _cub(t, yaw, scale):
    return np.concatenate([t, _q(2, yaw), scale])


def _cases(n=200, seed=0):
    rng = np.random.default_rng(seed)
    T = np.zeros((n, 7)); G = np.zeros((n, 10)); M = np.zeros((n, 10))
    for i in range(n):
        R = synth._rot(1, rng.uniform(-1, 1)) @ synth._rot(0, rng.uniform(-0.3, 0.3)) @ synth._rot(2, rng.uniform(-0.3, 0.3))
        T[i] = synth._pose7(R, rng.uniform(-2, 2, 3))
        Ro = synth._rot(2, rng.uniform(-np.pi, np.pi)) @ synth._rot(0, rng.normal(0, 0.05))
        G[i] = np.concatenate([synth._pose7(Ro, rng.uniform(-5, 5, 3)), rng.uniform(0.2, 2.0, 3)])
        # measurement: the global cuboid seen from the camera (cam frame), perturbed, possibly with another front face
        Rcw, tcw = R, T[i, :3]
        k = rng.integers(0, 4)
        Rl = Rcw @ Ro @ synth._rot(2, k * np.pi / 2 + rng.normal(0, 0.05))
        tl = Rcw @ G[i, :3] + tcw + rng.normal(0, 0.05, 3)
        sc = G[i, 7:].copy()
        if k % 2:
            sc[[0, 1]] = sc[[1, 0]]
        M[i] = np.concatenate([synth._pose7(Rl, tl), sc + rng.normal(0, 0.02, 3)])
    return T, G, M


def test_oracle_known_answers(oracle):
    # identical cuboid seen from an identity camera: zero error; a 90-degree front-face change with swapped scales is still zero
    T = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    g = _cub([1.0, 2.0, 3.0], 0.3, [0.5, 0.8, 1.1])[None]
    assert np.allclose(oracle.cuboid9_edge_linearize(T, g, g, jac=False), 0, atol=1e-12)
    m = _cub([1.0, 2.0, 3.0], 0.3 + np.pi / 2, [0.8, 0.5, 1.1])[None]
    assert np.allclose(oracle.cuboid9_edge_linearize(T, g, m, jac=False), 0, atol=1e-9)
    # pure translation offset: upsilon = R_meas^T (t_global - t_meas), scale difference in the tail
    m2 = _cub([1.1, 2.0, 3.0], 0.3, [0.4, 0.8, 1.1])[None]
    e = oracle.cuboid9_edge_linearize(T, g, m2, jac=False)[0]
    c, s = np.cos(0.3), np.sin(0.3)
    assert np.allclose(e[:3], 0, atol=1e-12) and np.allclose(e[3:6], [-0.1 * c, 0.1 * s, 0], atol=1e-12) and np.allclose(e[6:], [0.1, 0, 0], atol=1e-12)
    # oplus: pose * exp(update), scale + update
    out = oracle.cuboid9_oplus(g, np.array([[0, 0, 0.2, 0, 0, 0, 0.1, -0.1, 0.0]]))[0]
    assert np.allclose(out[:3], [1, 2, 3]) and np.allclose(out[3:7], _q(2, 0.5), atol=1e-12) and np.allclose(out[7:], [0.6, 0.7, 1.1])
    # numeric Jacobian wrt the scale part of the cuboid is the identity block
    _, Jc, Jq = oracle.cuboid9_edge_linearize(T, g, m2)
    assert np.allclose(Jq[0][6:, 6:], np.eye(3), atol=1e-5) and np.allclose(Jq[0][:6, 6:], 0, atol=1e-5)


@pytest.mark.gpu
def test_gpu_matches_oracle(ctx, oracle):
    from cube_slam_amd.optimizer import cuboid9_edge_linearize, cuboid9_oplus
    T, G, M = _cases(300, 1)
    err, Jc, Jq = cuboid9_edge_linearize(T, G, M, ctx=ctx)
    rerr, rJc, rJq = oracle.cuboid9_edge_linearize(T, G, M)
    assert np.allclose(err, rerr, rtol=1e-9, atol=1e-11)
    # delta = 1e-9 differences amplify 1-ulp differences of acos / tan / sin / cos by 1e9 * 1e-16: compare at 1e-5 relative of the scale
    assert np.abs(Jc - rJc).max() <= 2e-5 * max(1.0, np.abs(rJc).max()) and np.abs(Jq - rJq).max() <= 2e-5 * max(1.0, np.abs(rJq).max())
    upd = np.random.default_rng(2).normal(0, 0.1, (300, 9))
    assert np.allclose(cuboid9_oplus(G, upd, ctx=ctx), oracle.cuboid9_oplus(G, upd), rtol=0, atol=1e-12)
    assert np.array_equal(cuboid9_edge_linearize(T, G, M, jac=False, ctx=ctx), err)
