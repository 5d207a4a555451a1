"""GPU parity: object bundle adjustment (residuals, reduced camera system, LM trajectory) vs the CPU oracle."""
import numpy as np
import pytest

from cube_slam_amd import _lib, synth
from cube_slam_amd.ba import BundleAdjuster

pytestmark = pytest.mark.gpu
REL = 1e-5  # BASELINE.json north_star: BA residuals within 1e-5 relative


@pytest.fixture(scope="module")
def small():
    return synth.ba_problem(5, n_kf=40, n_points=1500, n_cuboids=8)


def test_residuals_and_chi2(ctx, oracle, small):
    ba = BundleAdjuster(small, ctx=ctx)
    chi, eo, ec, ep = ba.errors()
    rchi, reo, rec, rep = oracle.ba_errors(small)
    assert np.allclose(eo, reo, rtol=1e-12, atol=1e-12) and np.allclose(ec, rec, rtol=1e-11, atol=1e-11) and np.allclose(ep, rep, rtol=1e-11, atol=1e-13)
    assert abs(chi - rchi) <= 1e-11 * rchi
    ba.close()


def test_reduced_camera_system(ctx, oracle, small):
    ba = BundleAdjuster(small, ctx=ctx)
    lam = 3.7
    H, b = ba.reduced_dense(lam)
    rH, rb = oracle.ba_reduced_dense(small, 0, len(small["points"]), True, lam)
    # numeric (delta 1e-9) Jacobians of the cuboid edges carry ~1e-7 relative noise from the last bits of sin/cos
    scale = np.abs(rH).max()
    assert np.abs(H - rH).max() <= 2e-6 * scale
    assert np.abs(b - rb).max() <= 2e-6 * np.abs(rb).max()
    cams = 6 * int((1 - small["cam_fixed"]).sum())
    # camera-camera blocks touched only by reprojection edges: tight
    touched = np.zeros(len(small["cam_pose"]), bool); touched[small["cobs_cam"]] = True
    idx = np.cumsum(1 - small["cam_fixed"]) - 1
    free = [int(idx[i]) for i in range(len(touched)) if not touched[i] and not small["cam_fixed"][i]]
    for i in free[:10]:
        blk = slice(6 * i, 6 * i + 6)
        assert np.allclose(H[blk, :cams], rH[blk, :cams], rtol=1e-10, atol=1e-9 * scale)
    ba.close()


def test_build_ahead_equals_plain_order_with_rejected_trials(ctx, monkeypatch):
    """cs_ba_optimize enqueues the next iteration's system behind a trial's residual kernels before it knows the trial's verdict; a rejected trial rebuilds residuals and system from the
    restored estimates.  Same bits as the plain order (CUBESLAM_BA_AHEAD=0) -- also on graphs whose LM rejects trials (the points far off, the step overshoots)."""
    rejected = 0
    for seed, noise in ((3, 1.0), (4, 25.0), (6, 60.0)):
        d = dict(synth.ba_problem(seed, n_kf=30, n_points=900, n_cuboids=6, noise_px=noise))
        rng = np.random.default_rng(seed)
        d["points"] = d["points"] + rng.normal(0, 0.02 * noise, d["points"].shape)
        out = []
        for ahead in ("1", "0"):
            monkeypatch.setenv("CUBESLAM_BA_AHEAD", ahead)
            ba = BundleAdjuster(d, ctx=ctx)
            st = ba.optimize(12)
            out.append((st, ba.read()))
            ba.close()
        (sa, ra), (sb, rb) = out
        assert sa["iterations"] == sb["iterations"] and sa["lm_trials"] == sb["lm_trials"] and sa["chi2_trace"] == sb["chi2_trace"]
        assert all(x.tobytes() == y.tobytes() for x, y in zip(ra, rb))
        rejected += sa["lm_trials"] - sa["iterations"]
    assert rejected > 0, "no graph of this test made the LM reject a trial"


@pytest.mark.parametrize("kw", [dict(n_kf=40, n_points=1500, n_cuboids=8), dict(n_kf=25, n_points=800, n_cuboids=0), dict(n_kf=60, n_points=2500, n_cuboids=15)])
def test_lm_trajectory(ctx, oracle, kw):
    d = synth.ba_problem(11, **kw)
    ba = BundleAdjuster(d, ctx=ctx)
    st = ba.optimize(10)
    cam, pts, cub = ba.read()
    rcam, rpts, rcub, rst = oracle.ba_optimize(d, 10)
    assert st["iterations"] == rst["iterations"] and st["lm_trials"] == rst["lm_trials"]
    # g2o differentiates the cuboid edges numerically with delta = 1e-9: a 1-ulp difference of the state (device vs host
    # sin/cos, or simply the reduced system factored in another order) becomes ~1e-7 relative noise in those Jacobians, which
    # wobbles the intermediate iterates (observed up to 2e-5) before both runs settle on the same minimum; without cuboid edges
    # the traces agree to 1e-12.  The reference does the same to itself: its own function text moves its results by 3e-6 when
    # only the order of its edges changes (tests/test_ref_graph_pins.py).
    assert np.allclose(st["chi2_trace"], rst["chi2_trace"], rtol=(1e-4 if kw["n_cuboids"] else 1e-9))
    assert abs(st["chi2_final"] - rst["chi2_final"]) <= REL * rst["chi2_final"]
    assert st["chi2_final"] < 0.1 * st["chi2_init"]
    tol = 1e-4 if kw["n_cuboids"] else 1e-8
    assert np.allclose(cam, rcam, rtol=tol, atol=tol) and np.allclose(pts, rpts, rtol=tol, atol=tol)
    if len(cub):
        assert np.allclose(cub, rcub, rtol=1e-4, atol=1e-4)
    ba.close()


def test_all_cameras_fixed_but_one_and_stop_flag(ctx, oracle):
    import ctypes as C
    d = synth.ba_problem(3, n_kf=12, n_points=400, n_cuboids=3)
    d["cam_fixed"][:6] = 1
    ba = BundleAdjuster(d, ctx=ctx)
    st = ba.optimize(5)
    _, _, _, rst = oracle.ba_optimize(d, 5)
    assert np.allclose(st["chi2_trace"], rst["chi2_trace"], rtol=1e-4)
    stop = C.c_int(1)
    st2 = ba.optimize(5, C.byref(stop))  # forceStopFlag already raised: no iteration runs
    assert st2["iterations"] == 0
    ba.close()


def test_stop_flag_raised_by_another_thread_ends_the_solve(ctx):
    """g2o polls the caller's `bool *pbStopFlag` between iterations and LM trials (sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149;
    LocalMapping::InterruptBA raises it from another thread).  cs_ba_set_stop_flag_bool hands the library that byte itself: a flag raised while
    cs_ba_optimize runs ends it early, with the estimates of the last finished iteration."""
    import ctypes as C
    import threading
    import time
    d = synth.ba_problem(11, n_kf=200, n_points=20000, n_cuboids=50)
    ba = BundleAdjuster(d, ctx=ctx)
    ba.optimize(1)  # (kernels paged in)
    flag = C.c_ubyte(0)
    ba.set_stop_flag_bool(flag)
    t0 = time.perf_counter()
    full = ba.optimize(3)
    per_it = (time.perf_counter() - t0) / max(full["iterations"], 1)
    assert full["iterations"] == 3

    def raise_later():
        time.sleep(max(4 * per_it, 0.002))
        flag.value = 1
    th = threading.Thread(target=raise_later)
    th.start()
    st = ba.optimize(4000)  # (seconds if the flag were ignored)
    th.join()
    assert 0 < st["iterations"] < 4000, st["iterations"]
    assert ba.optimize(5)["iterations"] == 0  # still raised: nothing runs (Optimizer.cc:1386-1388)
    flag.value = 0
    assert ba.optimize(2)["iterations"] >= 1
    ba.set_stop_flag_bool(None)
    ba.close()


def test_band_and_sparse_solvers_agree(ctx, small, monkeypatch):
    """The two reduced-solve paths (cuboid elimination + LDS-window block-band Cholesky; minimum-degree sparse block Cholesky)
    solve the same damped system: identical LM decisions, chi2 within round-off."""
    res = {}
    for solver in ("band", "sparse"):
        monkeypatch.setenv("CUBESLAM_BA_SOLVER", solver)
        ba = BundleAdjuster(small, ctx=ctx)
        st = ba.optimize(6)
        cam, pts, cub = ba.read()
        res[solver] = (st, cam, pts, cub)
        ba.close()
    a, b = res["band"], res["sparse"]
    assert a[0]["iterations"] == b[0]["iterations"] and a[0]["lm_trials"] == b[0]["lm_trials"]
    assert np.allclose(a[0]["chi2_trace"], b[0]["chi2_trace"], rtol=1e-4)
    assert abs(a[0]["chi2_final"] - b[0]["chi2_final"]) <= 1e-5 * b[0]["chi2_final"]
    assert np.allclose(a[1], b[1], rtol=0, atol=1e-4) and np.allclose(a[3], b[3], rtol=0, atol=1e-4)
    # a graph whose cameras are not narrow-banded (one landmark seen by the first and the last keyframe) takes the sparse path
    wide = dict(small)
    far = int(np.nonzero(1 - np.asarray(small["cam_fixed"]))[0][-1])
    wide["obs_cam"] = np.concatenate([small["obs_cam"], [1, far]]).astype(np.int32)
    wide["obs_point"] = np.concatenate([small["obs_point"], [0, 0]]).astype(np.int32)
    wide["obs_uv"] = np.concatenate([small["obs_uv"], [[600.0, 170.0], [610.0, 171.0]]])
    wide["obs_inv_sigma2"] = np.concatenate([small["obs_inv_sigma2"], [1.0, 1.0]])
    monkeypatch.setenv("CUBESLAM_BA_SOLVER", "band")
    with pytest.raises(Exception):
        BundleAdjuster(wide, ctx=ctx)
    monkeypatch.delenv("CUBESLAM_BA_SOLVER")
    ba = BundleAdjuster(wide, ctx=ctx)
    st = ba.optimize(2)
    assert st["chi2_final"] <= st["chi2_init"]
    ba.close()


def test_allreduce_callback_on_device_pointer(ctx, small):
    """The multi-GPU exchange of bench.py on one GPU: a 1-rank RCCL group all-reduces the reduced camera system in place through
    the device pointer the library hands to the callback (zero-copy torch view); with one rank the sum is the identity, so the
    result must equal the run without a callback."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

    def allreduce(ptr, n):
        t = torch.as_tensor(_Dev(ptr, n), device="cuda")
        assert t.data_ptr() == ptr and t.dtype == torch.float64
        dist.all_reduce(t)
        torch.cuda.synchronize()
        calls.append(n)

    try:
        # rank 0 of a 2-rank job owns the landmarks [0, L/2) and all pose edges; with a 1-rank group its partial sums are the
        # whole sum, i.e. the problem restricted to those landmarks -- which a plain single-rank solver must reproduce
        from cube_slam_amd.ba import shard_landmarks
        lo, hi = shard_landmarks(len(small["points"]), 0, 2)
        keep = small["obs_point"] < hi
        half = dict(small)
        half["points"] = small["points"][:hi]
        for k in ("obs_cam", "obs_point", "obs_uv", "obs_inv_sigma2"):
            half[k] = small[k][keep]
        ref = BundleAdjuster(half, ctx=ctx)
        st_ref = ref.optimize(3); cam_ref, pts_ref, cub_ref = ref.read(); ref.close()
        ba = BundleAdjuster(small, ctx=ctx, rank=0, world=2, allreduce=allreduce)
        st = ba.optimize(3)
        cam, pts, cub = ba.read()
        ba.close()
        assert len(calls) >= 3 and st["iterations"] == st_ref["iterations"]
        assert np.allclose(st["chi2_trace"], st_ref["chi2_trace"], rtol=1e-9)
        assert np.allclose(cam, cam_ref, rtol=0, atol=1e-9) and np.allclose(pts[:hi], pts_ref, rtol=0, atol=1e-9) and np.allclose(cub, cub_ref, rtol=0, atol=1e-9)
    finally:
        if created:
            dist.destroy_process_group()


def test_two_ranks_on_one_gpu_sum_to_the_single_rank_solve(small):
    """The sharded BA with world = 2 on ONE GPU: two cs_ba instances (rank 0 and rank 1, each with its own context = stream) driven in lock-step by two
    threads; their cs_ba_set_allreduce callbacks meet at a barrier and add the two device buffers in place (what ncclAllReduce(ncclSum) does between two
    GPUs).  Proves on hardware that the shard systems sum to the single-rank system: same LM trace, same estimates on both ranks as one rank alone.
    (tests/test_rccl_gpu.py runs the same split over RCCL when two GPUs are visible.)"""
    import threading

    import torch

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

    one = BundleAdjuster(small, ctx=_lib.Context(0))
    st_one = one.optimize(4); cam_one, pts_one, cub_one = one.read(); one.close()

    meet = threading.Barrier(2, timeout=60)
    slot = [None, None]
    n_calls = [0, 0]

    def make_allreduce(rank):
        def allreduce(ptr, n):  # the library drained its stream before the call
            slot[rank] = torch.as_tensor(_Dev(ptr, n), device="cuda")
            meet.wait()
            if rank == 0:
                total = slot[0] + slot[1]
                slot[0].copy_(total); slot[1].copy_(total)
                torch.cuda.synchronize()
            meet.wait()
            n_calls[rank] += 1
        return allreduce

    out = [None, None]
    err = []

    def run(rank):
        try:
            ba = BundleAdjuster(small, ctx=_lib.Context(0), rank=rank, world=2, allreduce=make_allreduce(rank))
            st = ba.optimize(4)
            out[rank] = (st, *ba.read())
            ba.close()
        except Exception as e:  # pragma: no cover
            err.append(e)
            meet.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not err, err
    assert n_calls[0] == n_calls[1] and n_calls[0] >= 4
    (st0, cam0, pts0, cub0), (st1, cam1, pts1, cub1) = out
    assert st0["iterations"] == st1["iterations"] == st_one["iterations"] and st0["lm_trials"] == st_one["lm_trials"]
    assert st0["chi2_trace"] == st1["chi2_trace"], "the ranks hold the same sums bit for bit"
    # one rank alone adds the same terms in another order (per-rank partial sums first); the numeric Jacobians (delta = 1e-9) carry that round-off
    # into the iterates: first chi2 to 1e-14, the fourth to 1e-8 -- the level the reference's own text moves at when its heap layout changes
    assert abs(st0["chi2_trace"][0] - st_one["chi2_trace"][0]) <= 1e-12 * st_one["chi2_trace"][0]
    assert np.allclose(st0["chi2_trace"], st_one["chi2_trace"], rtol=1e-6)
    assert np.array_equal(cam0, cam1) and np.array_equal(cub0, cub1)
    assert np.allclose(cam0, cam_one, rtol=0, atol=1e-5) and np.allclose(cub0, cub_one, rtol=0, atol=1e-5)
    from cube_slam_amd.ba import shard_landmarks
    lo1, hi1 = shard_landmarks(len(small["points"]), 1, 2)
    assert np.allclose(pts0[:lo1], pts_one[:lo1], rtol=0, atol=1e-5) and np.allclose(pts1[lo1:hi1], pts_one[lo1:hi1], rtol=0, atol=1e-5), "every rank owns its landmarks' estimates"


def test_stereo_edges_match_oracle(ctx, oracle):
    """Mixed EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ graph (Optimizer.cc:120-184): residuals incl. the float reciprocal of
    the stereo projection, the reduced system and the LM trajectory."""
    d = synth.ba_problem(21, n_kf=30, n_points=1200, n_cuboids=0, stereo_frac=0.6)
    assert (d["obs_ur"] >= 0).sum() > 1000 and (d["obs_ur"] < 0).sum() > 1000
    ba = BundleAdjuster(d, ctx=ctx)
    chi, eo, _, _ = ba.errors()
    rchi, reo, _, _ = oracle.ba_errors(d)
    assert eo.shape == reo.shape == (len(d["obs_cam"]), 3)
    assert np.array_equal(eo[:, 2] == 0, d["obs_ur"] < 0)
    assert np.allclose(eo, reo, rtol=1e-12, atol=1e-11) and abs(chi - rchi) <= 1e-11 * rchi
    H, b = ba.reduced_dense(2.5)
    rH, rb = oracle.ba_reduced_dense(d, 0, len(d["points"]), True, 2.5)
    assert np.allclose(H, rH, rtol=1e-9, atol=1e-10 * np.abs(rH).max()) and np.allclose(b, rb, rtol=1e-9, atol=1e-10 * np.abs(rb).max())
    st = ba.optimize(10)
    cam, pts, _ = ba.read()
    rcam, rpts, _, rst = oracle.ba_optimize(d, 10)
    assert st["iterations"] == rst["iterations"] and st["lm_trials"] == rst["lm_trials"]
    assert np.allclose(st["chi2_trace"], rst["chi2_trace"], rtol=1e-8)
    assert abs(st["chi2_final"] - rst["chi2_final"]) <= REL * rst["chi2_final"] and st["chi2_final"] < 0.1 * st["chi2_init"]
    assert np.allclose(cam, rcam, rtol=0, atol=1e-7) and np.allclose(pts, rpts, rtol=0, atol=1e-6)
    ba.close()


def test_all_mono_ur_array_is_the_mono_problem(ctx, small):
    """obs_ur = all -1 (no stereo match anywhere) must reproduce the monocular run bit for bit."""
    m = dict(small); m["obs_ur"] = np.full(len(small["obs_cam"]), -1.0); m["bf"] = 386.1448; m["huber_stereo"] = 7.815 ** 0.5
    a = BundleAdjuster(small, ctx=ctx); b = BundleAdjuster(m, ctx=ctx)
    sa, sb = a.optimize(5), b.optimize(5)
    assert sa["chi2_trace"] == sb["chi2_trace"]
    assert all(np.array_equal(x, y) for x, y in zip(a.read(), b.read()))
    a.close(); b.close()


def test_two_sided_band_solver_agrees_with_one_sided(ctx, monkeypatch):
    """Long chains are eliminated from both ends at once (ba_band_twist_factor / ba_band_mid / ba_band_twist_back); same damped
    system as the one-workgroup band solver and the sparse solver: identical LM decisions, chi2 within round-off."""
    d = synth.ba_problem(31, n_kf=160, n_points=6000, n_cuboids=30)
    res = {}
    for solver in ("band", "band1", "sparse", "cr", None):
        if solver is None:
            monkeypatch.delenv("CUBESLAM_BA_SOLVER")
        else:
            monkeypatch.setenv("CUBESLAM_BA_SOLVER", solver)
        ctx.timing(True); ctx.timing_reset()
        ba = BundleAdjuster(d, ctx=ctx)
        st = ba.optimize(6)
        res[solver] = (st, ba.read(), ctx.timing_get("ba_band_twist_factor")[1], ctx.timing_get("ba_band_chol")[1], ctx.timing_get("ba_cr_eliminate")[1])
        ctx.timing(False)
        ba.close()
    assert res["band"][2] > 0 and res["band"][3] == 0 and res["band"][4] == 0, "CUBESLAM_BA_SOLVER=band: the two-sided chain"
    assert res["band1"][2] == 0 and res["band1"][3] > 0
    assert res["cr"][4] > 0 and res["cr"][2] == 0 and res[None][4] > 0, "nested dissection (ba_cr.hip) is the default of a 160-keyframe chain"
    a = res["band1"]
    for other in ("band", "sparse", "cr"):
        b = res[other]
        assert a[0]["iterations"] == b[0]["iterations"] and a[0]["lm_trials"] == b[0]["lm_trials"]
        assert np.allclose(a[0]["chi2_trace"], b[0]["chi2_trace"], rtol=1e-4 if other == "sparse" else 1e-6)
        assert abs(a[0]["chi2_final"] - b[0]["chi2_final"]) <= 1e-6 * b[0]["chi2_final"]
        tol = 1e-4  # numeric-Jacobian noise of the cuboid edges (see test_lm_trajectory); the cuboid-free run below is tight
        assert np.abs(a[1][0] - b[1][0]).max() <= tol and np.abs(a[1][2] - b[1][2]).max() <= tol, (other, np.abs(a[1][0] - b[1][0]).max(), np.abs(a[1][2] - b[1][2]).max())
    # without cuboid edges every Jacobian is analytic: the two band variants must then agree to solver round-off
    d0 = synth.ba_problem(32, n_kf=160, n_points=6000, n_cuboids=0)
    out = {}
    for solver in ("band", "band1", "cr"):
        monkeypatch.setenv("CUBESLAM_BA_SOLVER", solver)
        ba = BundleAdjuster(d0, ctx=ctx)
        out[solver] = (ba.optimize(8), ba.read())
        ba.close()
    for other in ("band", "cr"):
        assert out[other][0]["lm_trials"] == out["band1"][0]["lm_trials"]
        assert np.allclose(out[other][0]["chi2_trace"], out["band1"][0]["chi2_trace"], rtol=1e-10)
        assert np.abs(out[other][1][0] - out["band1"][1][0]).max() <= 1e-8 and np.abs(out[other][1][1] - out["band1"][1][1]).max() <= 1e-7


@pytest.mark.parametrize("k_obs", [3, 5, 7, 9])
def test_nested_dissection_for_every_super_block_size(ctx, monkeypatch, k_obs):
    """ba_cr.hip is instantiated for super-blocks of 2, 4, 6, 8 and 10 cameras (the band half-width rounded up): chains whose landmarks are seen by
    k_obs consecutive key frames have half-width k_obs - 1; a chain length that is not a multiple of the super-block exercises the identity padding."""
    d = synth.ba_problem(60 + k_obs, n_kf=163, n_points=5000, n_cuboids=0, k_obs=k_obs)
    out = {}
    for solver in ("band1", "cr"):
        monkeypatch.setenv("CUBESLAM_BA_SOLVER", solver)
        ctx.timing(True); ctx.timing_reset()
        ba = BundleAdjuster(d, ctx=ctx)
        st = ba.optimize(5)
        out[solver] = (st, ba.read(), ctx.timing_get("ba_cr_eliminate")[1])
        ctx.timing(False)
        ba.close()
    assert out["cr"][2] > 0 and out["band1"][2] == 0
    assert out["cr"][0]["lm_trials"] == out["band1"][0]["lm_trials"]
    assert np.allclose(out["cr"][0]["chi2_trace"], out["band1"][0]["chi2_trace"], rtol=1e-10)
    assert np.abs(out["cr"][1][0] - out["band1"][1][0]).max() <= 1e-8 and np.abs(out["cr"][1][1] - out["band1"][1][1]).max() <= 1e-7
