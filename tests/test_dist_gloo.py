"""world_size-2 tests of the sharded paths on CPU (gloo).  The compute inside each rank is the CPU oracle (there is no GPU
here); what is under test is the partitioning / exchange logic the GPU ranks use: frame blocks with host-side gathering, and the
all-reduce of per-shard reduced camera systems in the object BA."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from cube_slam_amd import shard, synth
from cube_slam_amd.ba import shard_landmarks


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _frames_worker(rank, world, port, n_frames, q):
    _init(rank, world, port)
    from oracle import pyoracle as po
    lo, hi = shard.frame_block(n_frames, rank, world)
    local = []
    for f in range(lo, hi):
        s = synth.cuboid_scene(4000 + f)
        cubs, _ = po.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=po.cuboid_opts())
        local.append(np.concatenate([c for c in cubs if len(c)]).tobytes() if any(len(c) for c in cubs) else b"")
    allr = shard.gather_frame_results(local, n_frames)
    if rank == 0:
        q.put(allr)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_blocks_cover_everything():
    for n in (0, 1, 7, 512):
        for world in (1, 2, 3, 8):
            blocks = [shard.frame_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.frame_block(4, 2, 2)


def test_sharded_frames_equal_serial(oracle):
    n_frames, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_frames_worker, args=(world, _free_port(), n_frames, q), nprocs=world, join=True)
    got = q.get()
    for f in range(n_frames):
        s = synth.cuboid_scene(4000 + f)
        cubs, _ = oracle.detect_cuboid(s["gray"], s["K"], s["Twc"], s["boxes"], s["lines"], opts=oracle.cuboid_opts())
        ref = np.concatenate([c for c in cubs if len(c)]).tobytes() if any(len(c) for c in cubs) else b""
        assert got[f] == ref
    assert any(len(g) for g in got)


def _ba_worker(rank, world, port, q):
    _init(rank, world, port)
    from oracle import pyoracle as po
    d = synth.ba_problem(11, n_kf=7, n_points=160, n_cuboids=2)
    lo, hi = shard_landmarks(len(d["points"]), rank, world)
    lam = 0.25
    H, b = po.ba_reduced_dense(d, lo, hi, rank == 0, lam)  # pose edges and the pose damping live on rank 0
    buf = np.concatenate([H.reshape(-1), b])
    shard.allreduce_sum_f64(buf)
    n = len(b)
    Hs, bs = buf[:n * n].reshape(n, n), buf[n * n:]
    dx = np.linalg.solve(Hs, bs)  # every rank solves the same reduced system redundantly
    q.put((rank, Hs.tobytes(), dx.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def _ba8_worker(rank, world, port, q):
    _init(rank, world, port)
    from oracle import pyoracle as po
    d = synth.ba_problem(12, n_kf=24, n_points=1200, n_cuboids=4)
    lo, hi = shard_landmarks(len(d["points"]), rank, world)
    H, b = po.ba_reduced_dense(d, lo, hi, rank == 0, 0.5)
    buf = np.concatenate([H.reshape(-1), b])
    shard.allreduce_sum_f64(buf)
    import hashlib
    if rank == 0:
        np.save(os.path.join(os.environ["CS_TEST_TMP"], "reduced8.npy"), buf)
    q.put((rank, hi - lo, hashlib.sha256(buf.tobytes()).hexdigest()))  # (a digest: eight systems do not fit the queue's pipe before the parent reads)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_config5_sizes_and_allreduce_bit_identical(oracle, tmp_path, monkeypatch):
    """What the first 8-GPU run of config 5 will meet, without the GPUs: (1) the landmark and frame partitions at config 5's sizes (2 000 key frames, 100 k points, 500 cuboids; 512
    frames) give every one of 8 ranks work, cover everything exactly once and differ by at most one unit; (2) eight gloo ranks all-reduce their shards' reduced camera
    systems of a mid-sized graph: the eight results are bit-identical and equal the unsharded system."""
    for n, world in ((100000, 8), (100000, 7), (512, 8), (2000, 8)):
        blocks = [shard_landmarks(n, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
        fb = [shard.frame_block(n, r, world) for r in range(world)]
        assert fb[0][0] == 0 and fb[-1][1] == n and all(fb[r][1] == fb[r + 1][0] for r in range(world - 1)) and min(hi - lo for lo, hi in fb) >= 1
    world = 8
    monkeypatch.setenv("CS_TEST_TMP", str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_ba8_worker, args=(world, _free_port(), q), nprocs=world, join=True)
    res = sorted([q.get() for _ in range(world)])
    assert all(r[1] >= 1 for r in res), "a rank without landmarks"
    assert all(r[2] == res[0][2] for r in res), "the ranks' all-reduced systems differ"
    d = synth.ba_problem(12, n_kf=24, n_points=1200, n_cuboids=4)
    H, b = oracle.ba_reduced_dense(d, 0, len(d["points"]), True, 0.5)
    got = np.load(os.path.join(str(tmp_path), "reduced8.npy"))
    assert np.allclose(got[:H.size].reshape(H.shape), H, rtol=1e-10, atol=1e-9 * np.abs(H).max()) and np.allclose(got[H.size:], b, rtol=1e-10, atol=1e-9 * np.abs(b).max())


def test_ba_allreduce_of_shard_systems_equals_full(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_ba_worker, args=(world, _free_port(), q), nprocs=world, join=True)
    res = sorted([q.get() for _ in range(world)])
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2], "ranks hold bit-identical reduced systems and steps"
    d = synth.ba_problem(11, n_kf=7, n_points=160, n_cuboids=2)
    H, b = oracle.ba_reduced_dense(d, 0, len(d["points"]), True, 0.25)
    Hs = np.frombuffer(res[0][1]).reshape(H.shape)
    assert np.allclose(Hs, H, rtol=1e-10, atol=1e-9 * np.abs(H).max())
    dx = np.frombuffer(res[0][2])
    assert np.allclose(dx, np.linalg.solve(H, b), rtol=1e-6, atol=1e-9)
