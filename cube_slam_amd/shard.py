"""Sharding helpers of the multi-GPU paths (one process per GPU, torch.distributed; DESIGN.md section "Multi-GPU").

Front-end (ORB / LSD+LBD / cuboid): frames are independent units -> contiguous block of frames per rank, no data-path collective;
the fixed-size result records are gathered on the host only when the caller wants the whole batch in one place.
Object BA: landmarks (and their observation edges) are sharded by landmark id (cube_slam_amd.ba.shard_landmarks), poses and
cuboids replicated; the ranks exchange the reduced camera system with one all-reduce per LM trial.
"""
import numpy as np


def frame_block(n_frames, rank, world):
    """[begin, end) of the frames owned by `rank`: contiguous blocks whose sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return n_frames * rank // world, n_frames * (rank + 1) // world


def gather_frame_results(local_results, n_frames, group=None):
    """Concatenate per-frame results (any picklable per-frame objects) of all ranks in frame order.
    Every rank passes the list for its own frame_block; every rank gets the full list (len n_frames)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        assert len(local_results) == n_frames
        return list(local_results)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = frame_block(n_frames, rank, world)
    if len(local_results) != hi - lo:
        raise ValueError("rank %d owns frames [%d, %d) but passed %d results" % (rank, lo, hi, len(local_results)))
    parts = [None] * world
    dist.all_gather_object(parts, list(local_results), group=group)
    out = [x for p in parts for x in p]
    assert len(out) == n_frames
    return out


def allreduce_sum_f64(array, group=None):
    """In-place sum over ranks of a float64 numpy array (host staging path used by the CPU tests; on GPUs the BA solver hands the
    device pointer of its reduced system to an RCCL all-reduce instead, see bench.py)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(array))
    dist.all_reduce(t, group=group)
    array[...] = t.numpy()
    return array


def comm_init_from_process_group(ctx, rank, world, device=None, group=None):
    """The library's own RCCL communicator of this rank (cs_comm_init) from a torch.distributed process group that is already up: rank 0 makes the ncclUniqueId, a broadcast of
    128 bytes carries it (on `device` when the group's backend is nccl, on the host for gloo).  bench.py --gpus N and tests/test_rccl_gpu.py's two-rank worker both come through
    here, so the first multi-GPU run of the bench exercises nothing the test has not."""
    import torch
    import torch.distributed as dist
    from cube_slam_amd import _lib
    uid = torch.zeros(128, dtype=torch.uint8, device=device if device is not None else "cpu")
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(_lib.Context.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, 0, group=group)
    ctx.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
