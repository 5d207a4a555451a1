"""RCCL inside the library (cs_comm_*): the communicator is created from an ncclUniqueId on the context's own stream; a sharded cs_ba without a
callback all-reduces its reduced camera system through it.  One GPU: a single-rank communicator (dlopen + init + ncclAllReduce on our stream).
Two or more GPUs (skipped on a one-GPU box): two processes, landmarks sharded, the LM trace equals the single-GPU one."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_allreduce(ctx):
    import torch
    from cube_slam_amd import _lib
    c = _lib.Context(0)
    c.comm_init(0, 1, _lib.Context.comm_unique_id())
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    _lib.check(c.ptr, _lib.lib().cs_comm_allreduce_f64(c.ptr, C.c_void_p(x.data_ptr()), C.c_long(x.numel())), "cs_comm_allreduce_f64")
    c.sync()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    c.close()


WORKER = r'''
import json, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))  # (what bench.py --gpus N brings up)
from cube_slam_amd import _lib, shard, synth
from cube_slam_amd.ba import BundleAdjuster
ctx = _lib.Context(rank)
shard.comm_init_from_process_group(ctx, rank, world, device="cuda")  # bench.py's own call
d = synth.ba_problem(11, n_kf=40, n_points=3000, n_cuboids=6)
st = BundleAdjuster(d, ctx=ctx, rank=rank, world=world).optimize(4)
if rank == 0:
    print("RESULT " + json.dumps({"chi2": st["chi2_final"], "trace": list(st["chi2_trace"][:st["iterations"]]), "trials": st["lm_trials"]}))
'''


def test_two_ranks_sharded_ba_equals_one_gpu(ctx, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from cube_slam_amd import synth
    from cube_slam_amd.ba import BundleAdjuster
    d = synth.ba_problem(11, n_kf=40, n_points=3000, n_cuboids=6)
    ref = BundleAdjuster(d, ctx=ctx).optimize(4)
    w = tmp_path / "w.py"
    w.write_text(WORKER % ROOT)
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631", str(w)],
                                  text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["trials"] == ref["lm_trials"]
    assert np.allclose(res["chi2"], ref["chi2_final"], rtol=1e-5)
