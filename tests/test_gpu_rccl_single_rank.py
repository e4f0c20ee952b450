"""The RCCL side of the N-GPU layout on the one card a test box has: a process group with backend "nccl" (= RCCL on ROCm) of world
size 1, and through it exactly the calls bench.py makes at N > 1 with device tensors — `all_gather_points` / `all_gather_points_async`
(bulletproofspp_amd/dist.py: int64 `all_gather_into_tensor` on cuda:0), the MAX all-reduce of the step time and the barriers.  Two
ranks cannot share a GPU under RCCL, so the two-rank tests (tests/test_gpu_two_ranks.py) exchange over gloo; this one makes sure the
device-tensor branch itself runs on the real backend.  A fresh process: a process group is process-wide state."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import torch.distributed as dist
import bulletproofspp_amd as b
from bulletproofspp_amd.capi import points_to_array, scalars_to_array, array_to_point
from bulletproofspp_amd.dist import all_gather_points, all_gather_points_async, shard_range
import pyoracle as O
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
gpu = b.Bppp(0)
pts = O.hash_points(b"rccl", 40); sc = [(7919 * i + 3) % O.N for i in range(40)]
lo, hi = shard_range(40, dist.get_rank(), dist.get_world_size())
part = gpu.msm(scalars_to_array(sc[lo:hi]), points_to_array(pts[lo:hi]))
row = points_to_array([part])[0]
got = all_gather_points(row, dist, dev)
assert got.shape == (1, 8) and (got[0] == row).all(), got
pend = all_gather_points_async(row, dist, dev)
assert (pend.result()[0] == row).all()
assert gpu.sum_points(got) == part == O.PyEC().inner_product(list(zip(sc, pts)))
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
assert float(t.item()) == 1.25
dist.destroy_process_group()
print("rccl ok", dist.is_nccl_available())
"""


def test_device_tensor_collectives_run_on_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + random.randrange(1500)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0 and "rccl ok True" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
