"""Multi-GPU layout of the hot path: MSM terms shard across ranks with no data-path
collective; the only exchange is an all-gather of one 64-byte affine partial point per rank (RCCL has no
mod-p reduction), after which every rank adds the N points locally."""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n terms owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_points(partial_xy: np.ndarray, dist, device=None) -> np.ndarray:
    """partial_xy: (8,) uint64 affine point of this rank.  Returns (world, 8) uint64, same on every rank.
    `dist` is torch.distributed (backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    import torch
    world = dist.get_world_size()
    mine = torch.from_numpy(np.ascontiguousarray(partial_xy).view(np.int64).reshape(1, 8))
    if device is not None:
        mine = mine.to(device)
    out = torch.zeros((world, 8), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().view(np.uint64)


class PendingGather:
    """An all-gather of partial points in flight: `result()` waits for it and returns the (world, 8) uint64 array."""

    def __init__(self, out, work):
        self.out, self.work = out, work

    def result(self) -> np.ndarray:
        if self.work is not None:
            self.work.wait()
        return self.out.cpu().numpy().view(np.uint64)


def all_gather_points_async(partial_xy: np.ndarray, dist, device=None) -> PendingGather:
    """Non-blocking variant: the 64-B exchange of step i overlaps the kernels of step i+1 (the MSM call itself returns only once
    its point is on the host, so the collective is the only thing left to hide)."""
    import torch
    world = dist.get_world_size()
    mine = torch.from_numpy(np.ascontiguousarray(partial_xy).view(np.int64).reshape(1, 8))
    if device is not None:
        mine = mine.to(device)
    out = torch.zeros((world, 8), dtype=torch.int64, device=mine.device)
    work = dist.all_gather_into_tensor(out, mine, async_op=True)
    pend = PendingGather(out, work)
    pend._keep = mine
    return pend


def sharded_msm(n: int, rank: int, world: int, local_msm: Callable[[int, int], np.ndarray],
                sum_points: Callable[[np.ndarray], np.ndarray], dist, device=None) -> np.ndarray:
    """local_msm(lo, hi) -> (8,) partial point of this rank's slice; sum_points((world, 8)) -> (8,) total."""
    lo, hi = shard_range(n, rank, world)
    part = local_msm(lo, hi)
    allp = all_gather_points(part, dist, device)
    return sum_points(allp)
