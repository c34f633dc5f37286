"""Multi-GPU layout of the path: one process per GPU, model-axis sharding, no data-path collective.

Decisions against one snapshot are independent and the snapshot is ~1 MB, so
every rank holds the full snapshot and owns a contiguous slice of the request table.  The only
collectives are control-plane: a barrier around timed regions, a MAX over ranks of elapsed time,
and (when the caller wants the full result table in one place) an all_gather of the 16-byte
results.  `torch.distributed` with backend "nccl" is RCCL on ROCm; the same code runs on "gloo"
(tests/test_dist_gloo.py).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n requests for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_requests(reqs: np.ndarray, rank: int, world: int) -> np.ndarray:
    lo, hi = shard_bounds(len(reqs), rank, world)
    return reqs[lo:hi]


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a python float over the default process group (identity when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local: np.ndarray, n_total: int, device=None) -> np.ndarray:
    """all_gather per-rank result slices (structured arrays) back into request order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    item = local.dtype.itemsize
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros(longest * item, dtype=torch.uint8, device=device)
    mine = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy())
    buf[: mine.numel()] = mine.to(buf.device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.zeros(n_total, dtype=local.dtype)
    for r, (lo, hi) in enumerate(sizes):
        out[lo:hi] = np.frombuffer(parts[r].cpu().numpy().tobytes()[: (hi - lo) * item], dtype=local.dtype)
    assert sizes[rank][1] - sizes[rank][0] == len(local)
    return out
