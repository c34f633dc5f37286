"""Multi-GPU layouts of the path: one process per GPU (torch.distributed; backend "nccl" IS RCCL on
ROCm, the same code runs on "gloo" — tests/test_dist_gloo.py).

1. Model axis (the throughput layout): decisions against one snapshot are
   independent and the snapshot is ~1 MB, so every rank holds the full snapshot and owns a contiguous
   slice of the request table.  No data-path collective; control-plane only (barrier, MAX of elapsed
   time, optional all_gather of the 16-byte results).

2. Pod axis (BASELINE.json north_star / config C4): rank g owns a contiguous range
   of PLACEMENT_ORDER positions; every rank sees the whole batch.  `PodShardedPlacer` drives the
   library's phase kernels (include/mmplace.h, csrc/shard_kernels.hpp) and performs the all-reduce
   between two phases: MIN of per-shard best positions / break positions / owner-supplied rows,
   SUM of per-shard candidate counts.  Six all-reduces per batch in the general protocol — but the head
   of the order decides almost every request, so a batch first takes the speculative form: every shard
   decides on its own slice, ONE all-reduce(MIN) of 2 int64 per decision picks the lowest shard holding
   an eligible pod, and only the decisions that shard could not finish alone go through the six phases
   (as a compacted sub-batch, identical on every shard).
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


# --------------------------------------------------------------------------------------------------
# pod-axis sharding

N_PHASES = 7  # phases 1..6 end in an all-reduce, phase 7 writes the result rows


class SolverShardBackend:
    """The product backend: one libmmplace context in shard mode, buffers are torch CUDA tensors."""

    def __init__(self, solver, shard: int, n_shards: int, device):
        import torch
        self.solver, self.shard, self.n_shards = solver, shard, n_shards
        self.device = torch.device(device)
        solver.shard_configure(shard, n_shards)

    @property
    def n_pods(self) -> int:
        return self.solver.n_pods

    def xchg_slots(self, phase: int) -> int:
        return self.solver.shard_xchg_slots(phase)

    def rank_partial(self):
        import torch
        r = torch.zeros(max(self.n_pods, 1), dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        self.solver.shard_rank_dev(r.data_ptr())  # synchronises the library's stream before returning
        return r

    def commit(self, rank):
        import torch
        torch.cuda.synchronize(self.device)  # the all-reduce ran on torch's stream
        self.solver.shard_commit_dev(rank.data_ptr())

    def phase(self, ph, d_reqs, n, d_extra, now, xchg, d_outs):
        import torch
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.solver.shard_phase_dev(ph, d_reqs.data_ptr(), n, d_extra.data_ptr() if d_extra is not None else 0, now,
                                    [x.data_ptr() for x in xchg], d_outs.data_ptr(), st)

    # ---- the speculative single-exchange form --------------------------------------------------
    def fast_slots(self) -> int:
        return self.solver.shard_fast_slots()

    def fast(self, d_reqs, n, d_extra, now, xf):
        import torch
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.solver.shard_fast_dev(d_reqs.data_ptr(), n, d_extra.data_ptr() if d_extra is not None else 0, now,
                                   xf.data_ptr(), st)

    def fast_finish(self, d_reqs, n, xf, d_outs):
        """-> (n_rest, rest_reqs, rest_outs): the undecided requests, compacted in decision order, and
        the buffer their result rows go to (library-owned device memory)."""
        import torch
        st = torch.cuda.current_stream(self.device).cuda_stream
        n_rest, rr, ro = self.solver.shard_fast_finish_dev(d_reqs.data_ptr(), n, xf.data_ptr(), d_outs.data_ptr(), st)
        return n_rest, _DevPtr(rr), _DevPtr(ro)

    def fast_scatter(self, n_rest, rest_outs, d_outs):
        import torch
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.solver.shard_fast_scatter_dev(n_rest, d_outs.data_ptr(), st)


class _DevPtr:
    """A raw device pointer with the one method the backends use of a tensor."""

    def __init__(self, p: int):
        self._p = p

    def data_ptr(self) -> int:
        return self._p


def _dist_all_reduce(t, op: str):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.SUM)


class PodShardedPlacer:
    """Drives one shard through commit and through the phases of a batch.  `all_reduce(tensor, op)`
    (op in {"min", "sum"}) defaults to torch.distributed.all_reduce on the default group."""

    def __init__(self, backend, all_reduce=None, speculative: bool = True):
        self.b = backend
        self.all_reduce = all_reduce or _dist_all_reduce
        self._xchg, self._xn = None, -1
        self._xf, self._xfn = None, -1
        self.speculative = speculative and hasattr(backend, "fast")
        self.last_n_rest = 0  # decisions of the last batch that needed the six-phase protocol

    def commit_steps(self):
        r = self.b.rank_partial()
        yield r, "sum"
        self.b.commit(r)

    def commit(self):
        for t, op in self.commit_steps():
            self.all_reduce(t, op)

    def _buffers(self, n: int):
        import torch
        if self._xn < n:
            self._xchg = [torch.empty(max(n, 1) * self.b.xchg_slots(ph), dtype=torch.int64, device=self.b.device)
                          for ph in range(1, N_PHASES)]
            self._xn = n
        return self._xchg

    def place_steps(self, d_reqs, n: int, d_extra, now: int, d_outs):
        """Generator: launches a kernel, then yields (exchange tensor, op) for the caller to all-reduce."""
        if not self.speculative:
            self.last_n_rest = n
            yield from self._general_steps(d_reqs, n, d_extra, now, d_outs)
            return
        import torch
        slots = self.b.fast_slots()
        if self._xfn < n:
            self._xf = torch.empty(max(n, 1) * slots, dtype=torch.int64, device=self.b.device)
            self._xfn = n
        xf = self._xf[: n * slots]
        self.b.fast(d_reqs, n, d_extra, now, xf)
        yield xf, "min"
        n_rest, rest_reqs, rest_outs = self.b.fast_finish(d_reqs, n, xf, d_outs)
        self.last_n_rest = n_rest
        if n_rest:  # the same number, the same requests, in the same order on every shard
            yield from self._general_steps(rest_reqs, n_rest, d_extra, now, rest_outs)
            self.b.fast_scatter(n_rest, rest_outs, d_outs)

    def _general_steps(self, d_reqs, n: int, d_extra, now: int, d_outs):
        xchg = self._buffers(n)
        for ph in range(1, N_PHASES):
            self.b.phase(ph, d_reqs, n, d_extra, now, xchg, d_outs)
            yield xchg[ph - 1][: n * self.b.xchg_slots(ph)], ("sum" if ph == 5 else "min")
        self.b.phase(N_PHASES, d_reqs, n, d_extra, now, xchg, d_outs)

    def place(self, d_reqs, n: int, d_extra, now: int, d_outs):
        for t, op in self.place_steps(d_reqs, n, d_extra, now, d_outs):
            self.all_reduce(t, op)


def run_lockstep(generators):
    """Advance several shards' step generators together inside ONE process, doing the all-reduce
    with tensor ops (tests on a single GPU: G virtual shards; also how a single process would drive
    several devices)."""
    import torch
    gens = list(generators)
    while True:
        steps = []
        for g in gens:
            try:
                steps.append(next(g))
            except StopIteration:
                steps.append(None)
        if all(s is None for s in steps):
            return
        assert all(s is not None for s in steps), "shards fell out of step"
        op = steps[0][1]
        stack = torch.stack([t.to(steps[0][0].device) for t, _ in steps])
        red = stack.amin(dim=0) if op == "min" else stack.sum(dim=0)
        for t, _ in steps:
            t.copy_(red.to(t.device))
