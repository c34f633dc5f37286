"""world_size-2 `gloo` test of the N>1 path on CPU: model-axis sharding, MAX-over-ranks timing and
the all_gather of results (modelmesh_amd/dist.py).  The decision engine inside each rank is the
CPU oracle (allowed in tests); what is under test is that sharding + gathering reproduces the
single-process table exactly, for sizes that do not divide evenly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from modelmesh_amd import dist as mdist
    from modelmesh_amd import workload as wl
    from oracle.bind import OracleFleet
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fleet = wl.make_fleet("C2")
        reqs, extra = wl.make_requests(fleet, 99, n=n)
        lo, hi = mdist.shard_bounds(n, rank, world)
        local = OracleFleet(fleet).place(reqs[lo:hi], extra, fleet.now)
        full = mdist.gather_results(local, n)
        slowest = mdist.max_over_ranks(float(rank + 1))
        dist.barrier()
        if rank == 0:
            want = OracleFleet(fleet).place(reqs, extra, fleet.now)
            ok = all(np.array_equal(full[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))
            q.put((ok, slowest, lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 2])
def test_two_rank_model_axis_sharding(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok, slowest, lo, hi = q.get(timeout=5)
    assert ok and slowest == 2.0 and lo == 0 and hi == (n + 1) // 2


def test_shard_bounds_cover_everything():
    from modelmesh_amd.dist import shard_bounds
    for n in (0, 1, 7, 100_000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
