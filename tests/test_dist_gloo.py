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


# ---- pod-axis sharding: the product's PodShardedPlacer drives a CPU emulation of
# ---- one shard (oracle/py_shard.py) — what is under test is the exchange protocol and its host driver
def _order_full(fleet, orc):
    absent = np.setdiff1d(np.arange(fleet.n_pods, dtype=np.int32), orc.order)
    return np.concatenate([orc.order, absent]).astype(np.int32)


def _emul_fleet(seed, profile, pods):
    from modelmesh_amd import workload as wl
    fleet = wl.fuzz_fleet(seed, pods=pods, models=120, profile=profile)
    reqs, extra = wl.fuzz_requests(fleet, seed, 250)
    return fleet, reqs, extra


@pytest.mark.parametrize("speculative", [True, False])
@pytest.mark.parametrize("seed,profile,pods,G", [(3, None, 130, 2), (4, "prefer", 200, 3), (5, "full", 200, 2),
                                                 (6, "prefer", 70, 4), (8, "full", 300, 5)])
def test_pod_axis_protocol_lockstep_emulation(seed, profile, pods, G, speculative):
    from modelmesh_amd import dist as mdist
    from modelmesh_amd._lib import PLACE_OUT
    from oracle.bind import OracleFleet
    from oracle.py_shard import EmulShardBackend
    fleet, reqs, extra = _emul_fleet(seed, profile, pods)
    orc = OracleFleet(fleet)
    want = orc.place(reqs, extra, fleet.now)
    placers = [mdist.PodShardedPlacer(EmulShardBackend(fleet, _order_full(fleet, orc), g, G), speculative=speculative)
               for g in range(G)]
    mdist.run_lockstep([p.commit_steps() for p in placers])
    outs = [np.zeros(len(reqs), dtype=PLACE_OUT) for _ in range(G)]
    mdist.run_lockstep([p.place_steps(reqs, len(reqs), extra, fleet.now, o) for p, o in zip(placers, outs)])
    for o in outs:
        for f in ("chosen", "best", "n_candidates", "hash"):
            bad = np.nonzero(o[f] != want[f])[0]
            assert len(bad) == 0, (f, bad[:5], o[bad[:5]], want[bad[:5]])
    assert len({p.last_n_rest for p in placers}) == 1
    if speculative:
        assert placers[0].last_n_rest < len(reqs)  # the single exchange decided something ...
        if profile == "prefer":
            assert placers[0].last_n_rest > 0      # ... and these fleets also need the six phases
    if not speculative:
        assert placers[0].last_n_rest == len(reqs)


def _pod_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from modelmesh_amd import dist as mdist
    from modelmesh_amd._lib import PLACE_OUT
    from oracle.bind import OracleFleet
    from oracle.py_shard import EmulShardBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fleet, reqs, extra = _emul_fleet(11, "prefer", 260)
        orc = OracleFleet(fleet)
        placer = mdist.PodShardedPlacer(EmulShardBackend(fleet, _order_full(fleet, orc), rank, world))
        placer.commit()  # all_reduce(SUM) of the per-shard rank slices over gloo
        out = np.zeros(len(reqs), dtype=PLACE_OUT)
        placer.place(reqs, len(reqs), extra, fleet.now, out)  # 5 x all_reduce(MIN) + 1 x all_reduce(SUM)
        want = orc.place(reqs, extra, fleet.now)
        q.put((rank, all(np.array_equal(out[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))))
    finally:
        dist.destroy_process_group()


def test_two_rank_pod_axis_sharding_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pod_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def _barrier_worker(rank, world, port, arr, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = bench.NodeBarrier.create(rank, world, dist)
        if b is None:
            q.put((rank, "no barrier"))
            return
        bad = None
        for i in range(1, 4001):
            arr[rank] = i
            if rank == i % world and i % 500 == 0:
                import time
                time.sleep(0.002)  # a straggler: the others have to wait for it
            b.wait()
            seen = [arr[r] for r in range(world)]
            if min(seen) < i or max(seen) > i + 1:  # behind the barrier nobody is before it; nobody can be two ahead
                bad = (i, seen)
                break
        dist.barrier()
        b.close()
        q.put((rank, bad))
    finally:
        dist.destroy_process_group()


def test_node_barrier_of_the_bench_holds_three_ranks_together():
    """bench.NodeBarrier (the barrier of the timed region's brackets on one node): no rank passes barrier i before
    every rank has reached it, over 4000 barriers with stragglers; the page in /dev/shm is gone afterwards."""
    import glob

    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 3, _free_port()
    arr = ctx.Array("q", world, lock=False)
    procs = [ctx.Process(target=_barrier_worker, args=(r, world, port, arr, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, None) for r in range(world)], got
    assert not glob.glob(f"/dev/shm/mmp_bench_barrier_{port}*")
