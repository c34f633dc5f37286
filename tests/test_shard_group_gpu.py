"""The pod-axis group with the collectives INSIDE the library (include/mmplace.h: mmp_shard_group_init,
mmp_shard_commit, mmp_shard_place_batch): what a Java mesh — which holds no RCCL handles — calls.

* RCCL itself: a group of ONE shard built from a real ncclUniqueId (ncclCommInitRank + every ncclAllReduce of the
  protocol run on the library's stream; RCCL refuses two ranks on one device, and the GPU box has one).
* The group protocol with G > 1 — fast kernel, exchange, device-side rest count over the fixed-capacity sub-batch,
  the six phases, the rerun when the capacity is exceeded, the sharded commit — with G contexts on cuda:0 driven by
  G host threads through the library's own transport callback (mmp_shard_group_set_exchange): the all-reduce is done
  by the threads (D2H, numpy, H2D).  Results must be bit-identical to the CPU oracle on every shard.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu

XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)


class ThreadExchange:
    """An all-reduce among G host threads of one process over raw device pointers."""

    def __init__(self, G):
        import torch  # noqa: F401  (the HIP runtime torch loaded is the one the library uses)
        self.G = G
        self.bar = threading.Barrier(G)
        self.parts = [None] * G
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0

    def fn_for(self, rank):
        def fn(_user, dev_buf, count, elem64, op_min, stream):
            try:
                dt = np.int64 if elem64 else np.int32
                host = np.empty(count, dtype=dt)
                assert self.hip.hipStreamSynchronize(stream) == 0
                assert self.hip.hipMemcpy(host.ctypes.data, dev_buf, host.nbytes, 2) == 0  # D2H
                self.parts[rank] = host
                self.bar.wait()
                stack = np.stack(self.parts)
                red = stack.min(axis=0) if op_min else stack.sum(axis=0, dtype=dt)
                self.bar.wait()  # everyone has read every part before anyone overwrites its own
                assert self.hip.hipMemcpy(dev_buf, red.ctypes.data, red.nbytes, 1) == 0  # H2D
                if rank == 0:
                    self.calls += 1
                return 0
            except Exception:  # noqa: BLE001
                self.bar.abort()
                return 1
        return XFN(fn)


def _group_place(fleet, batches, G, rccl=False):
    """-> per batch (outs of shard 0, n_rest); asserts that every shard returns the same rows."""
    xc = ThreadExchange(G) if G > 1 else None
    results = [[None] * len(batches) for _ in range(G)]
    errors = []
    keep = []

    def run(g):
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            if xc is not None:
                cb = xc.fn_for(g)
                keep.append(cb)
                assert s.lib.mmp_shard_group_set_exchange(s.h, C.cast(cb, C.c_void_p), None) == 0
                s.shard_group_init(None, g, G)
            else:
                s.shard_group_init(s.shard_unique_id() if rccl else None, 0, 1)
            s.load_fleet(fleet, commit=False)
            s.shard_commit()
            for i, (reqs, extra) in enumerate(batches):
                results[g][i] = s.shard_place(reqs, extra, fleet.now)
            s.shard_group_destroy()
        except Exception as e:  # noqa: BLE001
            errors.append((g, repr(e)))
            if xc is not None:
                xc.bar.abort()
        finally:
            s.close()

    ths = [threading.Thread(target=run, args=(g,)) for g in range(G)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for g in range(1, G):
        for i in range(len(batches)):
            assert np.array_equal(results[g][i][0], results[0][i][0]) and results[g][i][1] == results[0][i][1]
    return results[0]


def _check(fleet, batches, G, rccl=False):
    orc = OracleFleet(fleet)
    got = _group_place(fleet, batches, G, rccl)
    for (reqs, extra), (outs, n_rest) in zip(batches, got):
        assert_same_decisions(fleet, reqs, outs, orc.place(reqs, extra, fleet.now, threads=8))
    return [n for _, n in got]


def test_rccl_group_of_one_shard_c2_and_fuzz():
    """ncclGetUniqueId -> ncclCommInitRank(world 1) -> the sharded commit's ncclAllReduce(SUM) and every
    ncclAllReduce(MIN / SUM) of a batch on the library's stream."""
    fleet = wl.make_fleet("C2")
    _check(fleet, [wl.make_requests(fleet, 3), wl.make_requests(fleet, 4, n=777)], 1, rccl=True)
    for seed in range(4):
        f = wl.fuzz_fleet(seed, profile=[None, "full"][seed % 2])
        _check(f, [wl.fuzz_requests(f, seed, 500)], 1, rccl=True)


@pytest.mark.parametrize("G", [2, 3, 8])
def test_group_of_virtual_shards_equals_the_oracle(G):
    fleet = wl.make_fleet("C2")
    rest = _check(fleet, [wl.make_requests(fleet, 11), wl.make_requests(fleet, 12, n=5000)], G)
    assert all(r < 1000 for r in rest), rest  # the single exchange decides almost everything
    for seed in range(6):
        f = wl.fuzz_fleet(100 + seed, pods=300, profile=[None, "full", "pref"][seed % 3])
        _check(f, [wl.fuzz_requests(f, seed, 700), wl.fuzz_requests(f, seed + 50, 64)], G)


def test_rest_beyond_the_sub_batch_capacity_is_rerun_at_its_exact_size():
    """A cluster in which every instance is full sends most decisions through the six phases: far more than the
    max(1024, n / 16) rows the first pass carries — the group reruns them at their exact size."""
    fleet = wl.make_fleet("C2")
    rng = np.random.default_rng(5)
    P = fleet.n_pods
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 4_000, P)
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-0.04, 0.04, P))).astype(np.int64)
    reqs, extra = wl.make_requests(fleet, 21)
    rest = _check(fleet, [(reqs, extra)], 2)
    assert rest[0] > max(1024, len(reqs) // 16), rest


def test_c3_group_of_two_full_size():
    fleet = wl.make_fleet("C3")
    rest = _check(fleet, [wl.make_requests(fleet, 0xBE7C0)], 2)
    assert rest[0] < 1000, rest


def _async_run(s, fleet, batches, commit_between=False):
    """Issue every batch with mmp_shard_place_batch_async_dev (each into its own result buffer), wait once at the end."""
    import torch
    from modelmesh_amd._lib import PLACE_OUT
    dev = torch.device("cuda:0")
    bufs = []
    for reqs, extra in batches:
        d_reqs = torch.from_numpy(np.ascontiguousarray(reqs).view(np.uint8).reshape(-1)).to(dev)
        d_extra = torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev)
        d_outs = torch.zeros(len(reqs) * 16, dtype=torch.uint8, device=dev)
        bufs.append((d_reqs, d_extra, d_outs))
    torch.cuda.synchronize(dev)
    for i, ((reqs, _), (d_reqs, d_extra, d_outs)) in enumerate(zip(batches, bufs)):
        s.shard_place_async_dev(d_reqs.data_ptr(), len(reqs), d_extra.data_ptr(), fleet.now, d_outs.data_ptr())
        if commit_between and i == 0:
            s.shard_commit()  # completes the open batch first (against the snapshot it was issued on), then re-ranks
    last_rest = s.shard_wait()
    return [np.frombuffer(d_outs.cpu().numpy().tobytes(), dtype=PLACE_OUT) for _, _, d_outs in bufs], last_rest


def _many_exclusions(fleet, reqs, extra, rng, k=40):
    """Requests with more exclusions than a lane carries: the shard's kernel leaves them to the six phases (a rest > 0)."""
    reqs, extra = reqs.copy(), np.ascontiguousarray(extra, dtype=np.int32)
    pool = [extra]
    off = len(extra)
    for i in rng.choice(len(reqs), size=min(k, len(reqs)), replace=False):
        ex = rng.choice(fleet.n_pods, size=12, replace=False).astype(np.int32)
        reqs["extra_off"][i], reqs["n_extra"][i] = off, len(ex)
        pool.append(ex)
        off += len(ex)
    return reqs, np.concatenate(pool)


@pytest.mark.parametrize("rccl", [False, True])
def test_async_batches_on_a_group_of_one_are_completed_by_the_next_call_and_by_wait(rccl):
    fleet = wl.make_fleet("C2")
    rng = np.random.default_rng(9)
    b0 = wl.make_requests(fleet, 31)
    b1 = _many_exclusions(fleet, *wl.make_requests(fleet, 32, n=3000), rng)  # leaves a rest: resolved when b2 is issued
    b2 = wl.make_requests(fleet, 33, n=500)
    b3 = _many_exclusions(fleet, *wl.make_requests(fleet, 34, n=900), rng)  # ... resolved by the wait
    batches = [b0, b1, b2, b3]
    orc = OracleFleet(fleet)
    for commit_between in (False, True):
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.shard_group_init(s.shard_unique_id() if rccl else None, 0, 1)
            s.load_fleet(fleet, commit=False)
            s.shard_commit()
            outs, last_rest = _async_run(s, fleet, batches, commit_between)
            assert last_rest > 0  # b3's
            for (reqs, extra), got in zip(batches, outs):
                assert_same_decisions(fleet, reqs, got, orc.place(reqs, extra, fleet.now, threads=8))
            assert s.shard_wait() == last_rest  # nothing open: the last count again
            s.shard_group_destroy()
        finally:
            s.close()


def test_async_batches_on_two_virtual_shards():
    fleet = wl.fuzz_fleet(140, pods=300, profile="pref")
    rng = np.random.default_rng(10)
    batches = [wl.fuzz_requests(fleet, 1, 700), _many_exclusions(fleet, *wl.fuzz_requests(fleet, 2, 600), rng), wl.fuzz_requests(fleet, 3, 64)]
    orc = OracleFleet(fleet)
    G = 2
    xc = ThreadExchange(G)
    results, errors, keep = [None] * G, [], []

    def run(g):
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            cb = xc.fn_for(g)
            keep.append(cb)
            assert s.lib.mmp_shard_group_set_exchange(s.h, C.cast(cb, C.c_void_p), None) == 0
            s.shard_group_init(None, g, G)
            s.load_fleet(fleet, commit=False)
            s.shard_commit()
            results[g] = _async_run(s, fleet, batches)
            s.shard_group_destroy()
        except Exception as e:  # noqa: BLE001
            errors.append((g, repr(e)))
            xc.bar.abort()
        finally:
            s.close()

    ths = [threading.Thread(target=run, args=(g,)) for g in range(G)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for (reqs, extra), a, b in zip(batches, results[0][0], results[1][0]):
        assert np.array_equal(a, b)
        assert_same_decisions(fleet, reqs, a, orc.place(reqs, extra, fleet.now, threads=8))


# ---- two real PROCESSES, the in-library group, the exchange words moved by gloo (both on cuda:0) ---------------------------
def _group_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = None
    try:
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]

        def fn(_user, dev_buf, count, elem64, op_min, stream):  # mmp_exchange_fn over the process group (bench.py's _gloo_exchange_callback)
            try:
                t = torch.empty(int(count), dtype=torch.int64 if elem64 else torch.int32)
                nb = t.numel() * t.element_size()
                if hip.hipStreamSynchronize(stream) != 0 or hip.hipMemcpy(t.data_ptr(), dev_buf, nb, 2) != 0:
                    return 1
                dist.all_reduce(t, op=dist.ReduceOp.MIN if op_min else dist.ReduceOp.SUM)
                return 0 if hip.hipMemcpy(dev_buf, t.data_ptr(), nb, 1) == 0 else 1
            except Exception:  # noqa: BLE001
                return 1
        cb = XFN(fn)
        fleet = wl.make_fleet("C2")
        batches = [wl.make_requests(fleet, 21), wl.make_requests(fleet, 22, n=3000)]
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        assert s.lib.mmp_shard_group_set_exchange(s.h, C.cast(cb, C.c_void_p), None) == 0
        s.shard_group_init(None, rank, world)
        s.load_fleet(fleet, commit=False)
        s.shard_commit()
        orc = OracleFleet(fleet)
        ok = True
        for reqs, extra in batches:
            outs, _ = s.shard_place(reqs, extra, fleet.now)
            want = orc.place(reqs, extra, fleet.now, threads=4)
            ok = ok and all(np.array_equal(outs[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))
        # asynchronous batches too (a batch is completed by the next call / the final wait): the same control flow on both ranks
        d_r = torch.from_numpy(np.ascontiguousarray(batches[0][0]).view(np.uint8).reshape(-1).copy()).to("cuda:0")
        d_x = torch.from_numpy(np.ascontiguousarray(batches[0][1])).to("cuda:0")
        d_o = torch.zeros(len(batches[0][0]) * 16, dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            s.shard_place_async_dev(d_r.data_ptr(), len(batches[0][0]), d_x.data_ptr(), fleet.now, d_o.data_ptr())
        s.shard_wait()
        from modelmesh_amd._lib import PLACE_OUT
        got = np.frombuffer(d_o.cpu().numpy().tobytes(), dtype=PLACE_OUT)
        want = orc.place(batches[0][0], batches[0][1], fleet.now, threads=4)
        ok = ok and all(np.array_equal(got[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))
        s.shard_group_destroy()
        q.put((rank, bool(ok)))
    finally:
        if s is not None:
            s.close()
        dist.destroy_process_group()


def test_two_processes_in_library_group_over_gloo_on_one_gpu():
    """The control flow of the driver's N-rank run (bench.py pod_axis_lib_leg) at world 2 on this one device: two processes,
    each with its own context and its shard of C2's instance table, the group protocol inside the library (sharded commit,
    speculative exchange, six-phase rest, asynchronous batches) with the exchange words moved by gloo through
    mmp_shard_group_set_exchange — RCCL refuses two ranks on one device, so this is every line of the 2-rank path but
    ncclAllReduce itself.  Both ranks' result rows equal the oracle."""
    import socket

    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]
