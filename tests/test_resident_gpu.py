"""The resident decision kernel (include/mmplace.h: mmp_resident): mmp_place_batch(n = 1) answered by a wavefront that
stays on the GPU and polls 64 pinned request slots — no kernel launch per request.  Results must be those of the
oracle / the launch path whatever happens around it: many threads at once, commits and registry events in between
(which stop it), idle exits (it leaves by itself and is started again), requests it hands back to the launch path."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet

pytestmark = pytest.mark.gpu
FIELDS = ("chosen", "best", "n_candidates", "hash")


def _stats(s):
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert s.lib.mmp_resident_stats(s.h, C.byref(a), C.byref(b), C.byref(c)) == 0
    return a.value, b.value, c.value


def _resident_solver(fleet, monkeypatch, idle_ms=50):
    monkeypatch.setenv("MMP_RESIDENT", "1")
    monkeypatch.setenv("MMP_RESIDENT_IDLE_MS", str(idle_ms))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    return s


def test_single_requests_equal_the_oracle_and_need_one_launch(monkeypatch):
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 31, n=3000, extra_frac=0.0)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    s = _resident_solver(fleet, monkeypatch)
    try:
        for i in range(len(reqs)):
            got = s.place(reqs[i:i + 1], None, fleet.now)
            for f in FIELDS:
                assert got[f][0] == want[f][i], (i, f)
        launches, served, punted = _stats(s)
        assert served + punted == len(reqs) and launches <= 3 and served > 0.9 * len(reqs), (launches, served, punted)
        # requests with exclusions of their own, and batches, keep taking the launch paths
        r2, e2 = wl.make_requests(fleet, 32, n=64, extra_frac=1.0)
        w2 = OracleFleet(fleet).place(r2, e2, fleet.now)
        for i in range(len(r2)):
            one = r2[i:i + 1].copy()
            ex = e2[one["extra_off"][0]: one["extra_off"][0] + one["n_extra"][0]].copy()
            one["extra_off"] = 0
            got = s.place(one, ex, fleet.now)
            assert got["chosen"][0] == w2["chosen"][i] and got["hash"][0] == w2["hash"][i]
    finally:
        s.close()


def test_fuzz_fleets_including_the_shapes_it_hands_back(monkeypatch):
    """Fuzz fleets hit case (b), the replay list, the replica-set retry: the resident wavefront answers those with
    'punt' and the library decides them with a launch — same answer either way."""
    punted_total = 0
    for seed in range(6):
        fleet = wl.fuzz_fleet(seed, pods=200, profile=[None, "full", "prefer"][seed % 3])
        reqs, extra = wl.fuzz_requests(fleet, seed, 400)
        reqs = reqs.copy()
        reqs["n_extra"], reqs["extra_off"] = 0, 0
        want = OracleFleet(fleet).place(reqs, None, fleet.now)
        s = _resident_solver(fleet, monkeypatch)
        try:
            for i in range(len(reqs)):
                got = s.place(reqs[i:i + 1], None, fleet.now)
                for f in FIELDS:
                    assert got[f][0] == want[f][i], (seed, i, f)
            punted_total += _stats(s)[2]
        finally:
            s.close()
    assert punted_total > 0


def test_many_threads_commits_and_idle_exits(monkeypatch):
    a = wl.make_fleet("C2")
    b = wl.make_fleet("C2")
    rng = np.random.default_rng(0xAB)
    perm = rng.permutation(b.n_pods)
    for f in ("lru_time", "capacity", "used", "count", "rpm", "loading_in_progress"):
        b.pods[f] = b.pods[f][perm]
    reqs, _ = wl.make_requests(a, 91, n=4000, extra_frac=0.0)
    wa = OracleFleet(a).place(reqs, None, a.now, threads=8)
    wb = OracleFleet(b).place(reqs, None, b.now, threads=8)
    s = _resident_solver(a, monkeypatch, idle_ms=5)
    errors, stop = [], threading.Event()

    def worker(tid):
        try:
            i = tid
            while not stop.is_set():
                got = s.place(reqs[i:i + 1], None, a.now)
                is_a = all(got[f][0] == wa[f][i] for f in FIELDS)
                is_b = all(got[f][0] == wb[f][i] for f in FIELDS)
                if not (is_a or is_b):
                    errors.append((i, got[0]))
                i = (i + 7) % len(reqs)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    try:
        for t in ths:
            t.start()
        for k in range(20):  # every commit stops the resident kernel; the workers start the next one
            s.load_pods((b if k % 2 == 0 else a).pods)
            s.commit()
            time.sleep(0.01)
        stop.set()
        for t in ths:
            t.join()
        assert not errors, errors[:3]
        time.sleep(0.05)  # longer than the idle limit: it has left the GPU by itself ...
        before = _stats(s)[0]
        got = s.place(reqs[:1], None, a.now)  # ... and this request starts it again
        assert _stats(s)[0] == before + 1
        table = a if (20 - 1) % 2 else b
        want = OracleFleet(table).place(reqs[:1], None, table.now)
        assert got["chosen"][0] == want["chosen"][0] and got["hash"][0] == want["hash"][0]
    finally:
        stop.set()
        s.close()
