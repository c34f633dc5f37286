"""Round 6: the recorded walks of the long shortlists (place_kernel.hpp: LongMemo, long_memo_try) against the oracle, bit-exact.

On a full cluster (every instance full, caches about equally old: getNext's LRU-window mode, MM.java:4911-4917) a shortlist spans the
table; commit records per type row and fresh-row bit what the walk yields for a request none of whose own positions STEERS it, and the
prefix-table kernels answer such requests from the record.  The cases here put the request's own positions exactly where the record
must NOT be used — the best / first eligible instance excluded or calling, the instance that ends the list excluded or calling — next
to thousands of requests it does answer, with up to eight exclusions (duplicates among them) inside the list.
Every comparison goes through the C ABI; MMP_NO_LONG_MEMO=1 is the same library without the records."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu


def _full(fleet, rng, spread=1_000_000):
    P = fleet.n_pods
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 2000, P)   # every instance full
    fleet.pods["lru_time"] = fleet.now - 36_000_000 - rng.integers(0, spread, P)  # and about equally old
    return fleet


def _hostile(fleet, order, reqs, extra, rng):
    """Rewrite a third of the requests so that their own positions fall on the head of the placement order (where every type's first
    eligible / best instance and, with the fresh-row break on, the instance that ends the list are): exclusions there, the caller there,
    duplicates, eight exclusions."""
    n = len(reqs)
    reqs = reqs.copy()
    head = np.asarray(order[:6], dtype=np.int32)
    tailp = np.asarray(order[-3:], dtype=np.int32)
    pool = [np.asarray(extra, dtype=np.int32)]
    off = len(extra)
    kind = rng.integers(0, 12, n)
    for i in np.nonzero(kind < 4)[0]:
        k = int(kind[i])
        if k == 0:    # the caller is one of the first instances of the order
            reqs["self_pod"][i] = head[rng.integers(0, len(head))]
            continue
        if k == 1:    # exclusions on the head of the order (one to three of them, maybe twice the same)
            ex = rng.choice(head, int(rng.integers(1, 4)))
        elif k == 2:  # many exclusions inside the list, a duplicate among them, the table's last instances too
            ex = np.concatenate([rng.integers(0, fleet.n_pods, int(rng.integers(3, 6))).astype(np.int32), tailp[:1]])
            ex = np.concatenate([ex, ex[:1]])
        else:         # the caller excluded as well
            ex = np.asarray([reqs["self_pod"][i] if reqs["self_pod"][i] >= 0 else 0, head[1]], dtype=np.int32)
        ex = ex[:4].astype(np.int32)
        reqs["extra_off"][i] = off
        reqs["n_extra"][i] = len(ex)
        pool.append(ex)
        off += len(ex)
    return reqs, np.concatenate(pool)


@pytest.mark.parametrize("seed", range(4))
def test_recorded_long_walks_next_to_requests_that_steer_the_walk(seed, monkeypatch):
    rng = np.random.default_rng(7700 + seed)  # (the fleets of test_long_shortlists_by_prefix_tables: their shortlists are long)
    pods = int(rng.choice([1500, 3000, 5000]))
    fleet = _full(wl.fuzz_fleet(seed + 700, pods=pods, models=400, profile="full" if seed % 2 else None), rng)
    reqs, extra = wl.fuzz_requests(fleet, seed, 6000)
    # fresh rows on both sides of the fresh-row test (:4913-4917: 45 s and a tenth of the best instance's age)
    reqs["fresh_lru"] = np.where(rng.random(len(reqs)) < 0.5, fleet.now - 36_000_000 - rng.integers(0, 1_000_000, len(reqs)),
                                 fleet.now - rng.choice([10_000, 50_000, 4_000_000], len(reqs)))
    monkeypatch.setenv("MMP_LONG_MODE", "1")
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        order = s.order()
        rows = s.long_shortlists()
        assert len(rows) == 2 * max(fleet.n_types, 1) and rows["valid"].any(), rows
        reqs, extra = _hostile(fleet, order, reqs, extra, rng)
        want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8)
        assert want["n_candidates"].max() > 512
        got = s.place(reqs, extra, fleet.now)
        assert_same_decisions(fleet, reqs, got, want)
        assert s.split_batches()[0] == 0
        # one request per call (the single-decision kernel reads the same records), the rewritten requests among them
        for i in list(range(40)) + [int(j) for j in np.nonzero(reqs["n_extra"] > 0)[0][:80]]:
            one = s.place(reqs[i:i + 1], extra, fleet.now)
            assert_same_decisions(fleet, reqs[i:i + 1], one, want[i:i + 1])
    finally:
        s.close()
    # the same batch as two launches: the records alone, then the walk for what they leave (place_long_memo_kernel + place_long_tail_kernel)
    monkeypatch.setenv("MMP_LONG_SPLIT_FROM", "0")
    for tails in ("16", "3"):
        monkeypatch.setenv("MMP_TAIL_BLOCKS", tails)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            for _ in range(2):  # (the second call finds the stream's lists as the first one's tail left them)
                assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), want)
            assert s.split_batches()[0] == 2 or s.split_batches()[1]
        finally:
            s.close()
    monkeypatch.delenv("MMP_TAIL_BLOCKS")
    monkeypatch.setenv("MMP_NO_LONG_MEMO", "1")
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        assert len(s.long_shortlists()) == 0
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), want)
    finally:
        s.close()


@pytest.mark.parametrize("n,split", [(100_000, False), (250_000, False), (250_000, True)])
def test_full_cluster_c3_batches_on_both_long_kernels(n, split, monkeypatch):
    """C3 with every instance full: a batch below and one above the size from which the 4-wavefront instantiation with its tables in
    LDS takes over (kLongDenseFrom), hostile rows mixed in; the records must exist for every type of the workload."""
    fleet = wl.make_full_cluster(wl.make_fleet("C3"))
    rng = np.random.default_rng(77)
    parts, ex_parts, off = [], [], 0
    for k in range(-(-n // fleet.n_models)):
        rq, ex = wl.make_requests(fleet, seed=0x10C0 + k)
        rq = rq.copy()
        rq["extra_off"] += off
        off += len(ex)
        parts.append(rq)
        ex_parts.append(ex)
    reqs, extra = np.concatenate(parts)[:n], np.concatenate(ex_parts)
    if split:
        monkeypatch.setenv("MMP_LONG_SPLIT_FROM", "0")
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        rows = s.long_shortlists()
        assert len(rows) == 2 * max(fleet.n_types, 1)
        assert rows["valid"][0::2].sum() >= max(fleet.n_types, 1) - 1, rows  # (a preferring type in case (b) has no record)
        reqs, extra = _hostile(fleet, s.order(), reqs, extra, rng)
        got = s.place(reqs, extra, fleet.now)
        assert (s.split_batches()[0] > 0) == split
    finally:
        s.close()
    want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=16)
    assert_same_decisions(fleet, reqs, got, want)
    assert got["n_candidates"].max() > 5000


def test_records_follow_the_snapshot():
    """A commit that changes the head of the order changes the records; decisions after it come from the new ones."""
    rng = np.random.default_rng(31)
    fleet = _full(wl.fuzz_fleet(941, pods=3000, models=300), rng)
    reqs, extra = wl.fuzz_requests(fleet, 5, 4000)
    reqs["fresh_lru"] = fleet.now - 36_000_000 - rng.integers(0, 1_000_000, len(reqs))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        before = s.long_shortlists()
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8))
        first = int(s.order()[0])
        fleet.pods["lru_time"][first] = fleet.now - 1000  # the oldest cache becomes the youngest: another best instance
        s.load_fleet(fleet)
        after = s.long_shortlists()
        assert before["valid"].any() and not np.array_equal(before, after)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), OracleFleet(fleet).place(reqs, extra, fleet.now, threads=8))
    finally:
        s.close()
