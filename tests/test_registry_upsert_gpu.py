"""mmp_models_upsert: registry events (whole ModelRecords replaced by index, MM.java:628 registry listener)
applied incrementally must leave the solver in exactly the state a full reload of the updated registry gives:
same rows and entries (logically — the entries live in an append-only arena that is squeezed now and then)
and the same load-target / serve decisions as the oracle on the updated fleet."""
import copy

import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu


def _logical(rows, ep, et):
    return [(int(r["type"]), int(r["last_used"]), int(r["n_loaded"]), int(r["n_failed"]),
             tuple(ep[r["ent_off"]: r["ent_off"] + r["n_loaded"] + r["n_failed"]].tolist()),
             tuple(et[r["ent_off"]: r["ent_off"] + r["n_loaded"] + r["n_failed"]].tolist())) for r in rows]


def _mutate(fleet, rng, n_changes, grow):
    """Replace n_changes random ModelRecords (new copy sets / failures / lastUsed), optionally append models.
    Returns the event batch (idx, rows, ent_pod, ent_time) and applies it to `fleet` (rebuilding its CSR)."""
    M, P = fleet.n_models, fleet.n_pods
    idx = rng.integers(0, M, n_changes).astype(np.int32)
    if grow:
        idx = np.concatenate([idx, np.arange(M, M + grow, dtype=np.int32), idx[:3]])  # appends + repeats (last wins)
    rows = np.zeros(len(idx), dtype=_lib.MODEL_ROW)
    pods_l, time_l = [], []
    off = 0
    for i in range(len(idx)):
        k = int(rng.choice([0, 1, 1, 2, 3, 5, 9]))
        f = int(rng.choice([0, 0, 1, 2]))
        k, f = min(k, P), min(f, max(P - k, 0))
        c = rng.choice(P, size=k + f, replace=False) if k + f else np.zeros(0, np.int64)
        seg_l = c[:k][np.argsort(fleet.pods["id_order"][c[:k]], kind="stable")]
        seg_f = c[k:][np.argsort(fleet.pods["id_order"][c[k:]], kind="stable")]
        rows[i]["type"] = rng.integers(0, max(fleet.n_types, 1))
        rows[i]["last_used"] = fleet.now - int(rng.integers(0, 10**8))
        rows[i]["n_loaded"], rows[i]["n_failed"], rows[i]["ent_off"] = k, f, off
        pods_l += list(seg_l) + list(seg_f)
        time_l += list(fleet.now - rng.integers(0, 10**7, k + f))
        off += k + f
    ent_pod, ent_time = np.asarray(pods_l, np.int32), np.asarray(time_l, np.int64)
    # apply to the structured fleet: last event per model wins
    new_m = M + grow
    recs = _logical(fleet.models, fleet.ent_pod, fleet.ent_time) + [None] * grow
    ev = _logical(rows, ent_pod, ent_time)
    for i, m in enumerate(idx):
        recs[int(m)] = ev[i]
    models = np.zeros(new_m, dtype=_lib.MODEL_ROW)
    ep, et, o = [], [], 0
    for j, r in enumerate(recs):
        models[j]["type"], models[j]["last_used"], models[j]["n_loaded"], models[j]["n_failed"] = r[0], r[1], r[2], r[3]
        models[j]["ent_off"] = o
        ep += list(r[4])
        et += list(r[5])
        o += len(r[4])
    fleet.models, fleet.ent_pod, fleet.ent_time = models, np.asarray(ep, np.int32), np.asarray(et, np.int64)
    return idx, rows, ent_pod, ent_time


@pytest.mark.parametrize("seed,pods,models", [(0, 40, 60), (1, 300, 500), (2, 2000, 4000)])
def test_upserts_equal_a_full_reload(seed, pods, models):
    rng = np.random.default_rng(6000 + seed)
    fleet = wl.fuzz_fleet(seed + 500, pods=pods, models=models)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        arena, squeezed = len(fleet.ent_pod), 0
        for rnd in range(14):  # enough rounds for the arena to fill with garbage and be squeezed
            grow = int(rng.choice([0, 0, 7, 150])) if rnd % 3 == 1 else 0
            ev = _mutate(fleet, rng, int(rng.choice([1, 30, models // 2, models * 2])), grow)
            s.upsert_models(*ev)
            if rnd % 4 == 3:
                s.commit()  # resolved positions are rebuilt against the new snapshot too
            rows, ep, et = s.get_models()
            assert len(rows) == fleet.n_models
            squeezed += len(ep) < arena + len(ev[2])  # the arena shrank instead of growing by this call's entries
            arena = len(ep)
            assert _logical(rows, ep, et) == _logical(fleet.models, fleet.ent_pod, fleet.ent_time), rnd
            reqs, extra = wl.fuzz_requests(fleet, seed * 100 + rnd, 1500)
            want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=4)
            assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), want)
        assert squeezed >= 1 or models < 4000, "the arena was never squeezed: the compaction path went untested"
        # a reload of the same registry gives the same answers as the upserted state
        ref = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            ref.load_fleet(fleet)
            reqs, extra = wl.fuzz_requests(fleet, 999, 3000)
            a, b = s.place(reqs, extra, fleet.now), ref.place(reqs, extra, fleet.now)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(a[f], b[f])
        finally:
            ref.close()
    finally:
        s.close()


def test_upsert_argument_checks():
    from modelmesh_amd.solver import MmpError
    fleet = wl.fuzz_fleet(3, pods=10, models=5)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        row = np.zeros(1, dtype=_lib.MODEL_ROW)
        with pytest.raises(MmpError):
            s.upsert_models([7], row, [], [])          # neither an existing model nor the next index
        row["n_loaded"] = 2
        with pytest.raises(MmpError):
            s.upsert_models([0], row, [1], [5])        # entry range beyond the arrays of the call
        s.upsert_models([5], np.zeros(1, dtype=_lib.MODEL_ROW), [], [])  # append an empty record
        rows, _, _ = s.get_models()
        assert len(rows) == 6 and rows["n_loaded"][5] == 0
    finally:
        s.close()
