"""The instance-table listener WITH type constraints and TypeConstraintManager's INCREMENTAL path (SURVEY.md §8 rows a5 + a18:
MM.java:1455-1568, TypeConstraintManager.java:395-600, 680-747) from the REFERENCE'S OWN TEXT — tests/golden/ref_tcm.npz, made
by oracle/ref_harness/tcm_harness.cc over the event streams of tests/ref_tcm_cases.py — against what the oracle (and, in
tests/test_ref_tcm_gpu.py, the device) computes from the instance table as it stands at each checkpoint.

What the reference's incremental bookkeeping leaves PATH-INDEPENDENT is held exactly at every checkpoint: the cluster's stats,
clusterState's order, every type's candidate set (`constrainTo` of getNext), every record's ProhibitedTypeSet; and — on fleets
where no two label sets share a partition — the partitions' capacity / free / instance / model-copy totals and typeSetStats.
What it leaves PATH-DEPENDENT is pinned by its rule instead:
* a partition shared by several label sets OVER-COUNTS: getInstanceSetStats (TypeConstraintManager.java:557-579) records the
  labels -> tracker mapping only for the label set that CREATED the tracker, so getStatsForLabels returns null for the others and
  the listener never subtracts their departing instances (MM.java:1466, :1527-1528).  On such fleets the reference's totals are
  >= the table's, and the library reports the table's (streams tcm_events_0..9; the exact streams are tcm_events_exact_*);
* a partition's globalLru is the cluster-wide minimum as of the last event whose record carried that partition's labels (the
  listener re-accumulates the LRU over ALL instances into the touched partition's tracker only, MM.java:1515-1542);
* the preferred sets (`prefer` of getNext) depend on the ORDER of past events, three ways: refreshPerTypeInstanceSets is run by
  instanceAdded BEFORE the new record is in clusterState and by instanceRemoved (if at all) before it is out (:513-549), and not at
  all by events that change no type's sets although they change the instance scores; updateInstanceSet (:489-505) adds an instance
  that matches a preferred label even when it satisfies the requirements, where fromInstanceSet (:416-447, the computation at
  configuration load, which mmp_types_from_labels performs) does not; and a type the configuration change does not touch keeps
  the sets its history left (:607-668).  They are therefore NOT a function of the table; the vectors record them, the test counts
  how many equal the from-the-table sets (some, not all), and a host that wants the manager's own sets hands them over itself
  (mmp_types_load, which is what GpuPlacementLB does)."""
import copy
import os

import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd.solver import bitmap_from_bool
from oracle import bind as ob
from oracle import py_types
from oracle.bind import OracleFleet
from tests import ref_fleets as rf
from tests import ref_tcm_cases as tc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_tcm.npz")
STAT = ("total_capacity", "total_free", "global_lru", "instance_count", "model_copy_count")
MAXL = np.iinfo(np.int64).max


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN)


def labels(bits):
    return [i for i in range(64) if (int(bits) >> i) & 1]


def static_fleet(case, table, cfg):
    """The fleet a from-scratch observer builds from the table: rows of the present instances, the per-type tables of
    oracle/py_types.py installed (row T = a type without configured constraints)."""
    fleet = case["fleet"]
    P, T = fleet.n_pods, len(cfg)
    f = copy.copy(fleet)
    f.pods = fleet.pods.copy()
    f.pods["flags"] = _lib.POD_TOMBSTONE
    for p, row in table.items():
        order = f.pods["id_order"][p]
        f.pods[p] = row
        f.pods["id_order"][p] = order
        f.pods["flags"][p] = _lib.POD_LIVE
    present = {p: set(labels(case["pod_bits"][p])) for p in table}
    conf = {t: (labels(r), [l for l in labels(q) if l not in labels(r)]) for t, (r, q) in cfg.items() if r or q}
    per, default = py_types.type_tables(present, conf)
    al, pf = np.zeros((T + 1, P), bool), np.zeros((T + 1, P), bool)
    ha, hp = np.zeros(T + 1, np.uint8), np.zeros(T + 1, np.uint8)
    tables = {}
    for t in range(T + 1):
        a, b = per[t] if t in conf else (None, default)
        tables[t] = (None if a is None else set(a), None if b is None else set(b))
        ha[t], hp[t] = a is not None, b is not None
        al[t, list(a or [])] = True
        pf[t, list(b or [])] = True
    f.n_types, f.allowed, f.prefer, f.has_allowed, f.has_prefer = T + 1, bitmap_from_bool(al), bitmap_from_bool(pf), ha, hp
    return f, tables, conf


def check_checkpoint(name, case, ck, f, tables, want_order, want_stats, pts, sets, pst, tss, after_config, last_lru):
    """One checkpoint of the reference text against from-the-table values (computed by the oracle or read from the device)."""
    P, T = case["fleet"].n_pods, len(case["req_bits"])
    assert ck["cluster"] == tuple(int(want_stats[x]) for x in STAT), (name, ck["event"])
    assert np.array_equal(ck["present"][:, 0], want_order), (name, ck["event"])
    mask_of = {k: sum(1 << int(t) for t in s) for k, s in enumerate(sets)}
    for p, m in ck["present"]:
        assert mask_of[int(pts[p])] == int(m), (name, ck["event"], p)
    n_pref_equal = 0
    for t in range(T + 1):
        ha, al, hp, pf, st = ck["types"][t]
        wa, wp = tables[t]
        assert (tc.bitset(al, P) if ha else None) == wa, (name, ck["event"], t)
        gp = tc.bitset(pf, P) if hp else None
        n_pref_equal += gp == wp
        w = tuple(int(tss[t][x]) for x in STAT) if t < T else tuple(int(want_stats[x]) for x in STAT)
        if ha and t < T and not case["exact_partitions"]:
            assert st[3] >= w[3] and st[0] >= w[0], (name, ck["event"], t)
        elif ha and t < T:  # a constrained type: the sum over the partitions that do not prohibit it (lru: the minimum of theirs)
            assert (st[0], st[1], st[3], st[4]) == (w[0], w[1], w[3], w[4]), (name, ck["event"], t)
            ok = [last_lru.get(m, MAXL) for m, _, c in ck["parts"] if not (m >> t) & 1]
            assert st[2] == (min(ok) if ok else MAXL), (name, ck["event"], t)
        else:
            assert st == w, (name, ck["event"], t)
    live = {m: (s, c) for m, s, c in ck["parts"] if c > 0}
    want_parts = {mask_of[k]: tuple(int(pst[k][x]) for x in STAT) for k in range(len(sets)) if int(pst[k]["instance_count"]) > 0}
    if not case["exact_partitions"]:  # shared partitions: the reference's totals only ever exceed the table's (docstring)
        for m, w in want_parts.items():
            assert m in live and live[m][1] >= w[3] and live[m][0][0] >= w[0], (name, ck["event"], m)
        return n_pref_equal
    assert set(live) == set(want_parts), (name, ck["event"])
    for m, (s, c) in live.items():
        w = want_parts[m]
        assert (s[0], s[1], s[3], s[4]) == (w[0], w[1], w[3], w[4]) and c == w[3], (name, ck["event"], m)
        assert s[2] == last_lru.get(m, MAXL), (name, ck["event"], m, "partition lru = the cluster's as of its last event")
    return n_pref_equal


def walk(case, cks, evaluate):
    """Drives `evaluate(table, cfg) -> (f, tables, order, stats, pts, sets, pst, tss)` at the reference's checkpoints."""
    name = case.get("name", "")
    at = {ck["event"]: ck for ck in cks}
    last_lru, n_equal, n_rows = {}, 0, 0
    norm = lambda cfg: {t: (r, q & ~r) for t, (r, q) in cfg.items() if r or q}  # noqa: E731  (as ConfigTypeConstraints normalises, :92-95)
    prev = norm({t: (int(case["req_bits"][t]), int(case["pref_bits"][t])) for t in range(len(case["req_bits"]))})
    trackers = set()  # the partitions (by prohibited-type mask) that have a tracker: created with the first instance, dropped with the last
    for e, table, cfg, touched in tc.replay(case):
        ev = case["events"][e]
        need = e in at
        if touched is None and not need and ev["kind"] != tc.CONFIG:
            continue
        f, tables, conf = static_fleet(case, table, cfg)
        mask = lambda p: sum(1 << t for t in conf if tables[t][0] is not None  # noqa: E731
                             and not py_types.instance_matches(set(labels(case["pod_bits"][p])), conf[t][0], True))
        if ev["kind"] == tc.CONFIG and norm(cfg) != prev:
            last_lru = {}  # typeMappingsUpdated builds new trackers for the present instances; add() alone does not set their lru (:656-665)
            trackers = {mask(p) for p in table}
            prev = norm(cfg)
        stats = OracleFleet(f).stats() if touched is not None or need else None
        if touched is not None:  # the tracker of the event's labels (if there is one by now) takes the cluster-wide lru
            m = mask(touched)
            leaving = ev["kind"] == tc.DELETED or bool(ev["row"]["flags"] & _lib.POD_SHUTTING_DOWN)
            if not leaving:
                trackers.add(m)
            if m in trackers:
                last_lru[m] = int(stats["global_lru"])
            if leaving and not any(mask(p) == m for p in table):
                trackers.discard(m)  # instanceRemoved: "none left, remove them" (:533-540)
        if need:
            got = evaluate(f, table, cfg)
            n_equal += check_checkpoint(name, case, at[e], f, tables, *got, after_config=ev["kind"] == tc.CONFIG, last_lru=last_lru)
            n_rows += len(case["req_bits"]) + 1
    return n_equal, n_rows


def oracle_eval(f, table, cfg):
    orc = OracleFleet(f)
    pts, sets, pst = ob.partition_stats(f)
    return orc.order[: len(table)], orc.stats(), pts, sets, pst, ob.type_set_stats(f)


def test_listener_with_type_constraints_equals_the_reference_text(ref):
    n_ck = n_equal = n_rows = 0
    for name, case in tc.cases():
        ids = rf.string_ids(case["fleet"], 200)
        assert rf.digest(tc.input_blob(case, ids)) == bytes(ref[f"{name}/digest"]).decode(), name
        case["name"] = name
        cks, tail = tc.parse(ref[f"{name}/words"], case["fleet"].n_pods, len(case["req_bits"]))
        a, b = walk(case, cks, oracle_eval)
        n_ck, n_equal, n_rows = n_ck + len(cks), n_equal + a, n_rows + b
    assert n_ck >= 700
    # the preferred sets BETWEEN configuration changes are path-dependent in the reference: some, not all, equal the from-the-table sets
    assert 0 < n_equal < n_rows
