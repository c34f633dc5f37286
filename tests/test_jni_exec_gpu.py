"""The veneer's decision calls run on the GPU under the mock JVM (tests/jni_mock) and give the rows the direct C-ABI path
gives: fleet staging (podsLoad / typesLoad / replacedReplicaSetsLoad / modelsLoad / commit), getOrder, placeBatch,
serveBatch, gateBatch, clusterStats."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from tests import jni_mock as jm
from tests import ref_fleets as rf
from tests.test_jni_veneer import _java_natives

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def veneer(tmp_path_factory):
    return jm.Veneer(jm.build(tmp_path_factory.mktemp("jni")), _java_natives())


def _bb(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return jm.ByteBuffer(a) if a.size else None


def _stage(v, fleet):
    env = v.env
    env.clear()
    h = v.call("create", 0, int(fleet.min_space_units), int(fleet.min_churn_age_ms))
    assert h and env.pending() is None
    assert v.call("podsLoad", h, _bb(fleet.pods), fleet.n_pods) == 0
    if fleet.n_types:
        assert v.call("typesLoad", h, fleet.n_types, _bb(fleet.allowed, np.uint64), _bb(fleet.prefer, np.uint64),
                      _bb(fleet.has_allowed, np.uint8), _bb(fleet.has_prefer, np.uint8)) == 0
    rs = np.ascontiguousarray(fleet.replaced_rs, dtype=np.int32)
    if len(rs):
        assert v.call("replacedReplicaSetsLoad", h, _bb(rs), len(rs)) == 0
    assert v.call("modelsLoad", h, _bb(fleet.models), fleet.n_models, _bb(fleet.ent_pod, np.int32), _bb(fleet.ent_time, np.int64),
                  len(fleet.ent_pod)) == 0
    assert v.call("commit", h) == 0 and env.pending() is None
    return h


def _fleets():
    yield "C1", wl.make_fleet("C1"), 7
    yield "C2", wl.make_fleet("C2"), 5
    for seed in (1, 4, 8):
        yield f"fuzz{seed}", wl.fuzz_fleet(seed, profile=[None, "full", "pref"][seed % 3]), seed


def test_place_through_the_veneer_equals_the_c_abi(veneer):
    for name, fleet, seed in _fleets():
        reqs, extra = (wl.make_requests(fleet, seed) if name.startswith("C") else wl.fuzz_requests(fleet, seed, 2000))
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            want, order = s.place(reqs, extra, fleet.now), s.order()
            stats = s.stats()
        finally:
            s.close()
        h = _stage(veneer, fleet)
        try:
            ob = jm.ByteBuffer(np.zeros(max(fleet.n_pods, 1), np.int32))
            nb = jm.ByteBuffer(np.zeros(1, np.int32))
            assert veneer.call("getOrder", h, ob, nb) == 0
            assert np.array_equal(ob.arr[: int(nb.arr[0])], order), name
            outs = jm.ByteBuffer(np.zeros(len(reqs), dtype=_lib.PLACE_OUT))
            ex = np.ascontiguousarray(extra, dtype=np.int32)
            rc = veneer.call("placeBatch", h, _bb(reqs), len(reqs), _bb(ex), len(ex), int(fleet.now), outs)
            assert rc == 0 and veneer.env.pending() is None, veneer.env.pending()
            assert np.array_equal(outs.arr, want), name
            # one request at a time (what CacheMissForwardingLB.getNext does): the latency path
            for i in range(0, min(len(reqs), 40)):
                one = jm.ByteBuffer(np.zeros(1, dtype=_lib.PLACE_OUT))
                r1 = reqs[i:i + 1].copy()
                assert veneer.call("placeBatch", h, _bb(r1), 1, _bb(ex), len(ex), int(fleet.now), one) == 0
                assert one.arr[0] == want[i], (name, i)
            st = jm.ByteBuffer(np.zeros(1, dtype=_lib.STATS))
            assert veneer.call("clusterStats", h, st) == 0 and st.arr[0] == stats
        finally:
            veneer.call("destroy", h)


def test_serve_and_gates_through_the_veneer_equal_the_c_abi(veneer):
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            r2, counters = s.serve_counters(reqs, in_use, last_used)
            want = s.serve_k(r2, counters, xp, xt, fleet.now)
        finally:
            s.close()
        h = _stage(veneer, fleet)
        try:
            outs = jm.ByteBuffer(np.zeros(len(r2), dtype=_lib.SERVE_OUT))
            rc = veneer.call("serveBatch", h, _bb(r2), len(r2), _bb(counters), len(counters), _bb(xp, np.int32), _bb(xt, np.int64),
                             len(xp), int(fleet.now), outs)
            assert rc == 0 and veneer.env.pending() is None, veneer.env.pending()
            assert np.array_equal(outs.arr, want), name
        finally:
            veneer.call("destroy", h)
        break  # one serve fleet is the veneer's business; the semantics are tests/test_ref_vectors_gpu.py's
    for name, fleet, ids, r, xp, xt, expl, expiry in rf.gate_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            want = s.gates(r, xp, xt, expl, fleet.now, expiry)
        finally:
            s.close()
        h = _stage(veneer, fleet)
        try:
            outs = jm.ByteBuffer(np.zeros(len(r), dtype=_lib.GATE_OUT))
            rc = veneer.call("gateBatch", h, _bb(r), len(r), _bb(xp, np.int32), _bb(xt, np.int64), len(xp), _bb(expl, np.int32),
                             len(expl), int(fleet.now), int(expiry), outs)
            assert rc == 0 and veneer.env.pending() is None, veneer.env.pending()
            assert np.array_equal(outs.arr, want), name
        finally:
            veneer.call("destroy", h)
        break


def test_route_miss_and_delta_commits_through_the_veneer_equal_the_c_abi(veneer):
    """Round 4's natives: routeBatch (guards + serve target), missBatch (guards + load target) — one request at a time, as
    invokeModel asks — and deltaCommits."""
    name, fleet, ids, r, xp, xt, expl, expiry = next(iter(rf.gate_cases()))
    rng = np.random.default_rng(3)
    P = fleet.n_pods
    in_use = rng.integers(0, 3, P).astype(np.int32)
    last_used = (fleet.now - rng.choice([0, 5, 100, 10_000], P)).astype(np.int64)
    g = r[:24].copy()
    sr = np.zeros(len(g), dtype=_lib.SERVE_REQ)
    sr["model"], sr["self_pod"], sr["assume_completed_ms"] = g["model"], g["self_pod"], 3000
    sr["excl_off"], sr["n_excl"] = g["excl_off"], g["n_excl"]
    preqs, extra = wl.make_requests(fleet, 9, n=len(g), extra_frac=0.0)
    preqs["model"], preqs["self_pod"] = g["model"], g["self_pod"]
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        sr2, counters = s.serve_counters(sr, in_use, last_used)
        want_g = s.gates(g, xp, xt, expl, fleet.now, expiry)
        want_s = s.serve_k(sr2, counters, xp, xt, fleet.now)
        want_p = s.place(preqs, extra, fleet.now)
    finally:
        s.close()
    h = _stage(veneer, fleet)
    try:
        nb = jm.ByteBuffer(np.zeros(1, np.int64))
        assert veneer.call("deltaCommits", h, nb) == 0 and int(nb.arr[0]) == 0
        for i in range(len(g)):
            go = jm.ByteBuffer(np.zeros(1, dtype=_lib.GATE_OUT))
            so = jm.ByteBuffer(np.zeros(1, dtype=_lib.SERVE_OUT))
            g1, s1 = g[i:i + 1].copy(), sr2[i:i + 1].copy()
            c1 = counters[int(s1["cnt_off"][0]): int(s1["cnt_off"][0]) + int(s1["n_cnt"][0])].copy()
            s1["cnt_off"] = 0
            rc = veneer.call("routeBatch", h, _bb(g1), _bb(s1), 1, _bb(c1), len(c1), _bb(xp, np.int32), _bb(xt, np.int64), len(xp),
                             _bb(expl, np.int32), len(expl), int(fleet.now), int(expiry), go, so)
            assert rc == 0 and veneer.env.pending() is None, veneer.env.pending()
            assert go.arr[0] == want_g[i] and so.arr[0] == want_s[i], i
            po = jm.ByteBuffer(np.zeros(1, dtype=_lib.PLACE_OUT))
            rc = veneer.call("missBatch", h, _bb(g1), _bb(preqs[i:i + 1].copy()), 1, _bb(xp, np.int32), _bb(xt, np.int64), len(xp),
                             _bb(expl, np.int32), len(expl), None, 0, int(fleet.now), int(expiry), go, po)
            assert rc == 0 and veneer.env.pending() is None, veneer.env.pending()
            assert go.arr[0] == want_g[i] and po.arr[0] == want_p[i], i
        # a republished record, a commit: the insertion path through the veneer's podsUpsert / commit
        row = fleet.pods[:1].copy()
        row["count"] += 1
        assert veneer.call("podsUpsert", h, _bb(np.zeros(1, np.int32)), _bb(row), 1) == 0
        assert veneer.call("commit", h) == 0
        assert veneer.call("deltaCommits", h, nb) == 0 and int(nb.arr[0]) in (0, 1)
    finally:
        veneer.call("destroy", h)


def test_round5_natives_through_the_veneer_equal_the_c_abi(veneer):
    """placeBatchCaller (the single-caller form) and the latency-based rebalancers (scaleupPlanConc / scaledownPlanConc) under the
    mock JVM give the rows the ctypes path gives — on the reference-text cases of tests/ref_fleets.py."""
    name, fleet, ids, reqs, extra = next(iter(rf.caller_place_cases()))
    caller, rc = _lib.split_caller(reqs)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        want = s.place_c(caller, rc, extra, fleet.now)
    finally:
        s.close()
    h = _stage(veneer, fleet)
    try:
        ex = np.ascontiguousarray(extra, dtype=np.int32)
        outs = jm.ByteBuffer(np.zeros(len(rc), dtype=_lib.PLACE_OUT))
        assert veneer.call("placeBatchCaller", h, _bb(caller), _bb(rc), len(rc), _bb(ex), len(ex), int(fleet.now), outs) == 0
        assert veneer.env.pending() is None and np.array_equal(outs.arr, want)
        short = jm.ByteBuffer(np.zeros(3, dtype=_lib.PLACE_OUT))  # a direct buffer shorter than n results: refused, not overrun
        assert veneer.call("placeBatchCaller", h, _bb(caller), _bb(rc), len(rc), _bb(ex), len(ex), int(fleet.now), short) != 0
        veneer.env.clear()
    finally:
        veneer.call("destroy", h)

    name, fleet, ids, entries, conc, sp, cp = next(c for c in rf.scaleup_conc_cases() if c[0].endswith("_1_1"))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        w_out, w_co, w_ov, w_sk, w_res = s.scaleup_plan_conc(entries, conc, sp, cp)
    finally:
        s.close()
    h = _stage(veneer, fleet)
    try:
        o = jm.ByteBuffer(np.zeros(len(entries), dtype=_lib.SCALEUP_OUT))
        co = jm.ByteBuffer(np.zeros(len(entries), dtype=_lib.CONC_OUT))
        ov = jm.ByteBuffer(np.zeros(fleet.n_pods, np.uint8))
        sk = jm.ByteBuffer(np.zeros(1, np.int32))
        res = jm.ByteBuffer(np.zeros(1, dtype=_lib.CONC_RESULT))
        assert veneer.call("scaleupPlanConc", h, _bb(entries), _bb(conc), len(entries), _bb(sp), _bb(cp), o, co, ov, sk, res) == 0
        assert veneer.env.pending() is None
        assert np.array_equal(o.arr, w_out) and np.array_equal(co.arr, w_co) and np.array_equal(ov.arr, w_ov)
        assert int(sk.arr[0]) == w_sk and res.arr[0] == w_res
    finally:
        veneer.call("destroy", h)

    name, fleet, ids, entries, conc, dp, dyn = next(iter(rf.scaledown_conc_cases()))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        want = s.scaledown_plan_conc(entries, conc, dp, dyn)
    finally:
        s.close()
    h = _stage(veneer, fleet)
    try:
        rem = jm.ByteBuffer(np.zeros(len(entries), np.uint8))
        assert veneer.call("scaledownPlanConc", h, _bb(entries), _bb(conc), len(entries), _bb(dp), int(dyn), rem) == 0
        assert veneer.env.pending() is None and np.array_equal(rem.arr, want)
    finally:
        veneer.call("destroy", h)
