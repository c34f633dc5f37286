"""The pin of the oracle: tests/golden/ref_getnext.npz holds what the REFERENCE'S OWN TEXT decides — PLACEMENT_ORDER
(MM.java:4646-4703), CacheMissForwardingLB.filter / getNext (:4760-5003), ForwardingLB.getNext (:4315-4392),
MapFilteringSet.apply (:4282), isFull, age, InstanceRecord.getRemaining, Utils.STRING_ARRAY_COMP — extracted verbatim by
line range and compiled with g++ against stand-ins for the Java library types (oracle/ref_harness/, run where
/root/reference exists; the vectors travel).  Here the C restatement (oracle/mm_oracle.c) and the Python one must
reproduce them exactly on the fleets of tests/ref_fleets.py (the Python restatement is held to the C one on the same kind of
fleets by tests/test_oracle_cross.py); tests/test_ref_vectors_gpu.py holds the HIP path to the same vectors."""
import os

import numpy as np
import pytest

from oracle import bind as ob
from oracle.bind import OracleFleet
from tests import ref_fleets as rf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_getnext.npz")


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN)


def expected_audit(reqs, ref_place):
    """(chosen, n_candidates, hash) in this repository's conventions from the reference's (chosen, candidates.size(),
    survivors, hash): the early returns report n_candidates = hash = 0 (include/mmplace.h) — every null, and the
    ABORT_REQUESTs of a favourSelf request (:4894, :4932: only those can fire with favourSelf; :4990 needs !favourSelf)."""
    chosen = ref_place[:, 0]
    early = (chosen == -1) | ((chosen == -2) & ((reqs["flags"] & 1) != 0))
    n = np.where(early, 0, ref_place[:, 1])
    h = np.where(early, 0, ref_place[:, 3].astype(np.uint32))
    return chosen, n, h.astype(np.uint32)


def check_place(name, fleet, reqs, got, ref_place):
    chosen, n, h = expected_audit(reqs, ref_place)
    bad = np.nonzero((got["chosen"] != chosen) | (got["n_candidates"] != n) | (got["hash"] != h))[0]
    assert len(bad) == 0, (name, len(bad), [(int(i), got[i], ref_place[i], reqs[i]) for i in bad[:5]])
    # candidates.get(0) is the final best instance whenever the list has a first entry that is not case (b)'s (:4875: there
    # the best itself is not added): with a single candidate that was not filtered, chosen == best
    one = (n == 1) & (chosen >= 0)
    assert np.all(got["best"][one] >= 0)


def test_the_vectors_name_the_reference_text_they_came_from(ref):
    m = str(ref["manifest"])
    for piece in ("ModelMesh.java:4649-4701", "ModelMesh.java:4763-4771", "ModelMesh.java:4777-5003", "ModelMesh.java:4316-4391",
                  "ModelMesh.java:4282-4282", "InstanceRecord.java:204-204", "Utils.java:26-35"):
        assert piece in m, m
    assert len(ref["names"]) >= 60


def test_placement_order_and_load_targets_equal_the_reference_text(ref):
    n_cases = n_dec = 0
    for name, fleet, ids, reqs, extra in rf.place_cases():
        assert rf.digest(rf.input_blob(fleet, ids, reqs, extra)) == bytes(ref[f"{name}/digest"]).decode(), \
            f"{name}: the inputs differ from the ones the vectors were generated from (re-run oracle/ref_harness/make_ref_vectors.py)"
        orc = OracleFleet(fleet)
        assert orc.order_rc == 0, name
        # clusterState iteration order under the reference's comparator (instances present in the table)
        assert np.array_equal(orc.order, ref[f"{name}/order"]), (name, orc.order[:10], ref[f"{name}/order"][:10])
        got = orc.place(reqs, extra, fleet.now, threads=4)
        check_place(name, fleet, reqs, got, ref[f"{name}/place"])
        n_cases += 1
        n_dec += len(reqs)
    assert n_cases >= 60 and n_dec >= 140_000
    assert len(ref["C3_table/order"]) == 10_000 and len(ref["C4_table/order"]) == 50_000  # the bench's configurations, whole tables
    assert ref["C3_full_cluster/place"][:, 1].max() == 10_000  # whole-table shortlists under the reference's own text


def test_single_caller_batches_equal_the_reference_text(ref):
    """Batches of ONE calling instance (tests/ref_fleets.py:caller_place_cases) under the reference's getNext text; the device
    decides the same vectors in the single-caller form (tests/test_ref_vectors_gpu.py)."""
    n = 0
    for name, fleet, ids, reqs, extra in rf.caller_place_cases():
        assert rf.digest(rf.input_blob(fleet, ids, reqs, extra)) == bytes(ref[f"{name}/digest"]).decode(), name
        got = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=4)
        check_place(name, fleet, reqs, got, ref[f"{name}/place"])
        n += len(reqs)
    assert n >= 45_000


def test_serve_targets_equal_the_reference_text(ref):
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        assert rf.digest(rf.input_blob(fleet, ids, serve=(reqs, in_use, last_used, xp, xt))) == bytes(ref[f"{name}/digest"]).decode()
        want = ref[f"{name}/serve"]
        live = np.ascontiguousarray(((fleet.pods["flags"] & 2) != 0).astype(np.uint8))
        for i in range(len(reqs)):
            r = reqs[i]
            mm = fleet.models[r["model"]]
            pods = fleet.ent_pod[mm["ent_off"]: mm["ent_off"] + mm["n_loaded"]]
            times = fleet.ent_time[mm["ent_off"]: mm["ent_off"] + mm["n_loaded"]]
            keep = np.ones(len(pods), bool)
            for j in range(r["n_excl"]):
                p, t = xp[r["excl_off"] + j], xt[r["excl_off"] + j]
                keep &= ~((pods == p) & ((t == np.iinfo(np.int64).min) | (times == t)))  # MMP_ANY_TIME: a key exclude
            ch, ts = ob.serve(r["self_pod"], r["flags"] & 1, r["flags"] & 2, pods[keep], times[keep], fleet.now,
                              r["assume_completed_ms"], r["local_in_flight"], r["last_invoke_time"], live, in_use, last_used)
            assert ch == want[i, 0], (name, i, ch, ts, want[i], r)
            if ch >= 0:
                assert ts == want[i, 1], (name, i, ch, ts, want[i])


def oracle_gates(fleet, r, excl_pod, excl_time, explicit, in_use_expiry):
    """The C restatements of the guards (oracle/mm_gates_oracle.c), request by request -> (bits, initial_size)."""
    import ctypes as C

    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if len(a) else None
    lib = ob.load()
    orc = OracleFleet(fleet)
    P, now, m = fleet.n_pods, fleet.now, fleet.models
    tstats = np.ascontiguousarray(ob.type_set_stats(fleet))
    in_table = np.ascontiguousarray(((fleet.pods["flags"] & 4) == 0).astype(np.uint8))
    al = ob.unpack_bitmap(fleet.allowed, P) if fleet.n_types else None
    out = np.zeros((len(r), 2), np.int64)
    for i in range(len(r)):
        q = r[i]
        mr = m[q["model"]]
        lp = fleet.ent_pod[mr["ent_off"]: mr["ent_off"] + mr["n_loaded"]]
        lt = fleet.ent_time[mr["ent_off"]: mr["ent_off"] + mr["n_loaded"]]
        fp = fleet.ent_pod[mr["ent_off"] + mr["n_loaded"]: mr["ent_off"] + mr["n_loaded"] + mr["n_failed"]]
        ft = fleet.ent_time[mr["ent_off"] + mr["n_loaded"]: mr["ent_off"] + mr["n_loaded"] + mr["n_failed"]]
        keep = np.ones(len(lp), bool)
        for j in range(q["n_excl"]):
            keep &= lp != excl_pod[q["excl_off"] + j]
        ex = np.ascontiguousarray(explicit[q["explicit_off"]: q["explicit_off"] + q["n_explicit"]])
        fl, ty = int(q["flags"]), int(mr["type"])
        stats = tstats[(0 if ty < 0 or ty >= len(tstats) else ty):][:1]
        want = 0
        cp, ct = np.ascontiguousarray(lp[keep]), np.ascontiguousarray(lt[keep])
        if lib.orc_go_local(_p(cp), _p(ct), len(cp), int(q["self_pod"]), fl & 1, (fl >> 1) & 1, (fl >> 2) & 1, now):
            want |= 1
        ftc = np.ascontiguousarray(ft)
        if lib.orc_load_failures_breached(_p(ftc), len(ftc), now, in_use_expiry):
            want |= 2
        lpc = np.ascontiguousarray(lp)
        if lib.orc_load_locations_breached(_p(lpc), len(lpc), _p(ex), len(ex), _p(in_table)):
            want |= 4
        local_filtered = (q["self_pod"] in ex) or (q["self_pod"] in lp) or (q["self_pod"] in fp)
        blocked = bool(fleet.n_types and fleet.has_allowed[mr["type"]] and not al[mr["type"]][q["self_pod"]])
        if local_filtered or blocked:
            want |= 8
        if lib.orc_churn_reject(fleet.min_churn_age_ms, fleet.min_space_units, int(q["cache_capacity"]),
                                int(q["cache_weighted_size"]), int(q["cache_oldest_time"]), now):
            want |= 16
        rej = C.c_int(0)
        init = lib.orc_load_local_initial_size((fl >> 5) & 1, int(q["size_hint"]), int(q["loading_count"]),
                                               int(q["weight_predict_cutoff"]), int(q["loader_predicted"]),
                                               stats.ctypes.data_as(C.c_void_p), (fl >> 3) & 1,
                                               int(q["last_used_time"]), int(q["cache_capacity"]),
                                               int(q["cache_weighted_size"]), int(q["cache_oldest_time"]), C.byref(rej))
        if rej.value:
            want |= 32
        if lib.orc_reload_elsewhere((fl >> 4) & 1, int(q["loaded_time"]), int(q["load_timeout_ms"]), now,
                                    stats.ctypes.data_as(C.c_void_p)):
            want |= 64
        fresh = np.zeros(1, dtype=ob.ORC_POD)
        fresh["lru_time"], fresh["capacity"], fresh["used"] = q["fresh_lru"], q["fresh_capacity"], q["fresh_used"]
        fresh["count"], fresh["loading_threads"] = q["fresh_count"], q["fresh_loading_threads"]
        fresh["loading_in_progress"], fresh["rpm"] = q["fresh_in_progress"], q["fresh_rpm"]
        fresh["shutting_down"] = (fl >> 8) & 1
        curp = np.ascontiguousarray(orc.pods[q["self_pod"]: q["self_pod"] + 1]).copy()
        tomb = bool(fleet.pods["flags"][q["self_pod"]] & 4)
        curp["shutting_down"] = bool(fleet.pods["flags"][q["self_pod"]] & 1)
        if lib.orc_should_publish(None if tomb else curp.ctypes.data_as(C.c_void_p), fresh.ctypes.data_as(C.c_void_p),
                                  now, int(q["last_published"]), (fl >> 6) & 1, (fl >> 7) & 1, fleet.min_space_units):
            want |= 128
        out[i] = want, init
    return out


def check_gates(name, got_bits, got_init, ref_gate):
    """ref_gate: (MMP_GATE_* bits, initialSize) from the reference's fragments; the initial size is only observable where
    loadLocal does not return early (:5195)."""
    want_bits = ref_gate[:, 0].astype(np.uint32)
    bad = np.nonzero(np.asarray(got_bits, np.uint32) != want_bits)[0]
    assert len(bad) == 0, (name, len(bad), [(int(i), bin(int(got_bits[i])), bin(int(want_bits[i]))) for i in bad[:6]])
    sized = (want_bits & 32) == 0
    assert np.array_equal(np.asarray(got_init)[sized], ref_gate[sized, 1]), name
    assert int(np.bitwise_or.reduce(want_bits)) == 255, "some guard never fired in the reference's run"


def test_request_guards_equal_the_reference_text(ref):
    """goLocal (:3599-3626), checkLoadFailureCount / checkLoadLocationCount (:4593-4626), throwIfLocalLoadNotAllowed (:4011-4041),
    the churn guard (:3872-3884), loadLocal's sizing and early reject (:5159-5197), onEviction's reload rule (:2886-2897,
    :2918-2920), publishInstanceRecord's hysteresis (:5395-5468) with loadingChange / loadChange (:5537-5549)."""
    for name, fleet, ids, r, xp, xt, expl, expiry in rf.gate_cases():
        tstats = np.ascontiguousarray(ob.type_set_stats(fleet))
        assert rf.digest(rf.input_blob(fleet, ids, gates=(r, xp, xt, expl, expiry, tstats))) == bytes(ref[f"{name}/digest"]).decode(), name
        got = oracle_gates(fleet, r, xp, xt, expl, expiry)
        check_gates(name, got[:, 0], got[:, 1], ref[f"{name}/gate"])


def check_scaleup(name, out, ov, ref_scale, ref_ov):
    for k, f in enumerate(("action", "copies", "timestamp", "new_i1", "new_i2", "heavy")):
        got = np.asarray(out[f]).astype(np.int64)
        bad = np.nonzero(got != ref_scale[:, k])[0]
        assert len(bad) == 0, (name, f, len(bad), [(int(i), int(got[i]), int(ref_scale[i, k])) for i in bad[:5]])
    # getExcludeSet() (:5835) fills a LOCAL of the task: it is observable only through the exclude lists of the scale-ups it
    # feeds (:5787-5805), so it is compared when the run scaled something up (with a zero threshold the reference builds the set
    # and then dies on rpm / scaleUpRpms, :5792, caught at :5807 — no load, nothing to observe)
    if np.any(ref_scale[:, 0] == 2):
        assert np.array_equal(np.asarray(ov).astype(np.uint8), ref_ov), name


def test_scaleup_plan_equals_the_reference_text(ref):
    """rateTrackingTask (MM.java:5641-5806: the second-copy rule, the scale-up rule), getExcludeSet (:5836-5855), loadedSince
    (:5861-5870): per cache entry the load the reference's own loop body triggers (which, how many copies, with what timestamp),
    the usage-slice markers it leaves, lastHeavyTime touched; and the overloaded-instance set."""
    seen = set()
    for name, fleet, ids, entries, sp in rf.scaleup_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = OracleFleet(fleet).stats()
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        out, ov, _ = ob.scaleup_plan(fleet, entries, sp.view(ob.ORC_SCALEUP_PARAMS))
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])
        seen |= set(np.unique(ref[f"{name}/scale"][:, 0]))
    assert seen >= {0, 1, 2}


def test_scaleup_edge_cases_equal_the_reference_text(ref):
    """The rate task's remaining exits: no invocations since the last check (MM.java:5667-5669), a type confined to one instance
    (:5697-5699), a scale-up with nobody overloaded (ourExcludeSet = the model's own instances, :5789)."""
    for name, fleet, ids, entries, sp in rf.scaleup_edge_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = OracleFleet(fleet).stats()
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        out, ov, sk = ob.scaleup_plan(fleet, entries, sp.view(ob.ORC_SCALEUP_PARAMS))
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])
        assert bool(sk) == (len(entries) == 0), name
    assert np.any(ref["scaleup_edge_nothing_overloaded/scale"][:, 0] == 2) and not ref["scaleup_edge_nothing_overloaded/overloaded"].any()


def check_conc(name, couts, res, ref_conc, ref_avg, skipped):
    """Per MaxConcCacheEntry: what getRpmScaleThreshold(true) returned (MM.java:2766-2796; 0 = never called), whether it reset the
    counters (:2771-2773), priorSum / priorCount afterwards; and the task's averageModelParallelism after the run (:5815-5818) —
    a double, compared EXACTLY (it is a sum of ints, one IEEE division and a max; BASELINE's 1e-6 would be slack)."""
    for k, f in enumerate(("threshold", "reset", "new_prior_sum", "new_prior_count")):
        got = np.asarray(couts[f]).astype(np.int64)
        bad = np.nonzero(got != ref_conc[:, k])[0]
        assert len(bad) == 0, (name, f, len(bad), [(int(i), int(got[i]), int(ref_conc[i, k])) for i in bad[:5]])
    a, b = float(res["average_model_parallelism"]), float(ref_avg[0])
    assert a == b and abs(a - b) <= 1e-6, (name, a, b, skipped)


def test_scaleup_plan_latency_based_equals_the_reference_text(ref):
    """rateTrackingTask with limitModelConcurrency == true (latencyBased, MM.java:5677, :5702-5707, :5815-5818, :5836) over
    MaxConcCacheEntry entries whose getRpmScaleThreshold body (:2767-2795) is the reference's text too."""
    seen, resets, thresholds = set(), 0, set()
    for name, fleet, ids, entries, conc, sp, cp in rf.scaleup_conc_cases():
        cstats = np.zeros(1, dtype=ob.ORC_STATS)
        cstats[0] = OracleFleet(fleet).stats()
        blob = rf.input_blob(fleet, ids, scaleup=(entries, sp, cstats, np.ascontiguousarray(ob.type_set_stats(fleet))), conc=(conc, cp))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        out, couts, ov, sk, res = ob.scaleup_plan_conc(fleet, entries, conc, sp.view(ob.ORC_SCALEUP_PARAMS), cp)
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])
        check_conc(name, couts, res, ref[f"{name}/conc"], ref[f"{name}/average_model_parallelism"], sk)
        seen |= set(np.unique(ref[f"{name}/scale"][:, 0]))
        resets += int(ref[f"{name}/conc"][:, 1].sum())
        thresholds |= set(np.unique(ref[f"{name}/conc"][:, 0]))
    assert seen >= {0, 1, 2} and resets > 1000 and 2**31 - 1 in thresholds and len(thresholds) > 200


def test_scaledown_plan_conc_and_edge_cases_equal_the_reference_text(ref):
    """The janitor over MaxConcCacheEntry candidates (MM.java:6294-6305: getRpmScaleThreshold(false), queued requests), with a
    sample period too short to judge (:6286-6289), and with this instance shutting down / gone (removeSecondModelCopy :6322-6324)."""
    total = 0
    for name, fleet, ids, entries, conc, dp, dyn in rf.scaledown_conc_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))
        cp = np.zeros(1, dtype=rf._lib.CONC_PARAMS)
        cp["dynamic_rpm_scale_constant"], cp["average_model_parallelism"] = dyn, 1.0
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats), conc=(conc, cp))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        rem = ob.scaledown_plan_conc(fleet, entries, conc, dp.view(ob.ORC_SCALEDOWN_PARAMS), dyn)
        bad = np.flatnonzero(rem != ref[f"{name}/removed"])
        assert bad.size == 0, (name, bad[:8], entries[bad[:8]], conc[bad[:8]])
        plain = ob.scaledown_plan(fleet, entries, dp.view(ob.ORC_SCALEDOWN_PARAMS))
        assert not np.array_equal(plain, rem), name  # (the MaxConcCacheEntry rows matter on these fleets)
        total += int(rem.sum())
    for name, fleet, ids, entries, dp in rf.scaledown_edge_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        rem = ob.scaledown_plan(fleet, entries, dp.view(ob.ORC_SCALEDOWN_PARAMS))
        assert np.array_equal(rem, ref[f"{name}/removed"]), name
    assert total >= 100


def test_scaledown_plan_equals_the_reference_text(ref):
    """The janitor's pass over scaleCopiesCandidates (MM.java:6110-6140), removeModelCopies (:6197-6310) and
    removeSecondModelCopy (:6314-6335): per candidate, whether the reference's own text removes the LOCAL copy."""
    total = 0
    for name, fleet, ids, entries, dp in rf.scaledown_cases():
        istats = ob.instance_set_stats(fleet, int(dp["self_pod"][0]))
        blob = rf.input_blob(fleet, ids, scaledown=(entries, dp, istats))
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        rem = ob.scaledown_plan(fleet, entries, dp.view(ob.ORC_SCALEDOWN_PARAMS))
        bad = np.flatnonzero(rem != ref[f"{name}/removed"])
        assert bad.size == 0, (name, bad[:8], entries[bad[:8]])
        total += int(rem.sum())
    assert total >= 40


def plan_calls(plan, fleet, units, partitioned, n_parts):
    """The reaper's ensureLoadedInternal calls (model, lastUsed, subset) from a per-subset plan function
    plan(partition, skip_models) -> (models, last_used, info): one pass for the whole cluster, or one per partition in
    typeConstraints.getPartitionStats() order, each skipping what the earlier ones took (:6724)."""
    rows = []
    if not partitioned:
        m, lu, _ = plan(-1, None)
        return np.stack([m, lu, np.zeros_like(lu)], 1).astype(np.int64).reshape(-1, 3)
    taken = np.zeros(0, np.int32)
    for k in range(n_parts):
        m, lu, _ = plan(k, taken)
        rows.append(np.stack([m, lu, np.full_like(lu, k)], 1).astype(np.int64).reshape(-1, 3))
        taken = np.concatenate([taken, m.astype(np.int32)])
    return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)


def test_proactive_plan_equals_the_reference_text(ref):
    """The leader's reaper (MM.java:6456-6463 candidate switch, :6574-6577 candidate rule, :6473-6489 dispatch per instance
    subset, :6619-6746 triggerProactiveLoadsForInstanceSubset, ModelToLoad.compareTo :6407): the ensureLoadedInternal calls the
    reference's own text makes, in order, with their timestamps."""
    from oracle.ref_harness.make_ref_vectors import proactive_inputs
    total = 0
    for name, fleet, ids, units, partitioned in rf.proactive_cases():
        inp = proactive_inputs(fleet, units, partitioned)
        blob = rf.input_blob(fleet, ids, proactive=inp)
        assert rf.digest(blob) == bytes(ref[f"{name}/digest"]).decode(), name
        got = plan_calls(lambda k, skip: ob.proactive_plan(fleet, units, fleet.now, fleet.n_models, partition=k, skip_models=skip),
                         fleet, units, partitioned, len(inp[3]))
        want = ref[f"{name}/proactive"]
        assert got.shape == want.shape and np.array_equal(got, want), (name, got[:5], want[:5])
        total += len(want)
    assert total > 3000


STAT_FIELDS = ("total_capacity", "total_free", "global_lru", "instance_count", "model_copy_count")


def ref_checkpoints(ref, name):
    stats, lens, flat = ref[f"{name}/stats"], ref[f"{name}/order_len"], ref[f"{name}/orders"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [(stats[k], flat[offs[k]: offs[k + 1]]) for k in range(len(lens))]


def test_instance_table_listener_equals_the_reference_text(ref):
    """handleInstanceTableChange (MM.java:1456-1567) + InstanceSetStatsTracker (:54, :58-60, :64-71, :75-83, :87-91) fed an event
    stream: after every checkpoint, clusterStats (maintained by deltas, the LRU rescanned) and clusterState's order (a skip-list
    set under PLACEMENT_ORDER, updated entry by entry) equal what the oracle computes from the TABLE as it then stands."""
    import copy
    n = 0
    for name, fleet, ids, ev, ck, tables in rf.table_event_cases():
        assert rf.digest(rf.input_blob(fleet, ids, events=(ev, ck))) == bytes(ref[f"{name}/digest"]).decode(), name
        cps = ref_checkpoints(ref, name)
        assert len(cps) == len(tables)
        for k, (want_stats, want_order) in enumerate(cps):
            f = copy.copy(fleet)
            f.pods = tables[k]
            orc = OracleFleet(f)
            assert np.array_equal(orc.order, want_order), (name, k)
            st = orc.stats()
            assert [int(st[x]) for x in STAT_FIELDS] == [int(v) for v in want_stats], (name, k, st, want_stats)
            n += 1
    assert n > 150


def ref_upgrade_maps(ref, name):
    lens, flat = ref[f"{name}/map_len"], ref[f"{name}/maps"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [{int(k): int(v) for k, v in flat[offs[i]: offs[i + 1]]} for i in range(len(lens))]


def test_upgrade_tracker_equals_the_reference_text(ref):
    """UpgradeTracker.instanceAdded / instanceRemoved / doHousekeeping (UpgradeTracker.java:86-114, :121-186, :193-200) from the
    reference's text: getLikelyReplacedReplicaSets() after every call of six rolling-update streams, against the Python
    restatement the library's tracker is tested with (oracle/py_upgrade.py)."""
    from oracle.py_upgrade import UpgradeTracker
    nonempty = 0
    for name, ev in rf.upgrade_event_cases():
        want = ref_upgrade_maps(ref, name)
        t = UpgradeTracker()
        for i, e in enumerate(ev):
            if e["kind"] == 0:
                t.instanceAdded(int(e["labels_key"]), int(e["replica_set"]), int(e["start_time"]), int(e["now"]))
            elif e["kind"] == 1:
                t.instanceRemoved(int(e["labels_key"]), int(e["replica_set"]), int(e["now"]))
            else:
                t.doHousekeeping(int(e["now"]))
            assert t.likelyReplacedReplicaSets == want[i], (name, i, t.likelyReplacedReplicaSets, want[i])
            nonempty += bool(want[i])
    assert nonempty > 200


def _bits_to_set(words):
    return {64 * j + b for j, w in enumerate(words) for b in range(64) if (int(w) >> b) & 1}


def check_type_tables(name, ref, fleet, n_types, tables, partitions, type_stats):
    """tables(t) -> (allowed set | None, preferred set | None) for t in 0..T (row T: an unconfigured type);
    partitions -> (pod -> partition, [(5 stats, set of prohibited type rows)] per partition); type_stats(t) -> 5 stats."""
    P = fleet.n_pods
    W = (P + 63) // 64
    rows = ref[f"{name}/rows"]
    for t in range(n_types + 1):
        al, pf = tables(t)
        r = rows[t].astype(np.uint64)
        want_al = _bits_to_set(r[2: 2 + W]) if rows[t][0] else None
        want_pf = _bits_to_set(r[2 + W: 2 + 2 * W]) if rows[t][1] else None
        assert al == want_al, (name, t, "allowed")
        assert pf == want_pf, (name, t, "preferred", pf, want_pf)
    lens, flat = ref[f"{name}/part_types_len"], ref[f"{name}/part_types"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    ref_sig = [frozenset(int(x) for x in flat[offs[k]: offs[k + 1]]) for k in range(len(lens))]
    ref_stats = {ref_sig[k]: tuple(int(x) for x in ref[f"{name}/part_stats"][k]) for k in range(len(lens))}
    pts, parts = partitions
    got_stats = {sig: st for st, sig in parts}
    assert got_stats == ref_stats, (name, got_stats, ref_stats)
    ref_part = ref[f"{name}/pod_part"]
    for p in range(P):  # every instance in the partition with the same ProhibitedTypeSet
        if ref_part[p] < 0:
            assert pts[p] < 0, (name, p)
        else:
            assert parts[pts[p]][1] == ref_sig[int(ref_part[p])], (name, p)
    # the reaper's order (PARTITION_STATS_COMP: free desc, lru asc, capacity desc): the key sequence is what the comparator fixes
    key = lambda st: (-st[1], st[2], -st[0])  # noqa: E731
    want_keys = [key(ref_stats[ref_sig[int(k)]]) for k in ref[f"{name}/order"]]
    assert want_keys == sorted(key(st) for st in got_stats.values()), name
    for t in range(n_types):
        flag, *st = (int(x) for x in ref[f"{name}/tstats"][t])
        assert tuple(type_stats(t)) == (tuple(st) if flag else tuple(type_stats(None))), (name, t, "typeSetStats")


def test_type_constraint_tables_equal_the_reference_text(ref):
    """TypeConstraintManager's static computation from the reference's text (fromInstanceSet, instanceMatches, the
    ProhibitedTypeSet of every instance, the instance scores, inferPreferredInstances, candidateSubsetStats, the partitions'
    order) against the restatements the device tables are tested with: oracle/py_types.py for the per-type instance sets,
    oracle/bind.py's partitions / subset stats."""
    from oracle import py_types
    from modelmesh_amd.solver import bitmap_from_bool
    for name, fleet, ids, pod_bits, req_bits, pref_bits in rf.type_constraint_cases():
        assert rf.digest(rf.input_blob(fleet, ids, types=(pod_bits, req_bits, pref_bits))) == bytes(ref[f"{name}/digest"]).decode(), name
        P, T = fleet.n_pods, len(req_bits)
        labels = lambda bits: [i for i in range(64) if (int(bits) >> i) & 1]  # noqa: E731
        present = {p: set(labels(pod_bits[p])) for p in range(P) if not (fleet.pods["flags"][p] & 5)}
        cfg = {t: (labels(req_bits[t]), labels(pref_bits[t])) for t in range(T)}
        want, want_default = py_types.type_tables(present, cfg)
        tables = lambda t: ((None if want[t][0] is None else set(want[t][0]), None if want[t][1] is None else set(want[t][1]))  # noqa: E731
                            if t < T else (None, None if want_default is None else set(want_default)))
        # the tables installed in a fleet: partitions and stats as the oracle derives them
        import copy
        f = copy.copy(fleet)
        al = np.zeros((T + 1, P), bool)
        pf = np.zeros((T + 1, P), bool)
        ha, hp = np.zeros(T + 1, np.uint8), np.zeros(T + 1, np.uint8)
        for t in range(T + 1):
            a, b = tables(t)
            ha[t], hp[t] = a is not None, b is not None
            al[t, list(a or [])] = True
            pf[t, list(b or [])] = True
        f.n_types, f.allowed, f.prefer, f.has_allowed, f.has_prefer = T + 1, bitmap_from_bool(al), bitmap_from_bool(pf), ha, hp
        pts, sets, pst = ob.partition_stats(f)
        parts = [(tuple(int(pst[k][x]) for x in STAT_FIELDS), frozenset(int(t) for t in sets[k])) for k in range(len(sets))]
        tss = ob.type_set_stats(f)
        g = OracleFleet(f).stats()
        type_stats = lambda t: tuple(int((g if t is None else tss[t])[x]) for x in STAT_FIELDS)  # noqa: E731
        check_type_tables(name, ref, fleet, T, tables, (pts, parts), type_stats)


def test_preshutdown_migration_equals_the_reference_text(ref):
    """preShutdown's loop over the local cache (MM.java:6998-7046): for which entries triggerNewModelCopyElsewhere is called and
    which of those the shutdown waits for (CUTOFF_AGE_MS, :276), from the reference's text."""
    n_act = n_wait = 0
    for name, fleet, ids, entries, self_pod, now in rf.migration_cases():
        assert rf.digest(rf.input_blob(fleet, ids, migration=(entries, self_pod, now))) == bytes(ref[f"{name}/digest"]).decode(), name
        act, wait = ob.migration_plan(fleet, entries, self_pod, now)
        bits = ref[f"{name}/migration"]
        assert np.array_equal(act, bits & 1) and np.array_equal(wait, bits >> 1), name
        n_act, n_wait = n_act + int(act.sum()), n_wait + int(wait.sum())
    assert n_act > 500 and 0 < n_wait < n_act
