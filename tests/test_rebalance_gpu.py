"""GPU parity for the batch rebalancers: the device selects the
set, the caller then runs K x place.  a17 = leader reaper proactive loads, MM.java:6574-6577 and
:6616-6747."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle import bind as ob

pytestmark = pytest.mark.gpu


def _plan_fleet(seed, pods, models, used_frac, dup_frac=0.3):
    rng = np.random.default_rng(7000 + seed)
    fleet = wl.fuzz_fleet(seed, pods=pods, models=models)
    now = fleet.now
    p = fleet.pods
    p["flags"] = np.where(rng.random(pods) < 0.05, 1, 2)  # a few shutting down
    p["capacity"] = rng.choice([131072, 1_000_000], pods)
    p["used"] = (p["capacity"] * np.clip(rng.normal(used_frac, 0.1, pods), 0, 1.02)).astype(np.int64)
    p["loading_threads"] = rng.choice([0, 1, 8], pods)
    p["loading_in_progress"] = rng.choice([0, 1, 60, 500], pods)
    p["count"] = rng.integers(0, 40, pods)
    p["lru_time"] = np.where(p["count"] == 0, 2**63 - 1, now - rng.integers(1_000, 50_000_000, pods))
    m = fleet.models
    # most models unloaded; many share a lastUsed value (the TreeSet keeps only the first)
    unloaded = rng.random(models) < 0.8
    m["n_loaded"] = np.where(unloaded, 0, m["n_loaded"])
    m["n_failed"] = np.where(rng.random(models) < 0.1, 2, np.minimum(m["n_failed"], 1))
    lu = now - rng.integers(1_000, 60_000_000, models)
    dup = rng.random(models) < dup_frac
    lu = np.where(dup, now - rng.choice([5_000, 900_000, 30_000_000], models), lu)
    m["last_used"] = lu
    # entries must stay consistent with n_loaded/n_failed: rebuild
    tot = m["n_loaded"] + m["n_failed"]
    off = np.zeros(models + 1, np.int64)
    np.cumsum(tot, out=off[1:])
    m["ent_off"] = off[:-1]
    fleet.ent_pod = rng.integers(0, pods, int(off[-1])).astype(np.int32)
    fleet.ent_time = (now - rng.integers(1_000, 1_000_000, int(off[-1]))).astype(np.int64)
    return fleet


@pytest.mark.parametrize("seed,pods,models,used", [(0, 8, 300, 0.5), (1, 64, 3000, 0.2), (2, 300, 20000, 0.9),
                                                  (3, 300, 20000, 0.99), (4, 1000, 100_000, 0.6), (5, 5, 50, 0.0)])
def test_proactive_plan_matches_oracle(seed, pods, models, used):
    fleet = _plan_fleet(seed, pods, models, used)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        for default_units in (6400, 1):
            gm, gl, gi = s.proactive_plan(default_units, fleet.now, models)
            wm, wl_, wi = ob.proactive_plan(fleet, default_units, fleet.now, models)
            for f in ("size_estimate", "free_count", "total_count", "n_candidates", "n_selected", "error",
                      "space_to_fill", "cutoff"):
                assert int(gi[f]) == int(wi[f]), (f, gi, wi)
            assert np.array_equal(gm, wm) and np.array_equal(gl, wl_)
            assert np.all(np.diff(gl) < 0)  # strictly descending: equal lastUsed collapse (TreeSet)
            # truncated output keeps the prefix
            gm2, gl2, gi2 = s.proactive_plan(default_units, fleet.now, 3)
            assert np.array_equal(gm2, wm[:3]) and int(gi2["n_selected"]) == int(wi["n_selected"])
            # the selected models then go through K x getNext with lastUsedTime = their timestamp (:6727)
            if len(gm):
                reqs, extra = wl.make_requests(fleet, 3, n=len(gm))
                reqs["model"], reqs["last_used"] = gm, gl
                out = s.place(reqs, extra, fleet.now)
                want = ob.OracleFleet(fleet).place(reqs, extra, fleet.now)
                assert np.array_equal(out["chosen"], want["chosen"]) and np.array_equal(out["hash"], want["hash"])
    finally:
        s.close()


def _local_entries(fleet, self_pod, rng, n):
    """A plausible local cache of `self_pod`: mostly models it holds, plus noise."""
    from modelmesh_amd import _lib
    m = fleet.models
    e = np.zeros(n, dtype=_lib.CACHE_ENTRY)
    e["model"] = rng.integers(0, fleet.n_models, n)
    e["model"] = np.where(rng.random(n) < 0.03, -1, e["model"])
    e["weight"] = rng.choice([1, 2560, 6400, 60_000], n)
    e["last_used"] = np.where(rng.random(n) < 0.05, 0, fleet.now - rng.choice([500, 30_000, 4_000_000, 50_000_000], n))
    e["interval_count"] = rng.choice([0, 1, 50, 400, 5000], n)
    e["last_heavy_time"] = np.where(rng.random(n) < 0.4, 0, fleet.now - rng.choice([1_000, 700_000, 30_000_000], n))
    e["last_unload_time"] = np.where(rng.random(n) < 0.6, 0, fleet.now - rng.choice([10_000, 100_000], n))
    e["earlier_use_iteration"] = rng.integers(0, 120, n)
    e["last_used_iteration"] = e["earlier_use_iteration"] + rng.integers(0, 60, n)
    e["flags"] = (rng.random(n) < 0.05).astype(np.uint32)
    return e


def _rebalance_fleet(seed, pods, models, used):
    """Fleet where many models have 1-4 copies, often including `self_pod` = 0."""
    rng = np.random.default_rng(8000 + seed)
    fleet = wl.fuzz_fleet(seed + 70, pods=pods, models=models)
    now = fleet.now
    p = fleet.pods
    p["flags"] = np.where(rng.random(pods) < 0.05, 1, 2)
    p["flags"][0] = 2
    p["used"] = (p["capacity"] * np.clip(rng.normal(used, 0.02, pods), 0, 1.0)).astype(np.int64)
    p["rpm"] = rng.choice([0, 100, 3000, 9000, 50_000], pods)
    m = fleet.models
    k = rng.choice([0, 1, 1, 2, 2, 3, 4], models).clip(0, pods)
    f = rng.choice([0, 0, 0, 1], models).clip(0, max(pods - 4, 0))
    m["n_loaded"], m["n_failed"] = k, f
    off = np.zeros(models + 1, np.int64)
    np.cumsum(k + f, out=off[1:])
    m["ent_off"] = off[:-1]
    ent = np.zeros(int(off[-1]), np.int32)
    for i in range(models):
        c = rng.choice(pods, size=k[i] + f[i], replace=False)
        if k[i] and rng.random() < 0.7 and 0 not in c:
            c[0] = 0
        ent[off[i]: off[i + 1]] = c
    fleet.ent_pod = ent
    fleet.ent_time = (now - rng.choice([1_000, 25_000, 2_000_000, 90_000_000], len(ent))).astype(np.int64)
    return fleet, rng


@pytest.mark.parametrize("seed,pods,used", [(0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (3, 1, 0.5)])
def test_scaleup_scaledown_migration_plans_match_oracle(seed, pods, used):
    from modelmesh_amd import _lib
    fleet, rng = _rebalance_fleet(seed, pods, 600, used)
    now = fleet.now
    entries = _local_entries(fleet, 0, rng, 800)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        seen_actions = set()
        for thr, our_rpm, last_check in ((2000, 100, now - 10_000), (100, 50_000, now - 9_000), (0, 0, now - 10_000),
                                         (2000, 0, now - 1_000)):
            sp = np.zeros(1, dtype=_lib.SCALEUP_PARAMS)
            sp["self_pod"], sp["iteration_counter"] = 0, 130
            sp["second_copy_max_age_iters"], sp["second_copy_min_age_iters"] = 240, 42
            sp["scale_up_rpm_threshold"], sp["our_rpm"] = thr, our_rpm
            sp["now"], sp["last_check_time"], sp["rate_check_interval_ms"] = now, last_check, 10_000
            sp["second_copy_lru_threshold_ms"], sp["assume_completed_ms"] = 72_000_000, 30_000
            g_out, g_ov, g_sk = s.scaleup_plan(entries, sp)
            w_out, w_ov, w_sk = ob.scaleup_plan(fleet, entries, sp.view(ob.ORC_SCALEUP_PARAMS))
            assert g_sk == w_sk
            # getExcludeSet() is built lazily in the Java (MM.java:5771-5774); its content only matters
            # (and is only compared) when some entry is actually scaled up
            if np.any(w_out["action"] == 2):
                assert np.array_equal(g_ov, w_ov)
            for f in ("action", "copies", "timestamp", "new_i1", "new_i2", "heavy", "rpm"):
                if g_sk and f == "rpm":
                    continue
                assert np.array_equal(g_out[f], w_out[f]), (f, thr, np.nonzero(g_out[f] != w_out[f])[0][:5])
            seen_actions |= set(np.unique(g_out["action"]))
        if pods >= 200:  # (small fleets with type constraints: a type's subset may leave no candidate instance)
            assert seen_actions >= {0, 1, 2}

        for thr, cap, sd in ((2000, 200_000, 0), (2000, 1_000, 0), (10, 10_000_000, 0), (2000, 200_000, 1)):
            dp = np.zeros(1, dtype=_lib.SCALEDOWN_PARAMS)
            dp["self_pod"], dp["shutting_down"], dp["now"] = 0, sd, now
            dp["last_check_time"], dp["rate_check_interval_ms"] = now - 7_000, 10_000
            dp["adjusted_cache_capacity"], dp["scale_up_rpm_threshold"] = cap, thr
            got = s.scaledown_plan(entries, dp)
            want = ob.scaledown_plan(fleet, entries, dp.view(ob.ORC_SCALEDOWN_PARAMS))
            assert np.array_equal(got, want), (thr, cap, sd, np.nonzero(got != want)[0][:5])

        ga, gw = s.migration_plan(entries, 0, now)
        wa, ww = ob.migration_plan(fleet, entries, 0, now)
        assert np.array_equal(ga, wa) and np.array_equal(gw, ww)
        if pods > 1:
            assert ga.sum() > 0 and gw.sum() > 0 and (ga & ~gw).sum() > 0
        # a21 then issues triggerNewModelCopyElsewhere = getNext with lastUsedTime = lruTime and
        # excludes = current holders ∪ self (MM.java:6913-6928)
        idx = np.nonzero(ga)[0]
        if len(idx):
            reqs, _ = wl.make_requests(fleet, 5, n=len(idx))
            reqs["model"], reqs["last_used"] = entries["model"][idx], entries["last_used"][idx]
            reqs["self_pod"], reqs["flags"] = 0, 1
            reqs["extra_off"], reqs["n_extra"] = 0, 1
            extra = np.zeros(1, np.int32)
            out = s.place(reqs, extra, now)
            want = ob.OracleFleet(fleet).place(reqs, extra, now)
            assert np.array_equal(out["chosen"], want["chosen"])
            assert not np.any(out["chosen"] == 0)  # never back onto the instance that is shutting down
    finally:
        s.close()


@pytest.mark.parametrize("seed,pods", [(0, 30), (1, 200), (2, 700), (3, 64), (4, 5)])
def test_partitions_and_subset_stats_match_the_restatement(seed, pods):
    """With type constraints the mesh uses per-partition / per-type ClusterStats (TypeConstraintManager;
    typeSetStats MM.java:1432-1439, instanceSetStats :1446-1448): partitions by ProhibitedTypeSet, their
    stats, and every type row's candidate-subset stats, rebuilt at commit, against the numpy restatement."""
    fleet = wl.fuzz_fleet(seed + 900, pods=pods, models=50, profile="prefer" if seed % 2 else None)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        pts, parts = s.partitions()
        wpts, wsets, wst = ob.partition_stats(fleet)
        assert np.array_equal(pts, wpts)
        assert len(parts) == len(wsets)
        for k, (st, bits) in enumerate(parts):
            assert bits == sum(1 << t for t in wsets[k]), k
            for f in ("total_capacity", "total_free", "global_lru", "instance_count", "model_copy_count"):
                assert int(st[f]) == int(wst[k][f]), (k, f)
        wt = ob.type_set_stats(fleet)
        for t in range(len(wt)):
            st = s.type_stats(t)
            for f in ("total_capacity", "total_free", "global_lru", "instance_count", "model_copy_count"):
                assert int(st[f]) == int(wt[t][f]), (t, f)
        if not fleet.n_types:
            assert len(parts) == 0 and int(s.type_stats(0)["instance_count"]) == int(s.stats()["instance_count"])
    finally:
        s.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_proactive_plan_per_partition_matches_oracle(seed):
    """One reaper pass over a fleet with type constraints (MM.java:6473-6488): a plan per ProhibitedTypeSet
    partition — its stats, its instances, its prohibited types excluded — each skipping the models the earlier
    partitions already took."""
    fleet = _plan_fleet(seed + 20, 300, 6000, [0.5, 0.9, 0.2][seed])
    if not fleet.n_types:  # make sure the fleet has type constraints
        rng = np.random.default_rng(seed)
        fleet.n_types = 3
        al = rng.random((3, fleet.n_pods)) < np.array([[1.0], [0.4], [0.7]])
        from modelmesh_amd.solver import bitmap_from_bool
        fleet.allowed, fleet.prefer = bitmap_from_bool(al), bitmap_from_bool(np.zeros_like(al))
        fleet.has_allowed, fleet.has_prefer = np.array([0, 1, 1], np.uint8), np.zeros(3, np.uint8)
        fleet.models["type"] = rng.integers(0, 3, fleet.n_models)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        _, parts = s.partitions()
        assert len(parts) >= 2
        taken = np.zeros(0, np.int32)
        n_sel = 0
        for k in range(len(parts)):
            gm, gl, gi = s.proactive_plan(6400, fleet.now, fleet.n_models, partition=k, skip_models=taken)
            wm, wl_, wi = ob.proactive_plan(fleet, 6400, fleet.now, fleet.n_models, partition=k, skip_models=taken)
            for f in ("size_estimate", "free_count", "total_count", "n_candidates", "n_selected", "error",
                      "space_to_fill", "cutoff"):
                assert int(gi[f]) == int(wi[f]), (k, f, gi, wi)
            assert np.array_equal(gm, wm) and np.array_equal(gl, wl_), k
            _, prohibited = parts[k]
            assert not any((prohibited >> int(t)) & 1 for t in fleet.models["type"][gm]), "a prohibited type was selected"
            assert not np.intersect1d(gm, taken).size
            taken = np.concatenate([taken, gm])
            n_sel += len(gm)
        assert n_sel > 0
    finally:
        s.close()


@pytest.mark.parametrize("mode", ["2", "2-launches", "1"])  # 2: the bucketed plan only (an overflow would raise) — as ONE launch
# (the default) and as its eight dependent launches (MMP_PLAN_FUSED=0); 1: the sorted path only
@pytest.mark.parametrize("seed,pods,models,used,dup,keys", [
    (11, 300, 20000, 0.2, 0.02, "ms"), (12, 300, 20000, 0.95, 0.0, "ms"), (13, 1000, 100_000, 0.6, 0.02, "ms"),
    (14, 64, 3000, 0.3, 0.5, "few"), (15, 300, 20000, 0.3, 0.0, "wide"), (16, 64, 600, 0.3, 1.0, "one")])
def test_proactive_plan_bucketed_and_sorted_paths(monkeypatch, mode, seed, pods, models, used, dup, keys):
    """The plan without the sort (key-range buckets, ranks by counting) and the sorted path it falls back to, each forced on the
    same registries: ties (the TreeSet keeps the first model seen), one single lastUsed value, keys across the whole int64 range."""
    monkeypatch.setenv("MMP_PLAN_SORTED", mode[0])
    monkeypatch.setenv("MMP_PLAN_FUSED", "0" if mode.endswith("launches") else "1")
    fleet = _plan_fleet(seed, pods, models, used, dup_frac=dup)
    rng = np.random.default_rng(seed)
    m = fleet.models
    if keys == "few":   # <= 1024 qualified per value: 3000 models over 8 values
        m["last_used"] = fleet.now - rng.choice(np.arange(1, 9) * 700_000, models)
    elif keys == "wide":
        m["last_used"] = rng.integers(-2**62, 2**62, models)
    elif keys == "one":
        m["last_used"] = fleet.now - 5_000
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        for default_units in (6400, 1):
            gm, gl, gi = s.proactive_plan(default_units, fleet.now, models)
            wm, wl_, wi = ob.proactive_plan(fleet, default_units, fleet.now, models)
            for f in ("size_estimate", "free_count", "total_count", "n_candidates", "n_selected", "error", "space_to_fill", "cutoff"):
                assert int(gi[f]) == int(wi[f]), (f, gi, wi)
            assert np.array_equal(gm, wm) and np.array_equal(gl, wl_)
    finally:
        s.close()


def test_proactive_plan_falls_back_when_a_bucket_overflows():
    fleet = _plan_fleet(3, 300, 20000, 0.2, dup_frac=0.6)  # thousands of unloaded models share three lastUsed values
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        gm, gl, gi = s.proactive_plan(6400, fleet.now, 20000)
        wm, wl_, wi = ob.proactive_plan(fleet, 6400, fleet.now, 20000)
        assert int(gi["n_selected"]) == int(wi["n_selected"]) and np.array_equal(gm, wm) and np.array_equal(gl, wl_)
    finally:
        s.close()


@pytest.mark.parametrize("seed,pods,models,dup,keys", [(21, 1000, 100_000, 0.02, "ms"), (22, 64, 3000, 0.5, "few"), (23, 300, 50_000, 0.004, "ms"), (24, 64, 3000, 0.0, "few40")])
def test_one_launch_plan_is_the_same_every_time(monkeypatch, seed, pods, models, dup, keys):
    """The one-launch plan hands data between its workgroups through device-scope loads and stores and grid-wide barriers of its own:
    300 plans of the same registry (every one reusing the scratch the one before left behind) equal the oracle's."""
    monkeypatch.setenv("MMP_PLAN_SORTED", "2")
    monkeypatch.setenv("MMP_PLAN_FUSED", "1")
    fleet = _plan_fleet(seed, pods, models, 0.4, dup_frac=dup)
    if keys == "few":    # buckets of ~300 entries: ranked from memory
        fleet.models["last_used"] = fleet.now - np.random.default_rng(seed).choice(np.arange(1, 9) * 700_000, models)
    elif keys == "few40":  # buckets of ~60: most of them cut by a wavefront's range
        fleet.models["last_used"] = fleet.now - np.random.default_rng(seed).choice(np.arange(1, 41) * 140_000, models)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        wm, wl_, wi = ob.proactive_plan(fleet, 6400, fleet.now, models)
        assert len(wm) > 0
        for k in range(300):
            gm, gl, gi = s.proactive_plan(6400, fleet.now, models)
            assert int(gi["n_selected"]) == int(wi["n_selected"]) and int(gi["n_candidates"]) == int(wi["n_candidates"]), k
            assert np.array_equal(gm, wm) and np.array_equal(gl, wl_), k
    finally:
        s.close()
