"""The comparator / shortlist / rpm-filter half of the path is NOT pinned by any reference test
(SURVEY.md §4, §8c), so the C oracle is cross-checked against a second restatement written
independently from the Java text (oracle/py_oracle.py) on adversarial random snapshots."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from oracle import bind as ob
from oracle import py_oracle as po


def _py_pods(fleet):
    out = []
    for r in fleet.pods:
        out.append(dict(lru_time=int(r["lru_time"]), capacity=int(r["capacity"]), used=int(r["used"]),
                        version=int(r["version"]), count=int(r["count"]), loading_threads=int(r["loading_threads"]),
                        loading_in_progress=int(r["loading_in_progress"]), rpm=int(r["rpm"]),
                        id_order=int(r["id_order"]), replica_set=int(r["replica_set"]),
                        shutting_down=bool(r["flags"] & 5)))
    return out


def _py_hash(order, shortlist, remaining):
    pos = {p: i for i, p in enumerate(order)}
    words = {}
    for c in shortlist:
        words[pos[c] >> 6] = words.get(pos[c] >> 6, 0) | (1 << (pos[c] & 63))
    M = (1 << 64) - 1

    def mul(w):  # csrc/wave.hpp audit_mul
        x = ((w + 1) * 0x9E3779B97F4A7C15) & M
        x ^= x >> 29
        x = (x * 0xBF58476D1CE4E5B9) & M
        x ^= x >> 32
        return x | 1
    h = 0
    for w, b in words.items():
        h = (h + b * mul(w)) & M
    return ((h ^ (h >> 32)) & 0xFFFFFFFF) ^ ((remaining * 0x9E3779B1) & 0xFFFFFFFF)


@pytest.mark.parametrize("profile", [None, "full", "prefer"])
@pytest.mark.parametrize("seed", range(12))
def test_place_c_vs_python(seed, profile):
    P = int(np.random.default_rng(seed).choice([1, 5, 40, 64, 130]))
    fleet = wl.fuzz_fleet(seed, pods=P, models=120, profile=profile)
    reqs, extra = wl.fuzz_requests(fleet, seed, 250)
    _compare(fleet, reqs, extra)


def test_place_c_vs_python_scenarios():
    for name, fleet, reqs, extra in wl.scenario_fleets():
        _compare(fleet, reqs, extra)


def _compare(fleet, reqs, extra):
    P = fleet.n_pods
    orc = ob.OracleFleet(fleet)
    want = orc.place(reqs, extra, fleet.now)

    mesh = po.Mesh(fleet.min_space_units, fleet.min_churn_age_ms, fleet.now)
    pods = _py_pods(fleet)
    order = mesh.sorted_cluster_state(pods)
    assert order == list(orc.order)
    live = {i for i in range(P) if fleet.pods["flags"][i] & 2}
    rs = set(int(x) for x in fleet.replaced_rs)
    T = fleet.n_types
    al = ob.unpack_bitmap(fleet.allowed, P) if T else None
    pf = ob.unpack_bitmap(fleet.prefer, P) if T else None
    for i, r in enumerate(reqs):
        m = fleet.models[r["model"]]
        t = int(m["type"])
        constrain = set(np.nonzero(al[t])[0]) if T and fleet.has_allowed[t] else None
        prefer = set(np.nonzero(pf[t])[0]) if T and fleet.has_prefer[t] else None
        ents = fleet.ent_pod[m["ent_off"]: m["ent_off"] + m["n_loaded"] + m["n_failed"]]
        loaded = set(int(x) for x in ents[: m["n_loaded"]])
        failed = set(int(x) for x in ents[m["n_loaded"]:])
        tried = set(int(x) for x in extra[r["extra_off"]: r["extra_off"] + r["n_extra"]])
        fresh = dict(lru_time=int(r["fresh_lru"]), capacity=int(r["fresh_capacity"]), used=int(r["fresh_used"]),
                     count=int(r["fresh_count"]), rpm=int(r["fresh_rpm"]))
        chosen, best, shortlist, remaining = po.get_next(
            mesh, pods, order, live, rs, constrain, prefer, [tried, loaded, failed], int(r["self_pod"]),
            bool(r["flags"] & 1), fresh, int(r["last_used"]), int(r["pick"]))
        w = want[i]
        assert (chosen, best, len(shortlist)) == (w["chosen"], w["best"], w["n_candidates"]), (i, r, w)
        if shortlist:
            assert _py_hash(order, shortlist, remaining) == w["hash"], (i, shortlist, remaining)


def test_serve_c_vs_python():
    rng = np.random.default_rng(5)
    P = 12
    live_arr = (rng.random(P) < 0.85).astype(np.uint8)
    live = set(np.nonzero(live_arr)[0])
    now = wl.NOW_MS
    for _ in range(3000):
        k = int(rng.integers(0, 5))
        pods = rng.choice(P, size=k, replace=False)
        times = now - rng.choice([100, 2_999, 3_000, 3_001, 9_000], k)
        in_use = rng.integers(0, 3, P).astype(np.int32)
        lu = (now - rng.choice([0, 5, 5, 70], P)).astype(np.int64)
        self_id = int(rng.integers(-1, P))
        ex, pr = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        lif, lit = int(rng.integers(0, 3)), int(now - rng.choice([0, 5, 70]))
        got = ob.serve(self_id, ex, pr, pods, times, now, 3000, lif, lit, live_arr, in_use, lu)
        want = po.serve_get_next(list(zip(map(int, pods), map(int, times))), self_id, ex, pr, live, now, 3000, lif,
                                 lit, in_use, lu)
        assert got[0] == want[0]
        if want[0] != -1:
            assert got[1] == want[1]


def test_cache_ops_c_vs_python():
    """Random sequences of putIfAbsent / get / weight update / remove on both clhm models."""
    rng = np.random.default_rng(9)
    now = wl.NOW_MS
    for trial in range(40):
        cap = int(rng.choice([50, 200, 1000]))
        c, p = ob.CCache(cap), po.Clhm(cap)
        for step in range(200):
            op = rng.integers(0, 10)
            key = int(rng.integers(0, 30))
            t = int(now + step - rng.choice([0, 0, 5, 5, 50, 1000]))
            if op < 5:
                w = int(rng.choice([1, 5, 20, 60]))
                lu = 0 if rng.random() < 0.3 else t
                assert c.put_if_absent(key, w, lu, now + step) == p.putIfAbsent(key, w, lu, now + step)
            elif op < 7:
                lu = 0 if rng.random() < 0.5 else t
                assert c.get(key, lu, now + step) == p.get(key, lu, now + step)
            elif op < 9:
                w = int(rng.choice([1, 5, 20, 60, 300]))
                nt = int(rng.choice([-1, 0, t]))
                assert c.update_weight(key, w, nt, now + step) == p.updateWeight(key, w, nt, now + step)
            else:
                assert c.remove(key) == p.remove(key)
            assert c.keys() == p.keys() and c.weighted_size == p.weightedSize
            assert c.oldest_time() == p.oldestTime()


def test_fuzz_fleets_reach_every_branch_of_get_next():
    """Run after the parametrised cases above in the same process: the union of the fuzz fleets must
    have visited every branch of getNext that the kernel has to reproduce."""
    if len(po.BRANCHES) == 0:
        pytest.skip("run together with test_place_c_vs_python")
    expected = {"none_no_candidates", "retry_without_replica_sets", "case_a_found", "case_a_stop_at_full",
                "case_a_not_found", "case_b_age_break", "case_b_preferred", "case_b_no_preferred",
                "case_b_self_null", "self_is_best", "skip_non_preferred", "full_mode", "full_mode_self",
                "break_lru", "break_rem", "break_rem_self", "break_count", "self_in_shortlist",
                "rpm_nulled_first", "rpm_nulled_other", "rpm_break_at_one", "older_than_five_days",
                "chosen_is_self"}
    missing = expected - po.BRANCHES
    assert not missing, f"fuzz fleets never reached: {sorted(missing)}"


@pytest.mark.parametrize("seed", range(6))
def test_unload_buffer_manager_ops_c_vs_python(seed):
    """Random ModelCacheUnloadBufManager method sequences (:130-349, including removeEntry,
    discardFailedEntry, insertFailedPlaceholderEntry and the capacity-shrinking unloadComplete(failure))
    through both restatements: evictions in listener order, deque, and manager fields must agree."""
    import ctypes as C
    rng = np.random.default_rng(900 + seed)
    NOW = 1_760_000_000_000
    cap, reserved = int(rng.choice([5_000, 25_600, 131_072])), int(rng.choice([0, 256, 2_560]))
    h = ob.CCache(cap, reserved, NOW)
    pc = po.Clhm(cap)
    pu = po.UnloadBufManager(pc, reserved, NOW)
    nxt = 0
    for step in range(400):
        live = [k for k in pc.keys() if k != po.UnloadBufManager.KEY]
        known = int(rng.choice(live)) if live and rng.random() < 0.8 else 99_000 + int(rng.integers(0, 5))
        t = int(rng.choice([0, NOW - 500, NOW - 2_000, NOW - 7_200_000]))
        op = int(rng.choice([4, 4, 4, 5, 6, 7, 8, 9, 10, 11, 12, 1]))
        w = int(rng.choice([1, 1, 640, 2_560]))
        before = len(pu.evicted)
        if op in (4, 12):
            key = nxt if rng.random() < 0.85 else known
            nxt += 1
            r_c, ev_c = h.apply(op, key, w, t, 0, NOW)
            r_p = (pu.insertNewEntry if op == 4 else pu.insertFailedPlaceholderEntry)(key, w, t, NOW)
        elif op == 5:
            if known not in pu.weights:
                continue
            inc = int(rng.choice([639, 2_559]))
            r_c, ev_c = h.apply(op, known, inc, 0, 0, NOW)
            pu.adjustNewEntrySpaceRequest(inc, known, NOW)
            r_p = r_c
        elif op in (6, 7):
            req = int(rng.choice([1, 640, 100_000]))
            r_c, ev_c = h.apply(op, 0, req, 0, 0, NOW)
            r_p = pu.cacheSpaceIsReady(req) if op == 6 else pu.claimRequestedSpaceIfReady(req, NOW)
        elif op == 8:
            if known not in pu.weights:
                continue
            delta = int(rng.choice([-600, -1, 1, 700, 20_000]))
            if pu.weights[known] + delta <= 0:
                continue
            r_c, ev_c = h.apply(op, known, delta, 0, 0, NOW)
            pu.adjustWeightAfterLoad(delta, known, NOW)
            r_p = r_c
        elif op == 9:
            ok = int(rng.random() < 0.8)
            r_c, ev_c = h.apply(op, 0, 640, 0, ok, NOW)
            pu.unloadComplete(640, bool(ok), NOW)
            r_p = r_c
        elif op == 10:
            r_c, ev_c = h.apply(op, known, 0, 0, 0, NOW)
            r_p = pu.removeEntry(known, NOW)
        elif op == 11:
            r_c, ev_c = h.apply(op, 0, 1, 0, 0, NOW)
            pu.discardFailedEntry(1, NOW)
            r_p = r_c
        else:
            r_c, ev_c = h.apply(1, known, 0, t, 0, NOW)
            r_p = pc.get(known, t, NOW)
        assert int(r_c) == int(r_p), (seed, step, op)
        assert ev_c == [-1000000 if k == po.UnloadBufManager.KEY else k for k, _ in pu.evicted[before:]], (seed, step, op)
        assert h.keys() == [k for k in pc.keys() if k != po.UnloadBufManager.KEY]
        assert (h.u.total_unloading, h.u.total_occupancy, h.u.cache_deficit, h.c.weighted_size, h.c.capacity) == \
            (pu.totalUnloadingWeight, pu.totalModelCacheOccupancy, pu.cacheDeficit, pc.weightedSize, pc.capacity)
        h.u.n_evicted = 0


def test_relabelling_instances_renames_the_decisions():
    """getNext never looks at an instance's index, only at its record and its id ORDER: with the instance table
    permuted (rows moved, instanceIds / self / exclusions renamed, id_order kept with the row) the oracle picks the
    renamed instance with the same shortlist size and audit hash.  (The GPU suite runs the same property over all
    1M decisions of C4, test_c4_full_size_relabelling_and_batch_split.)"""
    from modelmesh_amd.solver import bitmap_from_bool
    fleet = wl.make_fleet("C3", models=5000, pods=700)
    reqs, extra = wl.make_requests(fleet, 14)
    whole = ob.OracleFleet(fleet).place(reqs, extra, fleet.now)
    P = fleet.n_pods
    new_of = np.random.default_rng(99).permutation(P).astype(np.int32)
    old_of = np.empty(P, np.int32)
    old_of[new_of] = np.arange(P, dtype=np.int32)
    f2 = wl.make_fleet("C3", models=5000, pods=700)
    f2.pods, f2.ent_pod = fleet.pods[old_of].copy(), new_of[fleet.ent_pod]
    f2.allowed = bitmap_from_bool(ob.unpack_bitmap(fleet.allowed, P)[:, old_of])
    f2.prefer = bitmap_from_bool(ob.unpack_bitmap(fleet.prefer, P)[:, old_of])
    r2 = reqs.copy()
    r2["self_pod"] = np.where(reqs["self_pod"] >= 0, new_of[np.clip(reqs["self_pod"], 0, P - 1)], reqs["self_pod"])
    got = ob.OracleFleet(f2).place(r2, new_of[extra] if len(extra) else extra, f2.now)
    for f in ("chosen", "best"):
        assert np.array_equal(got[f], np.where(whole[f] >= 0, new_of[np.clip(whole[f], 0, P - 1)], whole[f]))
    assert np.array_equal(got["n_candidates"], whole["n_candidates"]) and np.array_equal(got["hash"], whole["hash"])


@pytest.mark.parametrize("seed", range(6))
def test_request_guards_c_vs_python(seed):
    """The guards around the two selections (goLocal MM.java:3598-3626, the failure / location caps :4590-4627, the
    churn guard :3870-3884, loadLocal's size prediction and early reject :5158-5197, onEviction's reload rule
    :2886-2920, the publish hysteresis :5388-5468) in both restatements on random and boundary inputs."""
    import ctypes as C
    from oracle import py_gates as pg
    lib = ob.load()
    rng = np.random.default_rng(9100 + seed)
    now = 1_760_000_000_000
    JMAX = (1 << 63) - 1

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        return a, (a.ctypes.data_as(C.c_void_p) if len(a) else None)

    seen = {k: set() for k in ("go", "fail", "loc", "churn", "reject", "reload", "publish")}
    for _ in range(1500):
        # goLocal
        k = int(rng.integers(0, 5))
        pods = rng.choice(12, size=k, replace=False)
        times = rng.choice([0, now - 100, now - 1400, now - 1600, now - 1600, now - 50_000, now + 10], size=k)
        self_pod = int(pods[rng.integers(0, k)]) if k and rng.random() < 0.7 else 11 - int(rng.integers(0, 3))
        fs, hc, dn = (int(rng.random() < 0.6) for _ in range(3))
        cp, cpp = arr(pods, np.int32)
        ct, ctp = arr(times, np.int64)
        got = bool(lib.orc_go_local(cpp, ctp, k, self_pod, fs, hc, dn, now))
        want = pg.go_local(list(zip(map(int, pods), map(int, times))), self_pod, fs, hc, dn, now)
        assert got == want, ("go_local", pods, times, self_pod, fs, hc, dn)
        seen["go"].add(got)
        # failure cap
        nf = int(rng.integers(0, 6))
        ft, ftp = arr(now - rng.choice([10, 449_999, 450_000, 450_001, 3_000_000], size=nf), np.int64)
        got = bool(lib.orc_load_failures_breached(ftp, nf, now, 450_000))
        assert got == pg.load_failures_breached([int(x) for x in ft], now, 450_000), ("failures", ft)
        seen["fail"].add(got)
        # location cap
        nl = int(rng.integers(0, 9))
        lp, lpp = arr(rng.choice(12, size=nl, replace=False), np.int32)
        ex, exp_ = arr(rng.choice(12, size=int(rng.integers(0, 4)), replace=False), np.int32)
        it, itp = arr(rng.random(12) < 0.85, np.uint8)
        got = bool(lib.orc_load_locations_breached(lpp, nl, exp_, len(ex), itp))
        assert got == pg.load_locations_breached([int(x) for x in lp], set(int(x) for x in ex), [bool(x) for x in it]), "locations"
        seen["loc"].add(got)
        # churn guard
        cap = int(rng.choice([131072, 1_000_000]))
        ws = int(cap * rng.choice([0.1, 0.94, 0.96, 0.999, 1.0, 1.2]))
        oldest = int(rng.choice([-1, 0, JMAX, now - 1_000, now - 599_999, now - 600_000, now - 600_001, now - 4_500_000]))
        mca = int(rng.choice([0, 1, 600_000]))
        got = bool(lib.orc_churn_reject(mca, 6553, cap, ws, oldest, now))
        assert got == pg.churn_reject(mca, 6553, cap, ws, oldest, now), ("churn", mca, cap, ws, oldest)
        seen["churn"].add(got)
        # loadLocal size prediction + early reject
        st = np.zeros(1, dtype=ob.ORC_STATS)
        st["total_capacity"] = int(rng.choice([0, 10_000_000, 5_000_000_000, 9_000_000_000]))
        st["total_free"] = int(rng.choice([0, 1_000_000, 4_000_000_000]))
        st["model_copy_count"] = int(rng.choice([0, 9, 10, 1000, 300_000]))
        st["instance_count"] = int(rng.choice([0, 1, 2, 50]))
        sd = {n: int(st[n][0]) for n in st.dtype.names}
        hh, hint = int(rng.random() < 0.3), int(rng.choice([0, 1, 6400, 2_000_000, -5]))
        lc, cut, pred = int(rng.integers(0, 20)), 10, int(rng.choice([6400, 1, 200_000, 0]))
        wc, lu = int(rng.random() < 0.7), int(rng.choice([0, now - 5_000, now - 4_000_000, -3]))
        rej = C.c_int(0)
        got = int(lib.orc_load_local_initial_size(hh, hint, lc, cut, pred, st.ctypes.data_as(C.c_void_p), wc, lu, cap, ws, oldest,
                                                  C.byref(rej)))
        want, wrej = pg.load_local_initial_size(hh, hint, lc, cut, pred, sd, wc, lu, cap, ws, oldest)
        assert (got, bool(rej.value)) == (want, wrej), ("initial size", hh, hint, lc, pred, sd, wc, lu, cap, ws, oldest)
        seen["reject"].add(bool(rej.value))
        # onEviction reload rule
        ef = int(rng.random() < 0.2)
        lt = int(rng.choice([-1, now - 1_000, now - 180_000, now - 180_001, now - 5_000_000]))
        got = bool(lib.orc_reload_elsewhere(ef, lt, 90_000, now, st.ctypes.data_as(C.c_void_p)))
        assert got == pg.reload_elsewhere(ef, lt, 90_000, now, sd), ("reload", ef, lt, sd)
        seen["reload"].add(got)
        # publish hysteresis
        cur = np.zeros(1, dtype=ob.ORC_POD)
        cur["capacity"] = int(rng.choice([131072, 1_000_000]))
        cur["used"] = int(cur["capacity"][0] * rng.choice([0.0, 0.5, 0.94, 0.96]))
        cur["lru_time"] = int(rng.choice([JMAX, now - 30_000, now - 400_000, now - 4_000_000]))
        cur["count"], cur["loading_threads"] = int(rng.choice([0, 5, 40, 100])), int(rng.choice([3, 8]))
        cur["loading_in_progress"], cur["rpm"] = int(rng.choice([0, 1, 8, 9, 12])), int(rng.choice([0, 50, 1000]))
        cur["shutting_down"] = int(rng.random() < 0.1)
        fresh = cur.copy()
        if rng.random() < 0.8:
            fresh["capacity"] = int(cur["capacity"][0] - rng.choice([0, 100, 50_000]))
            fresh["used"] = int(cur["used"][0] * rng.choice([1.0, 1.1, 1.25]) + rng.choice([0, 0, 1]))
            fresh["lru_time"] = int(cur["lru_time"][0] - rng.choice([0, 0, 10_000, 19_999, 20_000, 30_000])) \
                if cur["lru_time"][0] != JMAX else int(rng.choice([JMAX, now - 1000]))
            fresh["count"] = int(cur["count"][0] + rng.choice([0, 0, 1, 5, 9, 10, 16]))
            fresh["loading_threads"] = int(rng.choice([cur["loading_threads"][0], 3]))
            fresh["loading_in_progress"] = int(cur["loading_in_progress"][0] + rng.choice([0, 0, 1, 2, 3, -1]))
            fresh["rpm"] = int(cur["rpm"][0] + rng.choice([0, 0, 5, 99, 100, 120]))
            fresh["shutting_down"] = int(rng.random() < 0.1)
        absent = rng.random() < 0.1
        lp_ = now - int(rng.choice([500, 1_999, 2_000, 38_999, 39_000, 100_000, 160_000, 160_001, 170_000]))
        force, pre = int(rng.random() < 0.5), int(rng.random() < 0.15)
        got = bool(lib.orc_should_publish(None if absent else cur.ctypes.data_as(C.c_void_p), fresh.ctypes.data_as(C.c_void_p), now,
                                          lp_, force, pre, 6553))
        cd = None if absent else {n: int(cur[n][0]) for n in cur.dtype.names}
        want = pg.should_publish(cd, {n: int(fresh[n][0]) for n in fresh.dtype.names}, now, lp_, force, pre, 6553)
        assert got == want, ("publish", cd, fresh, now - lp_, force, pre)
        seen["publish"].add(got)
    assert all(v == {False, True} for v in seen.values()), seen


@pytest.mark.parametrize("seed,pods,used", [(0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (3, 1, 0.5), (4, 60, 0.99)])
def test_rebalancers_c_vs_python(seed, pods, used):
    """rateTrackingTask's scale-up (MM.java:5636-5856), the janitor's scale-down (:6110-6335) and preShutdown's
    migration order (:6985-7040) in both restatements, on the fleets and local caches the GPU parity test uses."""
    from modelmesh_amd import _lib
    from oracle import py_rebalance as pr
    from tests.test_rebalance_gpu import _local_entries, _rebalance_fleet
    fleet, rng = _rebalance_fleet(seed, pods, 600, used)
    now = fleet.now
    entries = _local_entries(fleet, 0, rng, 800)
    orc = ob.OracleFleet(fleet)
    order = [int(x) for x in orc.order]
    pos_of = {p: i for i, p in enumerate(order)}
    BIG = 2**31 - 1
    ppods = [dict(rpm=int(r["rpm"]), shutting_down=bool(r["flags"] & 1), in_table=not bool(r["flags"] & 4)) for r in fleet.pods]
    pmodels = []
    for m in fleet.models:
        o, k, f = int(m["ent_off"]), int(m["n_loaded"]), int(m["n_failed"])
        pmodels.append(dict(type=int(m["type"]), last_used=int(m["last_used"]),
                            loaded=[(int(fleet.ent_pod[o + i]), int(fleet.ent_time[o + i])) for i in range(k)],
                            failed=[(int(fleet.ent_pod[o + k + i]), int(fleet.ent_time[o + k + i])) for i in range(f)]))
    pent = [dict(model=int(e["model"]), weight=int(e["weight"]), last_used=int(e["last_used"]),
                 interval_count=int(e["interval_count"]), last_heavy_time=int(e["last_heavy_time"]),
                 last_unload_time=int(e["last_unload_time"]), earlier_use_iteration=int(e["earlier_use_iteration"]),
                 last_used_iteration=int(e["last_used_iteration"]), failed=bool(e["flags"] & 1)) for e in entries]
    sd = lambda st: {n: int(st[n]) for n in st.dtype.names}  # noqa: E731
    gstats = sd(orc.stats())
    tstats = [sd(t) for t in ob.type_set_stats(fleet)]
    seen = set()
    for thr, our_rpm, last_check in ((2000, 100, now - 10_000), (100, 50_000, now - 9_000), (0, 0, now - 10_000),
                                     (2000, 0, now - 1_000), (100, 0, now - 12_000)):
        sp = np.zeros(1, dtype=_lib.SCALEUP_PARAMS)
        sp["self_pod"], sp["iteration_counter"] = 0, 130
        sp["second_copy_max_age_iters"], sp["second_copy_min_age_iters"] = 240, 42
        sp["scale_up_rpm_threshold"], sp["our_rpm"] = thr, our_rpm
        sp["now"], sp["last_check_time"], sp["rate_check_interval_ms"] = now, last_check, 10_000
        sp["second_copy_lru_threshold_ms"], sp["assume_completed_ms"] = 72_000_000, 30_000
        w_out, w_ov, w_sk = ob.scaleup_plan(fleet, entries, sp.view(ob.ORC_SCALEUP_PARAMS))
        g_out, g_ov, g_sk = pr.scaleup(ppods, order, gstats, tstats, bool(fleet.n_types), pmodels, pent,
                                       {n: int(sp[n][0]) for n in sp.dtype.names})
        assert bool(w_sk) == g_sk
        if g_sk:
            continue
        for f in ("action", "copies", "timestamp", "new_i1", "new_i2", "heavy", "rpm"):
            got = np.array([o[f] for o in g_out])
            assert np.array_equal(got, w_out[f]), (f, thr, np.nonzero(got != w_out[f])[0][:5])
        if np.any(w_out["action"] == 2):
            assert set(np.nonzero(w_ov)[0]) == g_ov
        seen |= set(int(a) for a in w_out["action"])
    if pods >= 200:
        assert seen >= {0, 1, 2}

    removed_any = False
    for thr, cap, shut in ((2000, 200_000, 0), (2000, 1_000, 0), (10, 10_000_000, 0), (2000, 200_000, 1)):
        dp = np.zeros(1, dtype=_lib.SCALEDOWN_PARAMS)
        dp["self_pod"], dp["shutting_down"], dp["now"] = 0, shut, now
        dp["last_check_time"], dp["rate_check_interval_ms"] = now - 7_000, 10_000
        dp["adjusted_cache_capacity"], dp["scale_up_rpm_threshold"] = cap, thr
        want = ob.scaledown_plan(fleet, entries, dp.view(ob.ORC_SCALEDOWN_PARAMS))
        # instanceSetStats(): this instance's partition (bind.scaledown_plan does the same for the C call)
        st = gstats
        if fleet.n_types:
            pts, _, pst = ob.partition_stats(fleet)
            st = sd(pst[int(pts[0])]) if int(pts[0]) >= 0 else dict(total_capacity=0, total_free=0, global_lru=2**63 - 1,
                                                                    instance_count=0, model_copy_count=0)
        got = pr.scaledown(ppods, {p: pos_of.get(p, BIG) for p in range(fleet.n_pods)}, st, pmodels, pent,
                           {n: int(dp[n][0]) for n in dp.dtype.names})
        assert np.array_equal(np.array(got, np.uint8), want), (thr, cap, shut, np.nonzero(np.array(got, np.uint8) != want)[0][:5])
        removed_any |= bool(want.any())
    if pods >= 60 and used > 0.9:
        assert removed_any

    wa, ww = ob.migration_plan(fleet, entries, 0, now)
    ga, gw = pr.migration(pmodels, pent, 0, now, 3_600_000)
    assert np.array_equal(np.array(ga, np.uint8), wa) and np.array_equal(np.array(gw, np.uint8), ww)


def test_latency_based_rebalancers_c_vs_python():
    """limitModelConcurrency == true: getRpmScaleThreshold (MM.java:2766-2796) on random MaxConcCacheEntry rows — including
    extreme counters, negative prior counts and wrapping products —, the latency-based rate task (:5677-5818, :5836) and the
    janitor over MaxConcCacheEntry candidates (:6294-6305), in both restatements (the reference-text vectors of
    tests/test_ref_vectors.py pin the C one; this keeps the second, independent one equal to it)."""
    from modelmesh_amd import _lib
    from oracle import py_rebalance as pr
    import ctypes as C
    from tests import ref_fleets as rf
    lib = ob.load()
    rng = np.random.default_rng(77)
    n = 4000
    rows = rf.conc_rows(rng, n)
    wild = rng.random(n) < 0.3  # beyond anything a mesh produces: the arithmetic must still be Java's
    rows["count_and_time_sum"] = np.where(wild, rng.integers(-2**63, 2**63 - 1, n), rows["count_and_time_sum"])
    rows["prior_sum"] = np.where(wild, rng.integers(-2**40, 2**62, n), rows["prior_sum"])
    rows["max_conc"] = np.where(wild, rng.integers(-5, 2**31 - 1, n), rows["max_conc"])
    lib.orc_rpm_scale_threshold.restype = C.c_int32
    lib.orc_rpm_scale_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_void_p]
    for i in range(n):
        for and_reset in (0, 1):
            dyn = int(rng.choice([540_000, 60_000, 1, 2**40]))
            co = np.zeros(1, dtype=_lib.CONC_OUT)
            co["new_prior_sum"], co["new_prior_count"] = rows["prior_sum"][i], rows["prior_count"][i]
            got = lib.orc_rpm_scale_threshold(rows[i:i + 1].ctypes.data_as(C.c_void_p), and_reset, 2000, dyn, co.ctypes.data_as(C.c_void_p))
            m = {k: int(rows[k][i]) for k in rows.dtype.names}
            want = pr.rpm_scale_threshold(m, bool(and_reset), 2000, dyn)
            assert (got, int(co["reset"][0]), int(co["new_prior_sum"][0]), int(co["new_prior_count"][0])) == want, (m, and_reset, dyn)

    for name, fleet, ids, entries, conc, sp, cp in rf.scaleup_conc_cases():
        if not name.endswith(("_1", "_3")):
            continue
        orc = ob.OracleFleet(fleet)
        order = [int(x) for x in orc.order]
        ppods = [dict(rpm=int(r["rpm"]), shutting_down=bool(r["flags"] & 1), in_table=not bool(r["flags"] & 4)) for r in fleet.pods]
        pmodels = []
        for m in fleet.models:
            o, k, f = int(m["ent_off"]), int(m["n_loaded"]), int(m["n_failed"])
            pmodels.append(dict(type=int(m["type"]), last_used=int(m["last_used"]),
                                loaded=[(int(fleet.ent_pod[o + i]), int(fleet.ent_time[o + i])) for i in range(k)],
                                failed=[(int(fleet.ent_pod[o + k + i]), int(fleet.ent_time[o + k + i])) for i in range(f)]))
        pent = [{f: int(e[f]) for f in ("model", "weight", "last_used", "interval_count", "last_heavy_time", "last_unload_time",
                                        "earlier_use_iteration", "last_used_iteration")} for e in entries]
        pconc = [{k: int(r[k]) for k in conc.dtype.names} for r in conc]
        sd = lambda st: {n_: int(st[n_]) for n_ in st.dtype.names}  # noqa: E731
        w_out, w_co, w_ov, w_sk, w_res = ob.scaleup_plan_conc(fleet, entries, conc, sp.view(ob.ORC_SCALEUP_PARAMS), cp)
        g_out, g_ov, g_sk, g_res = pr.scaleup(ppods, order, sd(orc.stats()), [sd(t) for t in ob.type_set_stats(fleet)], bool(fleet.n_types),
                                              pmodels, pent, {n_: int(sp[n_][0]) for n_ in sp.dtype.names}, pconc,
                                              dict(dynamic_rpm_scale_constant=int(cp["dynamic_rpm_scale_constant"][0]),
                                                   average_model_parallelism=float(cp["average_model_parallelism"][0])))
        assert bool(w_sk) == g_sk, name
        for f in ("action", "copies", "timestamp", "new_i1", "new_i2", "heavy", "rpm"):
            assert np.array_equal(np.array([o[f] for o in g_out]), w_out[f]), (name, f)
        for f in ("threshold", "reset", "new_prior_sum", "new_prior_count"):
            assert np.array_equal(np.array([o[f] for o in g_out]), w_co[f]), (name, f)
        assert g_res["average_model_parallelism"] == float(w_res["average_model_parallelism"]), name
        assert g_res["model_parallelism_sum"] == int(w_res["model_parallelism_sum"]), name
        if np.any(w_out["action"] == 2):
            assert set(np.nonzero(w_ov)[0]) == g_ov, name
    for name, fleet, ids, entries, conc, dp, dyn in rf.scaledown_conc_cases():
        orc = ob.OracleFleet(fleet)
        pos_of = {int(p): i for i, p in enumerate(orc.order)}
        ppods = [dict(rpm=int(r["rpm"]), shutting_down=bool(r["flags"] & 1), in_table=not bool(r["flags"] & 4)) for r in fleet.pods]
        pmodels = []
        for m in fleet.models:
            o, k, f = int(m["ent_off"]), int(m["n_loaded"]), int(m["n_failed"])
            pmodels.append(dict(type=int(m["type"]), last_used=int(m["last_used"]),
                                loaded=[(int(fleet.ent_pod[o + i]), int(fleet.ent_time[o + i])) for i in range(k)],
                                failed=[(int(fleet.ent_pod[o + k + i]), int(fleet.ent_time[o + k + i])) for i in range(f)]))
        pent = [{f: int(e[f]) for f in ("model", "weight", "last_used", "interval_count", "last_heavy_time", "last_unload_time",
                                        "earlier_use_iteration", "last_used_iteration")} for e in entries]
        pconc = [{k: int(r[k]) for k in conc.dtype.names} for r in conc]
        st = {n_: int(v) for n_, v in zip(ob.ORC_STATS.names, ob.instance_set_stats(fleet, 0, orc)[0])}
        want = ob.scaledown_plan_conc(fleet, entries, conc, dp.view(ob.ORC_SCALEDOWN_PARAMS), dyn)
        got = pr.scaledown(ppods, {p: pos_of.get(p, 2**31 - 1) for p in range(fleet.n_pods)}, st, pmodels, pent,
                           {n_: int(dp[n_][0]) for n_ in dp.dtype.names}, pconc, dyn)
        assert np.array_equal(np.array(got, np.uint8), want), name


@pytest.mark.parametrize("seed,pods,models,used", [(0, 8, 300, 0.5), (1, 64, 3000, 0.2), (2, 300, 4000, 0.9), (3, 300, 4000, 0.99),
                                                  (5, 5, 50, 0.0)])
def test_proactive_plan_c_vs_python(seed, pods, models, used):
    """The reaper's proactive loads (MM.java:6455-6488, :6574-6577, :6616-6747) in both restatements, cluster-wide and
    per ProhibitedTypeSet partition with the skip list, on the GPU parity test's fleets."""
    from oracle import py_rebalance as pr
    from tests.test_rebalance_gpu import _plan_fleet
    fleet = _plan_fleet(seed, pods, models, used)
    now = fleet.now
    orc = ob.OracleFleet(fleet)
    sd = lambda st: {n: int(st[n]) for n in st.dtype.names}  # noqa: E731
    ppods = [dict(capacity=int(r["capacity"]), used=int(r["used"]), loading_threads=int(r["loading_threads"]),
                  loading_in_progress=int(r["loading_in_progress"]), shutting_down=bool(r["flags"] & 5)) for r in fleet.pods]
    pmodels = []
    for m in fleet.models:
        o, k, f = int(m["ent_off"]), int(m["n_loaded"]), int(m["n_failed"])
        pmodels.append(dict(type=int(m["type"]), last_used=int(m["last_used"]), loaded=list(range(k)), failed=list(range(f))))
    g = sd(orc.stats())
    plans = [(-1, g, None, None)]
    if fleet.n_types:
        pts, sets, pst = ob.partition_stats(fleet)
        plans += [(k, sd(pst[k]), [int(x) == k for x in pts], set(int(t) for t in sets[k])) for k in range(len(sets))]
    for default_units in (6400, 1):
        taken = set()
        for k, st, in_subset, prohibited in plans:
            wm, wl_, wi = ob.proactive_plan(fleet, default_units, now, models, partition=k,
                                            skip_models=np.array(sorted(taken), np.int32) if (k >= 0 and taken) else None)
            sel, info = pr.proactive(ppods, g, st, in_subset, prohibited, taken if k >= 0 else None, pmodels, default_units, now)
            for f in ("size_estimate", "free_count", "total_count", "error", "space_to_fill", "cutoff"):
                assert info[f] == int(wi[f]), (k, f, info, wi)
            if info["error"]:
                assert sel is None and len(wm) == 0
                continue
            assert info["n_candidates"] == int(wi["n_candidates"]) and info["n_selected"] == int(wi["n_selected"])
            assert [i for i, _ in sel] == list(wm) and [lu for _, lu in sel] == list(wl_), k
            if k >= 0:
                taken |= set(int(x) for x in wm)
