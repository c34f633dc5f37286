"""GPU parity for the request-level guards around instance selection (SURVEY.md §8 rows a10, a11,
a14, a20): goLocal MM.java:3603-3626, checkLoadFailureCount :4607-4627, checkLoadLocationCount
:4590-4604, throwIfLocalLoadNotAllowed :4003-4042, churn guard :3870-3884, loadLocal size prediction
and early reject :5158-5197, onEviction reload rule :2886-2920, publish hysteresis :5397-5468."""
import ctypes as C

import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle import bind as ob

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if len(a) else None


@pytest.mark.parametrize("seed", range(4))
def test_gates_match_oracle(seed):
    rng = np.random.default_rng(3000 + seed)
    fleet = wl.fuzz_fleet(seed + 40, pods=int(rng.choice([8, 64, 300])), models=300,
                          profile=None if seed % 2 else "prefer")
    P, now = fleet.n_pods, fleet.now
    fleet.ent_time[:] = now - rng.choice([100, 1_400, 1_600, 449_000, 451_000, 3_000_000], len(fleet.ent_time))
    # make some models have many copies / failures so the count guards trigger
    m = fleet.models
    big = np.nonzero(rng.random(len(m)) < 0.1)[0]
    ent_pod, ent_time = list(fleet.ent_pod), list(fleet.ent_time)
    for i in big:
        k, f = int(rng.integers(4, min(8, P) + 1)) if P >= 5 else min(P, 2), int(rng.integers(0, min(8, P)))  # (the kernel prefetches 4 of each)
        f = min(f, P - k) if P - k > 0 else 0
        pods = rng.choice(P, size=k + f, replace=False)
        m["ent_off"][i], m["n_loaded"][i], m["n_failed"][i] = len(ent_pod), k, f
        ent_pod += list(pods)
        ent_time += list(now - rng.choice([100, 449_000, 451_000], k + f))
    fleet.ent_pod = np.array(ent_pod, np.int32)
    fleet.ent_time = np.array(ent_time, np.int64)

    n = 3000
    r = np.zeros(n, dtype=_lib.GATE_REQ)
    r["model"] = rng.integers(0, fleet.n_models, n)
    r["self_pod"] = rng.integers(0, P, n)
    mm = m[r["model"]]
    has = mm["n_loaded"] > 0
    pick = (mm["ent_off"] + rng.integers(0, 8, n) % np.maximum(mm["n_loaded"], 1)).clip(0, len(fleet.ent_pod) - 1)
    r["self_pod"] = np.where(has & (rng.random(n) < 0.6), fleet.ent_pod[pick], r["self_pod"])
    r["flags"] = rng.integers(0, 512, n)
    r["size_hint"] = rng.choice([0, 1, 6400, 2_000_000], n)
    r["last_used_time"] = rng.choice([0, now - 5_000, now - 4_000_000], n)
    r["cache_capacity"] = rng.choice([131072, 1_000_000], n)
    r["cache_weighted_size"] = (r["cache_capacity"] * rng.choice([0.1, 0.96, 0.999, 1.0], n)).astype(np.int64)
    r["cache_oldest_time"] = rng.choice([-1, _lib.JAVA_LONG_MAX, now - 1_000, now - 4_500_000, now - 700_000], n)
    r["loader_predicted"] = rng.choice([6400, 1, 200_000], n)
    r["loading_count"] = rng.integers(0, 20, n)
    r["weight_predict_cutoff"] = 10
    r["loaded_time"] = rng.choice([-1, now - 1_000, now - 200_000, now - 5_000_000], n)
    r["load_timeout_ms"] = rng.choice([90_000, 720_000], n)
    cur = fleet.pods[r["self_pod"]]
    near = rng.random(n) < 0.6
    r["fresh_lru"] = np.where(near, cur["lru_time"], cur["lru_time"] - rng.choice([0, 10_000, 30_000], n))
    r["fresh_capacity"] = np.where(near, cur["capacity"], cur["capacity"] - rng.choice([0, 100, 50_000], n))
    r["fresh_used"] = np.where(near, cur["used"], (cur["used"] * rng.choice([1.0, 1.1, 1.3], n)).astype(np.int64))
    r["fresh_count"] = cur["count"] + rng.choice([0, 0, 1, 2, 10], n)
    r["fresh_loading_threads"] = np.where(rng.random(n) < 0.9, cur["loading_threads"], 3)
    r["fresh_in_progress"] = cur["loading_in_progress"] + rng.choice([0, 0, 1, 3], n)
    r["fresh_rpm"] = cur["rpm"] + rng.choice([0, 0, 5, 99, 100, 1000], n)
    r["last_published"] = now - rng.choice([500, 2_500, 38_000, 39_500, 100_000, 170_000], n)
    ne = np.where(rng.random(n) < 0.3, rng.integers(1, 8, n), 0).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(ne, out=off[1:])
    r["excl_off"], r["n_excl"] = off[:-1], ne
    excl_pod = rng.integers(0, P, int(off[-1])).astype(np.int32)
    excl_time = np.full(int(off[-1]), _lib.ANY_TIME, np.int64)
    nx = np.where(rng.random(n) < 0.4, rng.integers(1, 8, n), 0).astype(np.int32)
    xoff = np.zeros(n + 1, np.int64)
    np.cumsum(nx, out=xoff[1:])
    r["explicit_off"], r["n_explicit"] = xoff[:-1], nx
    explicit = rng.integers(0, P, int(xoff[-1])).astype(np.int32)

    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        got = s.gates(r, excl_pod, excl_time, explicit, now, 450_000)
    finally:
        s.close()

    lib = ob.load()
    orc = ob.OracleFleet(fleet)
    # typeSetStats(mr.getType()): the stats of the instances the model's type may be placed on (MM.java:5169, :2918)
    tstats = np.ascontiguousarray(ob.type_set_stats(fleet))
    in_table = np.ascontiguousarray(((fleet.pods["flags"] & 4) == 0).astype(np.uint8))
    al = ob.unpack_bitmap(fleet.allowed, P) if fleet.n_types else None
    opods = orc.pods
    seen = 0
    for i in range(n):
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
        fl = int(q["flags"])
        ty = int(mr["type"])
        stats = tstats[(0 if ty < 0 or ty >= len(tstats) else ty): ][:1]
        want = 0
        cp, ct = np.ascontiguousarray(lp[keep]), np.ascontiguousarray(lt[keep])
        if lib.orc_go_local(_p(cp), _p(ct), len(cp), int(q["self_pod"]), fl & 1, (fl >> 1) & 1, (fl >> 2) & 1, now):
            want |= 1
        ftc = np.ascontiguousarray(ft)
        if lib.orc_load_failures_breached(_p(ftc), len(ftc), now, 450_000):
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
        curp = np.ascontiguousarray(opods[q["self_pod"]: q["self_pod"] + 1]).copy()
        tomb = bool(fleet.pods["flags"][q["self_pod"]] & 4)
        curp["shutting_down"] = bool(fleet.pods["flags"][q["self_pod"]] & 1)
        if lib.orc_should_publish(None if tomb else curp.ctypes.data_as(C.c_void_p), fresh.ctypes.data_as(C.c_void_p),
                                  now, int(q["last_published"]), (fl >> 6) & 1, (fl >> 7) & 1, fleet.min_space_units):
            want |= 128
        assert got[i]["bits"] == want, (i, bin(got[i]["bits"]), bin(want), q)
        assert got[i]["initial_size"] == init, (i, got[i], init)
        seen |= want
    assert seen == 255, f"some gate never fired in the sample: {seen:#x}"
