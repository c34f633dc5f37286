"""Seeded synthetic fleets for the BASELINE.json configs and an
adversarial fuzz fleet for parity tests.  Pure numpy data generation — no
placement logic lives here.

Configs: C1 256x8, C2 10k x 1k (Zipf), C3 100k x 10k (log-normal), C4 1M x 50k.
Seed = 0x4D4D00 + config index; now_ms = 1_760_000_000_000.
"""
from __future__ import annotations

import numpy as np

from ._lib import (JAVA_LONG_MAX, MODEL_ROW, PLACE_REQ, POD_LIVE, POD_ROW, POD_SHUTTING_DOWN,  # noqa: F401
                   POD_TOMBSTONE)
from .solver import Fleet, bitmap_from_bool

NOW_MS = 1_760_000_000_000
DEFAULT_MODEL_UNITS = 6400  # 50 MiB in 8 KiB units

CONFIGS = {
    "C1": dict(index=1, models=256, pods=8, cap=131072, types=False),
    "C2": dict(index=2, models=10_000, pods=1_000, cap=8_388_608, types=False),
    "C3": dict(index=3, models=100_000, pods=10_000, cap=8_388_608, types=True),
    "C4": dict(index=4, models=1_000_000, pods=50_000, cap=8_388_608, types=True),
}


def min_space_units(default_units: int, threads: int, cap_units: int, unload_mgr: bool = True) -> int:
    """MM.java:765-771 for config construction only (the library exports the same formula)."""
    mn = default_units * (1 if (unload_mgr or threads <= 1) else 2)
    return max(mn, min(default_units * threads, cap_units // 20))


def _models(rng, n_models, n_pods, now, n_types, last_used, id_order):
    k = rng.choice(4, size=n_models, p=[0.3, 0.5, 0.15, 0.05]).astype(np.int32)
    k = np.minimum(k, n_pods)
    nf = np.where(rng.random(n_models) < 0.01, rng.integers(1, 3, n_models), 0).astype(np.int32)
    nf = np.minimum(nf, max(n_pods - 3, 0))
    tot = k + nf
    off = np.zeros(n_models + 1, dtype=np.int64)
    np.cumsum(tot, out=off[1:])
    n_ent = int(off[-1])
    # distinct pods per model: a random start + distinct strides keeps ids unique without a loop
    start = rng.integers(0, n_pods, n_models)
    step = rng.integers(1, max(n_pods // 4, 2), n_models)
    for dj in range(1, 5):  # a stride with dj*step == 0 (mod P) would repeat a pod: fall back to 1
        step = np.where((dj * step) % n_pods == 0, 1, step)
    seg = np.repeat(np.arange(n_models), tot)
    j = np.arange(n_ent) - off[seg]
    ent_pod = ((start[seg] + j * step[seg]) % n_pods).astype(np.int32)
    # guard against stride collisions for tiny fleets
    if n_pods < 64:
        for m in range(n_models):
            s, e = off[m], off[m + 1]
            if e - s > 0:
                ent_pod[s:e] = rng.choice(n_pods, size=e - s, replace=False)
    # instanceIds / loadFailedInstanceIds iterate in TreeMap (instance id) order
    for_sort = id_order[ent_pod].astype(np.int64) + (j >= k[seg]) * (1 << 40) + seg.astype(np.int64) * (1 << 42)
    order = np.argsort(for_sort, kind="stable")
    ent_pod = ent_pod[order]
    ent_time = (now - rng.integers(1_000, 86_400_000, n_ent)).astype(np.int64)
    models = np.zeros(n_models, dtype=MODEL_ROW)
    models["type"] = rng.choice(n_types, size=n_models, p=_type_p(n_types)) if n_types else 0
    models["ent_off"] = off[:-1]
    models["n_loaded"] = k
    models["n_failed"] = nf
    models["last_used"] = last_used
    return models, ent_pod, ent_time


def _type_p(n_types):
    if n_types <= 1:
        return [1.0]
    rest = 0.3 / (n_types - 1)
    return [0.7] + [rest] * (n_types - 1)


def make_fleet(name: str, models: int | None = None, pods: int | None = None) -> Fleet:
    cfg = CONFIGS[name]
    M = models or cfg["models"]
    P = pods or cfg["pods"]
    rng = np.random.Generator(np.random.PCG64(0x4D4D00 + cfg["index"]))
    now = NOW_MS
    cap = cfg["cap"]
    rows = np.zeros(P, dtype=POD_ROW)
    rows["capacity"] = cap
    rows["used"] = np.rint(cap * rng.beta(5, 2, P)).astype(np.int64)
    rows["count"] = rng.poisson(2 * M / P, P)
    lru = now - rng.lognormal(np.log(3.6e6), 1.5, P).astype(np.int64)
    rows["lru_time"] = np.where(rows["count"] == 0, JAVA_LONG_MAX, lru)
    rows["rpm"] = np.minimum(rng.lognormal(np.log(300), 1.2, P), 2_000_000).astype(np.int32)
    rows["loading_threads"] = 8
    rows["loading_in_progress"] = rng.binomial(8, 0.1, P)
    rows["version"] = 1
    rows["id_order"] = np.arange(P, dtype=np.uint32)  # ids "%06x-%05x" % (rs, i) sort by i
    rows["replica_set"] = 0
    rows["flags"] = POD_LIVE

    if name == "C2":  # Zipf(s=1) request rates over model rank
        rate = 1000.0 / np.arange(1, M + 1)  # req/min
        last_used = now - (rng.exponential(60_000.0 / rate)).astype(np.int64) - 1
    else:
        last_used = now - rng.lognormal(np.log(3.6e6), 2.0, M).astype(np.int64) - 1

    n_types = 0
    allowed = prefer = has_allowed = has_prefer = None
    if cfg["types"]:
        # three label groups; type 0 unconstrained, 1 requires group A, 2 prefers group B,
        # 3 requires A∪B and prefers C∩(A∪B) = none -> inferred-empty preference stays null
        n_types = 4
        group = rng.integers(0, 3, P)
        al = np.ones((n_types, P), bool)
        pf = np.zeros((n_types, P), bool)
        al[1] = group == 0
        pf[2] = group == 1
        al[3] = group != 2
        allowed, prefer = bitmap_from_bool(al), bitmap_from_bool(pf)
        has_allowed = np.array([0, 1, 0, 1], np.uint8)
        has_prefer = np.array([0, 0, 1, 0], np.uint8)
    mrows, ent_pod, ent_time = _models(rng, M, P, now, n_types, last_used, rows["id_order"])
    return Fleet(pods=rows, models=mrows, ent_pod=ent_pod, ent_time=ent_time,
                 min_space_units=min_space_units(DEFAULT_MODEL_UNITS, 8, cap),
                 min_churn_age_ms=600_000, now=now, n_types=n_types, allowed=allowed, prefer=prefer,
                 has_allowed=has_allowed, has_prefer=has_prefer)


def make_requests(fleet: Fleet, seed: int, n: int | None = None, *, favour_frac=0.05, extra_frac=0.05,
                  drift_frac=0.25):
    """One load-target decision per model: self = m mod P, lastUsedTime from the model row."""
    rng = np.random.Generator(np.random.PCG64(seed))
    M, P = fleet.n_models, fleet.n_pods
    n = n or M
    reqs = np.zeros(n, dtype=PLACE_REQ)
    m = np.arange(n) % M
    reqs["model"] = m
    self_pod = (np.arange(n) % P).astype(np.int32)
    reqs["self_pod"] = self_pod
    reqs["flags"] = (rng.random(n) < favour_frac).astype(np.uint32)
    reqs["pick"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    lu = fleet.models["last_used"][m].copy()
    kind = rng.random(n)
    lu = np.where(kind < 0.30, 0, lu)                       # inference-triggered: lastUsedTime 0 (MM.java:3484)
    lu = np.where(kind > 0.97, fleet.now + 20_000, lu)      # rpm scale-up stamps now+20s (MM.java:5675)
    reqs["last_used"] = lu
    # the caller's getFreshInstanceRecord(): its snapshot row, rpm never set (0), sometimes drifted
    sp = fleet.pods[self_pod]
    drift = rng.random(n) < drift_frac
    reqs["fresh_lru"] = sp["lru_time"]
    reqs["fresh_capacity"] = sp["capacity"]
    reqs["fresh_used"] = sp["used"] + np.where(drift, rng.integers(0, 200_000, n), 0)
    reqs["fresh_count"] = sp["count"] + np.where(drift, rng.integers(0, 3, n), 0)
    reqs["fresh_rpm"] = 0
    ne = np.where(rng.random(n) < extra_frac, rng.integers(1, 4, n), 0).astype(np.int32)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ne, out=off[1:])
    reqs["extra_off"] = off[:-1]
    reqs["n_extra"] = ne
    extra = rng.integers(0, P, int(off[-1])).astype(np.int32)
    return reqs, extra


def make_full_cluster(fleet: Fleet, seed: int = 5, spread: float = 0.04) -> Fleet:
    """The steady state of a mesh (in place): EVERY instance full and all caches about equally old (global LRU eviction) —
    getNext is then in its LRU-window mode (MM.java:4911-4917) and most shortlists are the whole table."""
    rng = np.random.default_rng(seed)
    P = fleet.n_pods
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 40_000, P)
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-spread, spread, P))).astype(np.int64)
    return fleet


def add_sparse_types(fleet: Fleet, share: float = 0.1, k: int = 12, n_new: int = 4, seed: int = 11) -> Fleet:
    """`n_new` more types (in place) that only `k` instances each may host — a label requirement few instances satisfy puts the
    type's candidates ~P/k positions apart in the placement order; `share` of the models move to them."""
    rng = np.random.default_rng(seed)
    P, T0 = fleet.n_pods, fleet.n_types
    T = T0 + n_new
    al = np.ones((T, P), bool)
    pf = np.zeros((T, P), bool)
    if T0:
        al[:T0] = np.unpackbits(fleet.allowed.view(np.uint8), bitorder="little").reshape(T0, -1)[:, :P].astype(bool)
        pf[:T0] = np.unpackbits(fleet.prefer.view(np.uint8), bitorder="little").reshape(T0, -1)[:, :P].astype(bool)
    for t in range(T0, T):
        al[t] = False
        al[t, rng.choice(P, k, replace=False)] = True
    fleet.allowed, fleet.prefer = bitmap_from_bool(al), bitmap_from_bool(pf)
    fleet.has_allowed = np.concatenate([fleet.has_allowed if T0 else np.zeros(0, np.uint8), np.ones(n_new, np.uint8)])
    fleet.has_prefer = np.concatenate([fleet.has_prefer if T0 else np.zeros(0, np.uint8), np.zeros(n_new, np.uint8)])
    fleet.n_types = T
    sparse = rng.random(fleet.n_models) < share
    fleet.models["type"] = np.where(sparse, rng.integers(T0, T, fleet.n_models), fleet.models["type"])
    return fleet


def sample_requests(reqs, k: int, seed: int):
    """A seeded sample of a batch, in batch order (the `extra` pool stays whole: offsets still index it)."""
    idx = np.sort(np.random.default_rng(seed).choice(len(reqs), min(k, len(reqs)), replace=False))
    return np.ascontiguousarray(reqs[idx])


# --------------------------------------------------------------------------
def fuzz_fleet(seed: int, pods: int = 200, models: int = 300, profile: str | None = None) -> Fleet:
    """Adversarial small fleet: ties everywhere, full pods, Long.MAX lru, dead /
    shutting-down / tombstoned pods, several versions, type masks, preferences,
    replaced replica sets.  Hits every branch of getNext with a few hundred requests."""
    rng = np.random.Generator(np.random.PCG64(seed))
    now = NOW_MS
    P, M = pods, models
    cap = int(rng.choice([131072, 1_000_000]))
    msu = int(rng.choice([6553, 51200, 2560]))
    rows = np.zeros(P, dtype=POD_ROW)
    rows["capacity"] = np.where(rng.random(P) < 0.8, cap, cap // 2)
    fullish = rng.random(P) < (rng.choice([0.9, 1.0]) if profile == "full" else rng.choice([0.05, 0.5, 0.95]))
    used_frac = np.where(fullish, rng.uniform(0.97, 1.05, P), rng.choice([0.1, 0.5, 0.9], P))
    rows["used"] = (rows["capacity"] * used_frac).astype(np.int64)
    rows["count"] = rng.choice([0, 1, 2, 9, 10, 11, 12, 13, 40], P)
    base_lru = now - rng.choice([1_000, 30_000, 44_000, 46_000, 119_000, 121_000, 600_000, 3_600_000,
                                 86_400_000], P) - rng.integers(0, 3, P)
    rows["lru_time"] = np.where(rows["count"] == 0, JAVA_LONG_MAX, base_lru)
    rows["rpm"] = rng.choice([0, 50, 99, 100, 101, 150, 151, 300, 1000, 5000], P)
    rows["loading_threads"] = rng.choice([1, 8], P)
    rows["loading_in_progress"] = rng.integers(0, 3, P)
    multi_version = rng.random() < 0.5
    rows["version"] = rng.choice([1, 2, 3], P) if multi_version else 7
    perm = rng.permutation(P)
    rows["id_order"] = perm.astype(np.uint32)
    n_rs = int(rng.integers(1, 4))
    rows["replica_set"] = rng.integers(0, n_rs, P)
    rows["replica_set"] = np.where(rng.random(P) < 0.05, -1, rows["replica_set"])
    flags = np.full(P, POD_LIVE, dtype=np.uint32)
    flags = np.where(rng.random(P) < 0.05, flags & ~np.uint32(POD_LIVE), flags)
    flags = np.where(rng.random(P) < 0.04, flags | POD_SHUTTING_DOWN, flags)
    flags = np.where(rng.random(P) < 0.03, (flags | POD_TOMBSTONE) & ~np.uint32(POD_LIVE), flags)
    rows["flags"] = flags
    replaced = np.zeros(0, np.int32)
    r = rng.random()
    if r < 0.3:
        replaced = np.array([0], np.int32)
    elif r < 0.4:
        replaced = np.arange(n_rs, dtype=np.int32)  # everything replaced -> retry path

    n_types = int(rng.choice([0, 1, 3, 5])) if profile is None else int(rng.choice([3, 5]))
    allowed = prefer = has_allowed = has_prefer = None
    if n_types:
        al = rng.random((n_types, P)) < rng.choice([0.1, 0.6, 1.0], (n_types, 1))
        pf = rng.random((n_types, P)) < rng.choice([0.0, 0.05, 0.5], (n_types, 1))
        has_allowed = (rng.random(n_types) < 0.6).astype(np.uint8)
        has_prefer = (rng.random(n_types) < 0.7).astype(np.uint8)
        if profile is not None:
            has_prefer[0] = 0  # keep one plain type so the simple case is exercised too
        allowed, prefer = bitmap_from_bool(al), bitmap_from_bool(pf)
    last_used = now - rng.choice([0, 500, 4_000, 6_000, 700_000, 800_000, 80_000_000, 90_000_000,
                                  431_000_000, 433_000_000], M) - 1
    mrows, ent_pod, ent_time = _models(rng, M, P, now, n_types, last_used, rows["id_order"])
    return Fleet(pods=rows, models=mrows, ent_pod=ent_pod, ent_time=ent_time, min_space_units=msu,
                 min_churn_age_ms=int(rng.choice([30_000, 600_000])), now=now, n_types=n_types,
                 allowed=allowed, prefer=prefer, has_allowed=has_allowed, has_prefer=has_prefer,
                 replaced_rs=replaced)


def fuzz_requests(fleet: Fleet, seed: int, n: int):
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    M, P = fleet.n_models, fleet.n_pods
    reqs = np.zeros(n, dtype=PLACE_REQ)
    reqs["model"] = rng.integers(0, M, n)
    sp = rng.integers(0, P, n).astype(np.int32)
    reqs["self_pod"] = np.where(rng.random(n) < 0.05, -1, sp)
    reqs["flags"] = (rng.random(n) < 0.4).astype(np.uint32)
    reqs["pick"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    lu = fleet.models["last_used"][reqs["model"]].copy()
    kind = rng.random(n)
    lu = np.where(kind < 0.2, 0, lu)
    lu = np.where(kind > 0.85, fleet.now + 20_000, lu)
    reqs["last_used"] = lu
    row = fleet.pods[sp]
    stale = rng.random(n) < 0.5
    reqs["fresh_lru"] = np.where(stale, row["lru_time"],
                                 fleet.now - rng.choice([10_000, 50_000, 130_000, 4_000_000], n))
    reqs["fresh_capacity"] = row["capacity"]
    reqs["fresh_used"] = np.where(stale, row["used"], (row["capacity"] * rng.choice([0.2, 0.8, 0.99], n)).astype(np.int64))
    reqs["fresh_count"] = row["count"] + rng.integers(0, 2, n)
    reqs["fresh_rpm"] = rng.choice([0, 0, 0, 120, 400], n)
    ne = np.where(rng.random(n) < 0.3, rng.integers(1, 6, n), 0).astype(np.int32)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ne, out=off[1:])
    reqs["extra_off"] = off[:-1]
    reqs["n_extra"] = ne
    extra = rng.integers(0, P, int(off[-1])).astype(np.int32)
    return reqs, extra


def scenario_fleets():
    """Hand-built fleets that force the rare branches of getNext random fuzzing seldom reaches.
    Yields (name, fleet, reqs, extra)."""
    now = NOW_MS
    msu = 1000

    def pods_of(spec):
        rows = np.zeros(len(spec), dtype=POD_ROW)
        for i, (count, rem, lru, rpm) in enumerate(spec):
            rows[i]["capacity"] = 1_000_000
            rows[i]["used"] = 1_000_000 - rem
            rows[i]["count"] = count
            rows[i]["lru_time"] = lru
            rows[i]["rpm"] = rpm
        rows["loading_threads"] = 8
        rows["version"] = 1
        rows["id_order"] = np.arange(len(spec), dtype=np.uint32)
        rows["flags"] = POD_LIVE
        return rows

    def fleet_of(rows, pref_pods, n_models=4):
        P = len(rows)
        pf = np.zeros((2, P), bool)
        pf[1, pref_pods] = True
        models = np.zeros(n_models, dtype=MODEL_ROW)
        models["type"] = 1
        models["last_used"] = now - 1000
        return Fleet(pods=rows, models=models, ent_pod=np.zeros(0, np.int32), ent_time=np.zeros(0, np.int64),
                     min_space_units=msu, min_churn_age_ms=600_000, now=now, n_types=2,
                     allowed=bitmap_from_bool(np.ones((2, P), bool)), prefer=bitmap_from_bool(pf),
                     has_allowed=np.array([0, 0], np.uint8), has_prefer=np.array([0, 1], np.uint8))

    def reqs_of(fleet, self_pods, favour):
        n = len(self_pods)
        r = np.zeros(n, dtype=PLACE_REQ)
        r["model"] = 0
        r["self_pod"] = self_pods
        r["flags"] = favour
        r["pick"] = np.arange(n, dtype=np.uint32) * 977_777_777
        r["last_used"] = now - 1000
        sp = fleet.pods[np.maximum(self_pods, 0)]
        r["fresh_lru"], r["fresh_capacity"], r["fresh_used"], r["fresh_count"] = (
            sp["lru_time"], sp["capacity"], sp["used"], sp["count"])
        return r

    # S1: case (a) re-designates a preferred best with 4x the space of bestEntry; the self pod then
    # breaks on bestEntry's (small) remaining (quirk B#2 'curInst = bestEntry.getValue()' for self)
    rows = pods_of([(0, msu + 10, now - 5000, 10), (5, 900_000, now - 5000, 20), (6, 900_000, now - 5000, 30),
                    (7, 900_000, now - 5000, 40)])
    f = fleet_of(rows, [1, 2, 3])
    yield "a_found_self_breaks_on_best_entry", f, reqs_of(f, np.array([2, 3, 0, 1, -1], np.int32),
                                                        np.array([0, 0, 0, 0, 0], np.uint32)), np.zeros(0, np.int32)
    # S2: case (b): everything full, best not preferred, self preferred inside the age window + favourSelf => null
    rows = pods_of([(3, 0, now - 900_000, 10), (3, 0, now - 890_000, 500), (3, 0, now - 880_000, 30),
                    (3, 0, now - 100_000, 40)])
    f = fleet_of(rows, [1, 2, 3])
    yield "b_self_preferred_favour_null", f, reqs_of(f, np.array([1, 2, 2, 3, 0], np.int32),
                                                   np.array([1, 1, 0, 1, 1], np.uint32)), np.zeros(0, np.int32)
    # S3: nothing eligible and no replaced replica sets => null straight away
    rows = pods_of([(1, 500_000, now - 5000, 10), (2, 500_000, now - 5000, 10)])
    f = fleet_of(rows, [0])
    r = reqs_of(f, np.array([0, 1], np.int32), np.array([0, 1], np.uint32))
    r["extra_off"], r["n_extra"] = 0, 2
    yield "all_excluded", f, r, np.array([0, 1], np.int32)
    # S4: rpm filter with a busy best: model used <5 s ago, best rpm 5000 vs others 0 (fresh rpm) => best nulled
    rows = pods_of([(1, 500_000, now - 5000, 5000), (1, 499_999, now - 5000, 0), (1, 499_998, now - 5000, 0)])
    f = fleet_of(rows, [])
    f.has_prefer[:] = 0
    yield "busy_best_is_filtered", f, reqs_of(f, np.array([-1, 1, 2, 0], np.int32),
                                            np.array([0, 0, 1, 0], np.uint32)), np.zeros(0, np.int32)
    # S5 (round 2; branches of getNext that a gcov run over the fuzz fleets showed were never taken):
    # case (b) with an OLD best: the next instance is more than 120 s younger but within a quarter of the best's age,
    # so the absolute clause of :4864 holds and the relative one decides (no break); then a preferred instance, then a
    # non-preferred one behind it (the `else if (have_replay)` edge with the replay list already dropped, :4877-4879);
    # and a caller that excluded ITSELF while it is the preferred entry (`!excludeSelf` false at :4869).
    day = 86_400_000
    rows = pods_of([(3, 0, now - day, 10), (3, 0, now - day + 3_600_000, 500), (3, 0, now - day + 7_200_000, 30),
                    (3, 0, now - day + 7_300_000, 40), (3, 0, now - day + 7_350_000, 45), (3, 0, now - 100_000, 50)])
    f = fleet_of(rows, [2, 3])
    r = reqs_of(f, np.array([1, 2, 2, 3, -1], np.int32), np.array([0, 0, 1, 0, 0], np.uint32))
    r["extra_off"] = np.array([0, 0, 0, 0, 0])
    r["n_extra"] = np.array([0, 1, 0, 0, 0])  # request 1: self (pod 2) is in its own exclusion list
    yield "b_old_best_relative_window", f, r, np.array([2], np.int32)
    # S6: the simple case in full mode with an old best: a candidate more than 45 s younger than the best but within a
    # tenth of its age stays in the shortlist (:4915 absolute clause true, relative clause false)
    rows = pods_of([(3, 0, now - day, 10), (3, 0, now - day + 600_000, 20), (3, 0, now - day + 8_000_000, 30),
                    (3, 0, now - 50_000, 40)])
    f = fleet_of(rows, [])
    f.has_prefer[:] = 0
    yield "full_mode_old_best_relative_window", f, reqs_of(f, np.array([1, 2, 3, -1], np.int32),
                                                          np.array([0, 0, 0, 0], np.uint32)), np.zeros(0, np.int32)


# --------------------------------------------------------------------------
class ChurnStream:
    """Config C5: a steady-state stream of load / evict / republish events over a fleet,
    cut into slices of `slice_ms` simulated time (INSTANCE_REC_PUBLISH_MIN_PERIOD_MS = 2 s, MM.java:232):
    45 % model loads (one load-target decision, then the chosen pod's used/count and the model's
    instanceIds change), 45 % cache-eviction evaluations (clhm insert + evict on one pod's cache),
    10 % instance-record republishes.  Data generation and bookkeeping only — every decision is made by
    whoever consumes the slices."""

    def __init__(self, fleet: Fleet, seed: int, events_per_slice: int = 20_000, slice_ms: int = 2_000):
        import copy
        self.f = copy.deepcopy(fleet)
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.n_events, self.slice_ms = events_per_slice, slice_ms
        M, P = self.f.n_models, self.f.n_pods
        # model sizes: LogNormal(ln 50 MiB, 1.0) clipped to [1 MiB, 8 GiB], in 8 KiB units (ModelLoader.java:58-62)
        mib = np.clip(self.rng.lognormal(np.log(50.0), 1.0, M), 1.0, 8192.0)
        self.size_units = np.ceil(mib * 128).astype(np.int64)
        # registry in COO form so that instanceIds can grow: (model, pod, time, failed)
        m = self.f.models
        seg = np.repeat(np.arange(M), m["n_loaded"] + m["n_failed"])
        j = np.arange(len(seg)) - m["ent_off"][seg]
        self.coo_model, self.coo_pod = seg.astype(np.int32), self.f.ent_pod.copy()
        self.coo_time, self.coo_failed = self.f.ent_time.copy(), (j >= m["n_loaded"][seg])
        # one clhm deque per pod: `count` entries whose weights add up to `used`, oldest first
        cnt = np.maximum(self.f.pods["count"], 0).astype(np.int64)
        self.seg_off = np.zeros(P + 1, np.int32)
        np.cumsum(cnt, out=self.seg_off[1:])
        E = int(self.seg_off[-1])
        pod_of = np.repeat(np.arange(P), cnt)
        share = self.rng.random(E) + 0.05
        tot = np.add.reduceat(share, self.seg_off[:-1][cnt > 0]) if E else np.zeros(0)
        tot_full = np.zeros(P)
        tot_full[cnt > 0] = tot
        self.cache_wt = np.maximum((share / np.maximum(tot_full[pod_of], 1e-9) *
                                    self.f.pods["used"][pod_of]).astype(np.int64), 1).astype(np.int32)
        age = self.rng.lognormal(np.log(3.6e6), 1.5, E).astype(np.int64)
        lu = self.f.now - age
        order = np.lexsort((lu, pod_of))
        self.cache_lu = lu[order]
        self.cache_cap = self.f.pods["capacity"].astype(np.int64).copy()
        self.changed_pods = np.zeros(0, np.int32)
        self.changed_models = np.zeros(0, np.int32)

    @property
    def fleet(self) -> Fleet:
        return self.f

    def next_slice(self):
        f, rng = self.f, self.rng
        M, P = f.n_models, f.n_pods
        n_load = int(self.n_events * 0.45)
        n_evict = int(self.n_events * 0.45)
        n_pub = self.n_events - n_load - n_evict
        reqs = np.zeros(n_load, dtype=PLACE_REQ)
        mdl = rng.integers(0, M, n_load)
        sp = rng.integers(0, P, n_load).astype(np.int32)
        reqs["model"], reqs["self_pod"] = mdl, sp
        reqs["flags"] = (rng.random(n_load) < 0.05).astype(np.uint32)
        reqs["pick"] = rng.integers(0, 2**32, n_load, dtype=np.uint64).astype(np.uint32)
        reqs["last_used"] = np.where(rng.random(n_load) < 0.8, 0, f.models["last_used"][mdl])
        row = f.pods[sp]
        reqs["fresh_lru"], reqs["fresh_capacity"] = row["lru_time"], row["capacity"]
        reqs["fresh_used"], reqs["fresh_count"] = row["used"], row["count"]
        ev = np.zeros(n_evict, dtype=np.dtype([("cache", "<i4"), ("weight", "<i4"), ("last_used", "<i8")]))
        ev["cache"] = rng.integers(0, P, n_evict)
        ev["weight"] = np.minimum(self.size_units[rng.integers(0, M, n_evict)], 2**31 - 1)
        ev["last_used"] = np.where(rng.random(n_evict) < 0.7, 0, f.now - rng.integers(1, 7_200_000, n_evict))
        pub = np.unique(rng.integers(0, P, n_pub)).astype(np.int32)
        return {"place_reqs": reqs, "extra": np.zeros(0, np.int32), "evict_reqs": ev, "republish": pub}

    def apply(self, sl, place_outs):
        """Book the slice's outcomes into the fleet (what the Java mesh would write to the KV store)."""
        f, rng = self.f, self.rng
        reqs = sl["place_reqs"]
        chosen = np.where(place_outs["chosen"] == -2, reqs["self_pod"], place_outs["chosen"])
        ok = chosen >= 0
        c, m = chosen[ok].astype(np.int64), reqs["model"][ok]
        np.add.at(f.pods["used"], c, self.size_units[m])
        np.add.at(f.pods["count"], c, 1)
        touched = f.pods["lru_time"][c] == JAVA_LONG_MAX
        f.pods["lru_time"][c[touched]] = f.now - 3_600_000  # first model on an empty pod (LASTUSED_AGE_ON_ADD_MS)
        self.coo_model = np.concatenate([self.coo_model, m.astype(np.int32)])
        self.coo_pod = np.concatenate([self.coo_pod, c.astype(np.int32)])
        self.coo_time = np.concatenate([self.coo_time, np.full(len(c), f.now, np.int64)])
        self.coo_failed = np.concatenate([self.coo_failed, np.zeros(len(c), bool)])
        # a model already on that pod keeps a single entry (instanceIds is a map)
        key = self.coo_model.astype(np.int64) * (f.n_pods + 1) + self.coo_pod
        _, first = np.unique(key, return_index=True)
        keep = np.sort(first)
        self.coo_model, self.coo_pod = self.coo_model[keep], self.coo_pod[keep]
        self.coo_time, self.coo_failed = self.coo_time[keep], self.coo_failed[keep]
        pub = sl["republish"]
        f.pods["rpm"][pub] = np.minimum(rng.lognormal(np.log(300), 1.2, len(pub)), 2_000_000).astype(np.int32)
        f.pods["loading_in_progress"][pub] = rng.binomial(8, 0.1, len(pub))
        has = f.pods["count"][pub] > 0
        f.pods["lru_time"][pub[has]] += rng.integers(0, 1000, int(has.sum()))
        self.changed_pods = np.unique(np.concatenate([c.astype(np.int32), pub]))
        # registry back to CSR: instanceIds in id order, then loadFailedInstanceIds
        order = np.lexsort((f.pods["id_order"][self.coo_pod], self.coo_failed, self.coo_model))
        self.coo_model, self.coo_pod = self.coo_model[order], self.coo_pod[order]
        self.coo_time, self.coo_failed = self.coo_time[order], self.coo_failed[order]
        M = f.n_models
        nl = np.bincount(self.coo_model[~self.coo_failed], minlength=M).astype(np.int32)
        nf = np.bincount(self.coo_model[self.coo_failed], minlength=M).astype(np.int32)
        off = np.zeros(M + 1, np.int64)
        np.cumsum(nl + nf, out=off[1:])
        f.models["ent_off"], f.models["n_loaded"], f.models["n_failed"] = off[:-1], nl, nf
        f.ent_pod, f.ent_time = self.coo_pod.copy(), self.coo_time.copy()
        self.changed_models = np.unique(m).astype(np.int32)
        f.now += self.slice_ms

    def model_events(self):
        """The registry events of the last slice: the changed ModelRecords as (idx, rows, ent_pod, ent_time) with
        rows' ent_off indexing the returned entry arrays (the form mmp_models_upsert takes)."""
        f, idx = self.f, self.changed_models
        rows = f.models[idx].copy()
        k = (rows["n_loaded"] + rows["n_failed"]).astype(np.int64)
        off = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(k, out=off[1:])
        src = np.repeat(rows["ent_off"].astype(np.int64) - off[:-1], k) + np.arange(int(off[-1]))
        rows["ent_off"] = off[:-1]
        return idx, rows, f.ent_pod[src], f.ent_time[src]
