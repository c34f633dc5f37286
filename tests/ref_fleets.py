"""Fleets and requests of the reference-text vectors (tests/golden/ref_getnext.npz): built here, deterministically from
seeds, so that the generator (oracle/ref_harness/make_ref_vectors.py — needs /root/reference) and the tests that consume
the committed vectors (CPU: the oracle; GPU: the HIP path) construct byte-identical inputs.  A digest of the inputs is stored
with the vectors and checked by the tests.

The harness runs the reference's own Java text, which orders instances by their id STRINGS (String.compareTo, MM.java:4696)
and derives the replica set from id.substring(0, 6) (MM.java:4770).  The fuzz fleets of modelmesh_amd.workload draw
`id_order` and `replica_set` independently, which no set of strings can realise — so every fleet here first gets real ids
("%06x-..." prefixes per replica set, short ids for instances without one) and `id_order` := the rank of its id, and
the registry entries (instanceIds / loadFailedInstanceIds iterate in id order) are re-sorted accordingly."""
from __future__ import annotations

import hashlib
import struct

import numpy as np

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl

B36 = "0123456789abcdefghijklmnopqrstuvwxyz"


def _b36(i: int, width: int) -> str:
    s = ""
    for _ in range(width):
        s = B36[i % 36] + s
        i //= 36
    return s


def string_ids(fleet, seed: int):
    """Give every instance an id consistent with its replica set; id_order := String.compareTo rank; registry entries re-sorted."""
    rng = np.random.default_rng(0x1D5 + seed)
    P = fleet.n_pods
    tag = rng.permutation(P)  # decouples the id order from the pod index
    ids = []
    for p in range(P):
        rs = int(fleet.pods["replica_set"][p])
        ids.append(f"{rs:06x}-{_b36(int(tag[p]), 5)}" if rs >= 0 else "s" + _b36(int(tag[p]), 4))  # |id| < 7: no replica set (:4769)
    assert len(set(ids)) == P
    order = sorted(range(P), key=lambda p: ids[p])  # ASCII: Python's str order == String.compareTo's
    rank = np.empty(P, np.uint32)
    rank[order] = np.arange(P, dtype=np.uint32)
    fleet.pods["id_order"] = rank
    # instanceIds (a TreeMap) and loadFailedInstanceIds iterate in id order
    ep, et = fleet.ent_pod.copy(), fleet.ent_time.copy()
    for m in fleet.models:
        for a, k in ((int(m["ent_off"]), int(m["n_loaded"])), (int(m["ent_off"] + m["n_loaded"]), int(m["n_failed"]))):
            if k > 1:
                o = np.argsort(rank[fleet.ent_pod[a:a + k]], kind="stable")
                ep[a:a + k], et[a:a + k] = fleet.ent_pod[a:a + k][o], fleet.ent_time[a:a + k][o]
    fleet.ent_pod, fleet.ent_time = ep, et
    return ids


def place_cases():
    """(name, fleet, ids, reqs, extra): the fuzz fleets x profiles of the GPU parity tests, the scenario fleets, C1, C2."""
    for profile in (None, "full", "prefer"):
        for seed in range(16):
            pods = int(np.random.default_rng(seed).choice([1, 7, 63, 64, 65, 200, 700, 5000]))
            fleet = wl.fuzz_fleet(seed, pods=pods, profile=profile)
            ids = string_ids(fleet, seed)
            reqs, extra = wl.fuzz_requests(fleet, seed, 2000)
            yield f"fuzz_{profile}_{seed}", fleet, ids, reqs, extra
    for name, fleet, reqs, extra in wl.scenario_fleets():
        ids = string_ids(fleet, 99)
        yield f"scenario_{name}", fleet, ids, reqs, extra
    for cfg, seed in (("C1", 11), ("C2", 12)):
        fleet = wl.make_fleet(cfg)
        ids = string_ids(fleet, seed)
        reqs, extra = wl.make_requests(fleet, seed)
        yield cfg, fleet, ids, reqs, extra
    yield from big_place_cases()


def caller_place_cases():
    """(name, fleet, ids, reqs, extra): batches issued by ONE instance — every request of a batch carries the same self /
    favourSelf / getFreshInstanceRecord() (MM.java:5369-5386), as the batches of the rate task, the janitor, the reaper and
    preShutdown do.  The harness decides them as ordinary getNext calls; the device in the single-caller form
    (mmp_place_batch_c: 24-byte rows, tests/test_ref_vectors_gpu.py)."""
    for k, (seed, profile, pods) in enumerate(((3, None, 200), (5, "full", 700), (8, "prefer", 300), (12, "full", 5000), (14, None, 65))):
        fleet = wl.fuzz_fleet(seed, pods=pods, profile=profile)
        ids = string_ids(fleet, 200 + k)
        rng = np.random.default_rng(900 + k)
        for j in range(3):
            reqs, extra = wl.fuzz_requests(fleet, 70 * k + j, 3000)
            sp = -1 if (k + j) % 5 == 4 else int(rng.integers(0, pods))
            row = fleet.pods[max(sp, 0)]
            reqs["self_pod"], reqs["flags"] = sp, (j == 1)
            reqs["fresh_lru"] = row["lru_time"] if j != 2 else fleet.now - 130_000
            reqs["fresh_capacity"] = row["capacity"]
            reqs["fresh_used"] = row["used"] if j == 0 else int(row["capacity"] * [0.2, 0.8, 0.99][(k + j) % 3])
            reqs["fresh_count"] = int(row["count"]) + j
            reqs["fresh_rpm"] = 0 if j < 2 else 120
            yield f"caller_{k}_{j}", fleet, ids, reqs, extra
    fleet = wl.make_full_cluster(wl.make_fleet("C3"))
    ids = string_ids(fleet, 14)
    reqs, extra = wl.make_requests(fleet, seed=0xBE7C1)
    reqs = wl.sample_requests(reqs, 4000, 5)
    row = fleet.pods[4321]
    reqs["self_pod"], reqs["flags"] = 4321, 0
    reqs["fresh_lru"], reqs["fresh_capacity"], reqs["fresh_used"] = row["lru_time"], row["capacity"], row["used"]
    reqs["fresh_count"], reqs["fresh_rpm"] = row["count"], 0
    yield "caller_C3_full_cluster", fleet, ids, reqs, extra


def big_place_cases():
    """The configurations the bench quotes (BASELINE.json configs[2], [3]) and the two regimes with paths of their own on the
    device: the whole 10k / 50k-instance order and a seeded sample of the one-decision-per-model batch.  C3 as it is (window
    path); C3 with every instance full — the bench's `full_cluster` fleet and batch (prefix tables, case (b) from the whole-window
    tables); C3 with types only a dozen instances may host (next-non-empty-word tables); C4's 50k-instance table."""
    fleet = wl.make_fleet("C3")
    ids = string_ids(fleet, 13)
    reqs, extra = wl.make_requests(fleet, 13)
    yield "C3_table", fleet, ids, wl.sample_requests(reqs, 20000, 1), extra
    fleet = wl.make_full_cluster(wl.make_fleet("C3"))
    ids = string_ids(fleet, 14)
    reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
    yield "C3_full_cluster", fleet, ids, wl.sample_requests(reqs, 6000, 2), extra
    fleet = wl.add_sparse_types(wl.make_fleet("C3"))
    ids = string_ids(fleet, 15)
    reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
    sparse = np.nonzero(fleet.models["type"][reqs["model"]] >= 4)[0]
    pick = np.sort(np.concatenate([sparse[:: max(1, len(sparse) // 2000)][:2000], np.random.default_rng(3).choice(len(reqs), 2000, replace=False)]))
    yield "C3_sparse_types", fleet, ids, np.ascontiguousarray(reqs[np.unique(pick)]), extra
    fleet = wl.make_fleet("C4")
    ids = string_ids(fleet, 16)
    reqs, extra = wl.make_requests(fleet, 16)
    yield "C4_table", fleet, ids, wl.sample_requests(reqs, 10000, 4), extra


def serve_cases():
    """(name, fleet, ids, reqs, in_use, last_used, excl_pod, excl_time): serve-target requests on fuzz fleets — load-start times
    that straddle the assume-completed cutoff and collide, models with many copies, tried pairs and key excludes."""
    for seed in range(6):
        rng = np.random.default_rng(1000 + seed)
        fleet = wl.fuzz_fleet(seed, pods=int(rng.choice([8, 64, 300])), models=400)
        P, now = fleet.n_pods, fleet.now
        fleet.ent_time[:] = now - rng.choice([500, 2_999, 3_000, 3_001, 10_000, 10_000, 60_000], len(fleet.ent_time))
        if P >= 8:
            ent_pod, ent_time = list(fleet.ent_pod), list(fleet.ent_time)
            for i in np.nonzero(rng.random(fleet.n_models) < 0.1)[0]:
                k = int(rng.integers(5, 9))
                fleet.models["ent_off"][i], fleet.models["n_loaded"][i], fleet.models["n_failed"][i] = len(ent_pod), k, 0
                ent_pod += list(rng.choice(P, size=k, replace=False))
                ent_time += list(now - rng.choice([500, 2_999, 3_000, 3_001, 10_000, 60_000], k))
            fleet.ent_pod = np.array(ent_pod, np.int32)
            fleet.ent_time = np.array(ent_time, np.int64)
        ids = string_ids(fleet, 50 + seed)
        n = 3000
        reqs = np.zeros(n, dtype=_lib.SERVE_REQ)
        reqs["model"] = rng.integers(0, fleet.n_models, n)
        reqs["self_pod"] = np.where(rng.random(n) < 0.1, -1, rng.integers(0, P, n))
        m = fleet.models[reqs["model"]]
        has = m["n_loaded"] > 0
        pickc = (m["ent_off"] + rng.integers(0, 8, n) % np.maximum(m["n_loaded"], 1)).clip(0, max(len(fleet.ent_pod) - 1, 0))
        if len(fleet.ent_pod):
            reqs["self_pod"] = np.where(has & (rng.random(n) < 0.5), fleet.ent_pod[pickc], reqs["self_pod"])
        reqs["flags"] = rng.integers(0, 4, n)
        reqs["local_in_flight"] = rng.integers(0, 3, n)
        reqs["last_invoke_time"] = now - rng.choice([0, 10, 1000], n)
        reqs["assume_completed_ms"] = rng.choice([3000, 30_000], n)
        in_use = rng.integers(0, 3, P).astype(np.int32)
        last_used = (now - rng.choice([0, 5, 5, 100, 10_000], P)).astype(np.int64)
        ne = np.where(rng.random(n) < 0.3, rng.integers(1, 8, n), 0).astype(np.int32)
        off = np.zeros(n + 1, np.int64)
        np.cumsum(ne, out=off[1:])
        reqs["excl_off"], reqs["n_excl"] = off[:-1], ne
        excl_pod = np.zeros(int(off[-1]), np.int32)
        excl_time = np.zeros(int(off[-1]), np.int64)
        for i in np.nonzero(ne)[0]:
            mm = fleet.models[reqs["model"][i]]
            for j in range(ne[i]):
                if mm["n_loaded"] > 0 and rng.random() < 0.8:
                    e = mm["ent_off"] + rng.integers(0, mm["n_loaded"])
                    excl_pod[off[i] + j] = fleet.ent_pod[e]
                    excl_time[off[i] + j] = _lib.ANY_TIME if rng.random() < 0.5 else fleet.ent_time[e] + rng.integers(0, 2)
                else:
                    excl_pod[off[i] + j] = rng.integers(0, P)
                    excl_time[off[i] + j] = _lib.ANY_TIME
        yield f"serve_{seed}", fleet, ids, reqs, in_use, last_used, excl_pod, excl_time


def gate_cases():
    """(name, fleet, ids, reqs, excl_pod, excl_time, explicit, in_use_expiry): the request-level guards (mmp_gate_req) on fuzz
    fleets — models with many copies / failures so that the count guards trigger, fresh rows near and far from the table's."""
    for seed in range(4):
        rng = np.random.default_rng(3000 + seed)
        fleet = wl.fuzz_fleet(seed + 40, pods=int(rng.choice([8, 64, 300])), models=300, profile=None if seed % 2 else "prefer")
        P, now = fleet.n_pods, fleet.now
        fleet.ent_time[:] = now - rng.choice([100, 1_400, 1_600, 449_000, 451_000, 3_000_000], len(fleet.ent_time))
        m = fleet.models
        big = np.nonzero(rng.random(len(m)) < 0.1)[0]
        ent_pod, ent_time = list(fleet.ent_pod), list(fleet.ent_time)
        for i in big:
            k, f = int(rng.integers(4, min(8, P) + 1)) if P >= 5 else min(P, 2), int(rng.integers(0, min(8, P)))
            f = min(f, P - k) if P - k > 0 else 0
            pods = rng.choice(P, size=k + f, replace=False)
            m["ent_off"][i], m["n_loaded"][i], m["n_failed"][i] = len(ent_pod), k, f
            ent_pod += list(pods)
            ent_time += list(now - rng.choice([100, 449_000, 451_000], k + f))
        fleet.ent_pod = np.array(ent_pod, np.int32)
        fleet.ent_time = np.array(ent_time, np.int64)
        ids = string_ids(fleet, 70 + seed)
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
        r["fresh_lru"] = np.where(rng.random(n) < 0.04, -1, r["fresh_lru"])  # an empty cache: oldestTime() == -1, read as Long.MAX_VALUE (:5423-5425)
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
        yield f"gates_{seed}", fleet, ids, r, excl_pod, excl_time, explicit, 450_000


def _rebalance_fleet(seed, pods, models, used):
    """Fleet where many models have 1-4 copies, often including self_pod = 0 (as tests/test_rebalance_gpu.py builds it)."""
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


def _local_entries(fleet, rng, n):
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


def scaleup_cases():
    """(name, fleet, ids, entries, params): runs of the rate-tracking task (rateTrackingTask, MM.java:5636-5832) over the local
    cache entries of instance 0 — thresholds and clocks that reach the second-copy rule, the scale-up rule with and without
    overloaded instances, and the early returns."""
    for seed, pods, used in ((0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (3, 1, 0.5)):
        fleet, rng = _rebalance_fleet(seed, pods, 600, used)
        ids = string_ids(fleet, 80 + seed)
        now = fleet.now
        entries = _local_entries(fleet, rng, 800)
        for j, (thr, our_rpm, last_check) in enumerate(((2000, 100, now - 10_000), (100, 50_000, now - 9_000), (0, 0, now - 10_000),
                                                        (2000, 0, now - 1_000))):
            sp = np.zeros(1, dtype=_lib.SCALEUP_PARAMS)
            sp["self_pod"], sp["iteration_counter"] = 0, 130
            sp["second_copy_max_age_iters"], sp["second_copy_min_age_iters"] = 240, 42
            sp["scale_up_rpm_threshold"], sp["our_rpm"] = thr, our_rpm
            sp["now"], sp["last_check_time"], sp["rate_check_interval_ms"] = now, last_check, 10_000
            sp["second_copy_lru_threshold_ms"], sp["assume_completed_ms"] = 72_000_000, 30_000
            yield f"scaleup_{seed}_{j}", fleet, ids, entries, sp


def conc_rows(rng, n):
    """MaxConcCacheEntry state per cache entry (MM.java:2641-2797): countAndTimeSum with counts on both sides of the 64 / 8
    limits of getRpmScaleThreshold (:2769, :2780), with and without prior samples, zero time sums, a few queued requests."""
    c = np.zeros(n, dtype=_lib.CONC_ENTRY)
    count = rng.choice([0, 1, 7, 8, 30, 63, 64, 65, 500, 20_000], n)
    per_call = rng.choice([0, 1, 8, 50, 400, 30_000], n)  # mean duration in 1/10 ms
    c["count_and_time_sum"] = (count * per_call) * (1 << _lib.CONC_COUNT_BITS) + count
    c["prior_count"] = rng.choice([0, 0, 5, 64, 900, -1], n)
    c["prior_sum"] = np.where(c["prior_count"] > 0, c["prior_count"] * rng.choice([0, 2, 60, 500], n), rng.choice([0, 0, 7], n))
    c["max_conc"] = rng.choice([1, 1, 2, 4, 8, 16, 64], n)
    c["queued_requests"] = rng.choice([0, 0, 0, 1, 2, 9], n)
    return c


def scaleup_conc_cases():
    """(name, fleet, ids, entries, conc, params, conc_params): the rate task of a mesh that limits model concurrency (latencyBased,
    MM.java:5677): every entry's threshold is mcce.getRpmScaleThreshold(true) (:5704), averageModelParallelism feeds getExcludeSet
    (:5836) and is recomputed after the loop (:5815-5818)."""
    for seed, pods, used in ((0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (4, 64, 0.9)):
        fleet, rng = _rebalance_fleet(seed, pods, 600, used)
        ids = string_ids(fleet, 80 + seed)
        now = fleet.now
        entries = _local_entries(fleet, rng, 800)
        conc = conc_rows(rng, len(entries))
        for j, (thr, our_rpm, last_check, avg, pct) in enumerate(((2000, 100, now - 10_000, 1.0, 90), (300, 50_000, now - 9_000, 3.75, 90),
                                                                  (2000, 0, now - 1_000, 2.5, 90), (50, 9_000, now - 12_000, 11.3, 50))):
            sp = np.zeros(1, dtype=_lib.SCALEUP_PARAMS)
            sp["self_pod"], sp["iteration_counter"] = 0, 130
            sp["second_copy_max_age_iters"], sp["second_copy_min_age_iters"] = 240, 42
            sp["scale_up_rpm_threshold"], sp["our_rpm"] = thr, our_rpm
            sp["now"], sp["last_check_time"], sp["rate_check_interval_ms"] = now, last_check, 10_000
            sp["second_copy_lru_threshold_ms"], sp["assume_completed_ms"] = 72_000_000, 30_000
            cp = np.zeros(1, dtype=_lib.CONC_PARAMS)
            cp["dynamic_rpm_scale_constant"], cp["average_model_parallelism"] = 600_000 * pct // 100, avg
            yield f"scaleup_conc_{seed}_{j}", fleet, ids, entries, conc, sp, cp


def scaleup_edge_cases():
    """(name, fleet, ids, entries, params): rate-task runs that reach the lines the fleets above do not: no invocations since the
    last check (MM.java:5667-5669), a type confined to a single instance (:5697-5699), a scale-up with nothing overloaded (the
    exclude set stays the model's own instances, :5789)."""
    fleet, rng = _rebalance_fleet(5, 40, 300, 0.5)
    ids = string_ids(fleet, 91)
    now = fleet.now
    sp = np.zeros(1, dtype=_lib.SCALEUP_PARAMS)
    sp["self_pod"], sp["iteration_counter"] = 0, 130
    sp["second_copy_max_age_iters"], sp["second_copy_min_age_iters"] = 240, 42
    sp["scale_up_rpm_threshold"], sp["our_rpm"] = 100, 10**9
    sp["now"], sp["last_check_time"], sp["rate_check_interval_ms"] = now, now - 10_000, 10_000
    sp["second_copy_lru_threshold_ms"], sp["assume_completed_ms"] = 72_000_000, 1_000
    yield "scaleup_edge_empty", fleet, ids, _local_entries(fleet, rng, 0), sp
    # nothing overloaded (our own rpm is far above everyone's), loads old enough, plenty of candidates: scale-ups with an empty exclude set
    f2, rng2 = _rebalance_fleet(6, 60, 300, 0.3)
    f2.ent_time[:] = now - 90_000_000
    e2 = _local_entries(f2, rng2, 400)
    e2["interval_count"] = rng2.choice([400, 5000, 90_000], len(e2))
    yield "scaleup_edge_nothing_overloaded", f2, string_ids(f2, 92), e2, sp
    # type constraints under which one type may only run on instance 0 and another nowhere: typeSetStats(type).instanceCount < 2
    f3, rng3 = _rebalance_fleet(7, 30, 200, 0.4)
    T = 3
    al = np.ones((T, f3.n_pods), bool)
    al[1, 1:] = False
    al[2, :] = False
    f3.n_types = T
    f3.allowed, f3.prefer = wl.bitmap_from_bool(al), wl.bitmap_from_bool(np.zeros((T, f3.n_pods), bool))
    f3.has_allowed, f3.has_prefer = np.array([0, 1, 1], np.uint8), np.zeros(T, np.uint8)
    f3.models["type"] = rng3.integers(0, T, f3.n_models)
    yield "scaleup_edge_confined_types", f3, string_ids(f3, 93), _local_entries(f3, rng3, 300), sp


def scaledown_conc_cases():
    """(name, fleet, ids, entries, conc, params, dyn_const): janitor passes over MaxConcCacheEntry candidates (MM.java:6294-6305): the
    threshold of a model with three or more copies is mcce.getRpmScaleThreshold(false), a copy with queued requests stays."""
    for seed, pods, used in ((1, 200, 0.97), (4, 300, 0.99)):
        fleet, rng = _rebalance_fleet(seed, pods, 600, used)
        ids = string_ids(fleet, 80 + seed)
        now = fleet.now
        entries = _local_entries(fleet, rng, 800)
        entries["last_heavy_time"] = np.where(rng.random(len(entries)) < 0.5, 0, entries["last_heavy_time"])
        fleet.ent_time[:] = now - 90_000_000  # no copy loaded within the last 30 minutes (:6269)
        conc = conc_rows(rng, len(entries))
        for j, (thr, cap, pct) in enumerate(((2000, 10_000_000, 90), (10, 10_000_000, 10))):
            dp = np.zeros(1, dtype=_lib.SCALEDOWN_PARAMS)
            dp["self_pod"], dp["shutting_down"], dp["now"] = 0, 0, now
            dp["last_check_time"], dp["rate_check_interval_ms"] = now - 7_000, 10_000
            dp["adjusted_cache_capacity"], dp["scale_up_rpm_threshold"] = cap, thr
            yield f"scaledown_conc_{seed}_{j}", fleet, ids, entries, conc, dp, 600_000 * pct // 100


def scaledown_edge_cases():
    """(name, fleet, ids, entries, params): the sample period too small to judge a model's load (MM.java:6286-6289), and this
    instance shutting down / not in the table when a second copy is old enough to go (removeSecondModelCopy, :6322-6324)."""
    for k, (self_flags, last_check_ago) in enumerate(((2, 500), (1, 7_000), (4, 7_000))):
        fleet, rng = _rebalance_fleet(4, 300, 600, 0.99)
        fleet.pods["flags"][0] = self_flags
        if k:  # without type constraints instanceSetStats() stays the cluster's when this instance is not in clusterState (:1446-1448)
            fleet.n_types, fleet.allowed, fleet.prefer, fleet.has_allowed, fleet.has_prefer = 0, None, None, None, None
            fleet.models["type"] = 0
        fleet.ent_time[:] = fleet.now - 90_000_000
        entries = _local_entries(fleet, rng, 800)
        entries["last_heavy_time"] = 0
        dp = np.zeros(1, dtype=_lib.SCALEDOWN_PARAMS)
        dp["self_pod"], dp["shutting_down"], dp["now"] = 0, 0, fleet.now
        dp["last_check_time"], dp["rate_check_interval_ms"] = fleet.now - last_check_ago, 10_000
        dp["adjusted_cache_capacity"], dp["scale_up_rpm_threshold"] = 10_000_000, 2000
        yield f"scaledown_edge_{k}", fleet, string_ids(fleet, 95 + k), entries, dp


def scaledown_cases():
    """(name, fleet, ids, entries, params): janitor passes over scaleCopiesCandidates (MM.java:6110-6140 -> removeModelCopies
    :6197-6310 -> removeSecondModelCopy :6314-6335) for the local cache entries of instance 0."""
    for seed, pods, used in ((0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (3, 1, 0.5), (4, 300, 0.99)):
        fleet, rng = _rebalance_fleet(seed, pods, 600, used)
        ids = string_ids(fleet, 80 + seed)
        now = fleet.now
        entries = _local_entries(fleet, rng, 800)
        for j, (thr, cap, sd) in enumerate(((2000, 200_000, 0), (2000, 1_000, 0), (10, 10_000_000, 0), (2000, 200_000, 1))):
            dp = np.zeros(1, dtype=_lib.SCALEDOWN_PARAMS)
            dp["self_pod"], dp["shutting_down"], dp["now"] = 0, sd, now
            dp["last_check_time"], dp["rate_check_interval_ms"] = now - 7_000, 10_000
            dp["adjusted_cache_capacity"], dp["scale_up_rpm_threshold"] = cap, thr
            yield f"scaledown_{seed}_{j}", fleet, ids, entries, dp


def _plan_fleet(seed, pods, models, used_frac, dup_frac=0.3):
    """Mostly unloaded models, many sharing a lastUsed value (the reaper's TreeSet keeps the first of them); instances with
    and without loading capacity (as tests/test_rebalance_gpu.py builds it)."""
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
    unloaded = rng.random(models) < 0.8
    m["n_loaded"] = np.where(unloaded, 0, m["n_loaded"])
    m["n_failed"] = np.where(rng.random(models) < 0.1, 2, np.minimum(m["n_failed"], 1))
    lu = now - rng.integers(1_000, 60_000_000, models)
    dup = rng.random(models) < dup_frac
    lu = np.where(dup, now - rng.choice([5_000, 900_000, 30_000_000], models), lu)
    m["last_used"] = lu
    m["n_loaded"] = np.minimum(m["n_loaded"], max(pods - 2, 0))
    tot = m["n_loaded"] + m["n_failed"]
    off = np.zeros(models + 1, np.int64)
    np.cumsum(tot, out=off[1:])
    m["ent_off"] = off[:-1]
    # instanceIds / loadFailedInstanceIds are MAPS keyed by the instance id: a model's entries name distinct instances
    ent = np.zeros(int(off[-1]), np.int32)
    for i in np.flatnonzero(tot):
        ent[off[i]: off[i + 1]] = rng.choice(pods, size=int(tot[i]), replace=False)
    fleet.ent_pod = ent
    fleet.ent_time = (now - rng.integers(1_000, 1_000_000, int(off[-1]))).astype(np.int64)
    return fleet


def proactive_cases():
    """(name, fleet, ids, defaultModelSizeUnits, partitioned): one run of the leader's reaper as far as proactive loading goes
    (MM.java:6456-6490, :6574-6577, :6616-6747); partitioned: with type constraints, one pass per ProhibitedTypeSet partition."""
    from modelmesh_amd.solver import bitmap_from_bool
    for seed, pods, models, used in ((0, 8, 300, 0.5), (1, 64, 3000, 0.2), (2, 300, 12000, 0.9), (3, 300, 12000, 0.99), (5, 5, 50, 0.0)):
        fleet = _plan_fleet(seed, pods, models, used)
        ids = string_ids(fleet, 90 + seed)
        for units in (6400, 1):
            yield f"proactive_{seed}_{units}", fleet, ids, units, False
    for seed, total_copies in ((6, 2), (7, 8)):  # modelCopyCount < 3: defaultModelSizeUnits; <= 10: averaged with it (:6622-6629)
        fleet = _plan_fleet(seed, 6, 400, 0.3)
        fleet.pods["flags"] = 2
        fleet.pods["count"] = 0
        fleet.pods["count"][:2] = [total_copies // 2, total_copies - total_copies // 2]
        fleet.pods["lru_time"] = np.where(fleet.pods["count"] == 0, 2**63 - 1, fleet.pods["lru_time"])
        for units in (6400, 100):
            yield f"proactive_{seed}_{units}", fleet, string_ids(fleet, 90 + seed), units, False
    for seed in (0, 1, 2):
        fleet = _plan_fleet(seed + 20, 300, 6000, [0.5, 0.9, 0.2][seed])
        if not fleet.n_types:
            rng = np.random.default_rng(seed)
            fleet.n_types = 3
            al = rng.random((3, fleet.n_pods)) < np.array([[1.0], [0.4], [0.7]])
            fleet.allowed, fleet.prefer = bitmap_from_bool(al), bitmap_from_bool(np.zeros_like(al))
            fleet.has_allowed, fleet.has_prefer = np.array([0, 1, 1], np.uint8), np.zeros(3, np.uint8)
            fleet.models["type"] = rng.integers(0, 3, fleet.n_models)
        yield f"proactive_parts_{seed}", fleet, string_ids(fleet, 95 + seed), 6400, True


TABLE_EVENT = np.dtype([("type", "<i4"), ("pod", "<i4"), ("row", _lib.POD_ROW)])  # type: 0 ENTRY_ADDED, 1 ENTRY_UPDATED, 2 ENTRY_DELETED


def table_event_cases():
    """(name, fleet, ids, events, checkpoint_every, tables): a stream of instance-table listener events (MM.java:1455-1568) from
    an empty table — every instance added, then records republished (load / used / count / lruTime change, an emptied cache,
    shutdown announcements), instances deleted and re-added.  tables[k] = the instance table (rows + flags) as it stands at
    checkpoint k, for the implementations that take a table rather than events."""
    for seed, pods, n_ev, ck in ((0, 6, 60, 1), (1, 40, 600, 7), (2, 300, 3000, 100), (3, 300, 3000, 250)):
        rng = np.random.default_rng(400 + seed)
        fleet = wl.fuzz_fleet(seed + 30, pods=pods, models=50, profile=[None, "full", "pref", None][seed])
        fleet.n_types, fleet.allowed, fleet.prefer = 0, None, None
        fleet.has_allowed = fleet.has_prefer = None
        ids = string_ids(fleet, 60 + seed)
        now = fleet.now
        rows = fleet.pods.copy()
        rows["flags"] = _lib_flag_live()
        cur = rows.copy()
        present = np.zeros(pods, bool)
        ev = np.zeros(pods + n_ev, dtype=TABLE_EVENT)
        tables = []

        def snapshot():
            t = cur.copy()
            t["flags"] = np.where(present, t["flags"], 4)  # absent: tombstone
            tables.append(t)
        k = 0
        for i in rng.permutation(pods):
            ev[k] = (0, i, cur[i])
            present[i] = True
            k += 1
            if k % ck == 0:
                snapshot()
        while k < len(ev):
            i = int(rng.integers(0, pods))
            u = rng.random()
            if not present[i]:
                r = rows[i].copy()
                r["flags"] = _lib_flag_live()
                cur[i] = r
                ev[k] = (0, i, r)
                present[i] = True
            elif u < 0.12:
                ev[k] = (2, i, cur[i])
                present[i] = False
            elif u < 0.18:  # announces its shutdown: the listener treats the update as a delete (:1462-1464)
                r = cur[i].copy()
                r["flags"] = r["flags"] | 1
                ev[k] = (1, i, r)
                present[i] = False
            elif u < 0.22:  # republished unchanged: "Identical instance record already exists" (:1505-1509)
                ev[k] = (1, i, cur[i])
            else:
                r = cur[i].copy()
                what = rng.integers(0, 6)
                if what == 0:
                    r["used"] = min(int(r["capacity"]), max(0, int(r["used"]) + int(rng.integers(-40_000, 40_000))))
                elif what == 1:
                    r["count"] = max(0, int(r["count"]) + int(rng.integers(-3, 4)))
                elif what == 2:
                    r["lru_time"] = now - int(rng.integers(1_000, 80_000_000))
                elif what == 3:
                    r["count"], r["used"], r["lru_time"] = 0, 0, 2**63 - 1  # emptied cache
                elif what == 4:
                    r["loading_in_progress"] = int(rng.integers(0, 9))
                    r["rpm"] = int(rng.choice([0, 50, 400, 9000]))
                else:
                    r["lru_time"] = 0  # a record that never reported one: not counted in the global LRU (InstanceSetStatsTracker:58)
                # (a FULL record without an lruTime does not occur — a full cache has an oldest entry — and it is the one shape
                # under which PLACEMENT_ORDER's version rule, :4654-4660, is not transitive: a skip-list set then keeps whatever
                # order its insertion history produced, and libmmplace refuses such a table with MMP_EORDER)
                if int(r["lru_time"]) == 0 and int(r["capacity"]) - int(r["used"]) < fleet.min_space_units:
                    r["lru_time"] = now - int(rng.integers(1_000, 80_000_000))
                cur[i] = r
                ev[k] = (1, i, r)
            k += 1
            if k % ck == 0 or k == len(ev):
                snapshot()
        yield f"table_events_{seed}", fleet, ids, ev, ck, tables


def migration_cases():
    """(name, fleet, ids, entries, self_pod, now): preShutdown of instance 0 (MM.java:6998-7046): its cache entries in
    descendingLruMap() order — held and not held by it according to the registry, failed, never used (lastUsed 0), older and
    younger than CUTOFF_AGE_MS."""
    for seed, pods, used in ((0, 12, 0.5), (1, 200, 0.97), (2, 200, 0.2), (3, 1, 0.5), (4, 300, 0.99)):
        fleet, rng = _rebalance_fleet(seed, pods, 600, used)
        entries = _local_entries(fleet, rng, 800)
        entries["last_used"] = np.where(rng.random(800) < 0.3, fleet.now - rng.choice([3_599_999, 3_600_000, 3_600_001, 7_000_000], 800),
                                        entries["last_used"])  # the cutoff's edges
        entries = entries[np.argsort(-entries["last_used"], kind="stable")]  # most recently used first
        yield f"migration_{seed}", fleet, string_ids(fleet, 50 + seed), entries, 0, fleet.now


def type_constraint_cases():
    """(name, fleet, ids, pod_bits, req_bits, pref_bits): a type-constraint configuration over an instance table (as
    tests/test_types_gpu.py draws them): bit i of an instance's word = it carries label i; per type the required and preferred labels."""
    for seed, P, T, n_labels, label_p in ((0, 5, 2, 2, 0.5), (1, 64, 4, 3, 0.4), (2, 200, 6, 5, 0.3), (3, 700, 8, 6, 0.15), (4, 65, 3, 2, 0.0),
                                           (5, 130, 5, 4, 0.9), (6, 300, 12, 8, 0.5), (7, 40, 1, 1, 0.5)):
        rng = np.random.default_rng(9000 + seed)
        fleet = wl.fuzz_fleet(seed, pods=P, models=50)
        fleet.n_types, fleet.allowed, fleet.prefer = 0, None, None
        fleet.has_allowed = fleet.has_prefer = None
        pod_bits = np.zeros(P, np.uint64)
        for p in range(P):
            pod_bits[p] = sum(1 << i for i in range(n_labels) if rng.random() < label_p)
        req_bits, pref_bits = np.zeros(T, np.uint64), np.zeros(T, np.uint64)
        for t in range(T):
            nr, nf = int(rng.choice([0, 0, 1, 2])), int(rng.choice([0, 1, 2]))
            req = rng.choice(n_labels, size=min(nr, n_labels), replace=False) if nr else []
            pref = [l for l in (rng.choice(n_labels, size=min(nf, n_labels), replace=False) if nf else []) if l not in req]  # kept disjoint (:96-99)
            req_bits[t] = sum(1 << int(l) for l in req)
            pref_bits[t] = sum(1 << int(l) for l in pref)
        yield f"type_constraints_{seed}", fleet, string_ids(fleet, 70 + seed), pod_bits, req_bits, pref_bits


UPGRADE_EVENT = np.dtype([("kind", "<i4"), ("replica_set", "<i4"), ("labels_key", "<i8"), ("start_time", "<i8"), ("now", "<i8")])


def upgrade_event_cases():
    """(name, events): rolling-update streams for UpgradeTracker (UpgradeTracker.java:85-200; as tests/test_upgrade_gpu.py draws
    them): instances of a few replica sets under two label sets come and go while the clock advances by 50 ms ... 7 min, with
    housekeeping passes in between.  kind 0 instanceAdded, 1 instanceRemoved, 2 doHousekeeping."""
    for seed in range(6):
        rng = np.random.default_rng(300 + seed)
        now = 1_760_000_000_000
        starts, members = {}, []
        ev = np.zeros(500, dtype=UPGRADE_EVENT)
        for step in range(500):
            now += int(rng.choice([50, 5_000, 60_000, 400_000]))
            r = rng.random()
            if r < 0.55 or not members:
                lk = int(rng.choice([0, 0, 0, 7]))
                rs = int(rng.choice([-1, 0, 1, 2, 3, 4]))
                if rs not in starts:
                    starts[rs] = now - int(rng.integers(0, 3_000_000)) * 7 - rs  # distinct per set: no ties in Stream.max
                st = starts[rs] + int(rng.integers(0, 100_000)) * 11
                ev[step] = (0, rs, lk, st, now)
                members.append((lk, rs))
            elif r < 0.9:
                lk, rs = members.pop(int(rng.integers(0, len(members))))
                if rng.random() < 0.1:
                    lk += 100  # a re-deserialised record: another labels array identity
                ev[step] = (1, rs, lk, 0, now)
            else:
                ev[step] = (2, -1, 0, 0, now)
        yield f"upgrade_events_{seed}", ev
    # rolling updates proper: a deployment's old replica set is replaced member by member by a new one (start times now),
    # a second deployment (other labels) follows with a lag, housekeeping runs every minute, then everything ages out
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        now = 1_760_000_000_000
        rows = []
        n_old = int(rng.integers(3, 9))
        for lk, rs_old in ((0, 0), (7, 2)):
            for _ in range(n_old):
                rows.append((0, rs_old, lk, now - 86_400_000 - int(rng.integers(0, 3_600_000)), now))
                now += int(rng.integers(10, 2_000))
        live = {(0, 0): n_old, (7, 2): n_old}
        plan = [(0, 0, 1)] * n_old + [(7, 2, 3)] * n_old
        order = rng.permutation(len(plan)) if seed % 2 else np.arange(len(plan))
        for j in order:
            lk, rs_old, rs_new = plan[j]
            now += int(rng.choice([5_000, 20_000, 45_000, 200_000]))
            rows.append((0, rs_new, lk, now - int(rng.integers(1_000, 9_000)), now))
            now += int(rng.integers(500, 30_000))
            if live[(lk, rs_old)] > 0:
                rows.append((1, rs_old, lk, 0, now))
                live[(lk, rs_old)] -= 1
            if rng.random() < 0.4:
                now += 60_000
                rows.append((2, -1, 0, 0, now))
        for _ in range(6):  # ... and the aftermath: the map's entries expire 15 min after the last change
            now += int(rng.choice([240_000, 600_000]))
            rows.append((2, -1, 0, 0, now))
            rows.append((0, int(rng.choice([1, 3])), int(rng.choice([0, 7])), now - 2_000, now))
        yield f"upgrade_rolling_{seed}", np.array(rows, dtype=UPGRADE_EVENT)


def _lib_flag_live():
    return 2  # MMP_POD_LIVE


def input_blob(fleet, ids, reqs=None, extra=None, serve=None, gates=None, scaleup=None, scaledown=None, proactive=None, events=None,
               upgrade=None, types=None, migration=None, conc=None) -> bytes:
    """The harness' input file (layout: oracle/ref_harness/harness.cc main())."""
    P, M = fleet.n_pods, fleet.n_models
    T = int(fleet.n_types)
    W = (P + 63) // 64
    reqs = np.zeros(0, _lib.PLACE_REQ) if reqs is None else np.ascontiguousarray(reqs)
    extra = np.zeros(0, np.int32) if extra is None else np.ascontiguousarray(extra, dtype=np.int32)
    if serve is None:
        sreqs, in_use, last_used = np.zeros(0, _lib.SERVE_REQ), np.zeros(0, np.int32), np.zeros(0, np.int64)
        sx_pod, sx_time = np.zeros(0, np.int32), np.zeros(0, np.int64)
    else:
        sreqs, in_use, last_used, sx_pod, sx_time = serve
    repl = np.ascontiguousarray(fleet.replaced_rs, dtype=np.int32)
    if gates is None:
        greqs, gx_pod, gx_time, gexpl, g_expiry, tstats = np.zeros(0, _lib.GATE_REQ), np.zeros(0, np.int32), np.zeros(0, np.int64), \
            np.zeros(0, np.int32), 0, None
    else:
        greqs, gx_pod, gx_time, gexpl, g_expiry, tstats = gates  # tstats: typeSetStats(type) per type row (oracle.bind.ORC_STATS)
    hdr = [P, M, len(fleet.ent_pod), T, W, len(repl), len(reqs), len(extra), int(fleet.min_space_units), int(fleet.min_churn_age_ms),
           int(fleet.now), len(sreqs), len(sx_pod), len(greqs), len(gx_pod), len(gexpl)]
    idbuf = b"".join(s.encode("ascii").ljust(16, b"\0") for s in ids)
    parts = [b"MMREF1\0\0", struct.pack("<16q", *hdr), np.ascontiguousarray(fleet.pods).tobytes(), idbuf,
             np.ascontiguousarray(fleet.models).tobytes(), np.ascontiguousarray(fleet.ent_pod, dtype=np.int32).tobytes(),
             np.ascontiguousarray(fleet.ent_time, dtype=np.int64).tobytes()]
    if T:
        parts += [np.ascontiguousarray(fleet.has_allowed, dtype=np.uint8).tobytes(), np.ascontiguousarray(fleet.has_prefer, dtype=np.uint8).tobytes(),
                  np.ascontiguousarray(fleet.allowed, dtype=np.uint64).tobytes(), np.ascontiguousarray(fleet.prefer, dtype=np.uint64).tobytes()]
    parts += [repl.tobytes(), reqs.tobytes(), extra.tobytes(), np.ascontiguousarray(sreqs).tobytes()]
    if len(sreqs):
        parts += [np.ascontiguousarray(in_use, dtype=np.int32).tobytes(), np.ascontiguousarray(last_used, dtype=np.int64).tobytes()]
    parts += [np.ascontiguousarray(sx_pod, dtype=np.int32).tobytes(), np.ascontiguousarray(sx_time, dtype=np.int64).tobytes()]
    parts += [np.ascontiguousarray(greqs).tobytes(), np.ascontiguousarray(gx_pod, dtype=np.int32).tobytes(),
              np.ascontiguousarray(gx_time, dtype=np.int64).tobytes(), np.ascontiguousarray(gexpl, dtype=np.int32).tobytes()]
    if len(greqs):
        assert tstats.dtype.itemsize == 32 and len(tstats) == max(T, 1)
        parts += [struct.pack("<q", int(g_expiry)), np.ascontiguousarray(tstats).tobytes()]
    if scaleup is None:
        parts += [struct.pack("<q", -1)]
    else:
        entries, sp, cstats, ststats = scaleup  # cstats: the cluster's stats (1 ORC_STATS row); ststats: typeSetStats per type row
        assert cstats.dtype.itemsize == 32 and len(cstats) == 1 and len(ststats) == max(T, 1)
        parts += [struct.pack("<q", len(entries)), np.ascontiguousarray(sp).tobytes(), np.ascontiguousarray(entries).tobytes(),
                  np.ascontiguousarray(cstats).tobytes(), np.ascontiguousarray(ststats).tobytes()]
    if scaledown is None:
        parts += [struct.pack("<q", -1)]
    else:
        entries, dp, istats = scaledown  # istats: instanceSetStats() (1 ORC_STATS row)
        assert istats.dtype.itemsize == 32 and len(istats) == 1
        parts += [struct.pack("<q", len(entries)), np.ascontiguousarray(dp).tobytes(), np.ascontiguousarray(entries).tobytes(),
                  np.ascontiguousarray(istats).tobytes()]
    if proactive is None:
        parts += [struct.pack("<q", -1)]
    else:
        units, gstats, pstats, ptypes, pod_part = proactive  # cluster stats; per partition: its stats + prohibited type rows; pod -> partition
        assert gstats.dtype.itemsize == 32 and len(gstats) == 1 and len(pstats) == len(ptypes)
        parts += [struct.pack("<qq", len(ptypes), int(units)), np.ascontiguousarray(gstats).tobytes()]
        for k in range(len(ptypes)):
            t = np.ascontiguousarray(sorted(ptypes[k]), dtype=np.int32)
            parts += [np.ascontiguousarray(pstats[k: k + 1]).tobytes(), struct.pack("<q", len(t)), t.tobytes()]
        if len(ptypes):
            parts += [np.ascontiguousarray(pod_part, dtype=np.int32).tobytes()]
    if events is None:
        parts += [struct.pack("<q", -1)]
    else:
        ev, ck = events
        assert ev.dtype.itemsize == 72
        parts += [struct.pack("<qq", len(ev), int(ck)), np.ascontiguousarray(ev).tobytes()]
    if upgrade is None:
        parts += [struct.pack("<q", -1)]
    else:
        assert upgrade.dtype.itemsize == 32
        parts += [struct.pack("<q", len(upgrade)), np.ascontiguousarray(upgrade).tobytes()]
    if types is None:
        parts += [struct.pack("<q", -1)]
    else:
        pod_bits, req_bits, pref_bits = types
        assert len(pod_bits) == fleet.n_pods and len(req_bits) == len(pref_bits)
        parts += [struct.pack("<q", len(req_bits)), np.ascontiguousarray(pod_bits, dtype=np.uint64).tobytes(),
                  np.ascontiguousarray(req_bits, dtype=np.uint64).tobytes(), np.ascontiguousarray(pref_bits, dtype=np.uint64).tobytes()]
    if migration is None:
        parts += [struct.pack("<q", -1)]
    else:
        entries, self_pod, now = migration
        parts += [struct.pack("<qqq", len(entries), int(self_pod), int(now)), np.ascontiguousarray(entries).tobytes()]
    if conc is not None:  # optional trailer: limitModelConcurrency == true (the inputs without it keep their round-3/4 digests)
        rows, cparams = conc  # CONC_ENTRY per cache entry of the a15 / a16 section; 1 CONC_PARAMS row
        assert rows.dtype.itemsize == 32 and cparams.dtype.itemsize == 16 and len(cparams) == 1
        parts += [struct.pack("<q", len(rows)), np.ascontiguousarray(cparams).tobytes(), np.ascontiguousarray(rows).tobytes()]
    return b"".join(parts)


def digest(blob: bytes) -> str:
    return hashlib.sha256(blob).hexdigest()[:16]
