"""ModelMeshFailureExpiryTest.failureExpiryTest (src/test/java/com/ibm/watson/modelmesh/ModelMeshFailureExpiryTest.java
:52-128) as a closed loop on this path — the one reference-held anchor round 1 had not used (VERDICT r1 §"What's
missing" 1).  One instance; LOAD_FAILURE_EXPIRY_MS = 2000 (IN_USE_LOAD_FAILURE_EXPIRY_MS = LOAD_FAILURE_EXPIRY_MS / 2,
MM.java:219-221), janitor every second; the model's only load fails at t0 and a client keeps predicting every 400 ms:

* while the failure record exists the instance is in CacheMissExcludeSet.failed (MM.java:4734-4743), so the
  load-target decision finds no instance (null) and the recorded failure is re-thrown: the predicts at ~0 s and at
  ~1.4 s fail;
* checkLoadFailureCount (MM.java:4607-4627) does not fire for one record (MAX_LOAD_FAILURES = 3), with or without
  the IN_USE expiry cutoff;
* the janitor (MM.java:6040-6052) drops the record once now - failedTime > expiryAge, where expiryAge is the full
  2000 ms because a failed load leaves no cache entry (lastUsed = -1): at its 3 s pass;
* the predict at ~4.4 s therefore decides again, elects the (only, own) instance — ABORT_REQUEST = load locally — and
  the second load succeeds.

The timeline below is the test's own (sleep 400 / 1000 / 3000 ms); the numbers live in tests/golden/reference_kats.json.
CPU: both restatements.  GPU: the device in lock step with the oracle (tests/test_reference_kat_gpu.py imports this)."""
import json
import os

import numpy as np

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Fleet
from oracle import bind as ob
from oracle import py_gates

NOW = wl.NOW_MS
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["failureExpiry"]


def single_instance_fleet(failed_at, now):
    rows = np.zeros(1, dtype=wl.POD_ROW)
    rows["capacity"], rows["used"] = 131072, 0
    rows["count"], rows["lru_time"] = 0, np.iinfo(np.int64).max
    rows["loading_threads"], rows["version"] = 8, 1
    rows["flags"] = wl.POD_LIVE
    models = np.zeros(1, dtype=wl.MODEL_ROW)
    models["n_failed"] = 0 if failed_at is None else 1
    models["last_used"] = now - 1
    ent = np.zeros(0 if failed_at is None else 1, np.int32)
    return Fleet(pods=rows, models=models, ent_pod=ent, ent_time=np.full(len(ent), failed_at or 0, np.int64),
                 min_space_units=6553, min_churn_age_ms=600_000, now=now)


def janitor_keeps_failure(failed_at, now, last_used):
    """MM.java:6044-6052: lastUsed = -1 when the model has no cache entry."""
    expiry = KAT["load_failure_expiry_ms"]
    in_use = expiry // 2                                                  # :221
    age = in_use if (last_used > 0 and now - last_used < 3 * 60_000) else expiry   # :6047-6048, :280
    return not (now - failed_at > age)


def request(now):
    r = np.zeros(1, dtype=wl.PLACE_REQ)
    r["model"], r["self_pod"], r["pick"], r["last_used"] = 0, 0, 0x9E3779B9, now - 1
    r["fresh_lru"], r["fresh_capacity"] = np.iinfo(np.int64).max, 131072
    return r


def run_timeline(place, breached):
    """place(fleet, req, now) -> chosen; breached(fail_times, now) -> bool.  Returns [(t_ms, outcome)] of the three
    asserted predicts; the background predicts every 400 ms are driven too (they must not disturb the record)."""
    t0 = NOW
    failed_at = t0                    # registerModel(loadNow, sync) -> LOADING_FAILED
    outcomes, checks = [], list(KAT["asserted_predicts_ms"])
    events = sorted({*range(0, 4800, KAT["background_predict_period_ms"]), *checks})
    next_janitor = KAT["janitor_period_ms"]
    for t in events:
        now = t0 + t
        while next_janitor <= t:      # janitor passes that ran before this predict
            jn = t0 + next_janitor
            if failed_at is not None and not janitor_keeps_failure(failed_at, jn, -1):
                failed_at = None
            next_janitor += KAT["janitor_period_ms"]
        fleet = single_instance_fleet(failed_at, now)
        fails = [] if failed_at is None else [failed_at]
        assert not breached(fails, now)          # one record never reaches MAX_LOAD_FAILURES
        chosen = place(fleet, request(now), now)
        if chosen == -2:                          # ABORT_REQUEST: load locally; the dummy runtime's second load succeeds
            outcome, failed_at = "loaded", None
        else:
            assert chosen == -1                   # null: nowhere to load -> the recorded failure is re-thrown
            outcome = "failed"
        if t in checks:
            outcomes.append((t, outcome))
    return outcomes


def test_failure_expiry_kat_on_both_restatements():
    in_use = KAT["load_failure_expiry_ms"] // 2

    def place_c(fleet, r, now):
        return int(ob.OracleFleet(fleet).place(r, None, now)["chosen"][0])

    def place_py(fleet, r, now):
        from oracle import py_oracle as po
        from tests.test_oracle_cross import _py_pods
        mesh = po.Mesh(fleet.min_space_units, fleet.min_churn_age_ms, now)
        pods = _py_pods(fleet)
        order = mesh.sorted_cluster_state(pods)
        m = fleet.models[0]
        failed = set(int(x) for x in fleet.ent_pod[m["n_loaded"]: m["n_loaded"] + m["n_failed"]])
        q = r[0]
        fresh = dict(lru_time=int(q["fresh_lru"]), capacity=int(q["fresh_capacity"]), used=int(q["fresh_used"]),
                     count=int(q["fresh_count"]), rpm=int(q["fresh_rpm"]))
        chosen, _, _, _ = po.get_next(mesh, pods, order, {0}, set(), None, None, [set(), set(), failed], 0,
                                      bool(q["flags"] & 1), fresh, int(q["last_used"]), int(q["pick"]))
        return int(chosen)

    def breached(fails, now):
        return py_gates.load_failures_breached(fails, now, in_use)

    want = [(t, o) for t, o in zip(KAT["asserted_predicts_ms"], KAT["asserted_outcomes"])]
    assert run_timeline(place_c, breached) == want
    assert run_timeline(place_py, breached) == want
    # the boundary of the janitor's rule is strict (now - failedTime > expiryAge) and a recently used cache entry halves it
    e = KAT["load_failure_expiry_ms"]
    assert janitor_keeps_failure(NOW, NOW + e, -1) and not janitor_keeps_failure(NOW, NOW + e + 1, -1)
    assert janitor_keeps_failure(NOW, NOW + e // 2, NOW) and not janitor_keeps_failure(NOW, NOW + e // 2 + 1, NOW)
