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
