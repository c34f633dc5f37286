"""GPU: mmp_miss_batch — the cache-miss route of a request in ONE call (the request guards + the load target: two launches on one
latency slot, one wait) — equals mmp_gate_batch + mmp_place_batch on the same requests, for slot-sized calls (1, 5, 256) and for
batches that take the two calls (257, 3000), with and without extra exclusions, and equals the oracle's load targets."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests import ref_fleets as rf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 256, 257, 3000])
@pytest.mark.parametrize("extras", [False, True])
def test_miss_equals_the_two_calls_and_the_oracle(n, extras):
    name, fleet, ids, r, xp, xt, expl, expiry = next(iter(rf.gate_cases()))
    rng = np.random.default_rng(n + 17 * extras)
    orc = OracleFleet(fleet)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        for rep in range(3):  # (a slot is reused: sequence numbers, rows of the call before)
            sel = rng.choice(len(r), size=n, replace=n > len(r))
            g = r[sel].copy()
            reqs, extra = wl.make_requests(fleet, 900 + rep, n=max(n, 1), extra_frac=0.3 if extras else 0.0)
            reqs = reqs[:n].copy()
            reqs["model"], reqs["self_pod"] = g["model"], g["self_pod"]
            if not extras:
                reqs["n_extra"], reqs["extra_off"] = 0, 0
                extra = np.zeros(0, np.int32)
            g1 = s.gates(g, xp, xt, expl, fleet.now, expiry)
            p1 = s.place(reqs, extra, fleet.now)
            g2, p2 = s.miss(g, reqs, xp, xt, expl, extra, fleet.now, expiry)
            assert np.array_equal(g1["bits"], g2["bits"]) and np.array_equal(g1["initial_size"], g2["initial_size"]), (n, rep)
            want = orc.place(reqs, extra, fleet.now)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(p1[f], p2[f]), (n, rep, f)
                assert np.array_equal(p2[f], want[f]), (n, rep, f)
    finally:
        s.close()


def test_miss_rejects_two_models_in_one_request():
    name, fleet, ids, r, xp, xt, expl, expiry = next(iter(rf.gate_cases()))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        g = r[:2].copy()
        reqs, extra = wl.make_requests(fleet, 1, n=2, extra_frac=0.0)
        reqs["model"] = g["model"]
        reqs["model"][1] = (g["model"][1] + 1) % fleet.n_models
        with pytest.raises(Exception):
            s.miss(g, reqs, xp, xt, expl, extra, fleet.now, expiry)
    finally:
        s.close()
