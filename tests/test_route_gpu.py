"""GPU: the cache-hit route in one launch (mmp_route_batch = the request guards + the serve target of every request,
route_batch_kernel) against (a) the reference's own text — the gate half on the reference-text guard cases, the serve half on the
reference-text serve cases (tests/golden/ref_getnext.npz) — and (b) the two separate calls on paired requests that share one
exclusion pool, as cacheHitExcludeTl's MapFilteringSet is shared by goLocal and ForwardingLB.getNext (MM.java:3634, :4316)."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd.solver import Solver
from tests import ref_fleets as rf
from tests.test_ref_vectors import GOLDEN, check_gates

pytestmark = pytest.mark.gpu


def _serve_reqs_for(rng, fleet, gate_reqs):
    n = len(gate_reqs)
    s = np.zeros(n, dtype=_lib.SERVE_REQ)
    s["model"] = gate_reqs["model"]
    s["self_pod"] = gate_reqs["self_pod"]
    s["flags"] = rng.integers(0, 4, n)
    s["local_in_flight"] = rng.integers(0, 3, n)
    s["last_invoke_time"] = fleet.now - rng.choice([0, 10, 1000], n)
    s["assume_completed_ms"] = rng.choice([3000, 30_000], n)
    s["excl_off"], s["n_excl"] = gate_reqs["excl_off"], gate_reqs["n_excl"]  # the same MapFilteringSet
    return s


def test_route_equals_the_two_calls_and_the_reference_text_guards():
    ref = np.load(GOLDEN)
    for name, fleet, ids, r, xp, xt, expl, expiry in rf.gate_cases():
        rng = np.random.default_rng(hash(name) & 0xFFFF)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            P = fleet.n_pods
            in_use = rng.integers(0, 3, P).astype(np.int32)
            last_used = (fleet.now - rng.choice([0, 5, 5, 100, 10_000], P)).astype(np.int64)
            sreqs, counters = s.serve_counters(_serve_reqs_for(rng, fleet, r), in_use, last_used)
            g1 = s.gates(r, xp, xt, expl, fleet.now, expiry)
            s1 = s.serve_k(sreqs, counters, xp, xt, fleet.now)
            g2, s2 = s.route(r, sreqs, counters, xp, xt, expl, fleet.now, expiry)
        finally:
            s.close()
        assert np.array_equal(g1["bits"], g2["bits"]) and np.array_equal(g1["initial_size"], g2["initial_size"]), name
        assert np.array_equal(s1["chosen"], s2["chosen"]) and np.array_equal(s1["chosen_load_start"], s2["chosen_load_start"]), name
        check_gates(name, g2["bits"], g2["initial_size"], ref[f"{name}/gate"])


def test_route_serve_half_equals_the_reference_text():
    ref = np.load(GOLDEN)
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        want = ref[f"{name}/serve"]
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            sreqs, counters = s.serve_counters(reqs, in_use, last_used)
            g = np.zeros(len(reqs), dtype=_lib.GATE_REQ)  # guards of a request that asks for nothing but the route
            g["model"], g["self_pod"] = reqs["model"], reqs["self_pod"]
            g["excl_off"], g["n_excl"] = sreqs["excl_off"], sreqs["n_excl"]
            g["loaded_time"] = -1
            _, got = s.route(g, sreqs, counters, xp, xt, np.zeros(0, np.int32), fleet.now)
        finally:
            s.close()
        assert np.array_equal(got["chosen"], want[:, 0]), (name, np.nonzero(got["chosen"] != want[:, 0])[0][:5])
        remote = want[:, 0] >= 0
        assert np.array_equal(got["chosen_load_start"][remote], want[remote, 1]), name


def test_the_host_mirror_follows_removed_pods_and_new_registry_records():
    """Solver.serve_counters reads a host mirror of the instance list and the registry (what a Java host holds anyway): a
    removed row leaves it, a record upserted beyond the loaded registry extends it, and replaced entries do not pile up."""
    name, fleet, ids, reqs, in_use, last_used, xp, xt = next(iter(rf.serve_cases()))
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        _, c0 = s.serve_counters(reqs, in_use, last_used)
        gone = np.unique(c0["pod"])[:3].astype(np.int32)
        s.remove_pods(gone)
        _, c1 = s.serve_counters(reqs, in_use, last_used)
        assert len(gone) and not np.isin(c1["pod"], gone).any() and len(c1) < len(c0)
        # a new record past the end of the loaded registry, with one copy on a listed instance
        M = s.n_models
        keep = int(np.setdiff1d(np.unique(c0["pod"]), gone)[0])
        row = np.zeros(1, dtype=_lib.MODEL_ROW)
        row["n_loaded"], row["last_used"] = 1, fleet.now
        for rep in range(3000):  # (and the same record again and again: the mirror's entry pool stays bounded)
            s.upsert_models(np.array([M], np.int32), row, np.array([keep], np.int32), np.array([fleet.now], np.int64))
        assert len(s._ent_pod) < len(fleet.ent_pod) * 2 + 2100
        q = reqs[:1].copy()
        q["model"] = M
        q2, c2 = s.serve_counters(q, in_use, last_used)
        assert q2["n_cnt"][0] == 1 and c2["pod"][0] == keep
    finally:
        s.close()


@pytest.mark.parametrize("n", [1, 5, 256, 257])
def test_small_route_calls_take_a_latency_slot_and_equal_the_two_calls(n):
    """invokeModel asks for one route at a time: calls of up to 256 requests ride a latency slot (pinned buffers, one launch, the
    pinned completion flag) — same rows as the staged path (257) and as the two separate calls."""
    name, fleet, ids, r, xp, xt, expl, expiry = next(iter(rf.gate_cases()))
    rng = np.random.default_rng(n)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_fleet(fleet)
        P = fleet.n_pods
        in_use = rng.integers(0, 3, P).astype(np.int32)
        last_used = (fleet.now - rng.choice([0, 5, 5, 100, 10_000], P)).astype(np.int64)
        for rep in range(3):  # (the slot is reused: sequence numbers, stale rows)
            sel = rng.choice(len(r), size=min(n, len(r)), replace=False)
            g = r[sel].copy()
            sreqs, counters = s.serve_counters(_serve_reqs_for(rng, fleet, g), in_use, last_used)
            g1 = s.gates(g, xp, xt, expl, fleet.now, expiry)
            s1 = s.serve_k(sreqs, counters, xp, xt, fleet.now)
            g2, s2 = s.route(g, sreqs, counters, xp, xt, expl, fleet.now, expiry)
            assert np.array_equal(g1["bits"], g2["bits"]) and np.array_equal(g1["initial_size"], g2["initial_size"]), (n, rep)
            assert np.array_equal(s1["chosen"], s2["chosen"]) and np.array_equal(s1["chosen_load_start"], s2["chosen_load_start"]), (n, rep)
    finally:
        s.close()
