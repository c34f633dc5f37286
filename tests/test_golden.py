"""Golden vectors committed under tests/golden/ (written by tests/golden/make_golden.py — read its header
for provenance).  CPU: the oracle reproduces them and honours the reference's own known answers
(reference_kats.json, driven generically from the file).  GPU: the HIP path reproduces them through the
C ABI, so device parity does not depend on the oracle being importable or unchanged."""
import json
import os

import numpy as np
import pytest

from oracle import bind as ob
from tests import golden_io as gio
from tests.test_oracle_kat import CDriver, PyDriver, load_model, NOW, HOUR

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _place_cases():
    z = np.load(os.path.join(GOLDEN, "place_fuzz.npz"))
    for i in range(int(z["n_fleets"])):
        p = f"f{i}_"
        yield gio.unpack_fleet(z, p), z[p + "reqs"], z[p + "extra"], z[p + "order"], z[p + "outs"], z[p + "stats"]


def test_golden_files_are_current():
    """The generator's fleet list and the committed file agree (someone edited one without the other)."""
    from tests.golden.make_golden import FLEETS, REFERENCE_KATS
    z = np.load(os.path.join(GOLDEN, "place_fuzz.npz"))
    assert int(z["n_fleets"]) == len(FLEETS)
    assert json.load(open(os.path.join(GOLDEN, "reference_kats.json"))) == json.loads(json.dumps(REFERENCE_KATS))


def test_oracle_reproduces_golden_place_vectors():
    for fleet, reqs, extra, order, outs, stats in _place_cases():
        orc = ob.OracleFleet(fleet)
        assert np.array_equal(np.asarray(orc.order, np.int32), order)
        got = orc.place(reqs, extra, fleet.now)
        for f in ("chosen", "best", "n_candidates", "hash"):
            assert np.array_equal(got[f], outs[f]), f
        assert orc.stats().tobytes() == stats.tobytes()


def test_oracle_reproduces_golden_evict_vectors():
    z = np.load(os.path.join(GOLDEN, "evict_serve.npz"))
    seg, now = z["seg_off"], int(z["now"])
    for i in range(len(z["ev_cache"])):
        c = int(z["ev_cache"][i])
        r = ob.evict_eval(z["cache_lu"][seg[c]:seg[c + 1]], z["cache_wt"][seg[c]:seg[c + 1]], int(z["cache_cap"][c]),
                          int(z["ev_weight"][i]), int(z["ev_last_used"][i]), now)
        assert r.tobytes() == z["ev_result"][i].tobytes(), i


def _kats():
    return json.load(open(os.path.join(GOLDEN, "reference_kats.json")))


def _ids(spec):
    """'myModel0..17' -> [0..17]; 'myModel18' -> [18]."""
    if ".." in spec:
        a, b = spec.replace("myModel", "").split("..")
        return list(range(int(a), int(b) + 1))
    return [int(spec.replace("myModel", ""))]


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_reference_basic_eviction_from_json(driver):
    """EvictionsModelMeshTest.basicEvictionTest driven from reference_kats.json (not from constants in
    the test): capacity arithmetic (MM.java:748-753, :765-771) and the step-by-step eviction order."""
    k = _kats()["basicEvictionTest"]
    cap, dflt, th = k["capacity_units"], k["default_model_units"], k["loading_threads"]
    reserve = max(min(th * dflt // (2 if th <= 2 else 4), cap // 10), cap // 100)
    assert reserve == k["reserve_units"]
    assert ob.load().orc_min_space_units(dflt, th, cap, 1) == k["min_space_units"]
    d = driver(cap, reserve)
    assert d.effective_capacity() == k["effective_units"]
    assert k["effective_units"] * 8192 // (1 << 20) == k["effective_MiB"]
    t, n_now, seen = NOW - HOUR, 0, []
    for step in k["steps"]:
        size = step["size_units"]
        for key in _ids(step["load"]):
            if step.get("last_used") == "now":
                when, n_now = NOW + n_now, n_now + 1
            else:
                t += 10  # registerModel stamps now - 1 h, the test sleeps between adds
                when = t
            immediately = load_model(d, key, size, when, unloads_done=[dflt] * step.get("unloads_finishing_while_waiting", 0))
            if "loads_immediately" in step:
                assert immediately is step["loads_immediately"], step
        seen += [i for name in step["evicts"] for i in _ids(name)]
        assert d.evicted() == seen, step
        for _ in range(step.get("unloads_finishing_after", 0)):
            d.unload_complete(dflt)
        if "sized_after_load_units" in step:
            d.after_load(_ids(step["load"])[0], step["sized_after_load_units"] - size)
            seen += [i for name in step["then_evicts"] for i in _ids(name)]
            assert d.evicted() == seen, step
            assert d.keys()[0] == _ids(step["oldest_survivor"])[0]


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_reference_standalone_capacity_from_json(driver):
    k = _kats()["standalone_evictions"]
    cap, sz, th = k["capacity_units"], k["model_units"], k["loading_threads"]
    reserve = max(min(th * sz // 4, cap // 10), cap // 100)
    assert reserve == k["reserve_units"] and cap - reserve == k["effective_units"]
    assert ob.load().orc_min_space_units(sz, th, cap, 1) == k["min_space_units"]
    d = driver(cap, reserve)
    assert d.effective_capacity() // sz == k["fits_models"]
    acked = 0
    for i in range(12):
        load_model(d, i, sz, NOW - HOUR + i)
        while acked < len(d.evicted()):
            d.unload_complete(sz)
            acked += 1
    assert sorted(d.keys()) == list(range(3, 12))  # "ids[3..12)"


# ---- device ----------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_hip_reproduces_golden_place_vectors():
    from modelmesh_amd.solver import Solver
    for fleet, reqs, extra, order, outs, stats in _place_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            assert np.array_equal(s.order(), order)
            got = s.place(reqs, extra, fleet.now)
            for f in ("chosen", "best", "n_candidates", "hash"):
                assert np.array_equal(got[f], outs[f]), f
            assert s.stats().tobytes() == stats.tobytes()
        finally:
            s.close()


@pytest.mark.gpu
def test_hip_reproduces_golden_evict_vectors():
    from modelmesh_amd import _lib
    from modelmesh_amd.solver import Solver
    z = np.load(os.path.join(GOLDEN, "evict_serve.npz"))
    reqs = np.zeros(len(z["ev_cache"]), dtype=_lib.EVICT_REQ)
    reqs["cache"], reqs["weight"], reqs["last_used"] = z["ev_cache"], z["ev_weight"], z["ev_last_used"]
    s = Solver(100, 1000)
    try:
        s.load_caches(z["seg_off"].astype(np.int32), z["cache_lu"], z["cache_wt"], z["cache_cap"])
        got = s.evict(reqs, int(z["now"]))
    finally:
        s.close()
    want = z["ev_result"]
    for f in ("insert_pos", "n_victims", "self_evicted", "weighted_size", "oldest_time"):
        assert np.array_equal(got[f], want[f]), f
