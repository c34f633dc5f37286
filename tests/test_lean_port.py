"""The cpu_baseline leg of bench.py times orc_place_lean (oracle/mm_oracle.c: the getNext restatement without the
checker's per-call allocations, pos_of rebuild and audit hash) on a persistent worker pool.  It shares its body with
the checker; these tests keep the two entry points and the pool's dynamic chunking honest."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from oracle.bind import OracleFleet


def _same(got, want):
    for f in ("chosen", "best", "n_candidates"):
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("seed", range(12))
def test_lean_port_equals_the_checker_on_fuzz_fleets(seed):
    fleet = wl.fuzz_fleet(seed, profile=[None, "full", "pref"][seed % 3])
    reqs, extra = wl.fuzz_requests(fleet, seed, 600)
    orc = OracleFleet(fleet)
    want = orc.place(reqs, extra, fleet.now)
    for threads in (1, 3):
        pool = orc.lean_pool(threads)
        try:
            got, lat = pool(reqs, extra, fleet.now, latencies=True)
            _same(got, want)
            assert (lat > 0).all()
            _same(pool(reqs[:7], extra, fleet.now), want[:7])  # a batch smaller than one chunk
        finally:
            pool.close()


def test_lean_port_equals_the_checker_on_scenarios_and_c2():
    for name, fleet, reqs, extra in wl.scenario_fleets():
        orc = OracleFleet(fleet)
        pool = orc.lean_pool(2)
        try:
            _same(pool(reqs, extra, fleet.now), orc.place(reqs, extra, fleet.now))
        finally:
            pool.close()
    fleet = wl.make_fleet("C2")
    reqs, extra = wl.make_requests(fleet, 5)
    orc = OracleFleet(fleet)
    pool = orc.lean_pool(4)
    try:
        _same(pool(reqs, extra, fleet.now), orc.place(reqs, extra, fleet.now, threads=4))
    finally:
        pool.close()
