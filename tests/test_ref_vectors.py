"""The pin of the oracle: tests/golden/ref_getnext.npz holds what the REFERENCE'S OWN TEXT decides — PLACEMENT_ORDER
(MM.java:4646-4703), CacheMissForwardingLB.filter / getNext (:4760-5003), ForwardingLB.getNext (:4315-4392),
MapFilteringSet.apply (:4282), isFull, age, InstanceRecord.getRemaining, Utils.STRING_ARRAY_COMP — extracted verbatim by
line range and compiled with g++ against stand-ins for the Java library types (oracle/ref_harness/, run where
/root/reference exists; the vectors travel).  Here the C restatement (oracle/mm_oracle.c) and the Python one must
reproduce them exactly on the fleets of tests/ref_fleets.py (the Python restatement is held to the C one on the same kind of
fleets by tests/test_oracle_cross.py); tests/test_ref_vectors_gpu.py holds the HIP path to the same vectors."""
import os

import numpy as np
import pytest

from oracle import bind as ob
from oracle.bind import OracleFleet
from tests import ref_fleets as rf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_getnext.npz")


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN)


def expected_audit(reqs, ref_place):
    """(chosen, n_candidates, hash) in this repository's conventions from the reference's (chosen, candidates.size(),
    survivors, hash): the early returns report n_candidates = hash = 0 (include/mmplace.h) — every null, and the
    ABORT_REQUESTs of a favourSelf request (:4894, :4932: only those can fire with favourSelf; :4990 needs !favourSelf)."""
    chosen = ref_place[:, 0]
    early = (chosen == -1) | ((chosen == -2) & ((reqs["flags"] & 1) != 0))
    n = np.where(early, 0, ref_place[:, 1])
    h = np.where(early, 0, ref_place[:, 3].astype(np.uint32))
    return chosen, n, h.astype(np.uint32)


def check_place(name, fleet, reqs, got, ref_place):
    chosen, n, h = expected_audit(reqs, ref_place)
    bad = np.nonzero((got["chosen"] != chosen) | (got["n_candidates"] != n) | (got["hash"] != h))[0]
    assert len(bad) == 0, (name, len(bad), [(int(i), got[i], ref_place[i], reqs[i]) for i in bad[:5]])
    # candidates.get(0) is the final best instance whenever the list has a first entry that is not case (b)'s (:4875: there
    # the best itself is not added): with a single candidate that was not filtered, chosen == best
    one = (n == 1) & (chosen >= 0)
    assert np.all(got["best"][one] >= 0)


def test_the_vectors_name_the_reference_text_they_came_from(ref):
    m = str(ref["manifest"])
    for piece in ("ModelMesh.java:4649-4701", "ModelMesh.java:4763-4771", "ModelMesh.java:4777-5003", "ModelMesh.java:4316-4391",
                  "ModelMesh.java:4282-4282", "InstanceRecord.java:204-204", "Utils.java:26-35"):
        assert piece in m, m
    assert len(ref["names"]) >= 60


def test_placement_order_and_load_targets_equal_the_reference_text(ref):
    n_cases = n_dec = 0
    for name, fleet, ids, reqs, extra in rf.place_cases():
        assert rf.digest(rf.input_blob(fleet, ids, reqs, extra)) == bytes(ref[f"{name}/digest"]).decode(), \
            f"{name}: the inputs differ from the ones the vectors were generated from (re-run oracle/ref_harness/make_ref_vectors.py)"
        orc = OracleFleet(fleet)
        assert orc.order_rc == 0, name
        # clusterState iteration order under the reference's comparator (instances present in the table)
        assert np.array_equal(orc.order, ref[f"{name}/order"]), (name, orc.order[:10], ref[f"{name}/order"][:10])
        got = orc.place(reqs, extra, fleet.now, threads=4)
        check_place(name, fleet, reqs, got, ref[f"{name}/place"])
        n_cases += 1
        n_dec += len(reqs)
    assert n_cases >= 56 and n_dec >= 100_000


def test_serve_targets_equal_the_reference_text(ref):
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        assert rf.digest(rf.input_blob(fleet, ids, serve=(reqs, in_use, last_used, xp, xt))) == bytes(ref[f"{name}/digest"]).decode()
        want = ref[f"{name}/serve"]
        live = np.ascontiguousarray(((fleet.pods["flags"] & 2) != 0).astype(np.uint8))
        for i in range(len(reqs)):
            r = reqs[i]
            mm = fleet.models[r["model"]]
            pods = fleet.ent_pod[mm["ent_off"]: mm["ent_off"] + mm["n_loaded"]]
            times = fleet.ent_time[mm["ent_off"]: mm["ent_off"] + mm["n_loaded"]]
            keep = np.ones(len(pods), bool)
            for j in range(r["n_excl"]):
                p, t = xp[r["excl_off"] + j], xt[r["excl_off"] + j]
                keep &= ~((pods == p) & ((t == np.iinfo(np.int64).min) | (times == t)))  # MMP_ANY_TIME: a key exclude
            ch, ts = ob.serve(r["self_pod"], r["flags"] & 1, r["flags"] & 2, pods[keep], times[keep], fleet.now,
                              r["assume_completed_ms"], r["local_in_flight"], r["last_invoke_time"], live, in_use, last_used)
            assert ch == want[i, 0], (name, i, ch, ts, want[i], r)
            if ch >= 0:
                assert ts == want[i, 1], (name, i, ch, ts, want[i])
