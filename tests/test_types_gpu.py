"""GPU parity for row a18: per-type allowed / preferred instance sets rebuilt from labels on the
device vs the CPU restatement of TypeConstraintManager (oracle/py_types.py)."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle import py_types
from oracle.bind import OracleFleet, unpack_bitmap

pytestmark = pytest.mark.gpu


def _check(seed, P, T, n_labels, label_p):
    rng = np.random.default_rng(9000 + seed)
    fleet = wl.fuzz_fleet(seed, pods=P, models=50)
    names = [f"label-{i}" for i in range(n_labels)]
    pod_bits = np.zeros(P, np.uint64)
    pod_sets = {}
    for p in range(P):
        ls = {names[i] for i in range(n_labels) if rng.random() < label_p}
        pod_sets[p] = ls
        pod_bits[p] = sum(1 << names.index(l) for l in ls)
    cfg, req_bits, pref_bits = {}, np.zeros(T, np.uint64), np.zeros(T, np.uint64)
    for t in range(T):
        nr = int(rng.choice([0, 0, 1, 2]))
        nf = int(rng.choice([0, 1, 2]))
        req = sorted(rng.choice(names, size=min(nr, n_labels), replace=False)) if nr else []
        pref = sorted(rng.choice(names, size=min(nf, n_labels), replace=False)) if nf else []
        cfg[t] = (list(req), list(pref))
        req_bits[t] = sum(1 << names.index(l) for l in req)
        pref_bits[t] = sum(1 << names.index(l) for l in pref)
    present = {p: pod_sets[p] for p in range(P) if not (fleet.pods["flags"][p] & 5)}
    want, want_default = py_types.type_tables(present, cfg)

    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_pods(fleet.pods)
        al, pf, ha, hp = s.types_from_labels(req_bits, pref_bits, pod_bits)
        A, F = unpack_bitmap(al, P).astype(bool), unpack_bitmap(pf, P).astype(bool)
        for t in range(T):
            wa, wp = want[t]
            assert bool(ha[t]) == (wa is not None), (t, cfg[t])
            if wa is not None:
                assert set(np.nonzero(A[t])[0]) == set(wa), (t, cfg[t])
            assert bool(hp[t]) == (wp is not None), (t, cfg[t], wp)
            if wp is not None:
                assert set(np.nonzero(F[t])[0]) == set(wp), (t, cfg[t])
        assert not ha[T] and bool(hp[T]) == (want_default is not None)
        if want_default is not None:
            assert set(np.nonzero(F[T])[0]) == set(want_default)
        # the tables are installed: decisions made with them equal decisions made with host-loaded tables
        fleet.n_types, fleet.allowed, fleet.prefer, fleet.has_allowed, fleet.has_prefer = T + 1, al, pf, ha, hp
        fleet.models["type"] = rng.integers(0, T + 1, fleet.n_models)
        s.load_replaced_rs(fleet.replaced_rs)
        s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
        s.commit()
        reqs, extra = wl.fuzz_requests(fleet, seed, 1500)
        got = s.place(reqs, extra, fleet.now)
        wantp = OracleFleet(fleet).place(reqs, extra, fleet.now)
        for f in ("chosen", "best", "n_candidates", "hash"):
            assert np.array_equal(got[f], wantp[f]), f
    finally:
        s.close()


@pytest.mark.parametrize("seed,P,T,n_labels,label_p", [(0, 5, 2, 2, 0.5), (1, 64, 4, 3, 0.4), (2, 200, 6, 5, 0.3),
                                                       (3, 700, 8, 6, 0.15), (4, 65, 3, 2, 0.0), (5, 130, 5, 4, 0.9)])
def test_type_sets_from_labels(seed, P, T, n_labels, label_p):
    _check(seed, P, T, n_labels, label_p)


def test_required_label_on_one_instance():
    """Appendix C.4 — ModelMeshErrorPropagationTest.java:52-95: type my-type-1 requires my-label-1, only
    one replica carries it, so exactly that instance is ever chosen whatever the ingress pod."""
    fleet = wl.make_fleet("C1")
    P = fleet.n_pods
    pod_bits = np.zeros(P, np.uint64)
    pod_bits[3] = 1
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        s.load_pods(fleet.pods)
        s.types_from_labels(np.array([1], np.uint64), np.array([0], np.uint64), pod_bits)
        fleet.models["type"] = 0
        fleet.models["n_loaded"] = 0
        fleet.models["n_failed"] = 0
        s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
        s.commit()
        reqs, extra = wl.make_requests(fleet, 3, extra_frac=0.0)
        out = s.place(reqs, extra, fleet.now)
        assert set(np.unique(out["chosen"])) <= {3, -2}  # pod 3, or SELF when the caller is pod 3
        assert np.all(out["chosen"][reqs["self_pod"] != 3] == 3)
    finally:
        s.close()
