"""GPU: the HIP path against the reference-text vectors (tests/golden/ref_getnext.npz, see tests/test_ref_vectors.py): the
order of the instance table, every load-target decision (chosen, shortlist size, audit hash of the shortlist as the
reference's own `candidates` list gives it) and every serve-target decision, through the C ABI, bit-exact."""
import os

import numpy as np
import pytest

from modelmesh_amd.solver import Solver
from tests import ref_fleets as rf
from tests.test_ref_vectors import GOLDEN, check_place

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN)


def test_hip_order_and_load_targets_equal_the_reference_text(ref):
    n_dec = 0
    for name, fleet, ids, reqs, extra in rf.place_cases():
        assert rf.digest(rf.input_blob(fleet, ids, reqs, extra)) == bytes(ref[f"{name}/digest"]).decode(), name
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            want_order = ref[f"{name}/order"]
            assert np.array_equal(s.order()[: len(want_order)], want_order), name  # (absent rows sort behind the present ones)
            got = s.place(reqs, extra, fleet.now)
            check_place(name, fleet, reqs, got, ref[f"{name}/place"])
            n_dec += len(reqs)
        finally:
            s.close()
    assert n_dec >= 100_000


@pytest.mark.parametrize("env", [{}, {"MMP_LONG_MODE": "1"}, {"MMP_FORCE_WAVE": "1"}, {"MMP_MEMO_FROM": "0", "MMP_NO_SPLIT": "1"},
                                 {"MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0"}, {"MMP_NO_LONG_MEMO": "1"}, {"MMP_LONG_SPLIT_FROM": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_hip_single_caller_form_equals_the_reference_text(ref, env, monkeypatch):
    """mmp_place_batch_c (the caller's side once per call, 24-byte requests) on the batches of ONE calling instance the
    reference's own getNext text decided — through the kernels the library picks, the prefix-table kernels forced on, the
    wave path alone, and the per-type shortlists checked first for batches of every size, in both forms: in front of the lane phase
    of the same kernel (MMP_MEMO_FROM=0, MMP_NO_SPLIT=1) and as a launch of its own with a tail launch behind it (MMP_SPLIT_FROM=0);
    the full-cluster case without the recorded long walks (MMP_NO_LONG_MEMO=1) and with them in a launch of their own
    (MMP_LONG_SPLIT_FROM=0: place_long_memo_c_kernel + place_long_tail_c_kernel)."""
    from modelmesh_amd._lib import split_caller
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, fleet, ids, reqs, extra in rf.caller_place_cases():
        want = ref[f"{name}/place"]
        if "MMP_FORCE_WAVE" in env and name == "caller_C3_full_cluster":
            reqs, want = reqs[:1100], want[:1100]
        caller, rc = split_caller(reqs)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            got = s.place_c(caller, rc, extra, fleet.now)
        finally:
            s.close()
        check_place(name, fleet, reqs, got, want)


@pytest.mark.parametrize("env", [{"MMP_LONG_MODE": "1"}, {"MMP_LONG_MODE": "0"}, {"MMP_NO_CASEB": "1"}, {"MMP_NO_LONG_LDS": "1"},
                                 {"MMP_FORCE_WAVE": "1"}, {"MMP_LONG_MODE": "0", "MMP_MEMO_FROM": "0", "MMP_NO_SPLIT": "1"},
                                 {"MMP_LONG_MODE": "0", "MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0"},
                                 {"MMP_LONG_MODE": "0", "MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0", "MMP_TAIL_BLOCKS": "5"},
                                 {"MMP_NO_LONG_MEMO": "1"}, {"MMP_LONG_SPLIT_FROM": "0"}, {"MMP_LONG_SPLIT_FROM": "0", "MMP_TAIL_BLOCKS": "5"},
                                 {"MMP_LONG_DENSE_FROM": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_every_device_path_equals_the_reference_text_on_the_bench_configurations(ref, env, monkeypatch):
    """C3 / C3 with every instance full / C3 with sparse types / C4 (tests/ref_fleets.py:big_place_cases) through each path the
    library can take for them: the prefix-table kernel forced on (MMP_LONG_MODE=1) and off (0: window + lane + wave paths), case
    (b) without its whole-window tables (MMP_NO_CASEB=1), the long path's tables read from global memory, and the general
    wave-per-decision path alone (MMP_FORCE_WAVE=1; the full-cluster case there on a sample: whole-table shortlists take ~3 us
    each on it), and the per-type shortlists checked first — in front of the lane phase of the same kernel (MMP_MEMO_FROM=0,
    MMP_NO_SPLIT=1: place_batch_m_kernel) and split into the check's own launch + a tail launch (MMP_SPLIT_FROM=0: place_memo_kernel +
    place_tail_kernel, 16 and 5 tail workgroups); the full cluster without the recorded long walks (MMP_NO_LONG_MEMO=1), with them in
    a launch of their own (MMP_LONG_SPLIT_FROM=0: place_long_memo_kernel + place_long_tail_kernel) and inside the 4-wavefront
    instantiation (MMP_LONG_DENSE_FROM=0) — all equal to what the reference's own getNext text decided."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, fleet, ids, reqs, extra in rf.big_place_cases():
        want = ref[f"{name}/place"]
        if "MMP_FORCE_WAVE" in env and name == "C3_full_cluster":
            reqs, want = reqs[:1500], want[:1500]
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            got = s.place(reqs, extra, fleet.now)
        finally:
            s.close()
        check_place(name, fleet, reqs, got, want)


def test_hip_serve_targets_equal_the_reference_text(ref):
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        want = ref[f"{name}/serve"]
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            got = s.serve(reqs, in_use, last_used, xp, xt, fleet.now)
        finally:
            s.close()
        assert np.array_equal(got["chosen"], want[:, 0]), (name, np.nonzero(got["chosen"] != want[:, 0])[0][:5])
        remote = want[:, 0] >= 0  # (ABORT_REQUEST returns before filtered.add(chosenId, chosenTimeStamp), :4384-4389)
        assert np.array_equal(got["chosen_load_start"][remote], want[remote, 1]), name


def test_hip_request_guards_equal_the_reference_text(ref):
    from tests.test_ref_vectors import check_gates
    for name, fleet, ids, r, xp, xt, expl, expiry in rf.gate_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            got = s.gates(r, xp, xt, expl, fleet.now, expiry)
        finally:
            s.close()
        check_gates(name, got["bits"], got["initial_size"], ref[f"{name}/gate"])


def test_hip_scaleup_plan_equals_the_reference_text(ref):
    from tests.test_ref_vectors import check_scaleup
    for name, fleet, ids, entries, sp in rf.scaleup_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            out, ov, _ = s.scaleup_plan(entries, sp)
        finally:
            s.close()
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])


def test_hip_scaleup_edge_cases_equal_the_reference_text(ref):
    from tests.test_ref_vectors import check_scaleup
    for name, fleet, ids, entries, sp in rf.scaleup_edge_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            out, ov, sk = s.scaleup_plan(entries, sp)
        finally:
            s.close()
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])
        assert bool(sk) == (len(entries) == 0), name


def test_hip_scaleup_plan_latency_based_equals_the_reference_text(ref):
    """mmp_scaleup_plan_conc: the rate task of a mesh that limits model concurrency — thresholds, counter resets and
    averageModelParallelism (the path's one double: exact) against the reference's own text."""
    from tests.test_ref_vectors import check_conc, check_scaleup
    for name, fleet, ids, entries, conc, sp, cp in rf.scaleup_conc_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            out, couts, ov, sk, res = s.scaleup_plan_conc(entries, conc, sp, cp)
        finally:
            s.close()
        check_scaleup(name, out, ov, ref[f"{name}/scale"], ref[f"{name}/overloaded"])
        check_conc(name, couts, res, ref[f"{name}/conc"], ref[f"{name}/average_model_parallelism"], sk)


def test_hip_scaledown_plan_conc_and_edge_cases_equal_the_reference_text(ref):
    for name, fleet, ids, entries, conc, dp, dyn in rf.scaledown_conc_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            rem = s.scaledown_plan_conc(entries, conc, dp, dyn)
        finally:
            s.close()
        bad = np.flatnonzero(rem != ref[f"{name}/removed"])
        assert bad.size == 0, (name, bad[:8], entries[bad[:8]], conc[bad[:8]])
    for name, fleet, ids, entries, dp in rf.scaledown_edge_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            rem = s.scaledown_plan(entries, dp)
        finally:
            s.close()
        assert np.array_equal(rem, ref[f"{name}/removed"]), name


def test_hip_scaledown_plan_equals_the_reference_text(ref):
    for name, fleet, ids, entries, dp in rf.scaledown_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            rem = s.scaledown_plan(entries, dp)
        finally:
            s.close()
        bad = np.flatnonzero(rem != ref[f"{name}/removed"])
        assert bad.size == 0, (name, bad[:8], entries[bad[:8]])


def test_hip_proactive_plan_equals_the_reference_text(ref):
    from tests.test_ref_vectors import plan_calls
    for name, fleet, ids, units, partitioned in rf.proactive_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            n_parts = len(s.partitions()[1]) if partitioned else 0
            got = plan_calls(lambda k, skip: s.proactive_plan(units, fleet.now, fleet.n_models, partition=k, skip_models=skip),
                             fleet, units, partitioned, n_parts)
        finally:
            s.close()
        want = ref[f"{name}/proactive"]
        assert got.shape == want.shape and np.array_equal(got, want), (name, got[:5], want[:5])


def test_hip_instance_table_equals_the_reference_text(ref):
    """The listener's event stream against the library's table: per checkpoint the rows that changed since the last one go in
    through mmp_pods_upsert / mmp_pods_remove, a commit re-ranks; order and ClusterStats equal the reference text's."""
    from tests.test_ref_vectors import STAT_FIELDS, ref_checkpoints
    for name, fleet, ids, ev, ck, tables in rf.table_event_cases():
        cps = ref_checkpoints(ref, name)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            prev = None
            for k, (want_stats, want_order) in enumerate(cps):
                t = tables[k]
                if prev is None:
                    s.load_pods(t)
                else:
                    ch = np.flatnonzero(t != prev).astype(np.int32)
                    gone = ch[(t["flags"][ch] & 4) != 0]
                    upd = ch[(t["flags"][ch] & 4) == 0]
                    if len(upd):
                        s.upsert_pods(upd, t[upd])
                    if len(gone):
                        s.remove_pods(gone)
                s.commit()
                prev = t
                assert np.array_equal(s.order(), want_order), (name, k)
                st = s.stats()
                assert [int(st[x]) for x in STAT_FIELDS] == [int(v) for v in want_stats], (name, k, st, want_stats)
        finally:
            s.close()


def test_hip_library_upgrade_tracker_equals_the_reference_text(ref):
    """(host-side state of the context: no kernel — row a19)"""
    from tests.test_ref_vectors import ref_upgrade_maps
    for name, ev in rf.upgrade_event_cases():
        want = ref_upgrade_maps(ref, name)
        s = Solver(100, 1000)
        try:
            for i, e in enumerate(ev):
                if e["kind"] == 0:
                    s.upgrade_instance_added(int(e["labels_key"]), int(e["replica_set"]), int(e["start_time"]), int(e["now"]))
                elif e["kind"] == 1:
                    s.upgrade_instance_removed(int(e["labels_key"]), int(e["replica_set"]), int(e["now"]))
                else:
                    s.upgrade_housekeeping(int(e["now"]))
                assert s.upgrade_replaced() == want[i], (name, i)
        finally:
            s.close()


def test_hip_type_constraint_tables_equal_the_reference_text(ref):
    """mmp_types_from_labels (type_sets_kernel / type_prefer_kernel), the partitions and the subset stats of the device against
    TypeConstraintManager's text."""
    from oracle.bind import unpack_bitmap
    from tests.test_ref_vectors import STAT_FIELDS, check_type_tables
    for name, fleet, ids, pod_bits, req_bits, pref_bits in rf.type_constraint_cases():
        P, T = fleet.n_pods, len(req_bits)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_pods(fleet.pods)
            al, pf, ha, hp = s.types_from_labels(req_bits, pref_bits, pod_bits)
            s.load_replaced_rs(fleet.replaced_rs)
            s.load_models(fleet.models, fleet.ent_pod, fleet.ent_time)
            s.commit()
            A, F = unpack_bitmap(al, P).astype(bool), unpack_bitmap(pf, P).astype(bool)
            tables = lambda t: (set(np.nonzero(A[t])[0].tolist()) if ha[t] else None, set(np.nonzero(F[t])[0].tolist()) if hp[t] else None)  # noqa: E731
            pts, parts = s.partitions()
            parts = [(tuple(int(st[x]) for x in STAT_FIELDS), frozenset(t for t in range(T + 1) if (pr >> t) & 1)) for st, pr in parts]
            g = s.stats()
            type_stats = lambda t: tuple(int((g if t is None else s.type_stats(t))[x]) for x in STAT_FIELDS)  # noqa: E731
            check_type_tables(name, ref, fleet, T, tables, (pts, parts), type_stats)
        finally:
            s.close()


def test_hip_preshutdown_migration_equals_the_reference_text(ref):
    for name, fleet, ids, entries, self_pod, now in rf.migration_cases():
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            s.load_fleet(fleet)
            act, wait = s.migration_plan(entries, self_pod, now)
        finally:
            s.close()
        bits = ref[f"{name}/migration"]
        assert np.array_equal(act, bits & 1) and np.array_equal(wait, bits >> 1), name
