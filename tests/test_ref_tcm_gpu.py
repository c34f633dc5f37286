"""GPU: the library driven by the event streams of tests/ref_tcm_cases.py THROUGH THE C ABI (mmp_pods_upsert / mmp_pods_remove /
mmp_types_from_labels / mmp_snapshot_commit), read back at the reference's checkpoints (mmp_get_order, mmp_cluster_stats,
mmp_pod_partitions, mmp_partition_stats, mmp_type_stats) and held to what the REFERENCE'S OWN listener + TypeConstraintManager
text holds there (tests/golden/ref_tcm.npz) — the same checks, with the same documented exceptions, as tests/test_ref_tcm.py makes
of the oracle."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd.solver import Solver
from tests import ref_fleets as rf
from tests import ref_tcm_cases as tc
from tests.test_ref_tcm import GOLDEN, walk

pytestmark = pytest.mark.gpu


def test_device_instance_table_with_type_constraints_equals_the_reference_text():
    ref = np.load(GOLDEN)
    n_ck = 0
    for name, case in tc.cases():
        fleet = case["fleet"]
        rf.string_ids(fleet, 200)
        case["name"] = name
        P, T = fleet.n_pods, len(case["req_bits"])
        cks, _ = tc.parse(ref[f"{name}/words"], P, T)
        s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
        try:
            rows = fleet.pods.copy()
            rows["flags"] = _lib.POD_TOMBSTONE
            s.load_pods(rows)
            dev_table = {}

            def device_eval(f, table, cfg):
                # bring the device's table to `table`: the events since the last checkpoint, net
                gone = [p for p in dev_table if p not in table]
                if gone:
                    s.remove_pods(np.array(gone, np.int32))
                    for p in gone:
                        del dev_table[p]
                changed = [p for p in table if p not in dev_table or dev_table[p].tobytes() != f.pods[p].tobytes()]
                if changed:
                    s.upsert_pods(np.array(changed, np.int32), f.pods[changed])
                    for p in changed:
                        dev_table[p] = f.pods[p].copy()
                req = np.array([cfg[t][0] for t in range(T)], np.uint64)
                pref = np.array([cfg[t][1] & ~cfg[t][0] for t in range(T)], np.uint64)
                al, pf, ha, hp = s.types_from_labels(req, pref, case["pod_bits"])
                W = (P + 63) // 64
                assert np.array_equal(ha, f.has_allowed) and np.array_equal(hp, f.has_prefer), name
                assert np.array_equal(al[:, :W], f.allowed[:, :W]) and np.array_equal(pf[:, :W], f.prefer[:, :W]), name
                s.commit()
                pts, parts = s.partitions()
                sets = [{t for t in range(T + 1) if (m >> t) & 1} for _, m in parts]
                pst = [st for st, _ in parts]
                return s.order()[: len(table)], s.stats(), pts, sets, pst, [s.type_stats(t) for t in range(T)]

            walk(case, cks, device_eval)
            n_ck += len(cks)
        finally:
            s.close()
    assert n_ck >= 700
