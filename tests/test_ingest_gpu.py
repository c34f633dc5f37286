"""the instance table and the registry arrive as the reference's KV wire format
(Jackson JSON of InstanceRecord.java:37-69 / ModelRecord.java:61-114) and are parsed on the device.
Oracle of the parser = Python's json module; then the ingested snapshot must make exactly the decisions
the oracle makes on the structured fleet."""
import json

import numpy as np
import pytest

from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from oracle.bind import OracleFleet
from tests import wire
from tests.util import assert_same_decisions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,pods", [(0, 1), (1, 70), (2, 500), (3, 3000)])
def test_wire_format_ingestion_end_to_end(seed, pods):
    rng = np.random.default_rng(4000 + seed)
    fleet = wl.fuzz_fleet(seed + 60, pods=pods, models=400)
    fleet.pods["flags"] &= ~np.uint32(4)  # tombstones do not exist on the wire
    ids = wire.make_ids(rng, pods)
    wire.adopt_ids(fleet, ids)
    n_types = max(fleet.n_types, 1)
    type_names = ["NLCLASSIFIER"] + ["type-%d" % t for t in range(1, n_types)]
    start = (fleet.now - rng.integers(0, 10**9, pods)).astype(np.int64)
    lul = np.where(rng.random(fleet.n_models) < 0.3, fleet.now - rng.integers(1, 10**8, fleet.n_models), 0).astype(np.int64)
    pv = wire.pod_values(fleet, rng, start)
    mv = wire.model_values(fleet, ids, type_names, rng, lul)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    try:
        io, rs = s.load_pod_ids(ids)
        assert np.array_equal(io, fleet.pods["id_order"]) and np.array_equal(rs, fleet.pods["replica_set"])
        live = ((fleet.pods["flags"] & 2) != 0).astype(np.uint8)
        # ingest in two shuffled halves (upsert semantics), one value malformed on the first attempt
        perm = rng.permutation(pods)
        for part in (perm[: pods // 2], perm[pods // 2:]):
            vals = [pv[i] for i in part]
            if len(vals) > 2:
                vals[1] = vals[1][:-3]  # truncated JSON
            status, st = s.ingest_pods_json(vals, part, live[part])
            assert status.sum() == (1 if len(vals) > 2 else 0)
            if len(vals) > 2:
                assert status[1] == 1
                st2, stt2 = s.ingest_pods_json([pv[part[1]]], part[1:2], live[part[1:2]])
                assert st2[0] == 0
                st[1] = stt2[0]
            assert np.array_equal(st, start[part])
        want_rows, _ = wire.parsed_pod_rows(pv, fleet.pods)
        got_rows = s.get_pods()
        for f in got_rows.dtype.names:
            if f != "reserved":
                assert np.array_equal(got_rows[f], want_rows[f]), f
        assert np.array_equal(got_rows, fleet.pods)  # and that is the structured fleet itself

        s.load_type_names(type_names, unknown_type=0)
        status, got_lul = s.ingest_models_json(mv)
        assert not status.any() and np.array_equal(got_lul, lul)
        rows, ep, et = s.get_models()
        for f in ("type", "ent_off", "n_loaded", "n_failed", "last_used"):
            assert np.array_equal(rows[f], fleet.models[f]), f
        assert np.array_equal(ep, fleet.ent_pod) and np.array_equal(et, fleet.ent_time)

        s.load_types(fleet.n_types, fleet.allowed, fleet.prefer, fleet.has_allowed, fleet.has_prefer)
        s.load_replaced_rs(fleet.replaced_rs)
        s.commit()
        reqs, extra = wl.fuzz_requests(fleet, seed, 1500)
        orc = OracleFleet(fleet)
        assert np.array_equal(s.order(), orc.order)
        assert_same_decisions(fleet, reqs, s.place(reqs, extra, fleet.now), orc.place(reqs, extra, fleet.now, threads=4))
    finally:
        s.close()


def test_models_with_unknown_ids_and_malformed_values():
    s = Solver(100, 1000)
    try:
        s.load_pod_ids(["aaaaaa-00001", "aaaaaa-00002", "bbbbbb-00001"])
        s.load_type_names(["t0", "NLCLASSIFIER"], unknown_type=2)
        vals = [
            json.dumps({"type": "t0", "instanceIds": {"aaaaaa-00002": 5, "gone-pod-1": 6, "bbbbbb-00001": 7}, "lu": 9}),
            '{"instanceIds": {"aaaaaa-00001": 11}, "failedIn": {"bbbbbb-00001": 12}}',   # no type -> NLCLASSIFIER
            '{"type": "brand-new", "lu": 3, "instanceIds": {}}',                           # unknown type, no copies
            '{"type": "t0", "instanceIds": {"aaaaaa-00001": }}',                           # malformed
            '{"type": null, "autoDel": true, "refs": 2}',
        ]
        status, _ = s.ingest_models_json(vals)
        assert list(status) == [0, 0, 0, 1, 0]
        rows, ep, et = s.get_models()
        assert list(rows["type"]) == [0, 1, 2, rows["type"][3], 1]
        assert list(rows["n_loaded"]) == [3, 1, 0, 0, 0] and list(rows["n_failed"]) == [0, 1, 0, 0, 0]
        assert list(rows["ent_off"]) == [0, 3, 5, 5, 5]
        assert list(ep) == [1, -1, 2, 0, 2] and list(et) == [5, 6, 7, 11, 12]
        assert list(rows["last_used"]) == [9, 0, 3, 0, 0]
    finally:
        s.close()
