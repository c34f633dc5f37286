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


def _ids_for(n):
    return ["p%d" % i for i in range(n)]


def test_records_beyond_one_round_and_beyond_the_lds_tile():
    """> 64 map entries (several lane rounds of the wave path), > 64 fields, and values longer than the
    2 KB LDS tile (walked by one lane): same rows as a JSON library."""
    rng = np.random.default_rng(77)
    n_pods = 400
    ids = _ids_for(n_pods)
    s = Solver(100, 1000)
    try:
        s.load_pod_ids(ids)
        s.load_type_names(["NLCLASSIFIER", "t1"], unknown_type=0)
        recs = []
        for n_inst, n_fail, junk in ((100, 0, 0), (64, 65, 0), (3, 2, 90), (300, 5, 0), (0, 0, 0), (129, 128, 3)):
            inst = sorted(rng.choice(n_pods, n_inst, replace=False).tolist(), key=lambda p: ids[p])
            fail = sorted(rng.choice(n_pods, n_fail, replace=False).tolist(), key=lambda p: ids[p])
            d = {"type": "t1", "lu": int(rng.integers(1, 10**12))}
            for k in range(junk):  # > 64 top-level fields
                d["x%d" % k] = [k, {"y": "a,b:{c}"}]
            d["instanceIds"] = {ids[p]: int(rng.integers(1, 10**12)) for p in inst}
            d["failedIn"] = {ids[p]: int(rng.integers(1, 10**12)) for p in fail}
            d["lul"] = 7
            recs.append((d, inst, fail))
        vals = [json.dumps(d, separators=[(",", ":"), (", ", ": ")][i % 2]) for i, (d, _, _) in enumerate(recs)]
        assert max(map(len, vals)) > 2048 > min(map(len, vals))
        status, lul = s.ingest_models_json(vals)
        assert not status.any()
        rows, ep, et = s.get_models()
        off = 0
        for i, (d, inst, fail) in enumerate(recs):
            assert (rows["type"][i], rows["n_loaded"][i], rows["n_failed"][i], rows["ent_off"][i], rows["last_used"][i]) == \
                (1, len(inst), len(fail), off, d["lu"]), i
            assert list(ep[off: off + len(inst) + len(fail)]) == inst + fail, i
            assert list(et[off: off + len(inst) + len(fail)]) == list(d["instanceIds"].values()) + list(d["failedIn"].values())
            assert lul[i] == 7
            off += len(inst) + len(fail)
        # an InstanceRecord with a label list longer than the tile
        big = json.dumps({"count": 3, "labels": ["label-%04d" % k for k in range(300)], "cap": 99, "lruTime": 12345678901234})
        assert len(big) > 2048
        st, stt = s.ingest_pods_json([big, '{"cap":5}'], np.array([3, 4], np.int32), np.ones(2, np.uint8))
        assert list(st) == [0, 0]
        got = s.get_pods()
        assert (got["count"][3], got["capacity"][3], got["lru_time"][3]) == (3, 99, 12345678901234) and got["capacity"][4] == 5
    finally:
        s.close()


MALFORMED_MODELS = [
    '',                                                         # empty value
    '   ',
    '[]',                                                       # not an object
    '{"lu": 5',                                                 # truncated
    '{"lu": 5}}',                                               # closes twice
    '{"lu": 5} {"lu": 6}',                                      # two values
    '{"lu": 5,}',                                               # trailing comma
    '{"lu": 5 "lul": 6}',                                       # missing comma
    '{"lu": 5,, "lul": 6}',                                     # doubled comma
    '{"lu" 5}',                                                 # missing colon
    '{lu: 5}',                                                  # key is not a string
    '{"lu": "5"}',                                              # wrong type for a long
    '{"lu": 5.5}',
    '{"lu": 1e3}',
    '{"lu": -}',
    '{"lu": }',
    '{"instanceIds": 5}',                                       # map expected
    '{"instanceIds": ["p1"]}',
    '{"instanceIds": {"p1": 5]}',                               # wrong closer
    '{"instanceIds": {"p1": 5,}}',
    '{"instanceIds": {"p1": 5 "p2": 6}}',
    '{"instanceIds": {"p1": {"x": 1}}}',                        # entry value is not a long
    '{"instanceIds": {"p1": "5"}}',
    '{"instanceIds": {p1: 5}}',
    '{"failedIn": {"p1": }}',
    '{"type": "t1" x}',
    '{"type": "t1',                                             # unterminated string
    '{"mPath": "a\\"}',                                         # the escaped quote leaves the string open
]

WELL_FORMED_MODELS = [
    ('{}', (0, 0, 0, 0)),
    ('  {\n}\t', (0, 0, 0, 0)),
    ('{"lu":5,"lu":6}', (0, 0, 0, 6)),                          # a later duplicate wins (Jackson's default)
    ('{"instanceIds":{"p1":1},"instanceIds":{"p2":2,"p3":3}}', (0, 2, 0, 0)),
    ('{"instanceIds":null,"failedIn":null,"type":null}', (0, 0, 0, 0)),
    ('{"instanceIds":{ },"failedIn":{}}', (0, 0, 0, 0)),
    ('{"type":"t1","mPath":"a\\\\","lu":-3}', (1, 0, 0, -3)),   # the string ends after an escaped backslash
    ('{"x":{"instanceIds":{"p1":1}},"lu":2}', (0, 0, 0, 2)),    # a nested look-alike is not the field
    ('{"x":"\\"lu\\":9,","lu":2}', (0, 0, 0, 2)),               # nor is one inside a string
    ('{"fails":{"p1":{"msg":"}{][","t":5}},"failedIn":{"p1":4}}', (0, 0, 1, 0)),
]


def test_malformed_and_edge_case_values():
    s = Solver(100, 1000)
    try:
        s.load_pod_ids(_ids_for(8))
        s.load_type_names(["NLCLASSIFIER", "t1"], unknown_type=0)
        for v in MALFORMED_MODELS:
            try:
                d = json.loads(v)
                ok = isinstance(d, dict) and all(isinstance(d.get(k, 0), int) for k in ("lu", "lul")) and all(
                    d.get(k) is None or (isinstance(d[k], dict) and all(isinstance(x, int) for x in d[k].values()))
                    for k in ("instanceIds", "failedIn"))
            except ValueError:
                ok = False
            assert not ok, v  # the list really is malformed for the bean
        vals = MALFORMED_MODELS + [v for v, _ in WELL_FORMED_MODELS]
        status, _ = s.ingest_models_json(vals)
        nb = len(MALFORMED_MODELS)
        assert list(status[:nb]) == [1] * nb, [v for v, st in zip(vals, status) if not st][:5]
        assert not status[nb:].any(), [v for v, st in zip(vals[nb:], status[nb:]) if st]
        rows, ep, et = s.get_models()
        for i, (v, want) in enumerate(WELL_FORMED_MODELS):
            r = rows[nb + i]
            assert (r["type"], r["n_loaded"], r["n_failed"], r["last_used"]) == want, v
        assert list(rows["n_loaded"][:nb]) == [0] * nb and list(rows["n_failed"][:nb]) == [0] * nb
        # pods: same classes on the InstanceRecord parser
        pv = ['{"count":3,"cap":10}', '{"count":3,"cap":10', '{"count":"3"}', '{"shutdown":1}', '{"shutdown":true,"rpm":4}',
              '{"count":3 "cap":10}', '{}', '{"labels":["a","b"],"count":2}', '{"count":2,}', 'null']
        st, _ = s.ingest_pods_json(pv, np.arange(len(pv), dtype=np.int32) % 8, np.ones(len(pv), np.uint8))
        assert list(st) == [0, 1, 1, 1, 0, 1, 0, 0, 1, 1]
    finally:
        s.close()


def _rand_json(rng, depth=0):
    """A random JSON value (strings with escapes and structural characters inside, nesting, numbers)."""
    r = rng.random()
    if depth > 2 or r < 0.35:
        k = rng.integers(0, 6)
        if k == 0:
            return int(rng.integers(-10**12, 10**12))
        if k == 1:
            return None
        if k == 2:
            return bool(rng.integers(0, 2))
        if k == 3:
            return float(rng.integers(-1000, 1000)) / 8
        alphabet = ['a', 'b', ' ', '"', '\\', '{', '}', '[', ']', ':', ',', '\n', '\t', 'é', '/']
        return "".join(rng.choice(alphabet, int(rng.integers(0, 12))))
    if r < 0.65:
        return [_rand_json(rng, depth + 1) for _ in range(int(rng.integers(0, 4)))]
    return {"k%d" % i + ("\"" if rng.random() < 0.1 else ""): _rand_json(rng, depth + 1) for i in range(int(rng.integers(0, 4)))}


@pytest.mark.parametrize("seed", range(3))
def test_random_documents_agree_with_a_json_library(seed):
    """Property test.  Random well-formed ModelRecord values — known fields of the right type in random
    positions between random junk fields whose values nest, escape and contain every structural character —
    parse to exactly what json.loads says; every strict prefix of a value is rejected (truncation can never
    look well-formed: the closing brace of the record is missing)."""
    rng = np.random.default_rng(9100 + seed)
    n_pods = 50
    ids = _ids_for(n_pods)
    s = Solver(100, 1000)
    try:
        s.load_pod_ids(ids)
        s.load_type_names(["NLCLASSIFIER", "t1", "t2"], unknown_type=0)
        vals, docs = [], []
        for i in range(400):
            d = {}
            for j in range(int(rng.integers(0, 5))):
                d["junk%d" % j] = _rand_json(rng)
            if rng.random() < 0.7:
                d["type"] = str(rng.choice(["t1", "t2", "NLCLASSIFIER", "unheard-of"]))
            if rng.random() < 0.7:
                d["lu"] = int(rng.integers(0, 10**13))
            if rng.random() < 0.5:
                d["lul"] = int(rng.integers(-5, 10**13))
            for fld in ("instanceIds", "failedIn"):
                if rng.random() < 0.7:
                    pods = sorted(rng.choice(n_pods, int(rng.integers(0, 5)), replace=False).tolist(), key=lambda p: ids[p])
                    d[fld] = {ids[p]: int(rng.integers(1, 10**13)) for p in pods}
            items = list(d.items())
            rng.shuffle(items)
            d = dict(items)
            sep = [(",", ":"), (", ", ": "), (" ,\n ", " :\t")][int(rng.integers(0, 3))]
            vals.append(json.dumps(d, separators=sep, ensure_ascii=bool(rng.integers(0, 2))))
            docs.append(d)
        status, lul = s.ingest_models_json(vals)
        assert not status.any(), [v for v, st in zip(vals, status) if st][:3]
        rows, ep, et = s.get_models()
        tmap = {"NLCLASSIFIER": 0, "t1": 1, "t2": 2}
        for i, d in enumerate(docs):
            assert json.loads(vals[i]) == d
            inst, fail = d.get("instanceIds", {}), d.get("failedIn", {})
            assert rows["type"][i] == tmap.get(d.get("type", "NLCLASSIFIER"), 0), vals[i]
            assert (rows["n_loaded"][i], rows["n_failed"][i], rows["last_used"][i], lul[i]) == \
                (len(inst), len(fail), d.get("lu", 0), d.get("lul", 0)), vals[i]
            o = rows["ent_off"][i]
            want = [(ids.index(k), t) for k, t in list(inst.items()) + list(fail.items())]
            assert list(zip(ep[o:o + len(want)].tolist(), et[o:o + len(want)].tolist())) == want, vals[i]
        # truncations
        cut = []
        for v in vals[:120]:
            b = v.encode()
            k = int(rng.integers(0, len(b)))
            cut.append(b[:k])
        status, _ = s.ingest_models_json(cut)
        assert status.all(), [c for c, st in zip(cut, status) if not st][:3]
    finally:
        s.close()
