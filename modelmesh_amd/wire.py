"""Synthetic-input generator (not product logic): serialise a fleet the way the reference stores it in the
KV store (Jackson JSON of InstanceRecord.java:37-69 / ModelRecord.java:61-114: default-valued fields
omitted, TreeMap keys in order), with shuffled field order, stray whitespace, unknown fields, nested
`fails` objects and escaped strings to stress the device parser.  Used by the ingestion tests and by
bench.py's ingestion leg."""
import json

import numpy as np

from ._lib import JAVA_LONG_MAX


def make_ids(rng, n):
    """k8s-style instance ids: 6 chars of replica set + '-' + 5 chars; a few non-standard short ones."""
    rs = ["%06x" % int(x) for x in rng.integers(0, 16**6, 3)]
    ids, seen = [], set()
    while len(ids) < n:
        if rng.random() < 0.03:
            s = "p%d" % len(ids)  # |id| < 7: no replica set (MM.java:4769)
        else:
            s = "%s-%05x" % (rs[int(rng.integers(0, len(rs)))], int(rng.integers(0, 16**5)))
        if s not in seen:
            seen.add(s)
            ids.append(s)
    return ids


def adopt_ids(fleet, ids):
    """Make the fleet consistent with the ids: id_order = String.compareTo rank, replica_set interned in
    first-seen order, and every model's instanceIds / failedIn in TreeMap (id) order."""
    order = sorted(range(len(ids)), key=lambda i: ids[i])
    rank = np.zeros(len(ids), np.uint32)
    rank[order] = np.arange(len(ids), dtype=np.uint32)
    fleet.pods["id_order"] = rank
    intern = {}
    for i, s in enumerate(ids):
        fleet.pods["replica_set"][i] = intern.setdefault(s[:6], len(intern)) if len(s) >= 7 else -1
    m = fleet.models
    for j in range(len(m)):
        o, k, f = int(m["ent_off"][j]), int(m["n_loaded"][j]), int(m["n_failed"][j])
        for a, b in ((o, o + k), (o + k, o + k + f)):
            seg = np.argsort(rank[fleet.ent_pod[a:b]], kind="stable")
            fleet.ent_pod[a:b] = fleet.ent_pod[a:b][seg]
            fleet.ent_time[a:b] = fleet.ent_time[a:b][seg]
    if len(fleet.replaced_rs):
        fleet.replaced_rs = np.unique(fleet.pods["replica_set"][fleet.pods["replica_set"] >= 0])[: len(fleet.replaced_rs)]


def _dump(rng, fields):
    items = list(fields.items())
    rng.shuffle(items)
    if rng.random() < 0.3:
        items.insert(int(rng.integers(0, len(items) + 1)), ("x-new-field", {"a": [1, {"b": "}\\\"]"}], "c": None}))
    sep = [(",", ":"), (", ", ": "), (" ,\n ", " :\t")][int(rng.integers(0, 3))]
    return json.dumps(dict(items), separators=sep)


def pod_values(fleet, rng, start_times):
    out = []
    for i, r in enumerate(fleet.pods):
        f = {"lruTime": int(r["lru_time"]), "count": int(r["count"]), "cap": int(r["capacity"]), "used": int(r["used"]),
             "lThreads": int(r["loading_threads"]), "lInProg": int(r["loading_in_progress"]), "rpm": int(r["rpm"]),
             "shutdown": bool(r["flags"] & 1), "startTime": int(start_times[i]), "vers": int(r["version"]),
             "loc": "host-%d" % (i % 7), "zone": None if i % 3 else "zone \"a\"", "labels": ["gpu", "l%d" % (i % 4)]}
        # Jackson omits fields that hold the bean's default value
        f = {k: v for k, v in f.items() if v not in (0, False, None)}
        out.append(_dump(rng, f))
    return out


def model_values(fleet, ids, type_names, rng, last_unload):
    out = []
    m = fleet.models
    for j in range(len(m)):
        o, k, fl = int(m["ent_off"][j]), int(m["n_loaded"][j]), int(m["n_failed"][j])
        inst = {ids[int(p)]: int(t) for p, t in zip(fleet.ent_pod[o:o + k], fleet.ent_time[o:o + k])}
        failed = {ids[int(p)]: int(t) for p, t in zip(fleet.ent_pod[o + k:o + k + fl], fleet.ent_time[o + k:o + k + fl])}
        f = {"type": type_names[int(m["type"][j])], "mPath": "s3://bucket/m%d" % j, "instanceIds": inst, "refs": j % 3,
             "lu": int(m["last_used"][j]), "lul": int(last_unload[j])}
        if failed:
            f["failedIn"] = failed
            f["fails"] = {i: {"msg": "load \"failed\" {x}", "t": 5} for i in failed}
        f = {kk: v for kk, v in f.items() if v not in (0, False, None) and v != {}}
        if f.get("type") == "NLCLASSIFIER" and rng.random() < 0.5:
            del f["type"]  # absent type == DEFAULT_TYPE
        # keep TreeMap key order inside the id maps while the outer field order is shuffled
        out.append(_dump(rng, f))
    return out


assert JAVA_LONG_MAX == 2**63 - 1
