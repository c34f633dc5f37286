"""Event streams for the instance-table listener WITH type constraints (SURVEY.md §8 rows a5 + a18, the incremental path:
TypeConstraintManager.instanceAdded / instanceRemoved / updateInstance ..., MM.java:1455-1568 with typeConstraints != null).
Seeded; replayed by oracle/ref_harness/tcm_harness.cc (the reference's own text) into tests/golden/ref_tcm.npz.

Domain: an instance keeps its labels for its lifetime (they come from the pod's deployment), an instance leaves through a
shutting-down record or a delete event that carries its last record, and a type-constraint configuration change replaces the
whole map (TypeConstraintManager.typeMappingsUpdated)."""
import numpy as np

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl

EVENT = np.dtype([("kind", "<i4"), ("pod", "<i4"), ("labels", "<u8"), ("aux", "<u8"), ("row", _lib.POD_ROW)])
ADDED, UPDATED, DELETED, CONFIG = 0, 1, 2, 3


def stream(seed, P, T, n_labels, label_p, n_events, config_changes=0, checkpoint=25, debug=0, one_label_set_per_partition=False):
    rng = np.random.default_rng(7100 + seed)
    fleet = wl.fuzz_fleet(seed, pods=P, models=20)
    rows = fleet.pods.copy()
    rows["flags"] = _lib.POD_LIVE
    pod_bits = np.array([sum(1 << i for i in range(n_labels) if rng.random() < label_p) for _ in range(P)], np.uint64)
    req_bits, pref_bits = np.zeros(T, np.uint64), np.zeros(T, np.uint64)
    for t in range(T):
        nr, nf = int(rng.choice([0, 0, 1, 2])), int(rng.choice([0, 1, 2]))
        req = rng.choice(n_labels, size=min(nr, n_labels), replace=False) if nr else []
        pref = [l for l in (rng.choice(n_labels, size=min(nf, n_labels), replace=False) if nf else []) if l not in req]
        if one_label_set_per_partition and t < n_labels:  # type t requires exactly label t: an instance's ProhibitedTypeSet then
            req = [t]                                       # determines its label set, so no two label sets share a partition
            pref = [l for l in pref if l != t]
        req_bits[t] = sum(1 << int(l) for l in req)
        pref_bits[t] = sum(1 << int(l) for l in pref)
    ev = np.zeros(n_events, dtype=EVENT)
    present = {}
    cfg_at = set(rng.choice(n_events, config_changes, replace=False).tolist()) if config_changes else set()
    for e in range(n_events):
        if e in cfg_at:
            t = int(rng.integers(0, T))
            nr, nf = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1]))
            req = rng.choice(n_labels, size=min(nr, n_labels), replace=False) if nr else []
            if one_label_set_per_partition and t < n_labels:
                req = [t]
            pref = [l for l in (rng.choice(n_labels, size=min(nf, n_labels), replace=False) if nf else []) if l not in req]
            ev[e] = (CONFIG, t, sum(1 << int(l) for l in req), sum(1 << int(l) for l in pref), rows[0])
            continue
        p = int(rng.integers(0, P))
        r = rows[p].copy()
        r["used"] = min(int(r["capacity"]), max(0, int(r["used"]) + int(rng.integers(-50_000, 50_000))))
        r["count"] = max(0, int(r["count"]) + int(rng.integers(-2, 3)))
        if rng.random() < 0.5 and r["count"] > 0:
            r["lru_time"] = fleet.now - int(rng.integers(1, 50_000_000))
        r["flags"] = _lib.POD_LIVE
        roll = rng.random()
        if p not in present:
            kind = ADDED if roll < 0.85 else UPDATED  # an update for an instance the table has not seen: the listener's safety net (:1544-1547)
        elif roll < 0.6:
            kind = UPDATED
        elif roll < 0.8:
            kind, r["flags"] = UPDATED, _lib.POD_LIVE | _lib.POD_SHUTTING_DOWN  # the record announces the shutdown (:1462-1464)
        else:
            kind = DELETED
            r = present[p]
        ev[e] = (kind, p, pod_bits[p], 0, r)
        rows[p] = r
        if kind == DELETED or (r["flags"] & _lib.POD_SHUTTING_DOWN):
            present.pop(p, None)
        else:
            present[p] = r
    return dict(fleet=fleet, pod_bits=pod_bits, req_bits=req_bits, pref_bits=pref_bits, events=ev, checkpoint=checkpoint,
                local=int(rng.integers(0, P)), debug=debug, exact_partitions=one_label_set_per_partition or label_p == 0.0)


def cases():
    out = []
    for k, (P, T, nl, lp, n, cc) in enumerate([(6, 2, 2, 0.5, 200, 0), (40, 4, 3, 0.4, 600, 0), (120, 6, 5, 0.3, 1500, 0), (300, 8, 6, 0.15, 2500, 0),
                                                (60, 3, 2, 0.0, 400, 0), (90, 5, 4, 0.9, 900, 0), (200, 12, 8, 0.5, 2000, 0), (30, 1, 1, 0.5, 300, 0),
                                                (80, 5, 4, 0.4, 1200, 6), (150, 7, 5, 0.3, 1500, 10)]):
        out.append((f"tcm_events_{k}", stream(k, P, T, nl, lp, n, cc, debug=int(k == 1))))
    for k, (P, T, nl, lp, n, cc) in enumerate([(8, 3, 2, 0.5, 300, 0), (50, 5, 3, 0.5, 800, 0), (140, 8, 4, 0.4, 1600, 0), (260, 10, 5, 0.3, 2400, 0),
                                                (100, 6, 3, 0.5, 1200, 8), (180, 9, 4, 0.4, 1500, 12)]):
        out.append((f"tcm_events_exact_{k}", stream(100 + k, P, T, nl, lp, n, cc, one_label_set_per_partition=True)))
    return out


def input_blob(case, ids):
    f = case["fleet"]
    P, T = f.n_pods, len(case["req_bits"])
    hdr = np.array([P, T, len(case["events"]), case["checkpoint"], int(f.min_space_units), int(f.min_churn_age_ms), case["local"], case["debug"]], "<i8")
    idbuf = b"".join(s.encode("ascii").ljust(16, b"\0") for s in ids)
    return b"MMTCM1\0\0" + hdr.tobytes() + idbuf + case["req_bits"].tobytes() + case["pref_bits"].tobytes() + case["events"].tobytes()


def parse(words, P, T):
    """The harness' output words -> list of checkpoints (dicts) + the four upgradeTracker / republish counters."""
    w = np.asarray(words, "<i8")
    W = (P + 63) // 64
    n_ck, i, cks = int(w[0]), 1, []
    for _ in range(n_ck):
        ck = {"cluster": tuple(int(x) for x in w[i:i + 5])}
        i += 5
        n = int(w[i])
        i += 1
        ck["present"] = w[i:i + 2 * n].reshape(n, 2).copy()  # (instance, prohibited-type mask) in clusterState order
        i += 2 * n
        rows = []
        for _t in range(T + 1):
            ha, al = int(w[i]), w[i + 1:i + 1 + W].copy().view(np.uint64)
            i += 1 + W
            hp, pf = int(w[i]), w[i + 1:i + 1 + W].copy().view(np.uint64)
            i += 1 + W
            rows.append((ha, al, hp, pf, tuple(int(x) for x in w[i:i + 5])))
            i += 5
        ck["types"] = rows
        npart = int(w[i])
        i += 1
        ck["parts"] = [(int(w[i + 7 * k]), tuple(int(x) for x in w[i + 7 * k + 1:i + 7 * k + 6]), int(w[i + 7 * k + 6])) for k in range(npart)]
        i += 7 * npart
        ck["local"] = tuple(int(x) for x in w[i:i + 5])
        ck["n_refresh"], ck["event"] = int(w[i + 5]), int(w[i + 6])
        i += 7
        cks.append(ck)
    assert i + 4 == len(w)
    return cks, [int(x) for x in w[i:i + 4]]


def bitset(words, P):
    return {p for p in range(P) if (int(words[p >> 6]) >> (p & 63)) & 1}


def replay(case):
    """Yields (event index, table: instance -> row, cfg: type -> (required bits, preferred bits), touched: instance or None) after
    every event — the instance table as ANY observer of the stream holds it."""
    table, cfg = {}, {t: (int(case["req_bits"][t]), int(case["pref_bits"][t])) for t in range(len(case["req_bits"]))}
    same = ("lru_time", "capacity", "used", "version", "count", "loading_threads", "loading_in_progress", "rpm")
    for e, ev in enumerate(case["events"]):
        touched = None
        if ev["kind"] == CONFIG:
            cfg[int(ev["pod"])] = (int(ev["labels"]), int(ev["aux"]))
        else:
            touched = int(ev["pod"])
            if ev["kind"] == DELETED or (ev["row"]["flags"] & _lib.POD_SHUTTING_DOWN):
                table.pop(touched, None)
            else:
                if touched in table and all(table[touched][k] == ev["row"][k] for k in same):
                    touched = None  # an identical record: clusterState.add() refuses it and the listener returns (MM.java:1497-1502)
                else:
                    table[touched] = ev["row"].copy()
        yield e, table, cfg, touched
