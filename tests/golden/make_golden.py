#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/.

Provenance (read this before trusting a file here):

* `reference_kats.json` — numbers TRANSCRIBED FROM THE REFERENCE'S OWN TESTS (file:line in every
  entry; SURVEY.md Appendix C works the arithmetic).  The reference is Java 21 + un-vendored Maven
  artifacts (kv-utils 0.5.1, litelinks-core 1.7.2, pom.xml:62-63) and this image has no JDK, so the
  reference itself cannot be executed to produce vectors: these known answers are what pins the oracle.
  The file is written by hand-coded tables in this script, not computed by the oracle.
* `place_fuzz.npz`, `evict_serve.npz` — inputs + outputs of the CPU ORACLE (oracle/, both the C and the
  independently written Python restatement, which this script requires to agree before it writes
  anything).  They are regression vectors: they freeze today's restated semantics so that neither the
  oracle nor the HIP path can drift silently; they are not reference outputs.

usage: python tests/golden/make_golden.py      (rewrites the three files in place)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from modelmesh_amd import workload as wl  # noqa: E402
from oracle import bind as ob  # noqa: E402
from tests import golden_io as gio  # noqa: E402

FLEETS = [(0, None, 5), (1, None, 64), (2, None, 130), (3, "full", 40), (4, "full", 200), (5, "prefer", 64),
          (6, "prefer", 130), (7, "prefer", 333), (11, None, 1)]


def place_vectors():
    from tests.test_oracle_cross import _compare  # C restatement == Python restatement on these inputs
    from tools.oracle_coverage import uncovered
    miss = uncovered()  # gcov: every branch of the getNext restatement must be taken by the fleets the vectors come from
    if miss:
        raise SystemExit("place_fuzz.npz NOT written: branches of oracle/mm_oracle.c never taken:\n" +
                         "\n".join(f"  :{ln}: {note}: {text}" for ln, text, note in miss))
    out = {}
    for i, (seed, profile, pods) in enumerate(FLEETS):
        fleet = wl.fuzz_fleet(seed, pods=pods, models=160, profile=profile)
        reqs, extra = wl.fuzz_requests(fleet, seed, 300)
        _compare(fleet, reqs, extra)
        orc = ob.OracleFleet(fleet)
        want = orc.place(reqs, extra, fleet.now)
        gio.pack_fleet(out, f"f{i}_", fleet)
        out[f"f{i}_reqs"], out[f"f{i}_extra"] = reqs, extra
        out[f"f{i}_order"] = np.asarray(orc.order, np.int32)
        out[f"f{i}_outs"] = want
        out[f"f{i}_stats"] = np.asarray(orc.stats())
    out["n_fleets"] = np.int32(len(FLEETS))
    np.savez_compressed(os.path.join(HERE, "place_fuzz.npz"), **out)


def evict_serve_vectors():
    from oracle import py_oracle as po
    rng = np.random.default_rng(0x601D)
    now = wl.NOW_MS
    # clhm put evaluations: (deque, capacity, new weight, new lastUsed) -> (insertion index, victims, evicted weight)
    n_caches, seg = 64, [0]
    lu_all, wt_all, caps = [], [], []
    for c in range(n_caches):
        e = int(rng.choice([0, 1, 3, 17, 64, 65, 200]))
        lu = np.sort(now - rng.choice([1, 5, 5, 5, 900, 7_200_000], e) - rng.integers(0, 3, e)).astype(np.int64)
        wt = rng.choice([1, 640, 2560, 6400], e).astype(np.int32)
        lu_all.append(lu)
        wt_all.append(wt)
        caps.append(int(wt.sum() + rng.choice([-3000, 0, 1, 5000, 100000])))
        seg.append(seg[-1] + e)
    n = 600
    ev = dict(cache=rng.integers(0, n_caches, n).astype(np.int32),
              weight=rng.choice([1, 640, 6400, 60000], n).astype(np.int32),
              last_used=np.where(rng.random(n) < 0.4, 0, now - rng.choice([0, 5, 900, 8_000_000], n)).astype(np.int64))
    res = np.zeros(n, dtype=ob.ORC_EVICT_RESULT)
    for i in range(n):
        c = int(ev["cache"][i])
        r_c = ob.evict_eval(lu_all[c], wt_all[c], max(caps[c], 0), int(ev["weight"][i]), int(ev["last_used"][i]), now)
        # second restatement: replay the put on a Clhm built from the same deque
        p = po.Clhm(max(caps[c], 0))
        p.capacity = 1 << 62
        for k, (t, w) in enumerate(zip(lu_all[c], wt_all[c])):
            p.putIfAbsent(k, int(w), int(t), now)
        p.capacity = max(caps[c], 0)
        before = list(p.keys())
        p.putIfAbsent(10**6, int(ev["weight"][i]), int(ev["last_used"][i]), now)
        after = list(p.keys())
        n_victims = len(before) + 1 - len(after)
        assert int(r_c["n_victims"]) == n_victims, (i, r_c, n_victims)
        assert int(r_c["weighted_size"]) == p.weightedSize and int(r_c["oldest_time"]) == p.oldestTime(), (i, r_c)
        res[i] = r_c
    np.savez_compressed(os.path.join(HERE, "evict_serve.npz"), seg_off=np.asarray(seg, np.int64),
                        cache_lu=np.concatenate(lu_all), cache_wt=np.concatenate(wt_all),
                        cache_cap=np.maximum(np.asarray(caps, np.int64), 0), now=np.int64(now),
                        ev_cache=ev["cache"], ev_weight=ev["weight"], ev_last_used=ev["last_used"], ev_result=res)


REFERENCE_KATS = {
    "_provenance": "transcribed from the reference's own tests; see tests/golden/make_golden.py",
    "basicEvictionTest": {
        "source": "src/test/java/com/ibm/watson/modelmesh/EvictionsModelMeshTest.java:36-125; capacity/default size/"
                  "loading threads from src/test/java/com/ibm/watson/modelmesh/example/ExampleModelRuntime.java:192-195",
        "capacity_units": 131072, "default_model_units": 6400, "loading_threads": 6,
        "reserve_units": 9600, "effective_units": 121472, "effective_MiB": 949,
        "min_space_units": 6553,
        # one entry per ensureLoaded in the test; "unloads_*" = how many 1-second unloads of earlier victims
        # finish while this load waits for space / after it (the test's runtime unloads asynchronously)
        "steps": [
            {"load": "myModel0..17", "size_units": 6400, "loads_immediately": True, "evicts": []},
            {"load": "myModel18", "size_units": 6400, "loads_immediately": True, "evicts": ["myModel0"]},
            {"load": "myModel19", "size_units": 6400, "loads_immediately": True, "evicts": ["myModel1"]},
            {"load": "myModel20", "size_units": 6400, "loads_immediately": False, "evicts": ["myModel2"],
             "unloads_finishing_while_waiting": 1, "unloads_finishing_after": 2},
            {"load": "myModel0", "size_units": 6400, "last_used": "now", "evicts": ["myModel3"], "unloads_finishing_after": 1},
            {"load": "myModel21", "size_units": 6400, "last_used": "now", "evicts": ["myModel4"], "unloads_finishing_after": 1,
             "sized_after_load_units": 20480, "then_evicts": ["myModel5", "myModel6"], "oldest_survivor": "myModel7"}]},
    "concurrentEvictionTest": {
        "source": "EvictionsModelMeshTest.java:136-200",
        "preloaded": 18, "concurrent_adds": 10, "evicted": "the 10 oldest, nothing else"},
    "standalone_evictions": {
        "source": "src/test/java/com/ibm/watson/modelmesh/ModelMeshEvictionsTest.java:156-280; sizes from "
                  "DummyModelMesh / LocalInstanceParameters.java:28,36",
        "model_units": 2560, "capacity_units": 25600, "loading_threads": 8, "reserve_units": 2560,
        "effective_units": 23040, "fits_models": 9, "min_space_units": 2560,
        "sequential_12": {"survivors": "ids[3..12)"},
        "big_eviction": {"normal": 11, "big_factor": 4, "survivors": "ids[6..12)"},
        "reuse": {"fill": 9, "touch": [0, 1, 2], "add": 3, "survivors_include": [0, 1, 2, 9, 10, 11]}},
    "secondCopyTrigger": {
        "source": "ModelMeshEvictionsTest.java:411-447", "rate_check_ms": 100, "second_copy_window_ms": [4000, 10000],
        "uses_at_s": [0.06, 1.06, 12.56, 17.56], "copies_after_each": [1, 1, 1, 2]},
    "typeConstraint": {
        "source": "src/test/java/com/ibm/watson/modelmesh/ModelMeshErrorPropagationTest.java:52-95",
        "type": "my-type-1", "required_label": "my-label-1", "labelled_replicas": ["9000"], "copies": 1},
    "clusterEvictions": {
        "source": "src/test/java/com/ibm/watson/modelmesh/ModelMeshEvictionsTest.java:292-310 (testMultiLoadCluster), "
                  ":323-358 (testMultiLoadWithEvictionCluster), :371-409 (testMultiLoadWithEvictionClusterReuse); "
                  "clusterSize :558, wiggle room :340, expected ids :695-702",
        "cluster_size": 3, "multi_load": 9, "fits_models": 27, "beyond_capacity": 3, "wiggle_room": 6,
        "must_be_loaded_from_index": 9, "reuse": {"touch_first": 5, "add": 3}},
    "failureExpiry": {
        "source": "src/test/java/com/ibm/watson/modelmesh/ModelMeshFailureExpiryTest.java:52-128 (LOAD_FAILURE_EXPIRY 2000 :53, "
                  "janitor 1 s :52; predicts at 0 / +1000 / +3000 ms :99-118; background predict every 400 ms :86-96); "
                  "MM.java:219-221, :4607-4627, :6040-6052",
        "load_failure_expiry_ms": 2000,
        "janitor_period_ms": 1000,
        "background_predict_period_ms": 400,
        "asserted_predicts_ms": [0, 1400, 4400],
        "asserted_outcomes": ["failed", "failed", "loaded"],
    },
    "loadFailure": {
        "source": "src/test/java/com/ibm/watson/modelmesh/ModelMeshLoadFailureTest.java:432-492",
        "max_load_failures": 3, "failed_instances_never_retried_within_expiry": True},
}


def main():
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(REFERENCE_KATS, f, indent=1, sort_keys=True)
        f.write("\n")
    place_vectors()
    evict_serve_vectors()
    for name in sorted(os.listdir(HERE)):
        print(name, os.path.getsize(os.path.join(HERE, name)))


if __name__ == "__main__":
    main()
