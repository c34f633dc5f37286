"""The local cache (SURVEY.md §8 rows a12 + a13) held to the REFERENCE'S OWN TEXT: tests/golden/ref_clhm.npz is what
clhm/ConcurrentLinkedHashMap.java, clhm/LinkedDeque.java and ModelCacheUnloadBufManager.java — their method bodies compiled as
they stand by oracle/ref_harness (clhm_harness.cc; regenerate with oracle/ref_harness/make_clhm_vectors.py) — do on the operation
streams of tests/ref_clhm_cases.py.  Here: (1) the reference text itself reproduces what the reference's own tests assert
(EvictionsModelMeshTest, ModelMeshEvictionsTest); (2) the C oracle (oracle/mm_evict_oracle.c) equals the vectors operation by
operation and in its final state.  The device is held to the same file in tests/test_ref_clhm_gpu.py."""
import os

import numpy as np
import pytest

from oracle import bind as ob
from tests import ref_clhm_cases as rc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_clhm.npz")


@pytest.fixture(scope="module")
def vec():
    return np.load(GOLDEN)


def _evicted(v, name):
    outs, ev = v[f"{name}/outs"], v[f"{name}/evicted"]
    return [int(k) for o in outs for k in ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]]


def _final(v, name, c):
    hdr = v[f"{name}/final_hdr"]
    a = int(hdr[:c, 0].sum())
    b = a + int(hdr[c, 0])
    return hdr[c], v[f"{name}/final_key"][a:b], v[f"{name}/final_weight"][a:b], v[f"{name}/final_last_used"][a:b]


def test_the_streams_in_the_file_are_the_generators_streams(vec):
    """The committed operations are reproducible from the seeded generator (up to the documented cut where the reference would
    evict the pinned unload-buffer entry)."""
    for name, caps, reserved, ops in rc.cases():
        assert np.array_equal(vec[f"{name}/caps"], caps) and np.array_equal(vec[f"{name}/reserved"], reserved)
        kept = vec[f"{name}/ops"]
        assert len(kept) <= len(ops)
        want = {bytes(o.tobytes()) for o in ops}
        assert all(bytes(o.tobytes()) in want for o in kept[:: max(1, len(kept) // 50)])


def test_reference_text_reproduces_the_reference_tests_basic_eviction(vec):
    """EvictionsModelMeshTest.basicEvictionTest (:36-125): 18 x 50 MiB fit in 949 MiB; the 19th load evicts myModel0, the next
    two myModel1 and 2; myModel0 again evicts 3; the 160 MiB model evicts 4, 5, 6 and leaves 7."""
    assert _evicted(vec, "kat_basic_eviction") == [0, 1, 2, 3, 4, 5, 6]
    ops, outs = vec["kat_basic_eviction/ops"], vec["kat_basic_eviction/outs"]
    first = next(i for i, o in enumerate(outs) if o["n_evicted"])
    assert (ops[first]["op"], ops[first]["key"]) == (5, 18)  # growing the 19th model's placeholder to 50 MiB is what evicts
    _, keys, _, _ = _final(vec, "kat_basic_eviction", 0)
    assert 7 in keys and not set(range(1, 7)) & set(int(k) for k in keys)


def test_reference_text_reproduces_the_reference_tests_concurrent_eviction(vec):
    """EvictionsModelMeshTest.concurrentEvictionTest (:136-200): exactly the ten oldest go."""
    assert _evicted(vec, "kat_concurrent_eviction") == list(range(10))


def test_reference_text_reproduces_the_reference_tests_standalone_lru(vec):
    """ModelMeshEvictionsTest (:156-187, :240-280): 9 of 12 survive; touched entries outlive untouched ones."""
    ev = _evicted(vec, "kat_standalone_lru")
    assert ev[:3] == [0, 1, 2]
    assert set(ev[3:]) == {6, 7, 8}  # 3, 4, 5 were read and stay
    _, keys, _, _ = _final(vec, "kat_standalone_lru", 0)
    assert {3, 4, 5, 12, 13, 14} <= set(int(k) for k in keys)


def test_c_oracle_equals_the_reference_text(vec):
    n_ops = n_ev = 0
    for name in vec["names"]:
        caps, reserved, ops, outs, ev = (vec[f"{name}/{k}"] for k in ("caps", "reserved", "ops", "outs", "evicted"))
        caches = [ob.CCache(int(caps[c]), None if reserved[c] < 0 else int(reserved[c]), rc.NOW) for c in range(len(caps))]
        for i, op in enumerate(ops):
            h = caches[op["cache"]]
            res, evk = h.apply(int(op["op"]), int(op["key"]), int(op["arg"]), int(op["time"]), int(op["flag"]), rc.NOW)
            o = outs[i]
            want_ev = [int(k) for k in ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]]
            bw = h.lib.orc_ubm_buffer_weight(ob.C.byref(h.u)) if h.u is not None else 0
            assert (res, evk, h.c.weighted_size, h.oldest_time(), bw) == \
                (o["result"], want_ev, o["weighted_size"], o["oldest_time"], o["buffer_weight"]), (name, i, op)
            if h.u is not None and h.u.n_evicted > 900:
                h.u.n_evicted = 0
            n_ev += len(evk)
        n_ops += len(ops)
        for c, h in enumerate(caches):
            hdr, keys, wts, lus = _final(vec, name, c)
            lu, wt, key = h.nodes()
            assert np.array_equal(key, keys) and np.array_equal(wt, wts) and np.array_equal(lu, lus), (name, c)
            assert (h.c.capacity, h.c.weighted_size) == (hdr[1], hdr[2]), (name, c)
            if h.u is not None:
                assert (h.u.total_unloading, h.u.total_occupancy, h.u.cache_deficit) == (hdr[3], hdr[4], hdr[5]), (name, c)
    assert n_ops > 15_000 and n_ev > 2_000


def test_the_streams_exercise_what_the_kats_do_not(vec):
    """Ties on lastUsed (FIFO, LinkedDeque.java:267), touch = max (clhm :1358), the pinned Long.MAX_VALUE entry, deficits and the
    stale oldestTime after a failed unload's setCapacity (clhm :305-316) all occur in the vectors."""
    ties = stale = deficits = older_reads = 0
    for name in vec["names"]:
        hdr = vec[f"{name}/final_hdr"]
        deficits += int((hdr[:, 5] > 0).sum())
        ops, outs = vec[f"{name}/ops"], vec[f"{name}/outs"]
        lus = vec[f"{name}/final_last_used"]
        ties += int((np.diff(lus) == 0).sum())
        older_reads += int(((ops["op"] == 1) & (ops["time"] > 0) & (ops["time"] < rc.NOW - 1_000_000) & (outs["result"] == 1)).sum())
        # after a failed unload (op 9, flag 0) that evicted something: oldestTime differs from the deque's head iff it is stale
        for c in range(len(hdr)):
            idx = np.nonzero(ops["cache"] == c)[0]
            if len(idx) and ops[idx[-1]]["op"] == 9 and ops[idx[-1]]["flag"] == 0 and outs[idx[-1]]["n_evicted"]:
                _, _, _, l = _final(vec, name, c)
                stale += int(outs[idx[-1]]["oldest_time"] != (l[0] if len(l) else -1))
    assert ties > 50 and deficits > 0 and older_reads > 20
