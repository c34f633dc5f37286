"""GPU parity of the stateful per-pod caches (SURVEY.md §8 rows a12 + a13): random sequences of clhm
operations (ConcurrentLinkedHashMap.java :726-985 put / get / weight update / remove) and of
ModelCacheUnloadBufManager methods (:130-349) are replayed by one wavefront per cache and compared,
operation by operation, with the KAT-pinned C oracle: result, evicted keys in listener order, unload
buffer weight, weightedSize, oldestTime — and the final deque + manager fields."""
import numpy as np
import pytest

from modelmesh_amd import _lib
from modelmesh_amd.solver import Solver
from oracle import bind as ob

pytestmark = pytest.mark.gpu
NOW = 1_760_000_000_000


def _mk_caches(rng, n_caches, managed):
    """Oracle caches pre-filled through the oracle's own API, exported as the device's initial state."""
    caches = []
    for c in range(n_caches):
        cap = int(rng.choice([5_000, 25_600, 131_072, 400_000]))
        reserved = int(rng.choice([0, 256, 2_560, 9_600])) if managed[c] else None
        h = ob.CCache(cap, reserved, NOW)
        for k in range(int(rng.integers(0, 60))):
            w = int(rng.choice([1, 640, 2_560, 6_400]))
            t = NOW - int(rng.choice([10, 1_000, 1_000, 5_000, 3_600_000])) - int(rng.integers(0, 3))
            if managed[c]:
                h.lib.orc_ubm_insert_new_entry(ob.C.byref(h.u), 1000 * c + k, w, t, NOW)
            else:
                h.put_if_absent(1000 * c + k, w, t, NOW)
        if h.u is not None:
            h.u.n_evicted = 0
        caches.append(h)
    return caches


def _export(caches):
    seg = [0]
    lus, wts, keys, caps, ubm = [], [], [], [], np.zeros(len(caches), dtype=_lib.UBM_STATE)
    for i, h in enumerate(caches):
        lu, wt, key = h.nodes()
        lus.append(lu), wts.append(wt), keys.append(key)
        seg.append(seg[-1] + len(lu))
        caps.append(h.c.capacity)
        if h.u is not None:
            ubm[i] = (h.u.reserved, h.u.total_unloading, h.u.total_occupancy, h.u.cache_deficit, 0)
        else:
            ubm[i]["reserved"] = -1
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)  # noqa: E731
    return (np.array(seg, np.int32), cat(lus, np.int64), cat(wts, np.int32), cat(keys, np.int32),
            np.array(caps, np.int64), ubm)


def _random_ops(rng, caches, managed, n_ops, round_id):
    ops = np.zeros(n_ops, dtype=_lib.CACHE_OP)
    for i in range(n_ops):
        c = int(rng.integers(0, len(caches)))
        _, _, keys = caches[c].nodes()
        live = [int(k) for k in keys if k != _lib.UNLOADBUF_KEY]
        known = int(rng.choice(live)) if live and rng.random() < 0.8 else 1000 * c + 900 + int(rng.integers(0, 50))
        fresh = 1000 * c + 100 + round_id * 200 + i
        t = int(rng.choice([0, NOW - 500, NOW - 2_000, NOW - 7_200_000, NOW + 5]))
        w = int(rng.choice([1, 640, 2_560, 6_400, 30_000]))
        if managed[c]:
            op = int(rng.choice([4, 4, 5, 6, 7, 8, 8, 9, 10, 11, 12, 1]))
        else:
            op = int(rng.choice([0, 0, 0, 1, 1, 2, 2, 3]))
        key, arg, flag = known, w, 0
        if op in (0, 4, 12):
            key = fresh if rng.random() < 0.85 else known
            if op != 0:
                arg = 1 if rng.random() < 0.7 else w
        elif op == 2:
            t = int(rng.choice([-1, -1, 0, NOW - 100]))
        elif op == 5:
            arg = int(rng.choice([639, 2_559, 6_399]))
        elif op in (6, 7):
            arg = int(rng.choice([1, 640, 6_400, 100_000]))
        elif op == 8:
            arg = int(rng.choice([-600, -1, 0, 1, 700, 20_000]))
        elif op == 9:
            arg, flag = int(rng.choice([640, 2_560])), int(rng.random() < 0.85)
        elif op == 11:
            arg = int(rng.choice([1, 640]))
        ops[i] = (c, op, key, arg, t, flag, 0)
        # keep the oracle in step so that later ops see realistic keys
        yield_res = caches[c].apply(op, key, arg, t, flag, NOW)
        ops_res.append(yield_res)
    return ops


ops_res = []


@pytest.mark.parametrize("seed", range(8))
def test_replay_matches_oracle(seed):
    rng = np.random.default_rng(7000 + seed)
    n_caches = int(rng.choice([1, 5, 33]))
    managed = rng.random(n_caches) < 0.6
    caches = _mk_caches(rng, n_caches, managed)
    s = Solver(100, 1000)
    try:
        s.load_caches_keyed(*_export(caches))
        for rnd in range(4):
            ops_res.clear()
            ops = _random_ops(rng, caches, managed, int(rng.choice([1, 40, 300])), rnd)
            want = list(ops_res)
            outs, ev = s.cache_replay(ops, NOW)
            for i, (res, evk) in enumerate(want):
                o = outs[i]
                assert o["result"] == res, (seed, rnd, i, ops[i], o, res)
                got_ev = list(ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]])
                assert got_ev == evk, (seed, rnd, i, ops[i], got_ev, evk)
            # final state of every cache
            for c, h in enumerate(caches):
                st = s.cache_read(c)
                lu, wt, key = h.nodes()
                assert np.array_equal(st["key"], key), (seed, rnd, c)
                assert np.array_equal(st["last_used"], lu) and np.array_equal(st["weight"], wt), (seed, rnd, c)
                assert st["weighted_size"] == h.c.weighted_size and st["capacity"] == h.c.capacity
                if h.u is not None:
                    u = st["ubm"]
                    assert (u["total_unloading"], u["total_occupancy"], u["cache_deficit"]) == \
                        (h.u.total_unloading, h.u.total_occupancy, h.u.cache_deficit), (seed, rnd, c)
                    h.u.n_evicted = 0
            # per-op side outputs on the last op of each cache are the final values
            last = {}
            for i in range(len(ops)):
                last[int(ops[i]["cache"])] = i
            for c, i in last.items():
                h = caches[c]
                assert outs[i]["weighted_size"] == h.c.weighted_size
                assert outs[i]["oldest_time"] == h.oldest_time()
                if h.u is not None:
                    assert outs[i]["buffer_weight"] == h.lib.orc_ubm_buffer_weight(ob.C.byref(h.u))
    finally:
        s.close()


def test_reference_kat_on_device():
    """EvictionsModelMeshTest.basicEvictionTest arithmetic (:36-125) replayed on the
    device: capacity 131072, reserve 9600 -> 18 x 6400 fit, the 19th evicts myModel0."""
    cap, reserved = 131072, 9600
    h = ob.CCache(cap, reserved, NOW)
    s = Solver(100, 1000)
    try:
        s.load_caches_keyed(*_export([h]))
        ops = np.zeros(0, dtype=_lib.CACHE_OP)
        rows = []
        for m in range(19):
            t = NOW - 3_600_000 + 10 * m
            rows.append((0, _lib.COP_UBM_INSERT_NEW_ENTRY, m, 1, t, 0, 0))
            rows.append((0, _lib.COP_UBM_ADJUST_SPACE_REQUEST, m, 6399, 0, 0, 0))
        ops = np.array(rows, dtype=_lib.CACHE_OP)
        outs, ev = s.cache_replay(ops, NOW)
        evicted = [list(ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]) for o in outs]
        assert all(e == [] for e in evicted[:-1]) and evicted[-1] == [0]
        st = s.cache_read(0)
        assert set(st["key"]) == set(range(1, 19)) | {_lib.UNLOADBUF_KEY}
    finally:
        s.close()


def test_concurrent_eviction_kat_on_device():
    """EvictionsModelMeshTest.concurrentEvictionTest (:136-200) as ONE replay call: 18 x 50 MiB loaded,
    ten placeholders inserted together, then grown — exactly myModel0..9 are evicted, in that order."""
    h = ob.CCache(131072, 9600, NOW)
    s = Solver(100, 1000)
    try:
        s.load_caches_keyed(*_export([h]))
        rows = []
        t0 = NOW - 3_600_000
        for m in range(18):
            rows.append((0, _lib.COP_UBM_INSERT_NEW_ENTRY, m, 1, t0 + 10 * m, 0, 0))
            rows.append((0, _lib.COP_UBM_ADJUST_SPACE_REQUEST, m, 6399, 0, 0, 0))
            rows.append((0, _lib.COP_UBM_CLAIM_SPACE, 0, 6400, 0, 0, 0))
        for m in range(18, 28):
            rows.append((0, _lib.COP_UBM_INSERT_NEW_ENTRY, m, 1, t0 + 1000 + m, 0, 0))
        for m in range(18, 28):
            rows.append((0, _lib.COP_UBM_ADJUST_SPACE_REQUEST, m, 6399, 0, 0, 0))
        outs, ev = s.cache_replay(np.array(rows, dtype=_lib.CACHE_OP), NOW)
        evicted = [int(k) for o in outs for k in ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]]
        assert evicted == list(range(10))
        assert all(outs[i]["result"] == 1 for i in range(2, 54, 3))  # every claim of the first 18 succeeded
        st = s.cache_read(0)
        assert sorted(int(k) for k in st["key"] if k != _lib.UNLOADBUF_KEY) == list(range(10, 28))
    finally:
        s.close()
