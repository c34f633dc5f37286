"""Known-answer tests that pin the eviction / capacity half of the oracle to the reference's OWN
tests.  Both restatements (C: oracle/mm_evict_oracle.c, Python:
oracle/py_oracle.py) are driven through the call sequence ModelMesh makes for a load
(insertNewEntry → adjustNewEntrySpaceRequest → cacheSpaceIsReady/claim → adjustWeightAfterLoad,
MM.java:5063, :2094-2098, :2281-2292, :2425) and for an eviction (onEviction → entryRemoved →
unloadComplete, MM.java:2877, :2035)."""
import ctypes as C

import pytest

from oracle import bind as ob
from oracle import py_oracle as po

NOW = 1_760_000_000_000
HOUR = 3_600_000


class PyDriver:
    def __init__(self, capacity, reserved):
        self.c = po.Clhm(capacity)
        self.u = po.UnloadBufManager(self.c, reserved, NOW)

    def insert(self, key, t):
        return self.u.insertNewEntry(key, 1, t, NOW)

    def grow(self, key, inc):
        self.u.adjustNewEntrySpaceRequest(inc, key, NOW)

    def ready(self, req):
        return self.u.cacheSpaceIsReady(req)

    def claim(self, req):
        return self.u.claimRequestedSpaceIfReady(req, NOW)

    def after_load(self, key, delta):
        self.u.adjustWeightAfterLoad(delta, key, NOW)

    def unload_complete(self, w):
        self.u.unloadComplete(w, True, NOW)

    def touch(self, key, t):
        self.c.get(key, t, NOW)

    def evicted(self):
        return [k for k, _ in self.u.evicted]

    def keys(self):
        return [k for k in self.c.keys() if k != po.UnloadBufManager.KEY]

    def effective_capacity(self):
        return self.u.adjusted_capacity()

    def weighted_size(self):
        return self.c.weightedSize


class CDriver:
    def __init__(self, capacity, reserved):
        self.h = ob.CCache(capacity, reserved, NOW)
        self.lib = self.h.lib
        self.u = C.byref(self.h.u)

    def insert(self, key, t):
        return bool(self.lib.orc_ubm_insert_new_entry(self.u, key, 1, t, NOW))

    def grow(self, key, inc):
        self.lib.orc_ubm_adjust_new_entry_space_request(self.u, inc, key, NOW)

    def ready(self, req):
        return bool(self.lib.orc_ubm_cache_space_is_ready(self.u, req))

    def claim(self, req):
        return bool(self.lib.orc_ubm_claim_requested_space_if_ready(self.u, req, NOW))

    def after_load(self, key, delta):
        self.lib.orc_ubm_adjust_weight_after_load(self.u, delta, key, NOW)

    def unload_complete(self, w):
        self.lib.orc_ubm_unload_complete(self.u, w, 1, NOW)

    def touch(self, key, t):
        self.h.get(key, t, NOW)

    def evicted(self):
        return self.h.evicted()

    def keys(self):
        return self.h.keys()

    def effective_capacity(self):
        return self.h.c.capacity - self.lib.orc_ubm_buffer_weight(self.u)

    def weighted_size(self):
        return self.h.weighted_size


def load_model(d, key, size, t, unloads_done=()):
    """One loadLocal + CacheEntry.load + waitForSpaceToLoad + claim. Returns whether the space was
    ready immediately (i.e. the load did not have to wait for an unload to finish)."""
    d.insert(key, t)
    d.grow(key, size - 1)
    immediately = d.ready(size)
    for w in unloads_done:  # unloads that complete while we wait
        d.unload_complete(w)
    assert d.claim(size)
    return immediately


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_basic_eviction_kat(driver):
    """EvictionsModelMeshTest.basicEvictionTest (:36-125): capacity 1 GiB = 131072 units, default
    size 50 MiB = 6400, 6 loading threads ⇒ reserve clamp(6*6400/4, 1310, 13107) = 9600 units
    (MM.java:748-753) ⇒ effective 121472 units = 949 MiB (test comment :30-33)."""
    cap = 131072
    reserve = max(min(6 * 6400 // 4, cap // 10), cap // 100)
    assert reserve == 9600
    d = driver(cap, reserve)
    assert d.effective_capacity() == 121472 and 121472 * 8192 // (1 << 20) == 949
    t0 = NOW - HOUR  # registerModel stamps lastUsed = now - 1h (MM.java:3097-3101); adds 10 ms apart
    for i in range(18):
        assert load_model(d, i, 6400, t0 + 10 * i)
    assert d.evicted() == [] and d.weighted_size() == 18 * 6400 + 9600  # 18 x 50 MiB fit (:51-62)
    # 19th ⇒ 950 MiB > 949 ⇒ evicts the oldest (myModel0); load is NOT held up (:64-77)
    assert load_model(d, 18, 6400, t0 + 180) is True
    assert d.evicted() == [0]
    # 20th evicts myModel1, still loads at once (:85-88)
    assert load_model(d, 19, 6400, t0 + 190) is True
    assert d.evicted() == [0, 1]
    # 21st evicts myModel2 but must WAIT for the first unload to complete (:90-105)
    assert load_model(d, 20, 6400, t0 + 200, unloads_done=[6400]) is False
    assert d.evicted() == [0, 1, 2]
    d.unload_complete(6400)
    d.unload_complete(6400)
    # re-ensureLoaded(myModel0) evicts myModel3 (:107-109)
    load_model(d, 0, 6400, NOW)
    assert d.evicted() == [0, 1, 2, 3]
    d.unload_complete(6400)
    # 160 MiB model: predicted 50 MiB evicts myModel4; sized after load (+110 MiB) evicts 5 and 6;
    # myModel7 stays (:111-123)
    load_model(d, 21, 6400, NOW + 1)
    assert d.evicted() == [0, 1, 2, 3, 4]
    d.unload_complete(6400)
    d.after_load(21, 160 * 128 - 6400)
    assert d.evicted() == [0, 1, 2, 3, 4, 5, 6]
    assert d.keys()[0] == 7


def _dummy(driver):
    # DummyModelMesh: capacity 10*20 MiB = 25600 units, size 2560, 8 loading threads ⇒ reserve
    # clamp(8*2560/4=5120, 256, 2560) = 2560 ⇒ effective 23040 = 9 models (Appendix C.2)
    cap = 25600
    reserve = max(min(8 * 2560 // 4, cap // 10), cap // 100)
    assert reserve == 2560
    d = driver(cap, reserve)
    assert d.effective_capacity() // 2560 == 9
    return d


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_multi_load_with_eviction_standalone(driver):
    """ModelMeshEvictionsTest.testMultiLoadWithEvictionStandalone (:156-187): 12 loads, last 9 survive."""
    d = _dummy(driver)
    for i in range(12):
        load_model(d, i, 2560, NOW - HOUR + i)
        for _ in range(len(d.evicted()) - getattr(d, "_acked", 0)):
            d.unload_complete(2560)
        d._acked = len(d.evicted())
    assert d.evicted() == [0, 1, 2]
    assert sorted(d.keys()) == list(range(3, 12))


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_multi_load_with_big_eviction_standalone(driver):
    """:190-228 — 11 normal then one 4x model ⇒ survivors ids[6..12): five 1x + one 4x."""
    d = _dummy(driver)
    acked = 0
    for i in range(11):
        load_model(d, i, 2560, NOW - HOUR + i)
        while acked < len(d.evicted()):
            d.unload_complete(2560)
            acked += 1
    assert d.evicted() == [0, 1]
    # size override via encKey: predictSize returns the real size (DummyClassifierLoader.java:120-124)
    d.insert(11, NOW - HOUR + 11)
    d.grow(11, 4 * 2560 - 1)
    while acked < len(d.evicted()):
        d.unload_complete(2560)
        acked += 1
    assert d.claim(4 * 2560)
    assert sorted(d.keys()) == [6, 7, 8, 9, 10, 11]


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_multi_load_with_eviction_standalone_reuse(driver):
    """:240-280 — fill 9, touch the first three, add three ⇒ the touched and the new ones survive."""
    d = _dummy(driver)
    for i in range(9):
        load_model(d, i, 2560, NOW - HOUR + i)
    assert d.evicted() == []
    for i in range(3):
        d.touch(i, 0)  # useModel ⇒ runtimeCache.get(id, now)
    acked = 0
    for i in range(9, 12):
        load_model(d, i, 2560, NOW - HOUR + i)
        while acked < len(d.evicted()):
            d.unload_complete(2560)
            acked += 1
    assert d.evicted() == [3, 4, 5]
    assert set(d.keys()) >= {0, 1, 2, 9, 10, 11}


def test_min_space_units_kat():
    """Appendix C.1/C.2: minSpaceUnits (MM.java:765-771)."""
    lib = ob.load()
    assert lib.orc_min_space_units(6400, 6, 131072, 1) == 6553
    assert lib.orc_min_space_units(2560, 8, 25600, 1) == 2560
    assert lib.orc_min_space_units(6400, 8, 8_388_608, 1) == 51200
    assert lib.orc_min_space_units(6400, 8, 8_388_608, 0) == 51200
    assert lib.orc_min_space_units(6400, 1, 131072, 0) == 6400


@pytest.mark.parametrize("driver", [PyDriver, CDriver])
def test_concurrent_eviction_kat(driver):
    """EvictionsModelMeshTest.concurrentEvictionTest (:136-200): 18 x 50 MiB loaded in the 1 GiB cache,
    then 10 registrations at the same instant (unloads take 1 s, so none completes meanwhile): exactly the
    10 oldest are evicted — no eviction cascade — and myModel10..17 plus the 10 new ones stay."""
    cap = 131072
    d = driver(cap, 9600)
    t0 = NOW - HOUR
    for i in range(18):
        assert load_model(d, i, 6400, t0 + 10 * i)
    assert d.evicted() == []
    # the ten placeholders go in together (insertNewEntry, weight 1), then each grows to its
    # predicted size (adjustNewEntrySpaceRequest) before any unload has completed
    for i in range(10):
        d.insert(18 + i, t0 + 1000 + i)
    for i in range(10):
        d.grow(18 + i, 6399)
    assert d.evicted() == list(range(10))
    assert sorted(d.keys()) == list(range(10, 28))


def test_rpm_thresholds_in_integer_arithmetic():
    """MM.java:4958 `(int) (1.1 * minLoad)`, `(int) (1.5 * minLoad)`: the device evaluates them as
    min(x + x / 10, INT_MAX) and min(x + x / 2, INT_MAX) (place_kernel.hpp RpmRule::init).  Equal to the double
    products with Java's saturating narrowing for every 0 <= x < 2^31 — the exhaustive check of all 2^31 values
    takes two minutes; here: both ends, every multiple of ten near powers of two, and 20 million random values."""
    import numpy as np
    imax = 2**31 - 1
    rng = np.random.default_rng(11)
    parts = [np.arange(0, 3_000_000), np.arange(imax - 3_000_000, imax + 1), rng.integers(0, 2**31, 20_000_000)]
    for k in range(7, 31):
        parts.append(np.arange(max(2**k - 5000, 0), min(2**k + 5000, imax + 1)))
        parts.append((np.arange(-300, 300) + (2**k) // 10) * 10)
    x = np.unique(np.clip(np.concatenate(parts), 0, imax)).astype(np.int64)
    d = x.astype(np.float64)
    assert np.array_equal(np.minimum(np.floor(1.1 * d), imax).astype(np.int64), np.minimum(x + x // 10, imax))
    assert np.array_equal(np.minimum(np.floor(1.5 * d), imax).astype(np.int64), np.minimum(x + (x >> 1), imax))
