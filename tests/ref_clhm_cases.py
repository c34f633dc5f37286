"""Operation streams for the local cache (SURVEY.md §8 rows a12 + a13) that the reference-text harness
(oracle/ref_harness/clhm_harness.cc), the C oracle (oracle/mm_evict_oracle.c) and the device (cache_replay_kernel) all replay.
Seeded; the committed vectors (tests/golden/ref_clhm.npz) carry the streams themselves, so the tests do not regenerate them.

Domain of a stream = what ModelMesh guarantees about its own cache (MM.java:748-753, :1766): an unload reserve in
(0, capacity / 10], entry weights >= 1 at all times, and the pinned unload-buffer entry is never evicted.  make_clhm_vectors.py
cuts a cache's stream where the reference itself would evict that entry (a failed unload that shrinks the capacity below the
buffer)."""
import numpy as np

NOW = 1_760_000_000_000
CACHE_OP = np.dtype([("cache", "<i4"), ("op", "<i4"), ("key", "<i4"), ("arg", "<i4"), ("time", "<i8"), ("flag", "<i4"),
                     ("reserved", "<i4")])
UNLOADBUF_KEY = -1000000


def random_stream(seed, n_caches, n_ops, keys_per_cache=40, fail_unload=0.15):
    """Few distinct timestamps (ties are the rule: LinkedDeque.java:267), reads with older timestamps (touch = max, clhm :1358),
    weight growth after load, claims, failed unloads."""
    rng = np.random.default_rng(seed)
    caps = rng.choice([5_000, 25_600, 131_072, 400_000], n_caches).astype(np.int64)
    managed = rng.random(n_caches) < 0.6
    reserved = np.where(managed, np.minimum(rng.choice([64, 256, 2_560, 9_600], n_caches), caps // 10), -1).astype(np.int32)
    ops = np.zeros(n_ops, dtype=CACHE_OP)
    lb = {}  # a lower bound of every key's weight (negative deltas must leave it >= 1)
    for i in range(n_ops):
        c = int(rng.integers(0, n_caches))
        key = int(rng.integers(0, keys_per_cache)) + 1000 * c
        t = int(rng.choice([0, NOW - 500, NOW - 2_000, NOW - 2_000, NOW - 7_200_000, NOW + 5, NOW - int(rng.integers(0, 5000))]))
        w = int(rng.choice([1, 640, 2_560, 6_400, 30_000]))
        flag = 0
        if managed[c]:
            op = int(rng.choice([4, 4, 4, 5, 5, 6, 7, 8, 8, 9, 10, 11, 12, 1, 1]))
        else:
            op = int(rng.choice([0, 0, 0, 0, 1, 1, 2, 2, 3]))
        arg = w
        if op in (4, 12):
            arg = 1 if rng.random() < 0.7 else w
        elif op == 2:
            t = int(rng.choice([-1, -1, 0, NOW - 100]))
        elif op == 5:
            arg, flag = int(rng.choice([639, 2_559, 6_399])), int(rng.random() < 0.3)
        elif op in (6, 7):
            arg = int(rng.choice([1, 640, 6_400, 100_000]))
        elif op == 8:
            arg = int(rng.choice([-600, -1, 0, 1, 700, 20_000]))
            if arg < 0 and lb.get(key, 1) + arg < 1:
                arg = 0
        elif op == 9:
            arg, flag = int(rng.choice([640, 2_560])), int(rng.random() >= fail_unload)
        elif op == 11:
            arg = int(rng.choice([1, 640]))
        if op in (0, 4, 12, 2):
            lb[key] = min(lb.get(key, arg), arg)
        elif op in (5, 8):
            lb[key] = lb.get(key, 1) + arg
        ops[i] = (c, op, key, arg, t, flag, 0)
    return caps, reserved, ops


def kat_basic_eviction():
    """EvictionsModelMeshTest.basicEvictionTest (:36-125): capacity 131072 units, reserve 9600; models of
    6400 units registered 10 ms apart (lastUsed = now - 1 h + 10 ms * m); the 19th insert evicts myModel0, then 1, 2; re-adding
    myModel0 evicts 3; a model sized 160 MiB after its load evicts 4, 5, 6."""
    rows = []
    t0 = NOW - 3_600_000

    def load(m, t):  # ensureLoaded: placeholder (INSERTION_WEIGHT 1), predicted size, claim before the load starts, unloads complete
        rows.append((0, 4, m, 1, t, 0, 0))
        rows.append((0, 5, m, 6399, 0, 0, 0))
        rows.append((0, 7, 0, 6400, 0, 0, 0))
    for m in range(21):
        load(m, t0 + 10 * m)
        if m >= 18:
            rows.append((0, 9, 0, 6400, 0, 1, 0))  # the evicted model's unload completes
    load(0, NOW)                                  # myModel0 again: evicts 3
    rows.append((0, 9, 0, 6400, 0, 1, 0))
    load(21, NOW)                                 # predicted 50 MiB ...
    rows.append((0, 9, 0, 6400, 0, 1, 0))
    rows.append((0, 8, 21, 20480 - 6400, 0, 0, 0))  # ... sized 160 MiB after the load
    ops = np.array(rows, dtype=CACHE_OP)
    return np.array([131072], np.int64), np.array([9600], np.int32), ops


def kat_concurrent_eviction():
    """EvictionsModelMeshTest.concurrentEvictionTest (:136-200): 18 loaded, ten placeholders inserted together, then grown."""
    rows = []
    t0 = NOW - 3_600_000
    for m in range(18):
        rows += [(0, 4, m, 1, t0 + 10 * m, 0, 0), (0, 5, m, 6399, 0, 0, 0), (0, 7, 0, 6400, 0, 0, 0)]
    for m in range(18, 28):
        rows.append((0, 4, m, 1, t0 + 1000 + m, 0, 0))
    for m in range(18, 28):
        rows.append((0, 5, m, 6399, 0, 0, 0))
    return np.array([131072], np.int64), np.array([9600], np.int32), np.array(rows, dtype=CACHE_OP)


def kat_standalone_lru():
    """ModelMeshEvictionsTest (:156-280, Appendix C.2): capacity 25600, size 2560, reserve 2560; 12 sequential loads leave the
    last 9; then reads of three re-order the LRU before three more loads."""
    rows = []
    t0 = NOW - 60_000
    for m in range(12):
        rows += [(0, 4, m, 1, t0 + 10 * m, 0, 0), (0, 5, m, 2559, 0, 0, 0), (0, 9, 0, 2560, 0, 1, 0)]
    for m in (3, 4, 5):
        rows.append((0, 1, m, 0, 0, 0, 0))  # invoke: get(key) stamps now
    for m in range(12, 15):
        rows += [(0, 4, m, 1, NOW, 0, 0), (0, 5, m, 2559, 0, 0, 0), (0, 9, 0, 2560, 0, 1, 0)]
    return np.array([25600], np.int64), np.array([2560], np.int32), np.array(rows, dtype=CACHE_OP)


def head_reads():
    """A cache in use: every read is of the least recently used entry (a round-robin client), which moves the deque's head to
    its tail each time (LinkedDeque.reposition :243-255) — 30 entries, 900 reads, no insert in between; one managed, one not."""
    rows = []
    for c, op in ((0, 4), (1, 0)):
        for m in range(30):
            rows.append((c, op, 1000 * c + m, 100, NOW - 10_000 + m, 0, 0))
    for r in range(30):
        for c in (0, 1):
            for m in range(30):
                rows.append((c, 1, 1000 * c + m, 0, 0 if r % 2 else NOW + 1 + 30 * r + m, 0, 0))
    return np.array([100_000, 100_000], np.int64), np.array([2_560, -1], np.int32), np.array(rows, dtype=CACHE_OP)


def cases():
    """(name, capacities, reserves, ops)"""
    out = [("kat_basic_eviction",) + kat_basic_eviction(), ("kat_concurrent_eviction",) + kat_concurrent_eviction(),
           ("kat_standalone_lru",) + kat_standalone_lru(), ("head_reads",) + head_reads()]
    rng = np.random.default_rng(0xC1A)
    for k in range(36):
        n_caches = int(rng.choice([1, 3, 8, 24]))
        n_ops = int(rng.choice([60, 300, 1000, 1500]))
        out.append((f"stream{k:02d}",) + random_stream(9000 + k, n_caches, n_ops, keys_per_cache=int(rng.choice([12, 40, 90])),
                                                       fail_unload=float(rng.choice([0.0, 0.15, 0.4]))))
    return out


def initial_state(caps, reserved):
    """The device / oracle state the streams start from: empty caches; a managed one holds the manager's pinned entry
    (newInternalCacheEntry, MM.java:1617-1622: lastUsed = Long.MAX_VALUE, weight = the reserve)."""
    seg = [0]
    lus, wts, keys = [], [], []
    for r in reserved:
        if r >= 0:
            lus.append(np.iinfo(np.int64).max), wts.append(int(r)), keys.append(UNLOADBUF_KEY)
        seg.append(len(lus))
    return (np.array(seg, np.int32), np.array(lus, np.int64), np.array(wts, np.int32), np.array(keys, np.int32))
