"""ctypes binding of oracle/liboracle.so — the CPU parity checker.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the modelmesh_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

ORC_POD = np.dtype(
    [("lru_time", "<i8"), ("capacity", "<i8"), ("used", "<i8"), ("start_time", "<i8"), ("version", "<i8"),
     ("count", "<i4"), ("loading_threads", "<i4"), ("loading_in_progress", "<i4"), ("rpm", "<i4"),
     ("shutting_down", "<i4"), ("id_order", "<u4"), ("replica_set", "<i4"), ("pad_", "<i4")])
assert ORC_POD.itemsize == 72
ORC_STATS = np.dtype([("total_capacity", "<i8"), ("total_free", "<i8"), ("global_lru", "<i8"),
                      ("instance_count", "<i4"), ("model_copy_count", "<i4")])
ORC_NODE = np.dtype([("last_used", "<i8"), ("weight", "<i4"), ("key", "<i4")])


class OrcSnapshot(C.Structure):
    _fields_ = [("pods", C.c_void_p), ("n_pods", C.c_int32), ("order", C.c_void_p),
                ("min_space_units", C.c_int64), ("min_churn_age_ms", C.c_int64), ("n_types", C.c_int32),
                ("allowed", C.POINTER(C.c_void_p)), ("prefer", C.POINTER(C.c_void_p)), ("live", C.c_void_p),
                ("replaced_rs", C.c_void_p), ("n_replaced_rs", C.c_int32), ("n_rows", C.c_int32)]


class OrcServeReq(C.Structure):
    _fields_ = [("self", C.c_int32), ("exclude_self", C.c_int32), ("prefer_self", C.c_int32),
                ("n_copies", C.c_int32), ("copy_pod", C.c_void_p), ("copy_loaded", C.c_void_p),
                ("now", C.c_int64), ("assume_completed_ms", C.c_int64), ("local_in_flight", C.c_int32),
                ("pad_", C.c_int32), ("last_invoke_time", C.c_int64)]


class OrcCache(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n", C.c_int32), ("cap_nodes", C.c_int32),
                ("weighted_size", C.c_int64), ("capacity", C.c_int64), ("oldest_time", C.c_int64)]


class OrcUbm(C.Structure):
    _fields_ = [("cache", C.POINTER(OrcCache)), ("reserved", C.c_int32), ("total_unloading", C.c_int32),
                ("total_occupancy", C.c_int64), ("cache_deficit", C.c_int32), ("n_evicted", C.c_int32),
                ("evicted", C.c_int32 * 1024)]


_lib = None
_native = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load_native():
    """The same sources rebuilt ON THIS BOX with -O2 -march=native (oracle/_native/, git-ignored), for the
    cpu_baseline leg only: the shipped liboracle.so is built without -march because it travels to a host CPU
    that may differ.  Returns (lib, flags string); falls back to the shipped build when no compiler is here."""
    global _native
    if _native is None:
        out_dir = os.path.join(HERE, "_native")
        out = os.path.join(out_dir, "liboracle_native.so")
        srcs = [os.path.join(HERE, f) for f in ("mm_oracle.c", "mm_oracle_batch.c")]
        flags = "-O2 -march=native"
        try:
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-march=native", "-std=c11", "-fPIC", "-shared", "-o", out] + srcs +
                                  ["-lpthread"], stderr=subprocess.DEVNULL)
            lib = C.CDLL(out)
        except Exception:
            lib, flags = C.CDLL(LIB), "-O2 (shipped build; no compiler on this box)"
        lib.orc_pool_create.restype = C.c_void_p
        lib.orc_pool_create.argtypes = [C.c_int32, C.c_int32]
        lib.orc_pool_destroy.restype = None
        lib.orc_pool_destroy.argtypes = [C.c_void_p]
        lib.orc_pool_place.restype = C.c_int
        lib.orc_pool_place.argtypes = [C.c_void_p, C.POINTER(OrcSnapshot), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
        _native = (lib, flags)
    return _native


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.orc_min_space_units.restype = C.c_int64
        lib.orc_min_space_units.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int]
        lib.orc_sort_pods.restype = C.c_int
        lib.orc_sort_pods.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        lib.orc_place_batch.restype = C.c_int
        lib.orc_place_batch.argtypes = [C.POINTER(OrcSnapshot), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        lib.orc_serve.restype = C.c_int32
        lib.orc_serve.argtypes = [C.POINTER(OrcServeReq), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.orc_cluster_stats_of.restype = None
        lib.orc_cluster_stats_of.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
        lib.orc_cache_init.argtypes = [C.POINTER(OrcCache), C.c_int64]
        lib.orc_cache_free.argtypes = [C.POINTER(OrcCache)]
        lib.orc_cache_put_if_absent.restype = C.c_int32
        lib.orc_cache_put_if_absent.argtypes = [C.POINTER(OrcCache), C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                                C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        lib.orc_cache_get.restype = C.c_int
        lib.orc_cache_get.argtypes = [C.POINTER(OrcCache), C.c_int32, C.c_int64, C.c_int64]
        lib.orc_cache_update_weight.restype = C.c_int32
        lib.orc_cache_update_weight.argtypes = [C.POINTER(OrcCache), C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                                C.c_void_p, C.c_int32]
        lib.orc_cache_remove.restype = C.c_int
        lib.orc_cache_remove.argtypes = [C.POINTER(OrcCache), C.c_int32]
        lib.orc_cache_oldest_time.restype = C.c_int64
        lib.orc_cache_oldest_time.argtypes = [C.POINTER(OrcCache)]
        lib.orc_cache_find.restype = C.c_int32
        lib.orc_cache_find.argtypes = [C.POINTER(OrcCache), C.c_int32]
        U = C.POINTER(OrcUbm)
        lib.orc_ubm_init.argtypes = [U, C.POINTER(OrcCache), C.c_int32, C.c_int64]
        lib.orc_ubm_buffer_weight.restype = C.c_int32
        lib.orc_ubm_buffer_weight.argtypes = [U]
        lib.orc_ubm_insert_new_entry.restype = C.c_int
        lib.orc_ubm_insert_new_entry.argtypes = [U, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
        lib.orc_ubm_adjust_new_entry_space_request.argtypes = [U, C.c_int32, C.c_int32, C.c_int64]
        lib.orc_ubm_cache_space_is_ready.restype = C.c_int
        lib.orc_ubm_cache_space_is_ready.argtypes = [U, C.c_int32]
        lib.orc_ubm_claim_requested_space_if_ready.restype = C.c_int
        lib.orc_ubm_claim_requested_space_if_ready.argtypes = [U, C.c_int32, C.c_int64]
        lib.orc_ubm_adjust_weight_after_load.argtypes = [U, C.c_int32, C.c_int32, C.c_int64]
        lib.orc_ubm_unload_complete.argtypes = [U, C.c_int32, C.c_int, C.c_int64]
        lib.orc_ubm_remove_entry.restype = C.c_int32
        lib.orc_ubm_remove_entry.argtypes = [U, C.c_int32, C.c_int64]
        lib.orc_ubm_discard_failed_entry.argtypes = [U, C.c_int32, C.c_int64]
        lib.orc_ubm_insert_failed_placeholder_entry.restype = C.c_int
        lib.orc_ubm_insert_failed_placeholder_entry.argtypes = [U, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
        I32, I64, VP = C.c_int32, C.c_int64, C.c_void_p
        lib.orc_go_local.restype = C.c_int
        lib.orc_go_local.argtypes = [VP, VP, I32, I32, C.c_int, C.c_int, C.c_int, I64]
        lib.orc_load_failures_breached.restype = C.c_int
        lib.orc_load_failures_breached.argtypes = [VP, I32, I64, I64]
        lib.orc_load_locations_breached.restype = C.c_int
        lib.orc_load_locations_breached.argtypes = [VP, I32, VP, I32, VP]
        lib.orc_churn_reject.restype = C.c_int
        lib.orc_churn_reject.argtypes = [I64, I64, I64, I64, I64, I64]
        lib.orc_load_local_initial_size.restype = I32
        lib.orc_load_local_initial_size.argtypes = [C.c_int, I32, I32, I32, I32, VP, C.c_int, I64, I64, I64, I64,
                                                    C.POINTER(C.c_int)]
        lib.orc_reload_elsewhere.restype = C.c_int
        lib.orc_reload_elsewhere.argtypes = [C.c_int, I64, I64, I64, VP]
        lib.orc_should_publish.restype = C.c_int
        lib.orc_should_publish.argtypes = [VP, VP, I64, I64, C.c_int, C.c_int, I64]
        lib.orc_proactive_plan.restype = C.c_int32
        lib.orc_proactive_plan.argtypes = [VP, I32, VP, VP, VP, VP, I32, VP, VP, I32, I32, I64, VP, VP, I32, VP]
        lib.orc_scaleup_plan.restype = C.c_int
        lib.orc_scaleup_plan.argtypes = [VP, I32, VP, I32, VP, VP, I32, C.c_int, VP, VP, VP, VP, I32, VP, VP, VP]
        lib.orc_scaledown_plan.restype = None
        lib.orc_scaledown_plan.argtypes = [VP, VP, VP, VP, VP, VP, VP, VP, I32, VP, VP]
        lib.orc_scaleup_plan_conc.restype = C.c_int
        lib.orc_scaleup_plan_conc.argtypes = [VP, I32, VP, I32, VP, VP, I32, C.c_int, VP, VP, VP, VP, VP, I32, VP, VP, VP, VP, VP, VP]
        lib.orc_scaledown_plan_conc.restype = None
        lib.orc_scaledown_plan_conc.argtypes = [VP, VP, VP, VP, VP, VP, VP, VP, VP, I32, VP, C.c_int64, VP]
        lib.orc_migration_plan.restype = None
        lib.orc_migration_plan.argtypes = [VP, VP, VP, I32, I32, I64, I64, VP, VP]
        lib.orc_evict_eval.restype = None
        lib.orc_evict_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                       C.c_void_p]
        _lib = lib
    return _lib


ORC_EVICT_RESULT = np.dtype([("insert_pos", "<i4"), ("n_victims", "<i4"), ("self_evicted", "<i4"), ("pad_", "<i4"),
                             ("weighted_size", "<i8"), ("oldest_time", "<i8")])


def evict_eval(lu, wt, capacity, weight, last_used, now):
    lib = load()
    lu = np.ascontiguousarray(lu, dtype=np.int64)
    wt = np.ascontiguousarray(wt, dtype=np.int32)
    out = np.zeros(1, dtype=ORC_EVICT_RESULT)
    lib.orc_evict_eval(_p(lu) if len(lu) else None, _p(wt) if len(wt) else None, len(lu), int(capacity),
                       int(weight), int(last_used), int(now), _p(out))
    return out[0]


def serve(self_pod, exclude_self, prefer_self, copy_pod, copy_loaded, now, assume_completed_ms, local_in_flight,
          last_invoke_time, live, in_use, last_used):
    lib = load()
    copy_pod = np.ascontiguousarray(copy_pod, dtype=np.int32)
    copy_loaded = np.ascontiguousarray(copy_loaded, dtype=np.int64)
    r = OrcServeReq()
    r.self = int(self_pod)
    r.exclude_self = int(bool(exclude_self))
    r.prefer_self = int(bool(prefer_self))
    r.n_copies = len(copy_pod)
    r.copy_pod = copy_pod.ctypes.data if len(copy_pod) else None
    r.copy_loaded = copy_loaded.ctypes.data if len(copy_loaded) else None
    r.now = int(now)
    r.assume_completed_ms = int(assume_completed_ms)
    r.local_in_flight = int(local_in_flight)
    r.last_invoke_time = int(last_invoke_time)
    ts = C.c_int64(0)
    ch = lib.orc_serve(C.byref(r), _p(live), _p(in_use), _p(last_used), C.byref(ts))
    return ch, ts.value


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def to_orc_pods(pods: np.ndarray) -> np.ndarray:
    """mmp pod rows -> oracle rows. Tombstones are expressed as shutting-down
    (both mean "not in clusterState", MM.java:1462-1464)."""
    o = np.zeros(len(pods), dtype=ORC_POD)
    for f in ("lru_time", "capacity", "used", "version", "count", "loading_threads", "loading_in_progress",
              "rpm", "id_order", "replica_set"):
        o[f] = pods[f]
    o["shutting_down"] = (pods["flags"] & 5) != 0
    return o


def unpack_bitmap(bm: np.ndarray, n_pods: int) -> np.ndarray:
    """uint64 [T][W] -> uint8 [T][P]"""
    by = np.ascontiguousarray(bm).view(np.uint8).reshape(bm.shape[0], -1)
    bits = np.unpackbits(by, axis=1, bitorder="little")
    return np.ascontiguousarray(bits[:, :n_pods])


class OracleFleet:
    """The oracle's view of a modelmesh_amd.solver.Fleet (duck-typed)."""

    def __init__(self, fleet):
        self.lib = load()
        self.fleet = fleet
        P = fleet.n_pods
        self.pods = to_orc_pods(fleet.pods)
        present = self.pods["shutting_down"] == 0
        # clusterState only holds present rows; sort those with the literal comparator
        self.present_idx = np.nonzero(present)[0].astype(np.int32)
        sub = np.ascontiguousarray(self.pods[self.present_idx])
        order = np.zeros(max(len(sub), 1), dtype=np.int32)
        self.order_rc = self.lib.orc_sort_pods(_p(sub), len(sub), fleet.min_space_units,
                                               fleet.min_churn_age_ms, _p(order))
        self.order = self.present_idx[order[: len(sub)]].astype(np.int32)
        # orc_place walks `order` over the FULL pods array; absent rows simply are not in it.
        # It needs n_pods == len(order) for iteration, but indexes pods[] by pod index.
        self.live = np.ascontiguousarray(((fleet.pods["flags"] & 2) != 0).astype(np.uint8))
        T = fleet.n_types
        self._keep = []
        self.allowed_rows = self.prefer_rows = None
        if T > 0:
            al = unpack_bitmap(fleet.allowed, P) if fleet.allowed is not None else np.zeros((T, P), np.uint8)
            pf = unpack_bitmap(fleet.prefer, P) if fleet.prefer is not None else np.zeros((T, P), np.uint8)
            self._keep += [al, pf]
            self.allowed_rows = (C.c_void_p * T)()
            self.prefer_rows = (C.c_void_p * T)()
            for t in range(T):
                ha = fleet.has_allowed is not None and fleet.has_allowed[t]
                hp = fleet.has_prefer is not None and fleet.has_prefer[t]
                self.allowed_rows[t] = al[t].ctypes.data if ha else None
                self.prefer_rows[t] = pf[t].ctypes.data if hp else None
        self.rs = np.ascontiguousarray(fleet.replaced_rs, dtype=np.int32)
        s = OrcSnapshot()
        s.pods = self.pods.ctypes.data
        s.n_pods = len(self.order)
        s.order = self.order.ctypes.data
        s.min_space_units = fleet.min_space_units
        s.min_churn_age_ms = fleet.min_churn_age_ms
        s.n_types = T
        s.allowed = C.cast(self.allowed_rows, C.POINTER(C.c_void_p)) if T > 0 else None
        s.prefer = C.cast(self.prefer_rows, C.POINTER(C.c_void_p)) if T > 0 else None
        s.live = self.live.ctypes.data
        s.replaced_rs = self.rs.ctypes.data if len(self.rs) else None
        s.n_replaced_rs = len(self.rs)
        s.n_rows = P
        self.snap = s

    def place(self, reqs: np.ndarray, extra: np.ndarray, now: int, threads: int = 1, latencies: bool = False):
        from modelmesh_amd._lib import PLACE_OUT  # dtype only
        reqs = np.ascontiguousarray(reqs)
        extra = np.ascontiguousarray(extra if extra is not None and len(extra) else np.zeros(1, np.int32),
                                     dtype=np.int32)
        outs = np.zeros(len(reqs), dtype=PLACE_OUT)
        lat = np.zeros(len(reqs), dtype=np.float64) if latencies else None
        models = np.ascontiguousarray(self.fleet.models)
        ent = np.ascontiguousarray(self.fleet.ent_pod if len(self.fleet.ent_pod) else np.zeros(1, np.int32),
                                   dtype=np.int32)
        self.lib.orc_place_batch(C.byref(self.snap), _p(models), _p(ent), _p(reqs), _p(extra), len(reqs),
                                 int(now), _p(outs), int(threads), _p(lat) if latencies else None)
        return (outs, lat) if latencies else outs

    def lean_pool(self, threads: int):
        """A persistent worker pool over orc_place_lean (the algorithm without the checker's audit machinery), in
        the -march=native build.  Returns a callable place(reqs, extra, now, latencies=False) -> outs [, lat_ns];
        outs['hash'] carries n_remaining (this path computes no audit hash).  Call .close() when done."""
        lib, flags = load_native()
        pool = lib.orc_pool_create(int(threads), int(self.snap.n_pods))
        if not pool:
            raise RuntimeError("orc_pool_create failed")
        fleet, snap = self.fleet, self.snap
        models = np.ascontiguousarray(fleet.models)
        ent = np.ascontiguousarray(fleet.ent_pod if len(fleet.ent_pod) else np.zeros(1, np.int32), dtype=np.int32)

        out_cache = {}

        def place(reqs, extra, now, latencies=False):
            from modelmesh_amd._lib import PLACE_OUT  # dtype only
            reqs = np.ascontiguousarray(reqs)
            extra = np.ascontiguousarray(extra if extra is not None and len(extra) else np.zeros(1, np.int32), dtype=np.int32)
            outs = out_cache.get(len(reqs))  # reused: zero-filling 16 B per decision costs more than deciding it on 256 threads
            if outs is None:
                outs = out_cache[len(reqs)] = np.zeros(len(reqs), dtype=PLACE_OUT)
            lat = np.zeros(len(reqs), dtype=np.float64) if latencies else None
            rc = lib.orc_pool_place(pool, C.byref(snap), _p(models), _p(ent), _p(reqs), _p(extra), len(reqs), int(now),
                                    _p(outs), _p(lat) if latencies else None)
            if rc != 0:
                raise RuntimeError("orc_pool_place failed")
            return (outs, lat) if latencies else outs

        place.close = lambda: lib.orc_pool_destroy(pool)
        place.flags = flags
        place.keep = (models, ent, self)  # the snapshot's pointers live in this object's arrays
        return place

    def stats(self):
        out = np.zeros(1, dtype=ORC_STATS)
        self.lib.orc_cluster_stats_of(_p(self.pods), len(self.pods), self.fleet.min_space_units, _p(out))
        return out[0]


class CCache:
    """Pythonic handle on the C oracle's cache + unload-buffer manager (for the KAT tests)."""

    def __init__(self, capacity, reserved=None, now=0):
        self.lib = load()
        self.c = OrcCache()
        self.lib.orc_cache_init(C.byref(self.c), int(capacity))
        self.u = None
        self.now = now
        if reserved is not None:
            self.u = OrcUbm()
            self.lib.orc_ubm_init(C.byref(self.u), C.byref(self.c), int(reserved), int(now))

    def __del__(self):
        try:
            self.lib.orc_cache_free(C.byref(self.c))
        except Exception:
            pass

    # plain clhm operations
    def put_if_absent(self, key, weight, last_used, now):
        v = np.zeros(4096, np.int32)
        pos = C.c_int32(0)
        n = self.lib.orc_cache_put_if_absent(C.byref(self.c), key, weight, last_used, now, _p(v), len(v), C.byref(pos))
        return None if n < 0 else list(v[:n])

    def get(self, key, last_used, now):
        return bool(self.lib.orc_cache_get(C.byref(self.c), key, last_used, now))

    def update_weight(self, key, w, new_time, now):
        v = np.zeros(4096, np.int32)
        n = self.lib.orc_cache_update_weight(C.byref(self.c), key, w, new_time, now, _p(v), len(v))
        return None if n < 0 else list(v[:n])

    def remove(self, key):
        return bool(self.lib.orc_cache_remove(C.byref(self.c), key))

    def oldest_time(self):
        return self.lib.orc_cache_oldest_time(C.byref(self.c))

    def keys(self):
        nodes = np.ctypeslib.as_array(C.cast(self.c.nodes, C.POINTER(C.c_byte)), shape=(self.c.n * 16,)) if self.c.n else np.zeros(0, np.int8)
        arr = np.frombuffer(nodes.tobytes(), dtype=ORC_NODE)
        return [int(k) for k in arr["key"] if k != -1000000]

    @property
    def weighted_size(self):
        return self.c.weighted_size

    def nodes(self):
        """(last_used, weight, key) arrays in deque order (oldest first)"""
        if not self.c.n:
            return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
        raw = np.ctypeslib.as_array(C.cast(self.c.nodes, C.POINTER(C.c_byte)), shape=(self.c.n * 16,))
        arr = np.frombuffer(raw.tobytes(), dtype=ORC_NODE)
        return arr["last_used"].copy(), arr["weight"].copy(), arr["key"].copy()

    def apply(self, op, key, arg, time, flag, now):
        """One mmp_cache_op on the oracle; returns (result, evicted keys in listener order)."""
        L, c, u = self.lib, C.byref(self.c), (C.byref(self.u) if self.u is not None else None)
        ev0 = self.u.n_evicted if self.u is not None else 0
        plain = None
        if op == 0:
            plain = self.put_if_absent(key, arg, time, now)
            res = 0 if plain is None else 1
        elif op == 1:
            res = int(self.get(key, time, now))
        elif op == 2:
            plain = self.update_weight(key, arg, time, now)
            res = 0 if plain is None else 1
        elif op == 3:
            res = int(self.remove(key))
        elif op == 4:
            res = L.orc_ubm_insert_new_entry(u, key, arg, time, now)
        elif op == 5:
            res = 1 if L.orc_cache_find(c, key) >= 0 else 0
            L.orc_ubm_adjust_new_entry_space_request(u, arg, key, now)
        elif op == 6:
            res = L.orc_ubm_cache_space_is_ready(u, arg)
        elif op == 7:
            res = L.orc_ubm_claim_requested_space_if_ready(u, arg, now)
        elif op == 8:
            res = 1 if (L.orc_cache_find(c, key) >= 0 or arg == 0) else 0
            L.orc_ubm_adjust_weight_after_load(u, arg, key, now)
        elif op == 9:
            L.orc_ubm_unload_complete(u, arg, flag, now)
            res = 1 if flag else 0
        elif op == 10:
            res = L.orc_ubm_remove_entry(u, key, now)
        elif op == 11:
            L.orc_ubm_discard_failed_entry(u, arg, now)
            res = 1
        elif op == 12:
            res = L.orc_ubm_insert_failed_placeholder_entry(u, key, arg, time, now)
        else:
            raise ValueError(op)
        if plain is not None:
            return res, list(plain)
        if self.u is not None:
            assert self.u.n_evicted <= 1024, "oracle eviction log overflow"
            return res, [int(self.u.evicted[i]) for i in range(ev0, self.u.n_evicted)]
        return res, []

    # unload-buffer manager
    def evicted(self):
        return [int(self.u.evicted[i]) for i in range(self.u.n_evicted)]


ORC_PROACTIVE_INFO = np.dtype(
    [("size_estimate", "<i4"), ("free_count", "<i4"), ("total_count", "<i4"), ("n_candidates", "<i4"),
     ("n_selected", "<i4"), ("error", "<i4"), ("space_to_fill", "<i8"), ("cutoff", "<i8")])


# ---- instance partitions and subset stats (TypeConstraintManager), restated with numpy --------------------
def partitions(fleet):
    """ProhibitedTypeSet partitions of the present instances (TypeConstraintManager.java:553-578): an
    instance's set = the constrained types it cannot host.  -> (pts[P] with -1 for rows not in the table,
    [frozenset of prohibited type rows] per partition, interned in instance order)."""
    P = fleet.n_pods
    pts = np.full(P, -1, np.int32)
    sets, index = [], {}
    if not fleet.n_types:
        return pts, sets
    al = unpack_bitmap(fleet.allowed, P)
    present = (fleet.pods["flags"] & 5) == 0
    for p in range(P):
        if not present[p]:
            continue
        sig = frozenset(t for t in range(fleet.n_types) if fleet.has_allowed[t] and not al[t][p])
        if sig not in index:
            index[sig] = len(sets)
            sets.append(sig)
        pts[p] = index[sig]
    return pts, sets


def subset_stats(fleet, mask, global_stats):
    """InstanceSetStatsTracker over the present instances selected by `mask` (InstanceSetStatsTracker.java:53-92);
    its lru is the cluster-wide one (re-accumulated over ALL instances on every event, MM.java:1515-1542)."""
    pods = fleet.pods
    sel = mask & ((pods["flags"] & 5) == 0)
    rem = np.maximum(pods["capacity"] - pods["used"], 0)
    out = np.zeros(1, dtype=ORC_STATS)[0]
    out["total_capacity"] = int(pods["capacity"][sel].sum())
    out["total_free"] = int(rem[sel & (rem >= fleet.min_space_units)].sum())
    out["instance_count"] = int(sel.sum())
    out["model_copy_count"] = int(pods["count"][sel].sum())
    out["global_lru"] = int(global_stats["global_lru"]) if sel.any() else 2**63 - 1
    return out


def partition_stats(fleet):
    """-> (pts, sets, ORC_STATS[n_partitions])"""
    g = OracleFleet(fleet).stats()
    pts, sets = partitions(fleet)
    st = np.zeros(len(sets), dtype=ORC_STATS)
    for k in range(len(sets)):
        st[k] = subset_stats(fleet, pts == k, g)
    return pts, sets, st


def type_set_stats(fleet):
    """typeSetStats(type) for every type row (MM.java:1432-1439): candidateSubsetStats() of a constrained type =
    the sum over the partitions that do not prohibit it (TypeConstraintManager.java:356-377), the cluster's
    stats for an unconstrained one (and for every type when typeConstraints == null)."""
    g = OracleFleet(fleet).stats()
    T = max(fleet.n_types, 1)
    out = np.zeros(T, dtype=ORC_STATS)
    out[:] = g
    if fleet.n_types:
        pts, sets = partitions(fleet)
        for t in range(fleet.n_types):
            if fleet.has_allowed[t]:
                ok = np.array([t not in s for s in sets], bool)
                out[t] = subset_stats(fleet, np.isin(pts, np.nonzero(ok)[0]), g)
    return out


def proactive_plan(fleet, default_units, now, max_out, partition=-1, skip_models=None):
    """a17 on the oracle: returns (models, last_used, info); partition >= 0: for that partition only."""
    lib = load()
    orc = OracleFleet(fleet)
    gstats = np.zeros(1, dtype=ORC_STATS)
    gstats[0] = orc.stats()
    stats, in_subset, prohibited = gstats, None, None
    if partition >= 0:
        pts, sets, pst = partition_stats(fleet)
        stats = np.ascontiguousarray(pst[partition: partition + 1])
        in_subset = np.ascontiguousarray((pts == partition).astype(np.uint8))
        words = np.zeros(max(-(-fleet.n_types // 64), 1), np.uint64)
        for t in sets[partition]:
            words[t >> 6] |= np.uint64(1) << np.uint64(t & 63)
        prohibited = words
    skip = None
    if skip_models is not None and len(skip_models):
        skip = np.zeros(max(fleet.n_models, 1), np.uint8)
        skip[np.asarray(skip_models)] = 1
    om = np.zeros(max(max_out, 1), np.int32)
    ol = np.zeros(max(max_out, 1), np.int64)
    info = np.zeros(1, dtype=ORC_PROACTIVE_INFO)
    models = np.ascontiguousarray(fleet.models)
    n = lib.orc_proactive_plan(_p(orc.pods), len(orc.pods), _p(gstats), _p(stats), _p(in_subset) if in_subset is not None else None,
                               _p(prohibited) if prohibited is not None else None, int(fleet.n_types),
                               _p(skip) if skip is not None else None, _p(models), len(models), int(default_units),
                               int(now), _p(om), _p(ol), int(max_out), _p(info))
    n = min(n, max_out)
    return om[:n].copy(), ol[:n].copy(), info[0]


ORC_SCALEUP_PARAMS = np.dtype(
    [("self_pod", "<i4"), ("iteration_counter", "<i4"), ("second_copy_max_age_iters", "<i4"),
     ("second_copy_min_age_iters", "<i4"), ("scale_up_rpm_threshold", "<i4"), ("our_rpm", "<i4"), ("now", "<i8"),
     ("last_check_time", "<i8"), ("rate_check_interval_ms", "<i8"), ("second_copy_lru_threshold_ms", "<i8"),
     ("assume_completed_ms", "<i8")])
ORC_SCALEDOWN_PARAMS = np.dtype(
    [("self_pod", "<i4"), ("shutting_down", "<i4"), ("now", "<i8"), ("last_check_time", "<i8"),
     ("rate_check_interval_ms", "<i8"), ("adjusted_cache_capacity", "<i8"), ("scale_up_rpm_threshold", "<i4"),
     ("pad_", "<i4")])


def _ent(fleet):
    e = fleet.ent_pod if len(fleet.ent_pod) else np.zeros(1, np.int32)
    t = fleet.ent_time if len(fleet.ent_time) else np.zeros(1, np.int64)
    return np.ascontiguousarray(e, dtype=np.int32), np.ascontiguousarray(t, dtype=np.int64)


def scaleup_plan(fleet, entries, params):
    from modelmesh_amd._lib import SCALEUP_OUT
    lib = load()
    orc = OracleFleet(fleet)
    stats = np.zeros(1, dtype=ORC_STATS)
    stats[0] = orc.stats()
    entries = np.ascontiguousarray(entries)
    params = np.ascontiguousarray(params).reshape(1)
    outs = np.zeros(max(len(entries), 1), dtype=SCALEUP_OUT)
    ov = np.zeros(max(fleet.n_pods, 1), np.uint8)
    ep, et = _ent(fleet)
    models = np.ascontiguousarray(fleet.models)
    tstats = np.ascontiguousarray(type_set_stats(fleet))
    sk = lib.orc_scaleup_plan(_p(orc.pods), fleet.n_pods, _p(orc.order), len(orc.order), _p(stats), _p(tstats), len(tstats),
                              1 if fleet.n_types else 0, _p(models), _p(ep),
                              _p(et), _p(entries) if len(entries) else None, len(entries), _p(params), _p(outs), _p(ov))
    return outs[: len(entries)], ov[: fleet.n_pods], sk


def scaleup_plan_conc(fleet, entries, conc, params, conc_params):
    """The rate task with limitModelConcurrency == true: (outs, conc_outs, overloaded, returned_early, result)."""
    from modelmesh_amd._lib import CONC_OUT, CONC_RESULT, SCALEUP_OUT
    lib = load()
    orc = OracleFleet(fleet)
    stats = np.zeros(1, dtype=ORC_STATS)
    stats[0] = orc.stats()
    entries = np.ascontiguousarray(entries)
    conc = np.ascontiguousarray(conc)
    params = np.ascontiguousarray(params).reshape(1)
    cparams = np.ascontiguousarray(conc_params).reshape(1)
    n = len(entries)
    outs = np.zeros(max(n, 1), dtype=SCALEUP_OUT)
    couts = np.zeros(max(n, 1), dtype=CONC_OUT)
    res = np.zeros(1, dtype=CONC_RESULT)
    ov = np.zeros(max(fleet.n_pods, 1), np.uint8)
    ep, et = _ent(fleet)
    models = np.ascontiguousarray(fleet.models)
    tstats = np.ascontiguousarray(type_set_stats(fleet))
    sk = lib.orc_scaleup_plan_conc(_p(orc.pods), fleet.n_pods, _p(orc.order), len(orc.order), _p(stats), _p(tstats), len(tstats),
                                   1 if fleet.n_types else 0, _p(models), _p(ep), _p(et), _p(entries) if n else None,
                                   _p(conc) if n else None, n, _p(params), _p(cparams), _p(outs), _p(couts), _p(ov), _p(res))
    return outs[:n], couts[:n], ov[: fleet.n_pods], sk, res[0]


def scaledown_plan_conc(fleet, entries, conc, params, dyn_const):
    lib = load()
    orc = OracleFleet(fleet)
    stats = instance_set_stats(fleet, int(np.asarray(params).reshape(-1)[0]["self_pod"]), orc)
    pos_of = np.full(max(fleet.n_pods, 1), 2**31 - 1, np.int32)
    pos_of[orc.order] = np.arange(len(orc.order), dtype=np.int32)
    in_table = np.ascontiguousarray(((fleet.pods["flags"] & 4) == 0).astype(np.uint8))
    opods = orc.pods.copy()
    opods["shutting_down"] = (fleet.pods["flags"] & 1) != 0
    entries = np.ascontiguousarray(entries)
    conc = np.ascontiguousarray(conc)
    params = np.ascontiguousarray(params).reshape(1)
    n = len(entries)
    rem = np.zeros(max(n, 1), np.uint8)
    ep, et = _ent(fleet)
    models = np.ascontiguousarray(fleet.models)
    lib.orc_scaledown_plan_conc(_p(opods), _p(pos_of), _p(in_table), _p(stats), _p(models), _p(ep), _p(et),
                                _p(entries) if n else None, _p(conc) if n else None, n, _p(params), int(dyn_const), _p(rem))
    return rem[:n]


def instance_set_stats(fleet, self_pod, orc=None):
    """instanceSetStats() (MM.java:1446): the cluster's stats, or with type constraints the stats of this instance's
    partition (EMPTY_STATS when it is not in the table)."""
    orc = orc or OracleFleet(fleet)
    stats = np.zeros(1, dtype=ORC_STATS)
    stats[0] = orc.stats()
    if fleet.n_types:
        sp = int(self_pod)
        pts, sets, pst = partition_stats(fleet)
        k = int(pts[sp]) if 0 <= sp < fleet.n_pods else -1
        if k >= 0:
            stats[0] = pst[k]
        else:
            stats[0] = np.zeros(1, dtype=ORC_STATS)[0]
            stats[0]["global_lru"] = 2**63 - 1
    return stats


def scaledown_plan(fleet, entries, params):
    lib = load()
    orc = OracleFleet(fleet)
    stats = instance_set_stats(fleet, int(np.asarray(params).reshape(-1)[0]["self_pod"]), orc)
    pos_of = np.full(max(fleet.n_pods, 1), 2**31 - 1, np.int32)
    pos_of[orc.order] = np.arange(len(orc.order), dtype=np.int32)
    in_table = np.ascontiguousarray(((fleet.pods["flags"] & 4) == 0).astype(np.uint8))
    opods = orc.pods.copy()
    opods["shutting_down"] = (fleet.pods["flags"] & 1) != 0
    entries = np.ascontiguousarray(entries)
    params = np.ascontiguousarray(params).reshape(1)
    rem = np.zeros(max(len(entries), 1), np.uint8)
    ep, et = _ent(fleet)
    models = np.ascontiguousarray(fleet.models)
    lib.orc_scaledown_plan(_p(opods), _p(pos_of), _p(in_table), _p(stats), _p(models), _p(ep), _p(et),
                           _p(entries) if len(entries) else None, len(entries), _p(params), _p(rem))
    return rem[: len(entries)]


def migration_plan(fleet, entries, self_pod, now, cutoff_age_ms=3_600_000):
    lib = load()
    entries = np.ascontiguousarray(entries)
    act = np.zeros(max(len(entries), 1), np.uint8)
    wait = np.zeros(max(len(entries), 1), np.uint8)
    ep, _ = _ent(fleet)
    models = np.ascontiguousarray(fleet.models)
    lib.orc_migration_plan(_p(models), _p(ep), _p(entries) if len(entries) else None, len(entries), int(self_pod),
                           int(now), int(cutoff_age_ms), _p(act), _p(wait))
    return act[: len(entries)], wait[: len(entries)]
