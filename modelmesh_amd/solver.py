"""Thin Python veneer over the libmmplace C ABI (used by tests and bench.py).

Names follow the reference's domain: an *instance table* of pods
(InstanceRecord), a *registry* of models (ModelRecord), load-target decisions
(CacheMissForwardingLB.getNext, MM.java:4776), serve-target decisions
(ForwardingLB.getNext, MM.java:4315) and cache eviction (clhm).
Everything is computed by the HIP kernels behind the C ABI; nothing here
implements placement logic.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from ._lib import (EVICT_OUT, EVICT_REQ, MODEL_ROW, PLACE_OUT, PLACE_REQ, POD_ROW, SERVE_COUNTER, SERVE_OUT,
                   SERVE_REQ, STATS, MmpConfig, ptr)


class MmpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmmplace error {code}: {msg}")
        self.code = code


@dataclass
class Fleet:
    """One consistent view of clusterState + registry + typeConstraints."""
    pods: np.ndarray                      # POD_ROW[P]
    models: np.ndarray                    # MODEL_ROW[M]
    ent_pod: np.ndarray                   # int32: loaded ids then failed ids per model
    ent_time: np.ndarray                  # int64: load start / failure time per entry
    min_space_units: int
    min_churn_age_ms: int
    now: int
    n_types: int = 0                      # 0 == typeConstraints is null
    allowed: Optional[np.ndarray] = None  # uint64 [T][W] bit p = pod p allowed
    prefer: Optional[np.ndarray] = None
    has_allowed: Optional[np.ndarray] = None  # uint8 [T]
    has_prefer: Optional[np.ndarray] = None
    replaced_rs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    @property
    def n_pods(self) -> int:
        return int(self.pods.shape[0])

    @property
    def n_models(self) -> int:
        return int(self.models.shape[0])


def bitmap_from_bool(mask: np.ndarray) -> np.ndarray:
    """bool [T][P] -> uint64 [T][ceil(P/64)], bit p%64 of word p//64."""
    mask = np.atleast_2d(mask).astype(bool)
    t, p = mask.shape
    w = (p + 63) // 64
    padded = np.zeros((t, w * 64), dtype=np.uint8)
    padded[:, :p] = mask
    by = np.packbits(padded.reshape(t, w, 8, 8), axis=-1, bitorder="little").reshape(t, w, 8)
    return np.ascontiguousarray(by).view("<u8").reshape(t, w)


class Solver:
    def __init__(self, min_space_units: int, min_churn_age_ms: int, device: int = 0):
        self.lib = _lib.load()
        cfg = MmpConfig(device, 0, int(min_space_units), int(min_churn_age_ms))
        h = C.c_void_p()
        rc = self.lib.mmp_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise MmpError(rc, (self.lib.mmp_last_error(None) or b"").decode())
        self.h = h
        self.n_pods = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.mmp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int):
        if rc != 0:
            raise MmpError(rc, (self.lib.mmp_last_error(self.h) or b"").decode())

    # ---- snapshot ------------------------------------------------------
    def load_pods(self, pods: np.ndarray):
        pods = np.ascontiguousarray(pods, dtype=POD_ROW)
        self._ck(self.lib.mmp_pods_load(self.h, ptr(pods), len(pods)))
        self.n_pods = len(pods)
        self._live = (pods["flags"] & 2) != 0  # host mirror for serve_counters (what litelinks' instance list would hold)

    def upsert_pods(self, idx: np.ndarray, rows: np.ndarray):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        rows = np.ascontiguousarray(rows, dtype=POD_ROW)
        self._ck(self.lib.mmp_pods_upsert(self.h, ptr(idx), ptr(rows), len(idx)))
        self.n_pods = max(self.n_pods, int(idx.max()) + 1 if len(idx) else 0)
        if len(idx):
            live = getattr(self, "_live", np.zeros(0, bool))
            if self.n_pods > len(live):
                live = np.concatenate([live, np.zeros(self.n_pods - len(live), bool)])
            live[idx] = (rows["flags"] & 2) != 0
            self._live = live

    def remove_pods(self, idx: np.ndarray):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._ck(self.lib.mmp_pods_remove(self.h, ptr(idx), len(idx)))
        live = getattr(self, "_live", None)
        if live is not None and len(idx):
            live[idx[(idx >= 0) & (idx < len(live))]] = False  # a removed row leaves litelinks' list too

    def load_types(self, n_types, allowed=None, prefer=None, has_allowed=None, has_prefer=None):
        def u64(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.uint64)

        def u8(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
        allowed, prefer, has_allowed, has_prefer = u64(allowed), u64(prefer), u8(has_allowed), u8(has_prefer)
        self._ck(self.lib.mmp_types_load(self.h, int(n_types), ptr(allowed), ptr(prefer),
                                         ptr(has_allowed), ptr(has_prefer)))

    def types_from_labels(self, required, preferred, pod_labels):
        """a18: returns (allowed[T+1][W], prefer[T+1][W], has_allowed[T+1], has_prefer[T+1]) and installs them."""
        required = np.ascontiguousarray(required, dtype=np.uint64)
        preferred = np.ascontiguousarray(preferred, dtype=np.uint64)
        pod_labels = np.ascontiguousarray(pod_labels, dtype=np.uint64)
        T, W = len(required), (self.n_pods + 63) // 64
        al = np.zeros((T + 1, max(W, 1)), np.uint64)
        pf = np.zeros((T + 1, max(W, 1)), np.uint64)
        ha = np.zeros(T + 1, np.uint8)
        hp = np.zeros(T + 1, np.uint8)
        self._ck(self.lib.mmp_types_from_labels(self.h, T, ptr(required) if T else None, ptr(preferred) if T else None,
                                                ptr(pod_labels) if len(pod_labels) else None, ptr(al), ptr(pf),
                                                ptr(ha), ptr(hp)))
        return al[:, :W], pf[:, :W], ha, hp

    def load_replaced_rs(self, rs):
        rs = np.ascontiguousarray(rs, dtype=np.int32)
        self._ck(self.lib.mmp_replaced_rs_load(self.h, ptr(rs) if len(rs) else None, len(rs)))

    # ---- KV wire format ------------------------------------------------------
    @staticmethod
    def _pack(strings):
        bs = [x if isinstance(x, bytes) else x.encode() for x in strings]
        off = np.zeros(len(bs) + 1, np.int64)
        np.cumsum([len(b) for b in bs], out=off[1:])
        return b"".join(bs), off

    def load_pod_ids(self, ids):
        """ids: instance id strings in pod-index order; returns (id_order, replica_set)."""
        blob, off = self._pack(ids)
        off32 = off.astype(np.int32)
        n = len(ids)
        io = np.zeros(max(n, 1), np.uint32)
        rs = np.zeros(max(n, 1), np.int32)
        self._ck(self.lib.mmp_pod_ids_load(self.h, blob, ptr(off32), n, ptr(io), ptr(rs)))
        self.n_pods = n
        return io[:n], rs[:n]

    def ingest_pods_json(self, values, pod_idx, live=None):
        """values: InstanceRecord JSON (bytes/str) per record; returns (status, start_time)."""
        blob, off = self._pack(values)
        n = len(values)
        pod_idx = np.ascontiguousarray(pod_idx, dtype=np.int32)
        live = None if live is None else np.ascontiguousarray(live, dtype=np.uint8)
        st = np.zeros(max(n, 1), np.int64)
        status = np.zeros(max(n, 1), np.int32)
        self._ck(self.lib.mmp_pods_ingest_json(self.h, blob, ptr(off), n, ptr(pod_idx), ptr(live), ptr(st), ptr(status)))
        # the host mirror serve_counters reads (what litelinks' instance list would hold): the ingested rows' live flags
        mirror = getattr(self, "_live", None)
        if mirror is not None and n:
            top = int(pod_idx.max()) + 1
            if top > len(mirror):
                mirror = np.concatenate([mirror, np.zeros(top - len(mirror), bool)])
            ok = status[:n] == 0
            mirror[pod_idx[ok]] = True if live is None else live[ok].astype(bool)
            self._live = mirror
        return status[:n], st[:n]

    def load_type_names(self, names, unknown_type):
        blob, off = self._pack(names)
        off32 = off.astype(np.int32)
        self._ck(self.lib.mmp_type_names_load(self.h, blob, ptr(off32), len(names), int(unknown_type)))

    def ingest_models_json(self, values):
        """values: ModelRecord JSON per model; returns (status, last_unload_time)."""
        blob, off = self._pack(values)
        n = len(values)
        lul = np.zeros(max(n, 1), np.int64)
        status = np.zeros(max(n, 1), np.int32)
        self._ck(self.lib.mmp_models_ingest_json(self.h, blob, ptr(off), n, ptr(lul), ptr(status)))
        self.n_models = n
        self._models = self._ent_pod = None  # (the registry now lives on the device only: serve_counters needs load_models / upsert_models)
        return status[:n], lul[:n]

    def get_pods(self) -> np.ndarray:
        n = C.c_int32(0)
        self._ck(self.lib.mmp_pods_get(self.h, None, 0, C.byref(n)))
        rows = np.zeros(max(n.value, 1), dtype=POD_ROW)
        self._ck(self.lib.mmp_pods_get(self.h, ptr(rows), n.value, C.byref(n)))
        return rows[: n.value]

    def get_models(self):
        nm, ne = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.mmp_models_get(self.h, None, 0, None, None, 0, C.byref(nm), C.byref(ne)))
        rows = np.zeros(max(nm.value, 1), dtype=MODEL_ROW)
        ep = np.zeros(max(ne.value, 1), np.int32)
        et = np.zeros(max(ne.value, 1), np.int64)
        self._ck(self.lib.mmp_models_get(self.h, ptr(rows), nm.value, ptr(ep), ptr(et), ne.value, C.byref(nm), C.byref(ne)))
        return rows[: nm.value], ep[: ne.value], et[: ne.value]

    # UpgradeTracker (row a19)
    def upgrade_instance_added(self, labels_key, replica_set, start_time, now):
        self._ck(self.lib.mmp_upgrade_instance_added(self.h, int(labels_key), int(replica_set), int(start_time), int(now)))

    def upgrade_instance_removed(self, labels_key, replica_set, now):
        self._ck(self.lib.mmp_upgrade_instance_removed(self.h, int(labels_key), int(replica_set), int(now)))

    def upgrade_housekeeping(self, now):
        self._ck(self.lib.mmp_upgrade_housekeeping(self.h, int(now)))

    def upgrade_replaced(self) -> dict:
        rs = np.zeros(256, np.int32)
        ex = np.zeros(256, np.int64)
        n = C.c_int32(0)
        self._ck(self.lib.mmp_upgrade_replaced(self.h, ptr(rs), ptr(ex), 256, C.byref(n)))
        return {int(rs[i]): int(ex[i]) for i in range(min(n.value, 256))}

    def load_models(self, models, ent_pod, ent_time):
        models = np.ascontiguousarray(models, dtype=MODEL_ROW)
        ent_pod = np.ascontiguousarray(ent_pod, dtype=np.int32)
        ent_time = np.ascontiguousarray(ent_time, dtype=np.int64)
        self._ck(self.lib.mmp_models_load(self.h, ptr(models), len(models),
                                          ptr(ent_pod) if len(ent_pod) else None,
                                          ptr(ent_time) if len(ent_time) else None, len(ent_pod)))
        self.n_models = len(models)
        self._models, self._ent_pod = models.copy(), ent_pod.copy()  # host mirror for serve_counters

    def upsert_models(self, idx, rows, ent_pod, ent_time):
        """Registry events: rows[i] (its ent_off indexing ent_pod / ent_time of this call) replaces model idx[i]."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        rows = np.ascontiguousarray(rows, dtype=MODEL_ROW)
        ent_pod = np.ascontiguousarray(ent_pod, dtype=np.int32)
        ent_time = np.ascontiguousarray(ent_time, dtype=np.int64)
        self._ck(self.lib.mmp_models_upsert(self.h, ptr(idx), ptr(rows), len(idx), ptr(ent_pod) if len(ent_pod) else None,
                                            ptr(ent_time) if len(ent_time) else None, len(ent_pod)))
        if len(idx):
            self.n_models = max(getattr(self, "n_models", 0), int(idx.max()) + 1)
            # the host mirror serve_counters reads: the changed records' entries are appended, their rows rewritten
            if getattr(self, "_models", None) is not None:
                if self.n_models > len(self._models):  # new records: the mirror grows with the registry
                    self._models = np.concatenate([self._models, np.zeros(self.n_models - len(self._models), MODEL_ROW)])
                shifted = rows.copy()
                shifted["ent_off"] += len(self._ent_pod)
                self._ent_pod = np.concatenate([self._ent_pod, ent_pod])
                self._models[idx] = shifted
                live_entries = int(self._models["n_loaded"].sum())
                if len(self._ent_pod) > 2 * live_entries + 1024:  # the replaced records' entries: compact, as the device pool does
                    k = self._models["n_loaded"].astype(np.int64)
                    off = np.zeros(len(k) + 1, np.int64)
                    np.cumsum(k, out=off[1:])
                    seg = np.repeat(np.arange(len(k)), k)
                    self._ent_pod = self._ent_pod[self._models["ent_off"][seg] + (np.arange(int(off[-1])) - off[seg])]
                    self._models["ent_off"] = off[:-1]

    def commit(self):
        self._ck(self.lib.mmp_snapshot_commit(self.h))

    def load_fleet(self, f: Fleet, commit: bool = True):
        """Stage a whole fleet; commit=False leaves the commit to the caller (pod-axis shard mode)."""
        self.load_pods(f.pods)
        self.load_types(f.n_types, f.allowed, f.prefer, f.has_allowed, f.has_prefer)
        self.load_replaced_rs(f.replaced_rs)
        self.load_models(f.models, f.ent_pod, f.ent_time)
        if commit:
            self.commit()

    def order(self) -> np.ndarray:
        out = np.zeros(max(self.n_pods, 1), dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self.lib.mmp_get_order(self.h, ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def delta_commits(self) -> int:
        """Commits of this context that re-ranked by insertion (a few changed rows) instead of sorting."""
        n = C.c_int64(0)
        self._ck(self.lib.mmp_delta_commits(self.h, C.byref(n)))
        return n.value

    def shortlists(self) -> np.ndarray:
        """The published snapshot's per-type shortlists (mmp_shortlists): rows [2 * type + bit] of (valid, lo, hi, n_candidates)."""
        out = np.zeros(24, dtype=np.dtype([("valid", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("n_candidates", "<i4")]))
        n = C.c_int32(0)
        self._ck(self.lib.mmp_shortlists(self.h, ptr(out), len(out), C.byref(n)))
        return out[: n.value].copy()

    def long_shortlists(self, cap_types: int = 256) -> np.ndarray:
        """The published snapshot's recorded walks of the long shortlists (mmp_long_shortlists): rows [2 * type + bit]."""
        out = np.zeros(2 * cap_types, dtype=np.dtype([("valid", "<i4"), ("lo", "<i4"), ("hi", "<i4"), ("n_candidates", "<i4")]))
        n = C.c_int32(0)
        self._ck(self.lib.mmp_long_shortlists(self.h, ptr(out), len(out), C.byref(n)))
        return out[: min(n.value, len(out))].copy()

    def split_batches(self):
        """(batches this context decided as two launches — shortlist check + tail —, whether that is switched off): mmp_split_batches."""
        n, off = C.c_int64(0), C.c_int32(0)
        self._ck(self.lib.mmp_split_batches(self.h, C.byref(n), C.byref(off)))
        return int(n.value), bool(off.value)

    def stats(self) -> np.ndarray:
        out = np.zeros(1, dtype=STATS)
        self._ck(self.lib.mmp_cluster_stats(self.h, ptr(out)))
        return out[0]

    def type_stats(self, type_row: int) -> np.ndarray:
        """typeSetStats(type): the stats of the instances a type may be placed on (cluster-wide if unconstrained)."""
        out = np.zeros(1, dtype=STATS)
        self._ck(self.lib.mmp_type_stats(self.h, int(type_row), ptr(out)))
        return out[0]

    def partitions(self):
        """-> (pod -> partition array, [(stats, prohibited type rows as a python int bitset)] per partition)"""
        n = np.zeros(1, np.int32)
        self._ck(self.lib.mmp_partition_count(self.h, ptr(n)))
        npods = np.zeros(1, np.int32)
        self._ck(self.lib.mmp_pod_partitions(self.h, None, 0, ptr(npods)))
        pts = np.zeros(max(int(npods[0]), 1), np.int32)
        self._ck(self.lib.mmp_pod_partitions(self.h, ptr(pts), int(npods[0]), ptr(npods)))
        out = []
        for k in range(int(n[0])):
            st = np.zeros(1, dtype=STATS)
            words = np.zeros(16, np.uint64)
            self._ck(self.lib.mmp_partition_stats(self.h, k, ptr(st), ptr(words), len(words)))
            out.append((st[0], sum(int(w) << (64 * i) for i, w in enumerate(words))))
        return pts[: int(npods[0])], out

    # ---- decisions -----------------------------------------------------
    def place(self, reqs: np.ndarray, extra_pool: Optional[np.ndarray], now: int) -> np.ndarray:
        reqs = np.ascontiguousarray(reqs, dtype=PLACE_REQ)
        extra = np.zeros(0, np.int32) if extra_pool is None else np.ascontiguousarray(extra_pool, dtype=np.int32)
        outs = np.zeros(len(reqs), dtype=PLACE_OUT)
        self._ck(self.lib.mmp_place_batch(self.h, ptr(reqs), len(reqs), ptr(extra) if len(extra) else None,
                                          len(extra), int(now), ptr(outs)))
        return outs

    def place_dev(self, d_reqs: int, n: int, d_extra: int, now: int, d_outs: int, stream: int = 0):
        """Launch on raw device pointers (torch ``data_ptr()``), no sync."""
        self._ck(self.lib.mmp_place_batch_dev(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None),
                                              int(now), C.c_void_p(d_outs), C.c_void_p(stream or None)))

    def place_dev2(self, d_reqs: int, n: int, d_extra: int, n_extra_pool: int, now: int, d_outs: int, stream: int = 0):
        """place_dev with the pool's length: requests whose exclusion range leaves it are answered MMP_BAD_REQUEST, not followed."""
        self._ck(self.lib.mmp_place_batch_dev2(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None), int(n_extra_pool),
                                               int(now), C.c_void_p(d_outs), C.c_void_p(stream or None)))

    def place_c(self, caller, reqs_c, extra_pool: Optional[np.ndarray], now: int) -> np.ndarray:
        """The single-caller form (mmp_place_batch_c): caller = 1 PLACE_CALLER row, reqs_c = PLACE_REQ_C rows."""
        from ._lib import PLACE_CALLER, PLACE_REQ_C
        caller = np.ascontiguousarray(caller, dtype=PLACE_CALLER).reshape(1)
        reqs_c = np.ascontiguousarray(reqs_c, dtype=PLACE_REQ_C)
        extra = np.zeros(0, np.int32) if extra_pool is None else np.ascontiguousarray(extra_pool, dtype=np.int32)
        outs = np.zeros(len(reqs_c), dtype=PLACE_OUT)
        self._ck(self.lib.mmp_place_batch_c(self.h, ptr(caller), ptr(reqs_c) if len(reqs_c) else None, len(reqs_c),
                                            ptr(extra) if len(extra) else None, len(extra), int(now), ptr(outs) if len(reqs_c) else None))
        return outs

    def place_c_dev(self, caller, d_reqs: int, n: int, d_extra: int, n_extra_pool: int, now: int, d_outs: int, stream: int = 0):
        """The single-caller form on raw device pointers (24-byte rows), no sync; `caller` is a host PLACE_CALLER row."""
        from ._lib import PLACE_CALLER
        caller = np.ascontiguousarray(caller, dtype=PLACE_CALLER).reshape(1)
        self._ck(self.lib.mmp_place_batch_c_dev(self.h, ptr(caller), C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None),
                                                int(n_extra_pool), int(now), C.c_void_p(d_outs), C.c_void_p(stream or None)))

    def place_multi_dev(self, d_reqs, ns, d_extras, now: int, d_outs, stream: int = 0):
        """Several request arrays (raw device pointers), ONE launch, no sync: the same as place_dev per array."""
        k = len(d_reqs)
        arr = C.c_void_p * k
        r = arr(*[C.c_void_p(int(p)) for p in d_reqs])
        x = arr(*[C.c_void_p(int(p) or None) for p in d_extras])
        o = arr(*[C.c_void_p(int(p)) for p in d_outs])
        n = (C.c_int32 * k)(*[int(v) for v in ns])
        self._ck(self.lib.mmp_place_multi_dev(self.h, k, r, n, x, int(now), o, C.c_void_p(stream or None)))

    def serve_counters(self, reqs, in_use, last_used):
        """What a Java host assembles per request (MM.java:4343, :4356, :4360): for every copy of the request's model whose
        instance litelinks lists (here: the table's live flag), one (instance, inUse, lastUsed) entry.  -> (reqs with
        cnt_off / n_cnt filled, counters).  Needs the registry this Solver loaded (load_models)."""
        reqs = np.ascontiguousarray(reqs, dtype=SERVE_REQ).copy()
        in_use = np.ascontiguousarray(in_use, dtype=np.int32)
        last_used = np.ascontiguousarray(last_used, dtype=np.int64)
        models, ent_pod, live = getattr(self, "_models", None), getattr(self, "_ent_pod", None), getattr(self, "_live", None)
        if models is None or ent_pod is None or live is None:
            raise RuntimeError("serve_counters needs the host mirror of the registry and the instance list: load_models / load_pods "
                               "on this Solver (a registry ingested from JSON lives on the device only)")
        ok = (reqs["model"] >= 0) & (reqs["model"] < len(models))
        m = models[np.where(ok, reqs["model"], 0)]
        k = np.where(ok, m["n_loaded"], 0).astype(np.int64)
        off = np.zeros(len(reqs) + 1, np.int64)
        np.cumsum(k, out=off[1:])
        seg = np.repeat(np.arange(len(reqs)), k)
        j = np.arange(int(off[-1])) - off[seg]
        pod = ent_pod[m["ent_off"][seg] + j] if len(seg) else np.zeros(0, np.int32)
        listed = (pod >= 0) & (pod < len(live))
        listed[listed] &= live[pod[listed]]
        n_cnt = np.bincount(seg[listed], minlength=len(reqs)).astype(np.int32) if len(seg) else np.zeros(len(reqs), np.int32)
        c_off = np.zeros(len(reqs) + 1, np.int64)
        np.cumsum(n_cnt, out=c_off[1:])
        counters = np.zeros(int(c_off[-1]), dtype=SERVE_COUNTER)
        counters["pod"] = pod[listed]
        counters["in_use"] = in_use[pod[listed]]
        counters["last_used"] = last_used[pod[listed]]
        reqs["cnt_off"], reqs["n_cnt"] = c_off[:-1], n_cnt
        return reqs, counters

    def serve(self, reqs, in_use, last_used, excl_pod, excl_time, now) -> np.ndarray:
        """Convenience form for tests / benchmarks that hold per-INSTANCE counter arrays: builds the per-request counter
        entries (serve_counters) and calls serve_k."""
        reqs, counters = self.serve_counters(reqs, in_use, last_used)
        return self.serve_k(reqs, counters, excl_pod, excl_time, now)

    def serve_k(self, reqs, counters, excl_pod, excl_time, now) -> np.ndarray:
        """mmp_serve_batch: requests with their own (instance, inUse, lastUsed) entries — O(copies) per request."""
        reqs = np.ascontiguousarray(reqs, dtype=SERVE_REQ)
        counters = np.ascontiguousarray(counters, dtype=SERVE_COUNTER)
        excl_pod = np.ascontiguousarray(excl_pod, dtype=np.int32)
        excl_time = np.ascontiguousarray(excl_time, dtype=np.int64)
        outs = np.zeros(len(reqs), dtype=SERVE_OUT)
        self._ck(self.lib.mmp_serve_batch(self.h, ptr(reqs), len(reqs), ptr(counters) if len(counters) else None, len(counters),
                                          ptr(excl_pod) if len(excl_pod) else None,
                                          ptr(excl_time) if len(excl_time) else None, len(excl_pod),
                                          int(now), ptr(outs)))
        return outs

    def load_caches(self, seg_off, last_used, weight, capacity):
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int32)
        last_used = np.ascontiguousarray(last_used, dtype=np.int64)
        weight = np.ascontiguousarray(weight, dtype=np.int32)
        capacity = np.ascontiguousarray(capacity, dtype=np.int64)
        self._ck(self.lib.mmp_caches_load(self.h, len(capacity), ptr(seg_off),
                                          ptr(last_used) if len(last_used) else None,
                                          ptr(weight) if len(weight) else None, ptr(capacity)))

    def evict(self, reqs, now) -> np.ndarray:
        reqs = np.ascontiguousarray(reqs, dtype=EVICT_REQ)
        outs = np.zeros(len(reqs), dtype=EVICT_OUT)
        self._ck(self.lib.mmp_evict_batch(self.h, ptr(reqs), len(reqs), int(now), ptr(outs)))
        return outs

    # ---- stateful keyed caches (rows a12 + a13) -------------------------------------------------
    def load_caches_keyed(self, seg_off, last_used, weight, key, capacity, ubm=None):
        from ._lib import UBM_STATE
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int32)
        last_used = np.ascontiguousarray(last_used, dtype=np.int64)
        weight = np.ascontiguousarray(weight, dtype=np.int32)
        key = np.ascontiguousarray(key, dtype=np.int32)
        capacity = np.ascontiguousarray(capacity, dtype=np.int64)
        ubm = None if ubm is None else np.ascontiguousarray(ubm, dtype=UBM_STATE)
        self._kcache_slots = int(seg_off[-1])
        self._ck(self.lib.mmp_caches_load_keyed(self.h, len(capacity), ptr(seg_off),
                                                ptr(last_used) if len(last_used) else None,
                                                ptr(weight) if len(weight) else None, ptr(key) if len(key) else None,
                                                ptr(capacity), ptr(ubm)))

    def cache_replay(self, ops, now):
        """Apply clhm / ModelCacheUnloadBufManager operations in order; returns (outs, evicted_keys)."""
        from ._lib import CACHE_OP, CACHE_OP_OUT
        ops = np.ascontiguousarray(ops, dtype=CACHE_OP)
        outs = np.zeros(len(ops), dtype=CACHE_OP_OUT)
        self._kcache_slots = getattr(self, "_kcache_slots", 0) + len(ops)
        ev = np.zeros(max(self._kcache_slots, 1), np.int32)
        used = C.c_int32(0)
        self._ck(self.lib.mmp_cache_replay(self.h, ptr(ops) if len(ops) else None, len(ops), int(now),
                                           ptr(outs) if len(ops) else None, ptr(ev), len(ev), C.byref(used)))
        return outs, ev[: used.value]

    def cache_read(self, cache: int, max_entries: int = 4096):
        from ._lib import UBM_STATE
        lu = np.zeros(max_entries, np.int64)
        wt = np.zeros(max_entries, np.int32)
        key = np.zeros(max_entries, np.int32)
        n, cap, ws = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        ubm = np.zeros(1, dtype=UBM_STATE)
        self._ck(self.lib.mmp_cache_read(self.h, int(cache), max_entries, ptr(lu), ptr(wt), ptr(key), C.byref(n),
                                         C.byref(cap), C.byref(ws), ptr(ubm)))
        m = min(n.value, max_entries)
        return {"last_used": lu[:m], "weight": wt[:m], "key": key[:m], "capacity": cap.value,
                "weighted_size": ws.value, "ubm": ubm[0]}

    def gates(self, reqs, excl_pod, excl_time, explicit_pool, now, in_use_failure_expiry_ms=450_000) -> np.ndarray:
        from ._lib import GATE_OUT, GATE_REQ
        reqs = np.ascontiguousarray(reqs, dtype=GATE_REQ)
        excl_pod = np.ascontiguousarray(excl_pod, dtype=np.int32)
        excl_time = np.ascontiguousarray(excl_time, dtype=np.int64)
        explicit_pool = np.ascontiguousarray(explicit_pool, dtype=np.int32)
        outs = np.zeros(len(reqs), dtype=GATE_OUT)
        self._ck(self.lib.mmp_gate_batch(self.h, ptr(reqs), len(reqs), ptr(excl_pod) if len(excl_pod) else None,
                                         ptr(excl_time) if len(excl_time) else None, len(excl_pod),
                                         ptr(explicit_pool) if len(explicit_pool) else None, len(explicit_pool),
                                         int(now), int(in_use_failure_expiry_ms), ptr(outs)))
        return outs

    def route(self, gate_reqs, serve_reqs, counters, excl_pod, excl_time, explicit_pool, now, in_use_failure_expiry_ms=450_000):
        """The cache-hit route in one launch (mmp_route_batch): -> (gate outs, serve outs)."""
        from ._lib import GATE_OUT, GATE_REQ, SERVE_COUNTER, SERVE_OUT, SERVE_REQ
        gate_reqs = np.ascontiguousarray(gate_reqs, dtype=GATE_REQ)
        serve_reqs = np.ascontiguousarray(serve_reqs, dtype=SERVE_REQ)
        counters = np.ascontiguousarray(counters, dtype=SERVE_COUNTER)
        excl_pod = np.ascontiguousarray(excl_pod, dtype=np.int32)
        excl_time = np.ascontiguousarray(excl_time, dtype=np.int64)
        explicit_pool = np.ascontiguousarray(explicit_pool, dtype=np.int32)
        n = len(gate_reqs)
        gouts = np.zeros(n, dtype=GATE_OUT)
        souts = np.zeros(n, dtype=SERVE_OUT)
        self._ck(self.lib.mmp_route_batch(self.h, ptr(gate_reqs) if n else None, ptr(serve_reqs) if n else None, n,
                                          ptr(counters) if len(counters) else None, len(counters),
                                          ptr(excl_pod) if len(excl_pod) else None, ptr(excl_time) if len(excl_time) else None,
                                          len(excl_pod), ptr(explicit_pool) if len(explicit_pool) else None, len(explicit_pool),
                                          C.c_int64(int(now)), C.c_int64(int(in_use_failure_expiry_ms)),
                                          ptr(gouts) if n else None, ptr(souts) if n else None))
        return gouts, souts

    def miss(self, gate_reqs, place_reqs, excl_pod, excl_time, explicit_pool, extra_pool, now, in_use_failure_expiry_ms=450_000):
        """The cache-miss route in one call (mmp_miss_batch): the request guards + the load target -> (gate outs, place outs)."""
        from ._lib import GATE_OUT, GATE_REQ, PLACE_OUT, PLACE_REQ
        gate_reqs = np.ascontiguousarray(gate_reqs, dtype=GATE_REQ)
        place_reqs = np.ascontiguousarray(place_reqs, dtype=PLACE_REQ)
        excl_pod = np.ascontiguousarray(excl_pod, dtype=np.int32)
        excl_time = np.ascontiguousarray(excl_time, dtype=np.int64)
        explicit_pool = np.ascontiguousarray(explicit_pool, dtype=np.int32)
        extra_pool = np.ascontiguousarray(extra_pool, dtype=np.int32)
        n = len(gate_reqs)
        gouts = np.zeros(n, dtype=GATE_OUT)
        pouts = np.zeros(n, dtype=PLACE_OUT)
        self._ck(self.lib.mmp_miss_batch(self.h, ptr(gate_reqs) if n else None, ptr(place_reqs) if n else None, n,
                                         ptr(excl_pod) if len(excl_pod) else None, ptr(excl_time) if len(excl_time) else None,
                                         len(excl_pod), ptr(explicit_pool) if len(explicit_pool) else None, len(explicit_pool),
                                         ptr(extra_pool) if len(extra_pool) else None, len(extra_pool),
                                         C.c_int64(int(now)), C.c_int64(int(in_use_failure_expiry_ms)),
                                         ptr(gouts) if n else None, ptr(pouts) if n else None))
        return gouts, pouts

    def proactive_plan(self, default_model_size_units: int, now: int, max_out: int, partition: int = -1, skip_models=None):
        """a17: (models, last_used, info) the leader would proactively load, MRU first; partition >= 0: for that
        ProhibitedTypeSet partition only (one reaper call per partition when type constraints exist)."""
        from ._lib import PROACTIVE_INFO
        om = np.zeros(max(max_out, 1), np.int32)
        ol = np.zeros(max(max_out, 1), np.int64)
        info = np.zeros(1, dtype=PROACTIVE_INFO)
        skip = np.ascontiguousarray(skip_models if skip_models is not None else np.zeros(0, np.int32), dtype=np.int32)
        self._ck(self.lib.mmp_proactive_plan_subset(self.h, int(partition), ptr(skip) if len(skip) else None, len(skip),
                                                    int(default_model_size_units), int(now), int(max_out),
                                                    ptr(om), ptr(ol), ptr(info)))
        n = min(int(info[0]["n_selected"]), max_out)
        return om[:n].copy(), ol[:n].copy(), info[0]

    def scaleup_plan(self, entries, params):
        """a15: (outs, overloaded[P], skipped)"""
        from ._lib import CACHE_ENTRY, SCALEUP_OUT, SCALEUP_PARAMS
        entries = np.ascontiguousarray(entries, dtype=CACHE_ENTRY)
        params = np.ascontiguousarray(params, dtype=SCALEUP_PARAMS).reshape(1)
        outs = np.zeros(len(entries), dtype=SCALEUP_OUT)
        ov = np.zeros(max(self.n_pods, 1), np.uint8)
        sk = C.c_int32(0)
        self._ck(self.lib.mmp_scaleup_plan(self.h, ptr(entries) if len(entries) else None, len(entries), ptr(params),
                                           ptr(outs) if len(entries) else None, ptr(ov), C.byref(sk)))
        return outs, ov[: self.n_pods], sk.value

    def scaleup_plan_conc(self, entries, conc, params, conc_params):
        """a15 with limitModelConcurrency == true (latency-based): (outs, conc_outs, overloaded[P], skipped, result)"""
        from ._lib import CACHE_ENTRY, CONC_ENTRY, CONC_OUT, CONC_PARAMS, CONC_RESULT, SCALEUP_OUT, SCALEUP_PARAMS
        entries = np.ascontiguousarray(entries, dtype=CACHE_ENTRY)
        conc = np.ascontiguousarray(conc, dtype=CONC_ENTRY)
        assert len(conc) == len(entries)
        params = np.ascontiguousarray(params, dtype=SCALEUP_PARAMS).reshape(1)
        cparams = np.ascontiguousarray(conc_params, dtype=CONC_PARAMS).reshape(1)
        outs = np.zeros(len(entries), dtype=SCALEUP_OUT)
        couts = np.zeros(len(entries), dtype=CONC_OUT)
        res = np.zeros(1, dtype=CONC_RESULT)
        ov = np.zeros(max(self.n_pods, 1), np.uint8)
        sk = C.c_int32(0)
        n = len(entries)
        self._ck(self.lib.mmp_scaleup_plan_conc(self.h, ptr(entries) if n else None, ptr(conc) if n else None, n, ptr(params),
                                                ptr(cparams), ptr(outs) if n else None, ptr(couts) if n else None, ptr(ov),
                                                C.byref(sk), ptr(res)))
        return outs, couts, ov[: self.n_pods], sk.value, res[0]

    def scaledown_plan_conc(self, entries, conc, params, dynamic_rpm_scale_constant):
        from ._lib import CACHE_ENTRY, CONC_ENTRY, SCALEDOWN_PARAMS
        entries = np.ascontiguousarray(entries, dtype=CACHE_ENTRY)
        conc = np.ascontiguousarray(conc, dtype=CONC_ENTRY)
        assert len(conc) == len(entries)
        params = np.ascontiguousarray(params, dtype=SCALEDOWN_PARAMS).reshape(1)
        rem = np.zeros(max(len(entries), 1), np.uint8)
        n = len(entries)
        self._ck(self.lib.mmp_scaledown_plan_conc(self.h, ptr(entries) if n else None, ptr(conc) if n else None, n, ptr(params),
                                                  int(dynamic_rpm_scale_constant), ptr(rem)))
        return rem[:n]

    def scaledown_plan(self, entries, params):
        from ._lib import CACHE_ENTRY, SCALEDOWN_PARAMS
        entries = np.ascontiguousarray(entries, dtype=CACHE_ENTRY)
        params = np.ascontiguousarray(params, dtype=SCALEDOWN_PARAMS).reshape(1)
        rem = np.zeros(max(len(entries), 1), np.uint8)
        self._ck(self.lib.mmp_scaledown_plan(self.h, ptr(entries) if len(entries) else None, len(entries), ptr(params),
                                             ptr(rem)))
        return rem[: len(entries)]

    def migration_plan(self, entries, self_pod, now, cutoff_age_ms=3_600_000):
        from ._lib import CACHE_ENTRY
        entries = np.ascontiguousarray(entries, dtype=CACHE_ENTRY)
        act = np.zeros(max(len(entries), 1), np.uint8)
        wait = np.zeros(max(len(entries), 1), np.uint8)
        self._ck(self.lib.mmp_migration_plan(self.h, ptr(entries) if len(entries) else None, len(entries), int(self_pod),
                                             int(now), int(cutoff_age_ms), ptr(act), ptr(wait)))
        return act[: len(entries)], wait[: len(entries)]

    # ---- pod-axis shard mode (include/mmplace.h "pod-axis sharding"; orchestrated by dist.py) ----
    def shard_configure(self, shard: int, n_shards: int):
        self._ck(self.lib.mmp_shard_configure(self.h, int(shard), int(n_shards)))
        self.shard, self.n_shards = int(shard), int(n_shards)

    def shard_xchg_slots(self, phase: int) -> int:
        return int(self.lib.mmp_shard_xchg_slots(int(phase), int(self.n_shards)))

    def shard_rank_dev(self, d_rank: int):
        self._ck(self.lib.mmp_shard_rank_dev(self.h, C.c_void_p(d_rank)))

    def shard_commit_dev(self, d_rank: int):
        self._ck(self.lib.mmp_shard_commit_dev(self.h, C.c_void_p(d_rank)))

    def shard_phase_dev(self, phase: int, d_reqs: int, n: int, d_extra: int, now: int, d_xchg, d_outs: int,
                        stream: int = 0):
        arr = (C.c_void_p * 6)(*[C.c_void_p(int(x)) for x in d_xchg])
        self._ck(self.lib.mmp_shard_place_phase_dev(self.h, int(phase), C.c_void_p(d_reqs), int(n),
                                                    C.c_void_p(d_extra or None), int(now), arr,
                                                    C.c_void_p(d_outs or None), C.c_void_p(stream or None)))

    def shard_fast_slots(self) -> int:
        return int(self.lib.mmp_shard_fast_slots())

    def shard_fast_dev(self, d_reqs: int, n: int, d_extra: int, now: int, d_xf: int, stream: int = 0):
        self._ck(self.lib.mmp_shard_place_fast_dev(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None), int(now),
                                                   C.c_void_p(d_xf), C.c_void_p(stream or None)))

    def shard_fast_finish_dev(self, d_reqs: int, n: int, d_xf: int, d_outs: int, stream: int = 0):
        """-> (n_rest, device pointer of the compacted requests, device pointer of their result rows)"""
        n_rest, rr, ro = C.c_int32(0), C.c_void_p(0), C.c_void_p(0)
        self._ck(self.lib.mmp_shard_place_fast_finish_dev(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_xf),
                                                          C.c_void_p(d_outs), C.c_void_p(stream or None), C.byref(n_rest),
                                                          C.byref(rr), C.byref(ro)))
        return n_rest.value, rr.value or 0, ro.value or 0

    def shard_fast_scatter_dev(self, n_rest: int, d_outs: int, stream: int = 0):
        self._ck(self.lib.mmp_shard_place_fast_scatter_dev(self.h, int(n_rest), C.c_void_p(d_outs), C.c_void_p(stream or None)))

    # ---- the pod-axis group with RCCL inside the library (include/mmplace.h) ----
    def shard_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._ck(self.lib.mmp_shard_unique_id(buf))
        return buf.raw

    def shard_group_init(self, unique_id: bytes | None, rank: int, world: int):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self._ck(self.lib.mmp_shard_group_init(self.h, buf, int(rank), int(world)))
        self.n_shards = int(world)

    def shard_group_destroy(self):
        self._ck(self.lib.mmp_shard_group_destroy(self.h))

    def shard_commit(self):
        self._ck(self.lib.mmp_shard_commit(self.h))

    def shard_place(self, reqs, extra, now: int):
        """Collective over the group, host pointers -> (outs, n_rest)."""
        reqs = np.ascontiguousarray(reqs, dtype=PLACE_REQ)
        extra = np.ascontiguousarray(extra if extra is not None else np.zeros(0, np.int32), dtype=np.int32)
        outs = np.zeros(len(reqs), dtype=PLACE_OUT)
        n_rest = C.c_int32(0)
        self._ck(self.lib.mmp_shard_place_batch(self.h, ptr(reqs), len(reqs), ptr(extra) if len(extra) else None, len(extra),
                                                int(now), ptr(outs), C.byref(n_rest)))
        return outs, n_rest.value

    def shard_place_dev(self, d_reqs: int, n: int, d_extra: int, now: int, d_outs: int) -> int:
        """Collective over the group, device pointers; returns when d_outs is complete -> n_rest."""
        n_rest = C.c_int32(0)
        self._ck(self.lib.mmp_shard_place_batch_dev(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None), int(now),
                                                    C.c_void_p(d_outs), C.byref(n_rest)))
        return n_rest.value

    def shard_place_async_dev(self, d_reqs: int, n: int, d_extra: int, now: int, d_outs: int):
        """The same without the synchronisation at the end: completed by the next group call or by shard_wait()."""
        self._ck(self.lib.mmp_shard_place_batch_async_dev(self.h, C.c_void_p(d_reqs), int(n), C.c_void_p(d_extra or None), int(now),
                                                          C.c_void_p(d_outs)))

    def shard_wait(self) -> int:
        n_rest = C.c_int32(0)
        self._ck(self.lib.mmp_shard_wait(self.h, C.byref(n_rest)))
        return n_rest.value

    def sync(self):
        self._ck(self.lib.mmp_sync(self.h))

    def profile(self, enable: bool = True):
        """HIP-event timing of the kernels inside each host-pointer call (read with last_kernel_ms())."""
        self._ck(self.lib.mmp_profile(self.h, 1 if enable else 0))

    def last_kernel_ms(self) -> float:
        return float(self.lib.mmp_last_kernel_ms(self.h))
