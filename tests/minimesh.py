"""A closed-loop miniature mesh for the reference's CLUSTER tests (test infrastructure).

ModelMeshEvictionsTest's cluster cases (testMultiLoadCluster :292-310, testMultiLoadWithEvictionCluster
:323-358, testMultiLoadWithEvictionClusterReuse :371-409) exercise the hot path as a whole: every
`addModel(load=true, sync=true)` is one load-target decision (CacheMissForwardingLB.getNext, MM.java:4776)
at the instance the client's request lands on, the request guards around it (MM.java:3771-3774, :4003-4042,
:3870-3884, :5185-5197), one `loadLocal` on the chosen instance (insertNewEntry -> adjustNewEntrySpaceRequest
-> claim -> adjustWeightAfterLoad, MM.java:5063, :2094-2098, :2281-2292, :2425) whose evictions deregister
the evicted copies (onEviction, MM.java:2867-2933), and a republished InstanceRecord
(getFreshInstanceRecord, MM.java:5369-5386) that the NEXT decision sees.  This module replays that loop
over one or more BACKENDS in lock step — the CPU oracle, and the device through the C ABI — and checks after
every step that all backends took the same decision and hold the same caches.  The reference's assertions
(which models must still be loaded) are made by the tests that drive it.

DummyModelMesh (the reference's test runtime, DummyModelMesh.java:39-64): capacity 10 x 20 MiB = 25600
units, every model 2560 units, 8 loading threads => unload buffer 2560 units (MM.java:748-753), 9 models
fit per instance, minSpaceUnits 2560 (MM.java:765-771).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from modelmesh_amd import _lib
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Fleet
from oracle import bind as ob

CAPACITY, MODEL_UNITS, THREADS = 25600, 2560, 8
RESERVED = max(min(THREADS * MODEL_UNITS // 4, CAPACITY // 10), CAPACITY // 100)
MIN_SPACE = 2560
MIN_CHURN_AGE_MS = 600_000          # tas.min_churn_age_ms default, MM.java:697
LOAD_TIMEOUT_MS = 90_000
HOUR = 3_600_000                    # LASTUSED_AGE_ON_ADD_MS, MM.java:266
JMAX = _lib.JAVA_LONG_MAX


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if len(a) else None


class OracleBackend:
    """CPU restatement: one C-oracle cache + unload-buffer manager per instance, orc_place for decisions."""
    name = "oracle"

    def __init__(self, n_pods, now):
        self.caches = [ob.CCache(CAPACITY, RESERVED, now) for _ in range(n_pods)]
        self.lib = ob.load()
        self.fleet = None

    def publish(self, fleet, pod_idx, model_idx):
        self.fleet = fleet

    def place(self, req, now):
        o = ob.OracleFleet(self.fleet).place(req, None, now)
        return int(o["chosen"][0]), int(o["best"][0]), int(o["n_candidates"][0]), int(o["hash"][0])

    def gate(self, q, now):
        """The guards a load passes (bits of mmp_gate_out), from the oracle's single-guard functions."""
        f, lib = self.fleet, self.lib
        q = q[0]
        mr = f.models[q["model"]]
        lo, nl, nf = int(mr["ent_off"]), int(mr["n_loaded"]), int(mr["n_failed"])
        lp = np.ascontiguousarray(f.ent_pod[lo: lo + nl])
        fp = f.ent_pod[lo + nl: lo + nl + nf]
        ft = np.ascontiguousarray(f.ent_time[lo + nl: lo + nl + nf])
        stats = np.ascontiguousarray(ob.type_set_stats(f))[:1]
        in_table = np.ascontiguousarray(((f.pods["flags"] & 4) == 0).astype(np.uint8))
        fl, bits = int(q["flags"]), 0
        if lib.orc_load_failures_breached(_p(ft), len(ft), now, 450_000):
            bits |= _lib.GATE_FAILURES_BREACHED
        if lib.orc_load_locations_breached(_p(lp), len(lp), None, 0, _p(in_table)):
            bits |= _lib.GATE_LOCATIONS_BREACHED
        if (q["self_pod"] in lp) or (q["self_pod"] in fp):
            bits |= _lib.GATE_LOCAL_NOT_ALLOWED
        if lib.orc_churn_reject(f.min_churn_age_ms, f.min_space_units, int(q["cache_capacity"]),
                                int(q["cache_weighted_size"]), int(q["cache_oldest_time"]), now):
            bits |= _lib.GATE_CHURN_REJECT
        rej = C.c_int(0)
        init = lib.orc_load_local_initial_size((fl >> 5) & 1, int(q["size_hint"]), int(q["loading_count"]),
                                               int(q["weight_predict_cutoff"]), int(q["loader_predicted"]),
                                               stats.ctypes.data_as(C.c_void_p), (fl >> 3) & 1,
                                               int(q["last_used_time"]), int(q["cache_capacity"]),
                                               int(q["cache_weighted_size"]), int(q["cache_oldest_time"]), C.byref(rej))
        if rej.value:
            bits |= _lib.GATE_EARLY_REJECT
        if lib.orc_reload_elsewhere((fl >> 4) & 1, int(q["loaded_time"]), int(q["load_timeout_ms"]), now,
                                    stats.ctypes.data_as(C.c_void_p)):
            bits |= _lib.GATE_RELOAD_ELSEWHERE
        return bits, int(init)

    def cache_ops(self, rows, now):
        out = []
        for (c, op, key, arg, t, flag) in rows:
            h = self.caches[c]
            res, ev = h.apply(op, key, arg, t, flag, now)
            out.append((int(res), [int(k) for k in ev], int(h.c.weighted_size), int(h.oldest_time()),
                        int(h.lib.orc_ubm_buffer_weight(C.byref(h.u)))))
        return out

    def cache_keys(self, c):
        return sorted(self.caches[c].keys())

    def close(self):
        pass


class PyOracleBackend:
    """The SECOND, independently written restatement (oracle/py_oracle.py): PLACEMENT_ORDER, getNext, clhm and the
    unload-buffer manager in plain Python.  It has no restatement of the request guards (gate() returns None and
    is left out of the comparison); decisions and every cache operation are compared with the other backends."""
    name = "py-oracle"

    def __init__(self, n_pods, now):
        from oracle import py_oracle as po
        self.po = po
        self.c = [po.Clhm(CAPACITY) for _ in range(n_pods)]
        self.u = [po.UnloadBufManager(c, RESERVED, now) for c in self.c]
        self.fleet = None

    def publish(self, fleet, pod_idx, model_idx):
        self.fleet = fleet

    def place(self, req, now):
        po, f = self.po, self.fleet
        mesh = po.Mesh(f.min_space_units, f.min_churn_age_ms, now)
        pods = [dict(lru_time=int(r["lru_time"]), capacity=int(r["capacity"]), used=int(r["used"]), version=int(r["version"]),
                     count=int(r["count"]), loading_threads=int(r["loading_threads"]),
                     loading_in_progress=int(r["loading_in_progress"]), rpm=int(r["rpm"]), id_order=int(r["id_order"]),
                     replica_set=int(r["replica_set"]), shutting_down=bool(r["flags"] & 5)) for r in f.pods]
        order = mesh.sorted_cluster_state(pods)
        r = req[0]
        m = f.models[r["model"]]
        ents = f.ent_pod[m["ent_off"]: m["ent_off"] + m["n_loaded"] + m["n_failed"]]
        loaded, failed = set(int(x) for x in ents[: m["n_loaded"]]), set(int(x) for x in ents[m["n_loaded"]:])
        fresh = dict(lru_time=int(r["fresh_lru"]), capacity=int(r["fresh_capacity"]), used=int(r["fresh_used"]),
                     count=int(r["fresh_count"]), rpm=int(r["fresh_rpm"]))
        live = {i for i in range(f.n_pods) if f.pods["flags"][i] & 2}
        chosen, best, shortlist, remaining = po.get_next(mesh, pods, order, live, set(), None, None, [set(), loaded, failed],
                                                         int(r["self_pod"]), bool(r["flags"] & 1), fresh, int(r["last_used"]),
                                                         int(r["pick"]))
        h = 0
        if shortlist:
            from tests.test_oracle_cross import _py_hash
            h = _py_hash(order, shortlist, remaining)
        return int(chosen), int(best), len(shortlist), int(h)

    def gate(self, q, now):
        return None

    def cache_ops(self, rows, now):
        out = []
        for (c, op, key, arg, t, flag) in rows:
            u, ch = self.u[c], self.c[c]
            ev0 = len(u.evicted)
            if op == _lib.COP_UBM_INSERT_NEW_ENTRY:
                res = int(u.insertNewEntry(key, arg, t, now))
            elif op == _lib.COP_UBM_ADJUST_SPACE_REQUEST:
                res = 1 if ch._find(key) is not None else 0
                u.adjustNewEntrySpaceRequest(arg, key, now)
            elif op == _lib.COP_UBM_CLAIM_SPACE:
                res = int(u.claimRequestedSpaceIfReady(arg, now))
            elif op == _lib.COP_UBM_ADJUST_AFTER_LOAD:
                res = 1 if (ch._find(key) is not None or arg == 0) else 0
                u.adjustWeightAfterLoad(arg, key, now)
            elif op == _lib.COP_UBM_UNLOAD_COMPLETE:
                u.unloadComplete(arg, bool(flag), now)
                res = 1 if flag else 0
            elif op == _lib.COP_GET:
                res = int(ch.get(key, t, now))
            else:
                raise ValueError(op)
            ev = [(_lib.UNLOADBUF_KEY if k == u.KEY else int(k)) for k, _ in u.evicted[ev0:]]
            out.append((res, ev, int(ch.weightedSize), int(ch.oldestTime()), int(u.buffer_weight())))
        return out

    def cache_keys(self, c):
        return sorted(int(k) for k in self.c[c].keys() if k != self.u[c].KEY)

    def close(self):
        pass


class DeviceBackend:
    """The product: libmmplace through the C ABI (modelmesh_amd.solver.Solver is a ctypes veneer)."""
    name = "device"

    def __init__(self, n_pods, now):
        from modelmesh_amd.solver import Solver
        self.s = Solver(MIN_SPACE, MIN_CHURN_AGE_MS)
        ubm = np.zeros(n_pods, dtype=_lib.UBM_STATE)
        ubm["reserved"] = RESERVED
        # every instance starts with the unload-buffer entry only (ModelCacheUnloadBufManager ctor, :81-88)
        seed = ob.CCache(CAPACITY, RESERVED, now)
        lu, wt, key = seed.nodes()
        seg = np.arange(n_pods + 1, dtype=np.int32) * len(lu)
        ubm["total_unloading"], ubm["total_occupancy"], ubm["cache_deficit"] = \
            seed.u.total_unloading, seed.u.total_occupancy, seed.u.cache_deficit
        self.s.load_caches_keyed(seg, np.tile(lu, n_pods), np.tile(wt, n_pods), np.tile(key, n_pods),
                                 np.full(n_pods, CAPACITY, np.int64), ubm)
        self.loaded = False

    def publish(self, fleet, pod_idx, model_idx):
        """Instance-table and registry events as the KV listeners would deliver them, then a commit."""
        s = self.s
        if not self.loaded:
            s.load_fleet(fleet)
            self.loaded = True
            return
        if len(pod_idx):
            s.upsert_pods(np.asarray(pod_idx, np.int32), fleet.pods[pod_idx])
        if len(model_idx):
            rows = fleet.models[model_idx].copy()
            ep, et, off = [], [], 0
            for i, m in enumerate(model_idx):
                r = fleet.models[m]
                n = int(r["n_loaded"] + r["n_failed"])
                ep += list(fleet.ent_pod[r["ent_off"]: r["ent_off"] + n])
                et += list(fleet.ent_time[r["ent_off"]: r["ent_off"] + n])
                rows["ent_off"][i] = off
                off += n
            s.upsert_models(np.asarray(model_idx, np.int32), rows, np.asarray(ep, np.int32), np.asarray(et, np.int64))
        s.commit()

    def place(self, req, now):
        o = self.s.place(req, None, now)
        return int(o["chosen"][0]), int(o["best"][0]), int(o["n_candidates"][0]), int(o["hash"][0])

    def gate(self, q, now):
        o = self.s.gates(q, np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int32), now)[0]
        mask = (_lib.GATE_FAILURES_BREACHED | _lib.GATE_LOCATIONS_BREACHED | _lib.GATE_LOCAL_NOT_ALLOWED |
                _lib.GATE_CHURN_REJECT | _lib.GATE_EARLY_REJECT | _lib.GATE_RELOAD_ELSEWHERE)
        return int(o["bits"]) & mask, int(o["initial_size"])

    def cache_ops(self, rows, now):
        ops = np.array([(c, op, key, arg, t, flag, 0) for (c, op, key, arg, t, flag) in rows], dtype=_lib.CACHE_OP)
        outs, ev = self.s.cache_replay(ops, now)
        return [(int(o["result"]), [int(k) for k in ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]],
                 int(o["weighted_size"]), int(o["oldest_time"]), int(o["buffer_weight"])) for o in outs]

    def cache_keys(self, c):
        return sorted(int(k) for k in self.s.cache_read(c)["key"] if k != _lib.UNLOADBUF_KEY)

    def close(self):
        self.s.close()


class MiniMesh:
    def __init__(self, n_pods, backends, seed, now=wl.NOW_MS, ingress="round_robin"):
        self.P, self.now, self.ingress = n_pods, now, ingress
        self.backends = [b(n_pods, now) for b in backends]
        self.rng = np.random.default_rng(seed)
        self.copies = []        # per model: {pod: load timestamp}  (ModelRecord.instanceIds)
        self.last_used = []     # per model: ModelRecord.lastUsed
        self.trace = []         # (model, ingress, chosen, target, evicted keys)
        self.next_ingress = 0
        # what each instance last published (getFreshInstanceRecord): an empty cache
        self.cache_view = [dict(ws=RESERVED, oldest=JMAX, buf=RESERVED, count=0) for _ in range(n_pods)]
        self._publish(list(range(n_pods)), [])

    def close(self):
        for b in self.backends:
            b.close()

    # ---- state -> records ------------------------------------------------------------------------
    def _pod_row(self, p):
        v = self.cache_view[p]
        row = np.zeros(1, dtype=wl.POD_ROW)
        oldest = v["oldest"]
        row["lru_time"] = JMAX if oldest == -1 else oldest              # MM.java:5380-5382
        row["capacity"], row["used"] = CAPACITY - v["buf"], v["ws"] - v["buf"]  # :5373-5378
        row["count"], row["loading_threads"], row["version"] = v["count"], THREADS, 1
        row["id_order"], row["flags"] = p, wl.POD_LIVE
        return row[0]

    def _fleet(self):
        pods = np.array([self._pod_row(p) for p in range(self.P)], dtype=wl.POD_ROW)
        models = np.zeros(len(self.copies), dtype=wl.MODEL_ROW)
        ep, et = [], []
        for m, c in enumerate(self.copies):
            models["ent_off"][m], models["n_loaded"][m] = len(ep), len(c)
            for pod in sorted(c):   # instanceIds is a TreeMap: instance-id order
                ep.append(pod), et.append(c[pod])
        models["last_used"] = self.last_used
        return Fleet(pods=pods, models=models, ent_pod=np.array(ep, np.int32), ent_time=np.array(et, np.int64),
                     min_space_units=MIN_SPACE, min_churn_age_ms=MIN_CHURN_AGE_MS, now=self.now)

    def _publish(self, pod_idx, model_idx):
        fleet = self._fleet()
        for b in self.backends:
            b.publish(fleet, pod_idx, model_idx)

    def _all(self, what, fn):
        """Run one step on every backend; they must agree."""
        res = [fn(b) for b in self.backends]
        for b, r in zip(self.backends[1:], res[1:]):
            assert r is None or r == res[0], f"{what}: {b.name} {r} != {self.backends[0].name} {res[0]}"
        return res[0]

    def _ops(self, rows):
        out = self._all("cache ops", lambda b: b.cache_ops(rows, self.now))
        c = rows[-1][0]
        res, _, ws, oldest, buf = out[-1]
        self.cache_view[c].update(ws=ws, oldest=oldest, buf=buf)
        return out

    # ---- the calls a mesh makes ------------------------------------------------------------------
    def _gate_req(self, model, pod, last_used, flags=0, loaded_time=-1):
        v = self.cache_view[pod]
        q = np.zeros(1, dtype=_lib.GATE_REQ)
        q["model"], q["self_pod"], q["flags"] = model, pod, flags
        q["last_used_time"], q["cache_capacity"] = last_used, CAPACITY
        q["cache_weighted_size"], q["cache_oldest_time"] = v["ws"], v["oldest"]
        q["loader_predicted"], q["weight_predict_cutoff"] = MODEL_UNITS, THREADS + THREADS // 3
        q["loaded_time"], q["load_timeout_ms"] = loaded_time, LOAD_TIMEOUT_MS
        return q

    def add_model(self, tick_ms=25):
        """addModel(id, info, load=true, sync=true) at the instance the client's balancer picks."""
        self.now += tick_ms
        m = len(self.copies)
        t = self.now - HOUR                          # registerModel's lastUsed, MM.java:3097-3101
        self.copies.append({})
        self.last_used.append(t)
        self._publish([], [m])                       # the new ModelRecord reaches every registry view
        # where the client's own (litelinks) balancer lands the request
        if self.ingress == "round_robin":
            ingress = self.next_ingress
            self.next_ingress = (ingress + 1) % self.P
        elif self.ingress == "random":
            ingress = int(self.rng.integers(0, self.P))
        else:                                        # "single": every request enters through instance 0
            ingress = 0
        bits, _ = self._all("ingress guards", lambda b: b.gate(self._gate_req(m, ingress, t), self.now))
        assert not bits & (_lib.GATE_FAILURES_BREACHED | _lib.GATE_LOCATIONS_BREACHED)
        # the load-target decision: external request from a balanced source => favourSelf (MM.java:3526, :3782)
        v = self.cache_view[ingress]
        r = np.zeros(1, dtype=wl.PLACE_REQ)
        r["model"], r["self_pod"], r["flags"] = m, ingress, 1
        r["pick"], r["last_used"] = int(self.rng.integers(0, 2**32)), t
        fresh = self._pod_row(ingress)
        r["fresh_lru"], r["fresh_capacity"], r["fresh_used"] = fresh["lru_time"], fresh["capacity"], fresh["used"]
        r["fresh_count"], r["fresh_rpm"] = v["count"], 0
        chosen, _, _, _ = self._all("load-target decision", lambda b: b.place(r, self.now))
        target = ingress if chosen in (_lib.MMP_NONE, _lib.MMP_SELF) else chosen
        assert 0 <= target < self.P
        evicted = self._load_local(m, target, t)
        self.trace.append((m, ingress, chosen, target, evicted))
        return target

    def _load_local(self, m, pod, last_used):
        bits, init = self._all("target guards",
                               lambda b: b.gate(self._gate_req(m, pod, last_used, flags=8), self.now))  # weCreatedCacheEntry
        assert not bits & _lib.GATE_LOCAL_NOT_ALLOWED, "Nowhere available to load"
        assert not bits & _lib.GATE_CHURN_REJECT, "Cache churn threshold exceeded"
        assert not bits & _lib.GATE_EARLY_REJECT and init == MODEL_UNITS
        out = self._ops([(pod, _lib.COP_UBM_INSERT_NEW_ENTRY, m, 1, last_used, 0),
                         (pod, _lib.COP_UBM_ADJUST_SPACE_REQUEST, m, init - 1, 0, 0)])
        evicted = out[0][1] + out[1][1]
        assert m not in evicted, "the new entry itself was evicted"
        touched = [m]
        for k in evicted:                            # onEviction: deregister + unload, MM.java:2867-2933
            loaded_time = self.copies[k].pop(pod)
            touched.append(k)
            gb, _ = self._all("eviction guard", lambda b: b.gate(
                self._gate_req(k, pod, self.last_used[k], loaded_time=loaded_time), self.now))
            assert not gb & _lib.GATE_RELOAD_ELSEWHERE  # copies this young are not re-placed (:2895)
        rows = [(pod, _lib.COP_UBM_UNLOAD_COMPLETE, 0, MODEL_UNITS, 0, 1) for _ in evicted]
        rows += [(pod, _lib.COP_UBM_CLAIM_SPACE, 0, MODEL_UNITS, 0, 0),
                 (pod, _lib.COP_UBM_ADJUST_AFTER_LOAD, m, 0, 0, 0)]
        out = self._ops(rows)
        assert out[len(evicted)][0] == 1, "claimRequestedSpaceIfReady failed"
        assert not any(o[1] for o in out)
        self.copies[m][pod] = self.now
        self.cache_view[pod]["count"] += 1 - len(evicted)
        keys = self._all("cache contents", lambda b: b.cache_keys(pod))
        assert len(keys) == self.cache_view[pod]["count"]
        self._publish([pod], touched)                # InstanceRecord + ModelRecords republished
        return evicted

    def use_model(self, m, tick_ms=5):
        """useModel -> invokeModel on a loaded copy -> runtimeCache.get(id, now) (clhm touch)."""
        self.now += tick_ms
        pod = sorted(self.copies[m])[0]
        out = self._ops([(pod, _lib.COP_GET, m, 0, 0, 0)])
        assert out[0][0] == 1
        self.last_used[m] = self.now
        self._publish([pod], [m])

    def loaded(self):
        return [m for m, c in enumerate(self.copies) if c]
