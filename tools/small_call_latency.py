#!/usr/bin/env python
"""Wall time of the small host-pointer calls a mesh instance makes around one model load: the request guards
(mmp_gate_batch), the load target (mmp_place_batch), the cache-eviction evaluation (mmp_evict_batch), the serve target
(mmp_serve_batch) and the cache-hit route in one call (mmp_route_batch), n = 1 and n = 256, arguments marshalled once."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import _lib, workload as wl  # noqa: E402
from modelmesh_amd._lib import ptr  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
cs = wl.ChurnStream(fleet, 0xC5)
s.load_caches(cs.seg_off, cs.cache_lu, cs.cache_wt, cs.cache_cap)
now = fleet.now
reqs, _ = wl.make_requests(fleet, 3, n=256, extra_frac=0.0)


def timed(fn, args, reps=3000):
    for _ in range(200):
        fn(*args)
    t = np.zeros(reps)
    for i in range(reps):
        t0 = time.perf_counter()
        rc = fn(*args)
        t[i] = time.perf_counter() - t0
        assert rc == 0
    return np.percentile(t, 50) * 1e6, np.percentile(t, 99) * 1e6


for n in (1, 256):
    po = np.zeros(n, dtype=_lib.PLACE_OUT)
    r = reqs[:n].copy()
    print(f"n={n:4d} place  p50 %.1f us  p99 %.1f us" % timed(s.lib.mmp_place_batch, (s.h, ptr(r), C.c_int32(n), None, C.c_int32(0), C.c_int64(now), ptr(po))))
    ev = np.zeros(n, dtype=_lib.EVICT_REQ)
    ev["cache"] = np.arange(n) % fleet.n_pods
    ev["weight"] = 6400
    eo = np.zeros(n, dtype=_lib.EVICT_OUT)
    print(f"n={n:4d} evict  p50 %.1f us  p99 %.1f us" % timed(s.lib.mmp_evict_batch, (s.h, ptr(ev), C.c_int32(n), C.c_int64(now), ptr(eo))))
    g = np.zeros(n, dtype=_lib.GATE_REQ)
    g["model"] = np.arange(n)
    g["self_pod"] = np.arange(n) % fleet.n_pods
    g["cache_capacity"] = 8_388_608
    g["loader_predicted"] = 6400
    go = np.zeros(n, dtype=_lib.GATE_OUT)
    print(f"n={n:4d} gates  p50 %.1f us  p99 %.1f us" % timed(s.lib.mmp_gate_batch, (s.h, ptr(g), C.c_int32(n), None, None, C.c_int32(0), None, C.c_int32(0), C.c_int64(now), C.c_int64(450_000), ptr(go))))
    # the cache-hit route: guards + serve target — two calls (mmp_gate_batch, mmp_serve_batch) against one (mmp_route_batch)
    rng = np.random.default_rng(n)
    sr = np.zeros(n, dtype=_lib.SERVE_REQ)
    sr["model"], sr["self_pod"] = g["model"], g["self_pod"]
    sr["assume_completed_ms"] = 3000
    in_use = rng.integers(0, 3, fleet.n_pods).astype(np.int32)
    last_used = (now - rng.integers(0, 10_000, fleet.n_pods)).astype(np.int64)
    sr, cnt = s.serve_counters(sr, in_use, last_used)
    if len(cnt) == 0:
        cnt = np.zeros(1, dtype=_lib.SERVE_COUNTER)
    so = np.zeros(n, dtype=_lib.SERVE_OUT)
    nc = C.c_int32(int(sr["n_cnt"].sum()))
    serve_args = (s.h, ptr(sr), C.c_int32(n), ptr(cnt), nc, None, None, C.c_int32(0), C.c_int64(now), ptr(so))
    print(f"n={n:4d} serve  p50 %.1f us  p99 %.1f us" % timed(s.lib.mmp_serve_batch, serve_args))
    route_args = (s.h, ptr(g), ptr(sr), C.c_int32(n), ptr(cnt), nc, None, None, C.c_int32(0), None, C.c_int32(0), C.c_int64(now), C.c_int64(450_000),
                  ptr(go), ptr(so))
    print(f"n={n:4d} route  p50 %.1f us  p99 %.1f us   (guards + serve target in one call)" % timed(s.lib.mmp_route_batch, route_args))
    # the cache-miss route: guards + load target — two calls against one (mmp_miss_batch)
    miss_args = (s.h, ptr(g), ptr(r), C.c_int32(n), None, None, C.c_int32(0), None, C.c_int32(0), None, C.c_int32(0), C.c_int64(now),
                 C.c_int64(450_000), ptr(go), ptr(po))
    r["model"] = g["model"]
    print(f"n={n:4d} miss   p50 %.1f us  p99 %.1f us   (guards + load target in one call)" % timed(s.lib.mmp_miss_batch, miss_args))
s.close()
