#!/usr/bin/env python
"""The single-caller request form against the same decisions as 64-byte rows: launch time of the load-target kernel (one stream,
K launches between an event pair, 4 rotating buffers so that requests come from HBM), C3 and C3 with every instance full.
usage: tools/r5/caller_timing.py [n ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modelmesh_amd import _lib  # noqa: E402
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd._lib import PLACE_OUT  # noqa: E402
from modelmesh_amd.solver import Solver, ptr  # noqa: E402

ns = [int(x) for x in sys.argv[1:]] or [100_000, 800_000]
dev = torch.device("cuda", 0)
for full in (False, True):
    fleet = wl.make_fleet("C3")
    if full:
        wl.make_full_cluster(fleet)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    sets = -(-max(ns) // fleet.n_models)
    sp = 4321
    row = fleet.pods[sp]
    bufs = []
    for b in range(4):
        parts, ex_parts, off = [], [], 0
        for k in range(sets):
            rq, ex = wl.make_requests(fleet, seed=0xCA11E + b * 31 + k)
            rq = rq.copy()
            rq["extra_off"] += off
            off += len(ex)
            parts.append(rq)
            ex_parts.append(ex)
        rq = np.concatenate(parts)
        ex = np.concatenate(ex_parts)
        rq["self_pod"], rq["flags"] = sp, 0
        rq["fresh_lru"], rq["fresh_capacity"], rq["fresh_used"] = row["lru_time"], row["capacity"], row["used"] + 1000
        rq["fresh_count"], rq["fresh_rpm"] = row["count"], 0
        caller, rc = _lib.split_caller(rq)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)  # noqa: E731
        bufs.append((up(rq), up(rc), up(ex), len(ex), torch.zeros(len(rq) * 16, dtype=torch.uint8, device=dev),
                     torch.zeros(len(rq) * 16, dtype=torch.uint8, device=dev)))
    st = torch.cuda.Stream(dev)
    K = 200
    for n in ns:
        res = {}
        for form in ("rows", "caller"):
            if form == "rows":
                fn = s.lib.mmp_place_batch_dev
                args = [(s.h, C.c_void_p(r.data_ptr()), C.c_int32(n), C.c_void_p(e.data_ptr()), C.c_int64(fleet.now), C.c_void_p(o.data_ptr()),
                         C.c_void_p(st.cuda_stream)) for r, _, e, _, o, _ in bufs]
            else:
                fn = s.lib.mmp_place_batch_c_dev
                args = [(s.h, ptr(caller), C.c_void_p(r.data_ptr()), C.c_int32(n), C.c_void_p(e.data_ptr()), C.c_int32(ne), C.c_int64(fleet.now),
                         C.c_void_p(o.data_ptr()), C.c_void_p(st.cuda_stream)) for _, r, e, ne, _, o in bufs]
            for i in range(20):
                assert fn(*args[i % 4]) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(K):
                fn(*args[i % 4])
            e1.record(st)
            torch.cuda.synchronize()
            res[form] = e0.elapsed_time(e1) * 1e3 / K
        a = np.frombuffer(bufs[0][4].cpu().numpy().tobytes(), dtype=PLACE_OUT)[:n]
        b = np.frombuffer(bufs[0][5].cpu().numpy().tobytes(), dtype=PLACE_OUT)[:n]
        same = bool(np.array_equal(a, b))
        print(f"C3{' full cluster' if full else ''}: n {n:8d}  64-byte rows {res['rows']:7.2f} us ({n * 80 / res['rows'] / 1e6:5.2f} TB/s of requests + results)   "
              f"single-caller form {res['caller']:7.2f} us ({n * 40 / res['caller'] / 1e6:5.2f} TB/s)   {n / res['caller'] / 1e3:6.2f} G decisions/s   identical results: {same}")
    s.close()
