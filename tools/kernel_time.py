#!/usr/bin/env python
"""place_batch_kernel launch time on C3 (or argv[1]): K launches back to back on ONE stream between a HIP event pair,
(a) the same request batch every launch (inputs served from L2 / Infinity Cache) and (b) rotating through 48 distinct
batches (384 MB: inputs from HBM); then the same on 16 streams (step time).  One line per measurement.
Environment switches of the library apply (MMP_NO_HEADS=1: without the per-type head records)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
R = int(os.environ.get("KT_BATCHES", "48"))
fleet = wl.make_fleet(name)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
bufs = []
for b in range(R):
    rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + b)
    bufs.append((torch.from_numpy(rq.view(np.uint8).reshape(-1)).to(dev),
                 torch.from_numpy(np.ascontiguousarray(ex if len(ex) else np.zeros(1, np.int32))).to(dev),
                 torch.zeros(len(rq) * 16, dtype=torch.uint8, device=dev)))
n = len(rq)
fn = s.lib.mmp_place_batch_dev
tag = "no-heads" if os.environ.get("MMP_NO_HEADS") == "1" else "heads"


def args_of(b, st):
    r_, e_, o_ = bufs[b]
    return (s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now),
            C.c_void_p(o_.data_ptr()), C.c_void_p(st.cuda_stream))


def one_stream(rotate):
    st = torch.cuda.Stream(dev)
    a = [args_of(b if rotate else 0, st) for b in range(R)]
    for i in range(30):
        fn(*a[i % R])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(K):
        fn(*a[i % R])
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


def many_streams(ns, rotate, steps):
    sts = [torch.cuda.Stream(dev) for _ in range(ns)]
    a = [args_of((i % R) if rotate else (i % ns) % R, sts[i % ns]) for i in range(R * ns)]
    for i in range(max(2 * ns, 50)):
        fn(*a[i % len(a)])
    torch.cuda.synchronize()
    sched = [a[i % len(a)] for i in range(steps)]
    t0 = time.perf_counter()
    for x in sched:
        fn(*x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for rot in (False, True):
    us = one_stream(rot)
    print(f"{name} {tag} 1 stream, {'rotating %d batches' % R if rot else 'same batch':>19}: {us:7.2f} us per launch of {n} decisions "
          f"({n / us / 1e3:6.2f} G/s)", flush=True)
for ns in (4, 16):
    for rot in (False, True):
        for steps in (20, 1000):
            us = many_streams(ns, rot, steps)
            print(f"{name} {tag} {ns:2d} streams, {'rotating' if rot else 'one batch per stream':>20}, {steps:4d} steps: {us:7.2f} us per step "
                  f"({n / us / 1e3:6.2f} G/s)", flush=True)
pass


def graph_steps(ns, steps, reps=5):
    """the same schedule captured once as a hipGraph (ns branches forked from / joined into the capturing stream) and
    replayed: no host launch per step"""
    s2 = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s2.load_fleet(fleet)
    fn2 = s2.lib.mmp_place_batch_dev
    main = torch.cuda.Stream(dev)
    side = [torch.cuda.Stream(dev) for _ in range(ns)]

    def args2(b, st):
        r_, e_, o_ = bufs[b]
        return (s2.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now),
                C.c_void_p(o_.data_ptr()), C.c_void_p(st.cuda_stream))
    for i in range(2 * ns):
        fn2(*args2(i % R, side[i % ns]))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        for st in side:
            st.wait_stream(main)
        for i in range(steps):
            rc = fn2(*args2(i % R, side[i % ns]))
            assert rc == 0, rc
        for st in side:
            main.wait_stream(st)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e6)
    s2.close()
    return min(ts), float(np.median(ts))


if os.environ.get("KT_GRAPH", "1") == "1":
    for ns in (2, 4, 8):
        for steps in (20, 200, 1000):
            try:
                best, med = graph_steps(ns, steps)
                print(f"{name} {tag} hipGraph, {ns} branches, {steps:4d} steps: best {best:7.2f} median {med:7.2f} us per step "
                      f"({n / med / 1e3:6.2f} G/s)", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"{name} {tag} hipGraph, {ns} branches, {steps} steps: failed: {type(e).__name__}: {e}", flush=True)
