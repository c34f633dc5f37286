#!/usr/bin/env python
"""Where do the ~80 us go that a 20-step timed region costs beyond 20 x the steady-state step?  Splits a region into
host issue time and the wait for completion, for several stream counts, with torch.cuda.synchronize() and with a
spin on per-stream event queries in front of it."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
R = 48
bufs = []
for b in range(R):
    rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + b)
    bufs.append((torch.from_numpy(rq.view(np.uint8).reshape(-1)).to(dev),
                 torch.from_numpy(np.ascontiguousarray(ex if len(ex) else np.zeros(1, np.int32))).to(dev),
                 torch.zeros(len(rq) * 16, dtype=torch.uint8, device=dev)))
n = len(rq)
fn = s.lib.mmp_place_batch_dev


def args_of(b, st):
    r_, e_, o_ = bufs[b]
    return (s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now),
            C.c_void_p(o_.data_ptr()), C.c_void_p(st.cuda_stream))


def region(ns, steps, mode, reps=15):
    sts = [torch.cuda.Stream(dev) for _ in range(ns)]
    a = [args_of(i % R, sts[i % ns]) for i in range(R * ns)]
    evs = [torch.cuda.Event() for _ in range(ns)]
    for i in range(max(2 * ns, 50)):
        fn(*a[i % len(a)])
    torch.cuda.synchronize()
    out = []
    for rep in range(reps):
        sched = [a[(rep * steps + i) % len(a)] for i in range(steps)]
        t0 = time.perf_counter()
        for x in sched:
            fn(*x)
        t1 = time.perf_counter()
        if mode == "spin":
            for e, st in zip(evs, sts):
                e.record(st)
            for e in evs:
                while not e.query():
                    pass
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6))
        time.sleep(0.002)
    o = np.median(np.array(out), axis=0)
    return o


t0 = time.perf_counter(); torch.cuda.synchronize(); print(f"empty synchronize: {(time.perf_counter() - t0) * 1e6:.1f} us")
for mode in ("sync", "spin"):
    for ns in (1, 2, 4, 8, 16):
        for steps in (20, 200):
            iss, wait, tot = region(ns, steps, mode)
            print(f"{mode:4s} {ns:2d} streams {steps:4d} steps: issue {iss:7.1f} us, wait {wait:7.1f} us, total {tot:7.1f} us = {tot / steps:6.2f} us/step", flush=True)
