#!/usr/bin/env python
"""Anatomy of a 20-step timed region (the driver's `bench.py --steps 20 --warmup 5`): host issue time, the moment the host
KNOWS the device is done, and what torch.cuda.synchronize() costs after that — for several ways of closing the region."""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
pc = time.perf_counter


def args_of(b, st):
    r_, e_, o_ = bufs[b]
    return (s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now),
            C.c_void_p(o_.data_ptr()), C.c_void_p(st.cuda_stream))


def close(mode, sts, evs):
    """returns the time at which the host knows the device is done (before the closing synchronize)"""
    if mode == "sync":
        return pc()
    if mode == "spin":
        for e, st in zip(evs, sts):
            e.record(st)
        for e in evs:
            while not e.query():
                pass
    elif mode == "each":
        for st in sts:
            st.synchronize()
    elif mode in ("join", "joinspin"):
        for e, st in zip(evs[1:], sts[1:]):
            e.record(st)
        for e in evs[1:]:
            sts[0].wait_event(e)
        if mode == "join":
            sts[0].synchronize()
        else:
            evs[0].record(sts[0])
            while not evs[0].query():
                pass
    elif mode == "lastspin":
        # only the stream that got the last launch is polled first, then the others
        order = list(range(len(sts)))
        for i in order:
            evs[i].record(sts[i])
        for i in reversed(order):
            while not evs[i].query():
                pass
    return pc()


def region(ns, steps, mode, helpers=0, reps=21):
    sts = [torch.cuda.Stream(dev) for _ in range(ns)]
    a = [args_of(i % R, sts[i % ns]) for i in range(R * ns)]
    evs = [torch.cuda.Event() for _ in range(ns)]
    s.lib.mmp_issue_threads(s.h, helpers)
    for i in range(max(2 * ns, 50)):
        fn(*a[i % len(a)])
    s.lib.mmp_issue_flush(s.h)
    torch.cuda.synchronize()
    out = []
    for rep in range(reps):
        # like bench.py: warm-up steps, a fence, then the region
        for i in range(5):
            fn(*a[(rep * 7 + i) % len(a)])
        s.lib.mmp_issue_flush(s.h)
        close("spin", sts, evs)
        torch.cuda.synchronize()
        sched = [a[(rep * steps + i) % len(a)] for i in range(steps)]
        t0 = pc()
        for x in sched:
            fn(*x)
        s.lib.mmp_issue_flush(s.h)
        t1 = pc()
        t2 = close(mode, sts, evs)
        torch.cuda.synchronize()
        t3 = pc()
        out.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
    s.lib.mmp_issue_threads(s.h, 0)
    return np.median(np.array(out), axis=0)


def gpu_span(ns, steps, reps=11):
    """device-side span of the same region: event on the first stream before the first launch, events behind the last launches"""
    sts = [torch.cuda.Stream(dev) for _ in range(ns)]
    a = [args_of(i % R, sts[i % ns]) for i in range(R * ns)]
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(ns)]
    for i in range(50):
        fn(*a[i % len(a)])
    torch.cuda.synchronize()
    out = []
    for rep in range(reps):
        sched = [a[(rep * steps + i) % len(a)] for i in range(steps)]
        e0.record(sts[0])
        for x in sched:
            fn(*x)
        for e, st in zip(e1, sts):
            e.record(st)
        torch.cuda.synchronize()
        out.append(max(e0.elapsed_time(e) for e in e1) * 1e3)
    return float(np.median(out))


torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = pc(); torch.cuda.synchronize(); ts.append((pc() - t0) * 1e6)
print(f"synchronize on an idle device: p50 {np.median(ts):.1f} us  min {min(ts):.1f} us")
st = torch.cuda.Stream(dev)
fn(*args_of(0, st)); torch.cuda.synchronize()
ts = []
for _ in range(30):
    fn(*args_of(0, st))
    time.sleep(0.0005)  # the kernel is long done
    t0 = pc(); torch.cuda.synchronize(); ts.append((pc() - t0) * 1e6)
print(f"synchronize when one stream had a (finished) launch since the last one: p50 {np.median(ts):.1f} us")
ts = []
for _ in range(30):
    fn(*args_of(0, st))
    time.sleep(0.0005)
    t0 = pc(); st.synchronize(); t1 = pc(); torch.cuda.synchronize(); ts.append(((t1 - t0) * 1e6, (pc() - t1) * 1e6))
print(f"  stream.synchronize() then: {np.median([x[0] for x in ts]):.1f} us, device synchronize after it: {np.median([x[1] for x in ts]):.1f} us")
ev = torch.cuda.Event()
ts = []
for _ in range(30):
    fn(*args_of(0, st))
    t0 = pc(); ev.record(st); t1 = pc()
    while not ev.query():
        pass
    t2 = pc(); torch.cuda.synchronize(); ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (pc() - t2) * 1e6))
print(f"  one launch: event.record {np.median([x[0] for x in ts]):.1f} us, spin until done {np.median([x[1] for x in ts]):.1f} us, "
      f"device synchronize after it {np.median([x[2] for x in ts]):.1f} us")

for ns in (4, 8):
    print(f"device-side span of 20 steps on {ns} streams: {gpu_span(ns, 20):.1f} us; of 200: {gpu_span(ns, 200):.1f} us", flush=True)
for helpers in (0, 4):
    for ns in (4, 8):
        for mode in ("sync", "spin", "lastspin", "each", "join", "joinspin"):
            iss, known, syn, tot = region(ns, 20, mode, helpers)
            print(f"helpers {helpers} {ns} streams {mode:8s}: issue {iss:6.1f}  known-done +{known:6.1f}  synchronize +{syn:6.1f}  total {tot:6.1f} us"
                  f" = {tot / 20:5.2f} us/step", flush=True)
