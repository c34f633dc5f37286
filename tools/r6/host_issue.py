"""Host time of one mmp_place_batch_dev call (800k request rows, C3): split (two launches) against one launch — back to back on four
streams as the bench issues them, and one at a time with a synchronisation behind every call (the launch path alone).
usage: python tools/r6/host_issue.py   (env: MMP_NO_SPLIT=1 ...)"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

fleet = wl.make_fleet("C3")
parts, ex_parts, off = [], [], 0
for k in range(8):
    rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + k)
    rq = rq.copy()
    rq["extra_off"] += off
    off += len(ex)
    parts.append(rq)
    ex_parts.append(ex)
reqs, extra = np.concatenate(parts), np.concatenate(ex_parts)
n = len(reqs)
dev = torch.device("cuda", 0)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
d_extra = torch.from_numpy(np.ascontiguousarray(extra)).to(dev)
NS = int(os.environ.get("NSTREAMS", "4"))
outs = [torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in range(NS)]
sts = [torch.cuda.Stream(dev) for _ in range(NS)]
fn = s.lib.mmp_place_batch_dev
args = [(s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now), C.c_void_p(outs[i].data_ptr()),
         C.c_void_p(sts[i].cuda_stream)) for i in range(NS)]
for i in range(400):
    fn(*args[i % NS])
torch.cuda.synchronize()
pc = time.perf_counter
for label, sync in (("back to back, 4 streams", False), ("a synchronisation behind every call", True)):
    ts = []
    for rep in range(10):
        for i in range(20):
            t0 = pc()
            fn(*args[i % NS])
            ts.append(pc() - t0)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    ts = np.array(ts) * 1e6
    print(f"{label:40s}: host us per call  mean {ts.mean():6.2f}  p50 {np.percentile(ts, 50):6.2f}  p90 {np.percentile(ts, 90):6.2f}  max {ts.max():7.2f}   "
          f"split batches {s.split_batches()}", flush=True)
# what closing a 20-call region costs once the device is known to be done (an event per stream polled, then the synchronize)
evs = [torch.cuda.Event() for _ in sts]
close_us, span_us = [], []
for rep in range(30):
    t0 = pc()
    for i in range(20 * int(os.environ.get("CALLS_X", "1"))):
        fn(*args[i % NS])
    for e, st in zip(evs, sts):
        e.record(st)
    for e in reversed(evs):
        while not e.query():
            pass
    t1 = pc()
    torch.cuda.synchronize()
    t2 = pc()
    span_us.append((t1 - t0) * 1e6)
    close_us.append((t2 - t1) * 1e6)
print(f"20 calls on 4 streams: known done after {np.median(span_us):.1f} us (median), torch.cuda.synchronize() behind that {np.median(close_us):.1f} us "
      f"(p90 {np.percentile(close_us, 90):.1f})")
# the whole region (20 calls issued -> torch.cuda.synchronize() returned) by the way it is closed
for label, style in (("events polled (last stream first), then synchronize", "poll"), ("stream.synchronize() per stream, then synchronize", "streams"),
                     ("synchronize only", "sync"), ("streams in reverse order, then synchronize", "rstreams")):
    tot = []
    for rep in range(30):
        t0 = pc()
        for i in range(20):
            fn(*args[i % NS])
        if style == "poll":
            for e, st in zip(evs, sts):
                e.record(st)
            for e in reversed(evs):
                while not e.query():
                    pass
        elif style == "streams":
            for st in sts:
                st.synchronize()
        elif style == "rstreams":
            for st in reversed(sts):
                st.synchronize()
        torch.cuda.synchronize()
        tot.append((pc() - t0) * 1e6)
    print(f"region of 20 calls, closed by {label:55s}: {np.median(tot):.1f} us (median)  p90 {np.percentile(tot, 90):.1f}")
# positions inside a 20-call region
ts = np.zeros((30, 20))
for rep in range(30):
    for i in range(20):
        t0 = pc()
        fn(*args[i % NS])
        ts[rep, i] = pc() - t0
    torch.cuda.synchronize()
print("per position in a 20-call region (us, median over 30 regions):", " ".join(f"{v:.1f}" for v in np.median(ts, axis=0) * 1e6))
s.close()
