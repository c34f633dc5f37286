#!/usr/bin/env python
"""place_batch_kernel against batch size and stream count, device-resident requests, as CSV (stdout).

Every cell rotates through enough DISTINCT request batches that requests + results exceed the 256 MiB Infinity Cache
(>= 320 MB; at least 3 batches), so a launch reads its requests from HBM.  1 stream: K launches back to back between a
HIP event pair on that stream (us = average launch duration); > 1 streams: wall time of K steps issued round-robin
(us = step time).  frac = compulsory bytes (request 64 B + resolved model row + exclusions + result 16 B) / time / 8 TB/s.
usage: tools/place_sweep.py [workload] > profiles/rNN/place_sweep_<workload>.csv"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
fleet = wl.make_fleet(name)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
fn = s.lib.mmp_place_batch_dev
base = [wl.make_requests(fleet, seed=0xBE7C0 + b) for b in range(8)]
bytes_per_decision = bench.kernel_bytes(fleet, base[0][0]) / len(base[0][0])
print("workload,decisions_per_launch,streams,distinct_batches,us_per_launch_or_step,G_decisions_per_s,GBs,frac_of_8TBs")
M = fleet.n_models
for n in (25_000, 50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000):
    n_b = max(3, -(-320_000_000 // (n * 80)))
    bufs = []
    for b in range(n_b):
        parts, ex_parts, off = [], [], 0
        need = n
        k = b
        while need > 0:  # a batch of n decisions = slices / repeats of the seeded 100k-request batches
            rq, ex = base[k % len(base)]
            take = min(need, len(rq))
            sub = rq[:take].copy()
            used = int(sub["n_extra"].sum())
            sub["extra_off"] = np.concatenate([[0], np.cumsum(sub["n_extra"])[:-1]]) + off
            parts.append(sub)
            ex_parts.append(ex[: used])
            off += used
            need -= take
            k += 1
        rq = np.concatenate(parts)
        ex = np.concatenate(ex_parts) if off else np.zeros(1, np.int32)
        bufs.append((torch.from_numpy(rq.view(np.uint8).reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(ex)).to(dev),
                     torch.zeros(n * 16, dtype=torch.uint8, device=dev)))
    for ns in tuple(int(x) for x in os.environ.get('SWEEP_STREAMS', '1,2,4,8').split(',')):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        period = n_b * ns
        args = []
        for i in range(period):
            r_, e_, o_ = bufs[i % n_b]
            args.append((s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now),
                         C.c_void_p(o_.data_ptr()), C.c_void_p(streams[i % ns].cuda_stream)))
        steps = int(max(30, min(1000, 40_000_000 // n)))
        for i in range(max(2 * ns, n_b)):
            fn(*args[i % period])
        torch.cuda.synchronize()
        if ns == 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[0])
            for i in range(steps):
                fn(*args[i % period])
            e1.record(streams[0])
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / steps * 1e3
        else:
            sched = [args[i % period] for i in range(steps)]
            t0 = time.perf_counter()
            for a in sched:
                fn(*a)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
        gbs = n * bytes_per_decision / us / 1e3
        print(f"{name},{n},{ns},{n_b},{us:.3f},{n / us / 1e3:.3f},{gbs:.1f},{gbs / 8000:.4f}", flush=True)
    del bufs
    torch.cuda.empty_cache()
s.close()
