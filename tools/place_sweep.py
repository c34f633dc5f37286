#!/usr/bin/env python
"""place_batch_kernel throughput against batch size and stream count (device-resident requests).
usage: tools/place_sweep.py [workload]"""
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
fleet = wl.make_fleet(name)
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
d_extra = torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev)
fn = s.lib.mmp_place_batch_dev
for mult in (1, 4, 10, 40):
    big = np.tile(reqs, mult)
    n = len(big)
    d_reqs = torch.from_numpy(big.view(np.uint8).reshape(-1)).to(dev)
    for ns in (1, 2, 8):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        outs = [torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in range(ns)]
        args = [(s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now),
                 C.c_void_p(o.data_ptr()), C.c_void_p(st.cuda_stream)) for st, o in zip(streams, outs)]
        steps = max(20, 400 // mult)
        for i in range(10):
            fn(*args[i % ns])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(*args[i % ns])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} batch {n:>8} streams {ns}: {dt / steps * 1e6:8.2f} us/step  {n * steps / dt / 1e9:6.2f} G decisions/s", flush=True)
s.close()
