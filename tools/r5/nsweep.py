#!/usr/bin/env python
"""Launch time of the load-target kernel against the number of decisions per launch, finely around the sizes at which the
launch's workgroups fill whole rounds of the chip (6 wavefronts per SIMD x 1024 SIMDs x 64 lanes = 393 216 decisions per round).
One stream, K launches back to back between an event pair, rotating through 4 distinct buffers (requests from HBM).
usage: tools/r5/nsweep.py [n ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

MOD = int(os.environ.get("NSWEEP_MODELS_MOD", "0"))  # > 0: every request names a model below MOD (the resolved registry view shrinks to MOD rows)
ns = [int(x) for x in sys.argv[1:]] or [393_216, 400_000, 700_000, 786_432, 800_000, 1_179_648, 1_200_000, 1_572_864, 1_600_000]
fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
fn = s.lib.mmp_place_batch_dev
nmax = max(ns)
sets = -(-nmax // fleet.n_models)
bufs = []
for b in range(4):
    parts, ex_parts, off = [], [], 0
    for k in range(sets):
        rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + b * 31 + k)
        rq = rq.copy()
        if MOD:
            rq["model"] %= MOD
        rq["extra_off"] += off
        off += len(ex)
        parts.append(rq)
        ex_parts.append(ex)
    rq = np.concatenate(parts)
    ex = np.concatenate(ex_parts)
    bufs.append((torch.from_numpy(rq.view(np.uint8).reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(ex)).to(dev),
                 torch.zeros(len(rq) * 16, dtype=torch.uint8, device=dev)))
st = torch.cuda.Stream(dev)
K = 200
for n in ns:
    args = [(s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now), C.c_void_p(o_.data_ptr()),
             C.c_void_p(st.cuda_stream)) for r_, e_, o_ in bufs]
    for i in range(20):
        fn(*args[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(K):
        fn(*args[i % 4])
    e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / K
    print(f"n {n:9d}  workgroups {-(-n // 256):6d}  rounds@6/SIMD {n / 393216:5.2f}  {us:7.2f} us per launch  {us * 1e3 / n:6.3f} ns per 1000 decisions... {n / us / 1e3:6.2f} G/s")
s.close()
