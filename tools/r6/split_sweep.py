#!/usr/bin/env python
"""Launch time of the load-target batch without the per-type shortlists, with the check in front of the lane phase (one launch) and
split into the check's own launch + a tail launch (place_kernel.hpp: place_memo_kernel / place_tail_kernel), one process:
for every environment in VARIANTS a context of its own (the switches are read at mmp_create), K launches back to back on one
stream between an event pair, rotating through 6 distinct 800k-decision buffers (requests from HBM); the results of every
variant are compared, all rows, with the first one's (MMP_NO_MEMO=1: the ordinary path).  SWEEP_TAILS=4,64: more split variants with
that many tail workgroups; SWEEP_ONLY=0,2: a subset of the variants; SWEEP_K: launches per measurement; SWEEP_WORKLOAD=C4.
usage: tools/r6/split_sweep.py [n ...]     env MEMO_SWEEP_FORM=c: the single-caller form"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modelmesh_amd import _lib  # noqa: E402
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

VARIANTS = [{"MMP_NO_MEMO": "1"}, {"MMP_MEMO_FROM": "0", "MMP_NO_SPLIT": "1"}, {"MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0"}]
if os.environ.get("SWEEP_TAILS"):
    VARIANTS += [{"MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0", "MMP_TAIL_BLOCKS": t} for t in os.environ["SWEEP_TAILS"].split(",")]
if os.environ.get("SWEEP_LDS"):
    VARIANTS += [{"MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0", "MMP_MEMO_LDS_MIN": v} for v in os.environ["SWEEP_LDS"].split(",")]
if os.environ.get("SWEEP_NOTAIL"):
    VARIANTS += [{"MMP_MEMO_FROM": "0", "MMP_SPLIT_FROM": "0", "MMP_SPLIT_NOTAIL": "1"}]
if os.environ.get("SWEEP_ONLY"):
    VARIANTS = [VARIANTS[int(i)] for i in os.environ["SWEEP_ONLY"].split(",")]
FORM_C = os.environ.get("MEMO_SWEEP_FORM") == "c"
ns = [int(x) for x in sys.argv[1:]] or [100_000, 200_000, 400_000, 800_000, 1_600_000]
fleet = wl.make_fleet(os.environ.get("SWEEP_WORKLOAD", "C3"))
dev = torch.device("cuda", 0)
nmax = max(ns)
sets = -(-nmax // fleet.n_models)
bufs, caller = [], None
for b in range(6):
    parts, ex_parts, off = [], [], 0
    for k in range(sets):
        rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + b * 31 + k)
        rq = rq.copy()
        rq["extra_off"] += off
        off += len(ex)
        parts.append(rq)
        ex_parts.append(ex)
    rq = np.concatenate(parts)
    ex = np.concatenate(ex_parts)
    if FORM_C:
        CALLER = int(os.environ.get("MEMO_SWEEP_CALLER", "17"))  # pod 17 is FULL on C3 (its shortlists are the best instance alone); 4321 has room
        row = fleet.pods[CALLER]
        rq["self_pod"], rq["flags"], rq["fresh_rpm"] = CALLER, 0, 0
        rq["fresh_lru"], rq["fresh_capacity"], rq["fresh_used"], rq["fresh_count"] = row["lru_time"], row["capacity"], row["used"], row["count"]
        caller, rq = _lib.split_caller(rq)
    bufs.append((torch.from_numpy(rq.view(np.uint8).reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(ex)).to(dev),
                 torch.zeros(nmax * 16, dtype=torch.uint8, device=dev), len(ex)))
import time
st = torch.cuda.Stream(dev)
NST = int(os.environ.get("SWEEP_STREAMS", "4"))
sts = [st] + [torch.cuda.Stream(dev) for _ in range(NST - 1)]
K = int(os.environ.get("SWEEP_K", "200"))
ref = {}
for env in VARIANTS:
    for k in ("MMP_NO_MEMO", "MMP_MEMO_FROM", "MMP_NO_SPLIT", "MMP_SPLIT_FROM", "MMP_TAIL_BLOCKS", "MMP_SPLIT_NOTAIL", "MMP_MEMO_LDS_MIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    tag = " ".join(f"{k}={v}" for k, v in env.items()) or "default"
    for n in ns:
        if FORM_C:
            cp = np.ascontiguousarray(caller, dtype=_lib.PLACE_CALLER).reshape(1)
            fn = s.lib.mmp_place_batch_c_dev
            args = [(s.h, cp.ctypes.data_as(C.c_void_p), C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int32(ne),
                     C.c_int64(fleet.now), C.c_void_p(o_.data_ptr()), C.c_void_p(st.cuda_stream)) for r_, e_, o_, ne in bufs]
        else:
            fn = s.lib.mmp_place_batch_dev
            args = [(s.h, C.c_void_p(r_.data_ptr()), C.c_int32(n), C.c_void_p(e_.data_ptr()), C.c_int64(fleet.now), C.c_void_p(o_.data_ptr()),
                     C.c_void_p(st.cuda_stream)) for r_, e_, o_, ne in bufs]
        for o in bufs:
            o[2].zero_()
        for i in range(24):
            assert fn(*args[i % 6]) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(K):
            fn(*args[i % 6])
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        # the same launches round-robin on four streams (the bench's timed region): wall time per launch
        margs = [tuple(list(a[:-1]) + [C.c_void_p(sts[i % NST].cuda_stream)]) for i, a in enumerate(args * (NST * 6 // 6 if NST % 6 == 0 else NST))]
        for i in range(24):
            fn(*margs[i % len(margs)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4 * K):
            fn(*margs[i % len(margs)])
        torch.cuda.synchronize()
        us4 = (time.perf_counter() - t0) * 1e6 / (4 * K)
        outs = torch.cat([o[2][: n * 16] for o in bufs])
        same = "reference"
        if n in ref:
            same = "identical" if torch.equal(outs, ref[n]) else f"DIFFERENT ({int((outs != ref[n]).view(-1, 16).any(1).sum())} rows)"
        else:
            ref[n] = outs.clone()
        extra_note = f"  split batches {s.split_batches()}"
        print(f"{tag:58s} n {n:9d}  {us:7.2f} us per launch  {n / us / 1e3:6.2f} G/s  hbm_only {n * (40 if FORM_C else 80) / us / 1e3 / 8000:5.3f}  "
              f"{NST} streams {us4:6.2f} us per launch {n / us4 / 1e3:6.2f} G/s  results {same}{extra_note}", flush=True)
    s.close()
