#!/usr/bin/env python
"""Wall time of one load-target decision through mmp_place_batch(n = 1) with pre-marshalled arguments: the launch path
(a latency slot: one kernel launch + completion flag) against the resident kernel (MMP_RESIDENT=1); then the aggregate rate of
T host threads issuing single requests.  usage: python tools/resident_latency.py"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd._lib import PLACE_OUT, ptr  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

fleet = wl.make_fleet("C3")
reqs, _ = wl.make_requests(fleet, 0xBE7C0, extra_frac=0.0)
for mode in ("launch", "resident"):
    os.environ["MMP_RESIDENT"] = "1" if mode == "resident" else "0"
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    one, out = reqs[:1].copy(), np.zeros(1, dtype=PLACE_OUT)
    args = (s.h, ptr(one), C.c_int32(1), None, C.c_int32(0), C.c_int64(fleet.now), ptr(out))
    fn = s.lib.mmp_place_batch
    lat = []
    for i in range(6000):
        one[0] = reqs[i]
        t0 = time.perf_counter()
        fn(*args)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[500:]) * 1e6
    print(f"{mode:9s} 1 thread : p50 {np.percentile(lat, 50):6.2f} us  p99 {np.percentile(lat, 99):6.2f} us  min {lat.min():6.2f} us", flush=True)
    for T in (4, 16, 64):
        n_each = 3000
        def work(t):
            o1, ou = reqs[:1].copy(), np.zeros(1, dtype=PLACE_OUT)
            a = (s.h, ptr(o1), C.c_int32(1), None, C.c_int32(0), C.c_int64(fleet.now), ptr(ou))
            for i in range(n_each):
                o1[0] = reqs[(t * n_each + i) % len(reqs)]
                fn(*a)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{mode:9s} {T:2d} threads: {T * n_each / dt / 1e3:8.1f} k single decisions/s ({dt / n_each * 1e6:6.2f} us per call per thread)", flush=True)
    s.close()
