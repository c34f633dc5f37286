#!/usr/bin/env python
"""Ingest the C3 instance table and an M-model registry from their KV wire format, print the library's
kernel-time bracket; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
usage: tools/ingest_prof.py [models]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import wire, workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.Generator(np.random.PCG64(0x5EC0))
f = wl.make_fleet("C3", models=M)
P, now = f.n_pods, f.now
ids = wire.make_ids(rng, P)
wire.adopt_ids(f, ids)
names = ["NLCLASSIFIER"] + ["type-%d" % t for t in range(1, max(f.n_types, 1))]
t0 = time.time()
pv = wire.pod_values(f, rng, np.full(P, now - 1000, np.int64))
mv = wire.model_values(f, ids, names, rng, np.zeros(M, np.int64))
print("generated in %.1fs: pods %d B, models %d B" % (time.time() - t0, sum(map(len, pv)), sum(map(len, mv))))
s = Solver(f.min_space_units, f.min_churn_age_ms)
s.load_pod_ids(ids)
s.load_type_names(names, 0)
s.profile(True)
live = np.ones(P, np.uint8)
for rep in range(4):
    t0 = time.perf_counter()
    s.ingest_pods_json(pv, np.arange(P, dtype=np.int32), live)
    t1 = time.perf_counter()
    kp = s.last_kernel_ms()
    st, _ = s.ingest_models_json(mv)
    t2 = time.perf_counter()
    km = s.last_kernel_ms()
    print("rep %d pods: kernel %.3f ms wall %.2f ms | models: kernel span %.3f ms wall %.2f ms bad %d" %
          (rep, kp, (t1 - t0) * 1e3, km, (t2 - t1) * 1e3, int(st.sum())))
s.close()
