#!/usr/bin/env python
"""Which kernels a commit consists of, and how long each takes: N commits of one kind on a bench fleet, to be run under
`rocprofv3 --kernel-trace --stats` (tools/gpu_round.sh commit).  kind = delta (16 republished rows per commit: the insertion
re-rank) | full (a replaced table: ranks from scratch).   usage: tools/commit_breakdown.py [C3|C4] [delta|full] [n = 40]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C3"
kind = sys.argv[2] if len(sys.argv) > 2 else "delta"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
fleet = wl.make_fleet(workload)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
s.profile(True)
rng = np.random.default_rng(16)
P = fleet.n_pods
idx = np.sort(rng.choice(P, size=16, replace=False)).astype(np.int32)
span, wall = [], []
for i in range(n):
    if kind == "delta":
        rows = fleet.pods[idx].copy()
        rows["count"] += i & 1
        rows["rpm"] += 7 * (i & 1)
        s.upsert_pods(idx, rows)
    else:
        s.load_pods(fleet.pods)
    t0 = time.perf_counter()
    s.commit()
    wall.append((time.perf_counter() - t0) * 1e6)
    span.append(s.last_kernel_ms() * 1e3)
print(f"{workload} {kind}: {n} commits, {s.delta_commits()} by insertion; device span median {np.median(span[2:]):.1f} us, "
      f"commit() wall median {np.median(wall[2:]):.1f} us")
s.close()
