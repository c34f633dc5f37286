#!/usr/bin/env python
"""Per-kernel time of the in-library sharded commit at one shard (run under rocprofv3 --kernel-trace --stats).
usage: tools/shard_commit_breakdown.py [C3|C4] [n = 20]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fleet = wl.make_fleet(workload)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.shard_group_init(None, 0, 1)
s.load_fleet(fleet, commit=False)
wall = []
for i in range(n):
    t0 = time.perf_counter()
    s.shard_commit()
    wall.append((time.perf_counter() - t0) * 1e6)
print(f"{workload}: mmp_shard_commit at one shard: median {np.median(wall[2:]):.1f} us per call")
s.close()
