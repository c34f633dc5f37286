#!/usr/bin/env python
"""Per-kernel time of the reaper's proactive plan on a bench fleet: run under `rocprofv3 --kernel-trace --stats`.
usage: tools/plan_breakdown.py [C3|C4] [n = 20]"""
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
s.load_fleet(fleet)
s.profile(True)
span, wall = [], []
for i in range(n):
    t0 = time.perf_counter()
    m, lu, info = s.proactive_plan(6400, fleet.now, 4096)
    wall.append((time.perf_counter() - t0) * 1e6)
    span.append(s.last_kernel_ms() * 1e3)
print(f"{workload}: plan of {fleet.n_models} registry rows: n_candidates {int(info['n_candidates'])}, n_selected {int(info['n_selected'])}, "
      f"total_count {int(info['total_count'])}; device span median {np.median(span[2:]):.1f} us, call wall median {np.median(wall[2:]):.1f} us")
s.close()
