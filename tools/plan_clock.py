#!/usr/bin/env python
"""Where do the microseconds of the one-launch reaper plan go?  The phase-clock build of the library (tools/phase_clock.py build)
stores the 100 MHz clock at every phase boundary of proactive_plan_fused_kernel (workgroup 0); this prints the medians.
usage: python tools/plan_clock.py [C3|C4] [n = 40]      (GPU box)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMP_LIB_PATH"] = os.path.join(ROOT, "modelmesh_amd", "lib", "libmmplace_phase.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fleet = wl.make_fleet(workload)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
s.profile(True)
rd = s.lib.mmp_debug_plan_clock
rd.argtypes, rd.restype = [C.c_void_p, C.c_void_p], C.c_int
rows, span = [], []
for i in range(n):
    s.proactive_plan(6400, fleet.now, 4096)
    span.append(s.last_kernel_ms() * 1e3)
    t = np.zeros(16, np.int64)
    assert rd(s.h, t.ctypes.data) == 0
    rows.append(t[:16] - t[0])
med = np.median(np.array(rows[3:]), axis=0) / 100.0
names = ["A: space budget, candidate counts and key ranges", "barrier 1", "B: scalars, fold, histogram (slots)", "barrier 2 + bucket scan (every workgroup)",
         "C: pairs to offset + slot", "barrier 3", "D: ranks inside the chunks, chunk totals", "-", "E: wait for the chunks in front, emit"]
print(f"{workload}: one-launch plan, device span median {np.median(span[3:]):.1f} us (events around the launch); workgroup 0's clock:")
for k, nm in enumerate(names):
    print(f"  {nm:52s} {med[k + 1] - med[k]:7.2f} us")
print(f"  {'first to last marker':52s} {med[9]:7.2f} us")
print(f"  inside A (workgroup 0, lane 0): instance rows added up at {med[13]:.2f} us, registry rows taken at {med[14]:.2f} us, workgroup total at {med[15]:.2f} us")
print(f"  the workgroup of the LAST chunk is done {med[12]:.2f} us after workgroup 0's first marker")
print(f"  of barrier 2 + scan: the scan itself {med[11] - med[10]:.2f} us")
s.close()
