import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms); s.load_fleet(fleet)
cs = wl.ChurnStream(fleet, 0xC5)
sl = cs.next_slice(); got = s.place(sl["place_reqs"], sl["extra"], cs.fleet.now); cs.apply(sl, got)
ev = cs.model_events(); f = cs.fleet
print("changed models", len(ev[0]), "entries", len(ev[2]))
s.profile(True)
for rep in range(5):
    t0 = time.perf_counter(); s.upsert_models(*ev); t1 = time.perf_counter()
    k = s.last_kernel_ms()
    s.load_models(f.models, f.ent_pod, f.ent_time); t2 = time.perf_counter()
    lib = s.lib
    print(f"upsert {1e3*(t1-t0):.3f} ms (kernel {k:.4f})  full reload {1e3*(t2-t1):.3f} ms")
s.close()
