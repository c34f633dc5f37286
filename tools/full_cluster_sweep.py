#!/usr/bin/env python
"""place_batch_kernel on a FULL cluster (every instance below minSpaceUnits: the LRU-window mode of getNext,
MM.java:4911-4917), where shortlists can span the whole table.  usage: tools/full_cluster_sweep.py [frac_full]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd._lib import PLACE_OUT  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402
from oracle.bind import OracleFleet  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # > 0: every cache is about equally old (age 10 h +- spread)
fleet = wl.make_fleet("C3")
rng = np.random.default_rng(5)
full = rng.random(fleet.n_pods) < frac
fleet.pods["used"] = np.where(full, fleet.pods["capacity"] - rng.integers(0, 40_000, fleet.n_pods), fleet.pods["used"])
if spread > 0:  # the steady state under global LRU eviction: all caches have about the same age
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-spread, spread, fleet.n_pods))).astype(np.int64)
if os.environ.get("NO_PREF") == "1":
    fleet.has_prefer[:] = 0  # no preferred instances: no case (b)
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
n = len(reqs)
d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
d_extra = torch.from_numpy(np.ascontiguousarray(extra)).to(dev)
d_outs = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)
args = (s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now),
        C.c_void_p(d_outs.data_ptr()), C.c_void_p(st.cuda_stream))
fn = s.lib.mmp_place_batch_dev
for _ in range(5):
    fn(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    fn(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
got = np.frombuffer(d_outs.cpu().numpy().tobytes(), dtype=PLACE_OUT)
want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=os.cpu_count())
ok = all(np.array_equal(got[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))
if not ok:
    for f in ("chosen", "best", "n_candidates", "hash"):
        bad = np.nonzero(got[f] != want[f])[0]
        print(f, len(bad), bad[:5], got[f][bad[:5]], want[f][bad[:5]])
print(f"full fraction {frac} spread {spread}: {dt * 1e6:.1f} us per 100k decisions, mean shortlist {got['n_candidates'].mean():.1f}, "
      f"max {got['n_candidates'].max()}, parity {ok}")
s.close()
