#!/usr/bin/env python
"""place_batch on C3 (100k decisions x 10k instances) when a share of the models carries a type that only a few
instances may host (the candidate filter of MM.java:4889-4897): their first eligible instance lies anywhere in the
placement order.  usage: tools/sparse_type_sweep.py [share_of_models] [instances_per_sparse_type]
Run once with MMP_LONG_MODE unset (commit picks the long variant: prefix-table jumps) and once with MMP_LONG_MODE=0
(lean variant: such decisions fall to the wave path)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd._lib import PLACE_OUT  # noqa: E402
from modelmesh_amd.solver import Solver, bitmap_from_bool  # noqa: E402
from oracle.bind import OracleFleet  # noqa: E402

share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fleet = wl.make_fleet("C3")
rng = np.random.default_rng(11)
P, T0 = fleet.n_pods, fleet.n_types
T = T0 + 4  # four sparse types on top of the workload's own
al = np.ones((T, P), bool)
pf = np.zeros((T, P), bool)
if T0:
    al[:T0] = np.unpackbits(fleet.allowed.view(np.uint8), bitorder="little").reshape(T0, -1)[:, :P].astype(bool)
    pf[:T0] = np.unpackbits(fleet.prefer.view(np.uint8), bitorder="little").reshape(T0, -1)[:, :P].astype(bool)
for t in range(T0, T):
    al[t] = False
    al[t, rng.choice(P, k, replace=False)] = True
fleet.allowed, fleet.prefer = bitmap_from_bool(al), bitmap_from_bool(pf)
fleet.has_allowed = np.concatenate([fleet.has_allowed if T0 else np.zeros(0, np.uint8), np.ones(4, np.uint8)])
fleet.has_prefer = np.concatenate([fleet.has_prefer if T0 else np.zeros(0, np.uint8), np.zeros(4, np.uint8)])
fleet.n_types = T
sparse = rng.random(fleet.n_models) < share
fleet.models["type"] = np.where(sparse, rng.integers(T0, T, fleet.n_models), fleet.models["type"])
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
# the same batch on 8 streams round-robin (bench.py's timed region): what a host with independent batches sees
sts = [torch.cuda.Stream(dev) for _ in range(8)]
outs8 = [torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in sts]
args8 = [(s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now),
          C.c_void_p(o.data_ptr()), C.c_void_p(q.cuda_stream)) for o, q in zip(outs8, sts)]
for a in args8:
    fn(*a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(400):
    fn(*args8[i & 7])
torch.cuda.synchronize()
dt8 = (time.perf_counter() - t0) / 400
got = np.frombuffer(d_outs.cpu().numpy().tobytes(), dtype=PLACE_OUT)
want = OracleFleet(fleet).place(reqs, extra, fleet.now, threads=os.cpu_count())
ok = all(np.array_equal(got[f], want[f]) for f in ("chosen", "best", "n_candidates", "hash"))
print(f"share {share} k {k} MMP_LONG_MODE={os.environ.get('MMP_LONG_MODE', 'auto')}: {dt * 1e6:.1f} us per 100k decisions ({dt8 * 1e6:.1f} on 8 streams), "
      f"requests of a sparse type {sparse[reqs['model']].mean():.3f}, parity {ok}")
s.close()
