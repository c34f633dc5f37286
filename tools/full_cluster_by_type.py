import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from modelmesh_amd import workload as wl
from modelmesh_amd._lib import PLACE_OUT
from modelmesh_amd.solver import Solver
fleet = wl.make_fleet("C3")
rng = np.random.default_rng(5)
P = fleet.n_pods
fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 40_000, P)
fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-0.04, 0.04, P))).astype(np.int64)
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
t = fleet.models["type"][reqs["model"]]
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
for k in [0, 1, 2, 3, -1]:
    sel = reqs if k < 0 else np.ascontiguousarray(np.tile(reqs[t == k], 12)[:100000])
    n = len(sel)
    d_reqs = torch.from_numpy(sel.view(np.uint8).reshape(-1)).to(dev)
    d_extra = torch.from_numpy(np.ascontiguousarray(extra)).to(dev)
    d_outs = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    args = (s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now), C.c_void_p(d_outs.data_ptr()), C.c_void_p(st.cuda_stream))
    fn = s.lib.mmp_place_batch_dev
    for _ in range(5): fn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): fn(*args)
    torch.cuda.synchronize()
    print("type", k, "n", n, f"{(time.perf_counter()-t0)/30*1e6:.1f} us per launch")
s.close()
