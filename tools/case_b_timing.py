"""Is case (b) decided from the whole-window tables?  The fleet of tests/test_place_parity_gpu.py::test_case_b_... (every request
of the preferring type takes case (b)), one 30k-decision launch, with the tables and with MMP_NO_CASEB=1 (the wave path).
usage (GPU box): python tools/case_b_timing.py"""
import ctypes as C
import os
import subprocess
import sys
import time

if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import torch
    from modelmesh_amd import workload as wl
    from modelmesh_amd.solver import Solver, bitmap_from_bool
    from oracle.bind import OracleFleet, unpack_bitmap
    rng = np.random.default_rng(7700)
    fleet = wl.make_fleet("C3", models=30_000, pods=3_000)
    P = fleet.n_pods
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 40_000, P)
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-0.04, 0.04, P))).astype(np.int64)
    orc = OracleFleet(fleet)
    pf = unpack_bitmap(fleet.prefer, P).astype(bool)
    pf[:, orc.order[:3]] = False
    fleet.prefer = bitmap_from_bool(pf)
    fleet.models["type"] = np.where(rng.random(fleet.n_models) < 0.5, 2, fleet.models["type"])
    reqs, extra = wl.make_requests(fleet, 40)
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    dev = torch.device("cuda", 0)
    n = len(reqs)
    d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
    d_extra = torch.from_numpy(np.ascontiguousarray(extra)).to(dev)
    d_outs = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev)
    args = (s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now), C.c_void_p(d_outs.data_ptr()),
            C.c_void_p(st.cuda_stream))
    for _ in range(3):
        s.lib.mmp_place_batch_dev(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        s.lib.mmp_place_batch_dev(*args)
    torch.cuda.synchronize()
    print(f"{sys.argv[1]}: {(time.perf_counter() - t0) / 10 * 1e6:.1f} us per launch of {n} decisions (half of them case (b))")
    s.close()
else:
    for label, env in (("whole-window tables", {}), ("MMP_NO_CASEB=1 (wave path)", {"MMP_NO_CASEB": "1"})):
        subprocess.run([sys.executable, __file__, label], env=dict(os.environ, **env))
