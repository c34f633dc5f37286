import ctypes as C, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from modelmesh_amd import workload as wl
from modelmesh_amd.solver import Solver
from modelmesh_amd._lib import PLACE_OUT
fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms); s.load_fleet(fleet)
cs = wl.ChurnStream(fleet, 0xC5)
s.load_caches(cs.seg_off, cs.cache_lu, cs.cache_wt, cs.cache_cap)
single, _ = wl.make_requests(fleet, seed=0xC51, n=256); single = np.ascontiguousarray(single); single["n_extra"] = 0; single["extra_off"] = 0
fn = s.lib.mmp_place_batch
def per_request(tag):
    out = np.zeros(1, dtype=PLACE_OUT)
    lat = np.zeros((256, 30))
    outs = np.zeros(256, dtype=PLACE_OUT)
    for r in range(30):
        for i in range(256):
            p = single[i:i+1]
            t0 = time.perf_counter_ns()
            fn(s.h, p.ctypes.data_as(C.c_void_p), 1, None, 0, C.c_int64(int(fleet.now)), out.ctypes.data_as(C.c_void_p))
            lat[i, r] = (time.perf_counter_ns() - t0) / 1e3
            outs[i] = out[0]
    med = np.median(lat, axis=1)
    print(tag, "per-request median latency us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (np.percentile(med, 50), np.percentile(med, 90), np.percentile(med, 99), med.max()))
    slow = np.argsort(med)[-8:]
    print("   slowest:", [(int(i), round(float(med[i]), 1), int(outs[i]["n_candidates"]), int(outs[i]["chosen"])) for i in slow])
    print("   corr(n_candidates, latency) =", np.corrcoef(outs["n_candidates"], med)[0, 1], " mean n_cand", outs["n_candidates"].mean())
per_request("before churn")
for it in range(9):
    f = cs.fleet
    ev = cs.model_events() if it else None
    if it:
        s.upsert_pods(cs.changed_pods, f.pods[cs.changed_pods]); s.upsert_models(*ev); s.commit()
    sl = cs.next_slice()
    got = s.place(sl["place_reqs"], sl["extra"], f.now)
    s.evict(sl["evict_reqs"], f.now)
    cs.apply(sl, got)
per_request("after 8 slices of churn (quiet now)")
