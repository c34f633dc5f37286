"""Where does the tail of a single decision's latency come from while the C5 churn stream runs?  The churn leg of bench.py with
one of its library calls left out at a time (ablation), the same C prober thread issuing mmp_place_batch(n = 1) throughout.
usage (GPU box): python tools/churn_attribution.py [slices]   ->  a table: variant, calls, p50 / p99 / p99.9 / max (us)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

slices = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fleet = wl.make_fleet("C3")
VARIANTS = [("quiet (no churn calls at all)", set()),
            ("everything (the bench's churn leg)", {"pods", "models", "commit", "place", "evict"}),
            ("without the registry events (mmp_models_upsert)", {"pods", "commit", "place", "evict"}),
            ("without the commit", {"pods", "models", "place", "evict"}),
            ("without the slice's load-target batch (mmp_place_batch, 9k)", {"pods", "models", "commit", "evict"}),
            ("without the slice's eviction batch (mmp_evict_batch, 9k)", {"pods", "models", "commit", "place"}),
            ("only the registry events", {"models"}),
            ("only the commit (with the instance-record upserts)", {"pods", "commit"})]
print(f"{'variant':66s} {'calls':>8s} {'p50':>7s} {'p99':>7s} {'p99.9':>7s} {'max':>8s}   (us; C3, {slices} slices of 20k events)")
for name, on in VARIANTS:
    s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
    s.load_fleet(fleet)
    cs = wl.ChurnStream(fleet, 0xC5)
    s.load_caches(cs.seg_off, cs.cache_lu, cs.cache_wt, cs.cache_cap)
    single, _ = wl.make_requests(fleet, seed=0xC51, n=256)
    single = np.ascontiguousarray(single)
    single["n_extra"] = 0
    single["extra_off"] = 0
    prober = bench._single_prober()
    for it in range(slices + 1):
        f = cs.fleet
        ev = cs.model_events() if it else None
        if it:
            if "pods" in on:
                s.upsert_pods(cs.changed_pods, f.pods[cs.changed_pods])
            if "models" in on:
                s.upsert_models(*ev)
            if "commit" in on:
                s.commit()
        sl = cs.next_slice()
        if "place" in on or not it:
            got = s.place(sl["place_reqs"], sl["extra"], f.now)
        if "evict" in on:
            s.evict(sl["evict_reqs"], f.now)
        if not it:
            prober.prober_start(C.cast(s.lib.mmp_place_batch, C.c_void_p), s.h, single.ctypes.data_as(C.c_void_p), len(single),
                                C.c_int64(int(fleet.now)), C.c_int64(4_000_000))
        if not on:
            time.sleep(0.05)
        cs.apply(sl, got)
    buf = np.zeros(4_000_000, np.uint32)
    n = int(prober.prober_stop(buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf))))
    lat = buf[100:n].astype(np.float64) / 1e3
    print(f"{name:66s} {len(lat):8d} {np.percentile(lat, 50):7.1f} {np.percentile(lat, 99):7.1f} {np.percentile(lat, 99.9):7.1f} {lat.max():8.1f}")
    s.close()
