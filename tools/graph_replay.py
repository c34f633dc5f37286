#!/usr/bin/env python
"""Can a host capture its mmp_place_batch_dev launches into a HIP graph and replay them?  k launches of one request set each on a
capturing stream (torch.cuda.CUDAGraph), replayed; results against the eager launches; wall time per replay against k eager
launches and against one mmp_place_multi_dev launch.   usage: tools/graph_replay.py [k = 8]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
fleet = wl.make_fleet("C3")
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
sets = [wl.make_requests(fleet, seed=0x3A00 + i) for i in range(k)]
n = len(sets[0][0])
d_reqs = [torch.from_numpy(r.view(np.uint8).reshape(-1)).to(dev) for r, _ in sets]
d_extra = [torch.from_numpy(np.ascontiguousarray(x if len(x) else np.zeros(1, np.int32))).to(dev) for _, x in sets]
d_a = [torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in range(k)]
d_b = [torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in range(k)]
st = torch.cuda.Stream(dev)


def eager(outs, stream):
    for i in range(k):
        s.place_dev(d_reqs[i].data_ptr(), n, d_extra[i].data_ptr(), fleet.now, outs[i].data_ptr(), stream.cuda_stream)


eager(d_a, st)  # (first launches: LDS grants, module load — not inside a capture)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=st):
        eager(d_b, st)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, e)
    sys.exit(0)
for o in d_b:
    o.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay results identical to eager:", all(bool((a == b).all().item()) for a, b in zip(d_a, d_b)))
for name, fn in (("eager, %d launches" % k, lambda: eager(d_a, st)), ("graph replay", g.replay),
                 ("mmp_place_multi_dev, one launch", lambda: s.place_multi_dev([t.data_ptr() for t in d_reqs], [n] * k,
                                                                                [t.data_ptr() for t in d_extra], fleet.now,
                                                                                [t.data_ptr() for t in d_a], st.cuda_stream))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per {k} x {n} decisions")
s.close()
