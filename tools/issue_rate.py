"""Is the timed region of bench.py bound by the host's launch rate or by the GPU?  Issues the same
100k-decision C3 batches (mmp_place_batch_dev, one stream per issuer) from 1..T host threads and reports the
time the issue loop itself takes next to the time until the GPU is idle.
usage (GPU box): python tools/issue_rate.py [steps]"""
import ctypes as C
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
fleet = wl.make_fleet("C3")
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
n = len(reqs)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
d_extra = torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev)
fn = s.lib.mmp_place_batch_dev


def run(threads, streams_per_thread):
    streams = [[torch.cuda.Stream(dev) for _ in range(streams_per_thread)] for _ in range(threads)]
    outs = [[torch.zeros(n * 16, dtype=torch.uint8, device=dev) for _ in range(streams_per_thread)] for _ in range(threads)]
    args = [[(s.h, C.c_void_p(d_reqs.data_ptr()), C.c_int32(n), C.c_void_p(d_extra.data_ptr()), C.c_int64(fleet.now),
              C.c_void_p(o.data_ptr()), C.c_void_p(st.cuda_stream)) for st, o in zip(ss, oo)] for ss, oo in zip(streams, outs)]
    per = steps // threads
    issue_t = [0.0] * threads
    go = threading.Barrier(threads + 1)

    def work(t):
        sched = [args[t][i % streams_per_thread] for i in range(per)]
        go.wait()
        t0 = time.perf_counter()
        rc = 0
        for a in sched:
            rc |= fn(*a)
        issue_t[t] = time.perf_counter() - t0
        assert rc == 0

    for w in range(2):  # warm, then measured
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for x in th:
            x.start()
        torch.cuda.synchronize(dev)
        go.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        t_all = time.perf_counter() - t0
        go.reset()
    tot = per * threads
    print(f"threads {threads:2d} x streams {streams_per_thread:2d}: issue {t_issue / tot * 1e6:6.2f} us/step, "
          f"until idle {t_all / tot * 1e6:6.2f} us/step = {n * tot / t_all / 1e9:6.2f} G decisions/s")


for th, sp in [(1, 1), (1, 8), (2, 4), (4, 2), (4, 4), (8, 2), (8, 4), (16, 2)]:
    run(th, sp)
s.close()
