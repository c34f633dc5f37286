"""How long can a decision wait for a commit?  One thread issues single load-target decisions (latency path)
while the main thread re-commits the instance table of a config (default C3: 10k instances, 100k models); prints the decisions that started and finished INSIDE a
commit and the latency percentiles of the decisions that overlapped a commit against those that did not.
usage (GPU box): python tools/commit_block.py [C2|C3|C4]   (MMP_LIB_PATH selects another build of the library)"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
mode = sys.argv[2] if len(sys.argv) > 2 else "commit"   # or "upsert": registry events (8.6k changed ModelRecords per call)
fleet = wl.make_fleet(name)
reqs, extra = wl.make_requests(fleet, 5, n=2000, extra_frac=0.0)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
spans, commits = [], []
stop = threading.Event()


def placer():
    i = 0
    while not stop.is_set():
        t0 = time.perf_counter()
        s.place(reqs[i:i + 1], None, fleet.now)
        spans.append((t0, time.perf_counter()))
        i = (i + 1) % len(reqs)


th = threading.Thread(target=placer)
th.start()
time.sleep(0.05)
if mode == "upsert":
    rng = np.random.default_rng(1)
    uidx = np.sort(rng.choice(fleet.n_models, 8600, replace=False)).astype(np.int32)
    urows = fleet.models[uidx].copy()
    k = (urows["n_loaded"] + urows["n_failed"]).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(k)])
    sel = np.concatenate([np.arange(o, o + n) for o, n in zip(fleet.models["ent_off"][uidx], k)]).astype(np.int64)
    urows["ent_off"] = offs[:-1]
    uep, uet = fleet.ent_pod[sel], fleet.ent_time[sel]
for _ in range(300):
    t0 = time.perf_counter()
    if mode == "upsert":
        s.upsert_models(uidx, urows, uep, uet)   # the same contents again: decisions keep their answers
    else:
        s.commit()
    commits.append((t0, time.perf_counter()))
    time.sleep(0.001)
stop.set()
th.join()
sp = np.array(spans)
cm = np.array(commits)
lat = (sp[:, 1] - sp[:, 0]) * 1e6
# a decision overlaps a commit if the two intervals intersect
idx = np.searchsorted(cm[:, 0], sp[:, 1])          # commits that started before the decision ended
over = np.zeros(len(sp), bool)
inside = np.zeros(len(sp), bool)
for k, (d0, d1) in enumerate(sp):
    j = idx[k] - 1
    if j >= 0 and cm[j, 1] > d0:
        over[k] = True
        inside[k] = d0 > cm[j, 0] and d1 < cm[j, 1]
print(f"{mode} ({fleet.n_pods} instances, {fleet.n_models} models): mean {np.mean(cm[:, 1] - cm[:, 0]) * 1e3:.3f} ms; {len(sp)} single decisions, "
      f"{int(over.sum())} overlapped one, {int(inside.sum())} started and finished inside one")
for name, m in ((f"overlapping a {mode}", over), (f"no {mode} running", ~over)):
    if m.any():
        print(f"  {name:22s}: p50 {np.percentile(lat[m], 50):7.1f} us   p99 {np.percentile(lat[m], 99):7.1f} us   max {lat[m].max():8.1f} us")
s.close()
