"""When do the wavefronts of ONE launch start and end?  (phase-clock build: slots 14 / 15 of a wavefront's row hold the clock at the top
and at the end of place_block.)  Prints, for a 100k launch: the span first start -> last end, the spread of the starts (how long
the dispatcher takes to put the grid on the chip), per-wavefront lifetimes — against the launch's duration by an event pair.
usage: python tools/phase_clock.py build; [MMP_PHASE_FULL=1] python tools/r6/wave_timeline.py [n]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MMP_LIB_PATH"] = os.path.join(ROOT, "modelmesh_amd", "lib", "libmmplace_phase.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tick = float(os.environ.get("MMP_TICK_NS", "10"))
fleet = wl.make_fleet("C3")
if os.environ.get("MMP_PHASE_FULL") == "1":
    wl.make_full_cluster(fleet)
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
reqs = reqs[:n]
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
d_extra = torch.from_numpy(np.ascontiguousarray(extra if len(extra) else np.zeros(1, np.int32))).to(dev)
d_outs = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)
rd = s.lib.mmp_debug_phase_read
rd.argtypes, rd.restype = [C.c_void_p, C.c_int], C.c_int
buf = np.zeros((4096, 16), np.uint32)
with torch.cuda.stream(st):
    for i in range(50):
        s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(200):
        s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
    e1.record(st)
torch.cuda.synchronize()
print(f"{n} decisions per launch, back to back on one stream: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
spans, spreads, lives, lane, why_n, why_or, slow = [], [], [], [], [], {}, []
for rep in range(30):
    assert rd(buf.ctypes.data, 1) == 0
    s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    assert rd(buf.ctypes.data, 0) == 0
    live = buf[:, 11] > 0
    t0 = buf[live, 14].astype(np.int64)
    t1 = buf[live, 15].astype(np.int64)
    base = t0.min()
    t0 = (t0 - base) % (1 << 32)
    t1 = (t1 - base) % (1 << 32)
    spans.append((t1.max()) * tick)
    spreads.append(np.percentile(t0, [50, 90, 100]) * tick)
    lives.append(np.percentile(t1 - t0, [50, 90, 99, 100]) * tick)
    lane.append(buf[live, 8].mean() * 0.42)
    why_n.append(int(buf[live, 7].sum()))
    for v in buf[live, 9][buf[live, 7] > 0]:
        why_or[int(v)] = why_or.get(int(v), 0) + 1
    lt = (t1 - t0) * tick
    slow.append((int((lt > 9500).sum()), int(((lt > 9500) & (buf[live, 7] > 0)).sum())))
sp = np.array(spreads)
lv = np.array(lives)
print(f"wavefronts per launch {int(live.sum())}; first start -> last end: median {np.median(spans) / 1e3:.2f} us")
print(f"starts after the first one: p50 {np.median(sp[:, 0]) / 1e3:.2f}  p90 {np.median(sp[:, 1]) / 1e3:.2f}  last {np.median(sp[:, 2]) / 1e3:.2f} us")
print(f"lifetime of a wavefront (top of place_block -> its end): p50 {np.median(lv[:, 0]) / 1e3:.2f}  p90 {np.median(lv[:, 1]) / 1e3:.2f}  p99 {np.median(lv[:, 2]) / 1e3:.2f}  max {np.median(lv[:, 3]) / 1e3:.2f} us"
      f"   (lane phase, mean: {np.mean(lane) / 1e3:.2f} us)")
print(f"requests left to the walk per launch (long_memo_try): {np.mean(why_n):.1f}; wavefronts by the reasons or-ed together (1 shape, 2 no record, 4 caller is best and not full, "
      f"8 a steering position, 16 the caller's own break): {dict(sorted(why_or.items()))} over 30 launches")
if os.environ.get("WHY_DUMP"):
    for row in np.nonzero(live & (buf[:, 7] > 0))[0]:
        m = int(buf[row, 13])
        mr = fleet.models[m]
        pos = {int(p): i for i, p in enumerate(s.order())}
        print("  left to the walk: model", m, "type", int(mr["type"]), "loaded", int(mr["n_loaded"]), "failed", int(mr["n_failed"]),
              "requests of it:", [(int(q["self_pod"]), pos.get(int(q["self_pod"])), int(q["n_extra"]), int(q["flags"])) for q in reqs[reqs["model"] == m]][:4],
              "entries at positions", [pos.get(int(e)) for e in fleet.ent_pod[int(mr["ent_off"]):int(mr["ent_off"]) + int(mr["n_loaded"]) + int(mr["n_failed"])]] if hasattr(fleet, "ent_pod") else "")
print(f"wavefronts that live longer than 9.5 us per launch: {np.mean([a for a, b in slow]):.1f}, of them with a request left to the walk: {np.mean([b for a, b in slow]):.1f}")
s.close()
