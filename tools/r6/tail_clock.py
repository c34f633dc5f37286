"""Where does the tail launch of a split batch (place_kernel.hpp: place_tail_kernel) spend its time?  The phase-clock build of the
library (tools/phase_clock.py build), C3, 800k request rows split at every size; per launch the s_memtime deltas of the tail's
wavefronts (rows 0 .. 4 * workgroups - 1 of g_phase; the first launch of the pair carries no markers).
usage: python tools/r6/tail_clock.py [launches]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MMP_LIB_PATH"] = os.path.join(ROOT, "modelmesh_amd", "lib", "libmmplace_phase.so")
os.environ["MMP_MEMO_FROM"] = "0"
os.environ["MMP_SPLIT_FROM"] = "0"
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fleet = wl.make_fleet("C3")
parts, ex_parts, off = [], [], 0
for k in range(8):
    rq, ex = wl.make_requests(fleet, seed=0xBE7C0 + k)
    rq = rq.copy()
    rq["extra_off"] += off
    off += len(ex)
    parts.append(rq)
    ex_parts.append(ex)
reqs, extra = np.concatenate(parts), np.concatenate(ex_parts)
n = len(reqs)
s = Solver(fleet.min_space_units, fleet.min_churn_age_ms)
s.load_fleet(fleet)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(-1)).to(dev)
d_extra = torch.from_numpy(np.ascontiguousarray(extra)).to(dev)
d_outs = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)
rd = s.lib.mmp_debug_phase_read
rd.argtypes, rd.restype = [C.c_void_p, C.c_int], C.c_int
buf = np.zeros((4096, 16), np.uint32)
for i in range(20):
    s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
assert rd(buf.ctypes.data, 1) == 0
G = int(os.environ.get("MMP_TAIL_BLOCKS", "16"))
acc = np.zeros(16)
accmax = np.zeros(16)
rows = 0
for i in range(steps):
    s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    assert rd(buf.ctypes.data, 1) == 0
    b = buf[: 4 * G].astype(np.float64)
    live = b[:, 13] > 0
    acc += b[live].sum(axis=0)
    accmax += b[live].max(axis=0)
    rows += int(live.sum())
names = {13: "13 tail: the counts read", 14: "14 tail: this pass.s entries read", 0: "0 request + model row resolved",
         1: "1 first eligible", 2: "2 best row + preference", 3: "3 break scans + count + hash", 4: "4 -", 5: "5 rpm rule", 6: "6 select + orig[]",
         8: "8 lane phase as a whole", 9: "9 __syncthreads wait", 10: "10 long phase + wave path"}
tick_ns = float(os.environ.get("MMP_TICK_NS", "0.42"))
print(f"{steps} launches x {n} decisions, {G} tail workgroups; {rows / steps:.0f} tail wavefronts with markers per launch; split batches {s.split_batches()}")
for k in (13, 14, 0, 1, 2, 3, 5, 6, 8, 9, 10):
    print(f"  {names[k]:44s} mean {acc[k] / max(rows, 1) * tick_ns:9.1f} ns   slowest wavefront {accmax[k] / steps * tick_ns:9.1f} ns")
s.close()
