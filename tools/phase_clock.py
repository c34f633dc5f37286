"""Where does a wavefront of place_batch_kernel spend its cycles?  Builds the phase-clock variant of the library
(tools/micro/phase_clock.hip: -DMMP_PHASE_CLOCK) when asked to, runs C3 batches through it and prints, per
phase, the s_memtime ticks per wavefront (on this gfx950 the counter advances at the shader clock: 18.4k ticks for a
lane phase that rocprofv3 brackets at ~7.5 us, so MMP_TICK_NS defaults to 0.42).
usage: python tools/phase_clock.py build      (here: hipcc cross-compiles)
       python tools/phase_clock.py [steps]    (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "modelmesh_amd", "lib", "libmmplace_phase.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-function",
                           os.path.join(ROOT, "tools", "micro", "phase_clock.hip"), "-o", LIB, "-ldl", "-lpthread"])
    sys.exit(0)

os.environ["MMP_LIB_PATH"] = LIB
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from modelmesh_amd import workload as wl  # noqa: E402
from modelmesh_amd.solver import Solver  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
fleet = wl.make_fleet("C3")
if os.environ.get("MMP_PHASE_FULL") == "1":  # every instance full, all caches about equally old (bench.py: full_cluster_leg): the long path
    rng = np.random.default_rng(5)
    fleet.pods["used"] = fleet.pods["capacity"] - rng.integers(0, 40_000, fleet.n_pods)
    fleet.pods["lru_time"] = fleet.now - (36_000_000 * (1 + rng.uniform(-0.04, 0.04, fleet.n_pods))).astype(np.int64)
reqs, extra = wl.make_requests(fleet, seed=0xBE7C0)
n = len(reqs)
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
for i in range(20):
    s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
assert rd(buf.ctypes.data, 1) == 0
acc = np.zeros(16)
rows = 0
for i in range(steps):   # one launch at a time: each wavefront's row holds that launch's deltas
    s.place_dev(d_reqs.data_ptr(), n, d_extra.data_ptr(), fleet.now, d_outs.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    assert rd(buf.ctypes.data, 1) == 0
    live = buf[:, 11] > 0
    acc += buf[live].sum(axis=0)
    rows += int(live.sum())
names = ["0 resolve: request -> model row, self position", "1 first eligible pod (bitmap word of the type)",
         "2 best row + preference step", "3 break scans (fresh-row rule, self rule, count threshold)",
         "4 count + audit hash", "5 rpm rule", "6 survivor select + orig[]", "7 -",
         "8 lane phase as a whole (incl. result store)", "9 __syncthreads wait", "10 long phase + wave path"]
print(f"{steps} launches x {n} decisions; {rows / steps:.0f} wavefronts per launch; decisions on the wave path per launch: "
      f"{acc[12] / steps:.1f}")
tick_ns = float(os.environ.get("MMP_TICK_NS", "0.42"))
for k, nm in enumerate(names):
    print(f"  {nm:62s} {acc[k] / rows:9.1f} ticks = {acc[k] / rows * tick_ns:9.1f} ns per wavefront")
s.close()
