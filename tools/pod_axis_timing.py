"""The two pod-axis legs of bench.py alone, at one shard on one GPU (the protocol's own cost: no collective runs).
usage: python tools/pod_axis_timing.py [workload] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def fence():
    torch.cuda.synchronize(dev)


for leg in (bench.pod_axis_leg, bench.pod_axis_lib_leg):
    r = leg(wl, 0, 1, dev, steps, 5, fence)
    print(leg.__name__, json.dumps({k: r[k] for k in ("ms_per_step", "ms_per_step_async", "value", "parity_vs_oracle", "took_the_six_phase_protocol", "sharded_commit_ms") if k in r}))
