#!/usr/bin/env python
"""Register / scratch / LDS / occupancy figures of every kernel of libmmplace as the compiler reports them
(-Rpass-analysis=kernel-resource-usage, device-only compile of csrc/mmplace.hip for gfx950).  Needs no GPU.
usage: tools/kernel_resources.py [name-substring ...]   (markdown table on stdout)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "modelmesh_amd", "csrc", "mmplace.hip")
want = sys.argv[1:]
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "-Wno-unused-function",
                    "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"], capture_output=True, text=True)
rows, cur = [], None
for ln in p.stderr.split("\n"):
    m = re.search(r"remark: .*Function Name: (\S+)", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("mmp::", "")}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
cols = ["VGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"]
print("| kernel | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
for r in sorted(rows, key=lambda r: r["name"]):
    if want and not any(w in r["name"] for w in want):
        continue
    print(f"| `{r['name']}` | " + " | ".join(str(r.get(c, "")) for c in cols) + " |")
