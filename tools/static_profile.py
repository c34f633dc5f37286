#!/usr/bin/env python
"""Static instruction profile of one kernel by SOURCE LINE: compiles csrc/mmplace.hip for gfx950 with
-gline-tables-only, walks the kernel's assembly and counts VALU / SALU / LDS / VMEM instructions per `.loc` line.
Static counts, not executed ones — but the load-target kernel's common path is nearly straight-line, so the table shows
where a wavefront's instruction stream goes (the kernel is bound by VALU issue: DESIGN.md 9).  Needs no GPU.
usage: tools/static_profile.py [kernel-name-substring = place_batch_kernel] [top = 50]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernel = sys.argv[1] if len(sys.argv) > 1 else "place_batch_kernel"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
src = os.path.join(ROOT, "modelmesh_amd", "csrc", "mmplace.hip")
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-gline-tables-only", "-Wno-unused-function", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
files = {}
for ln in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
start = end = None
for i, ln in enumerate(lines):
    if start is None and re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", ln):
        start = i
    elif start is not None and ".end_amdhsa_kernel" in ln:
        end = i
        break
if start is None:
    sys.exit(f"no kernel matching {kernel}")
kinds = (("valu", ("v_",)), ("salu", ("s_",)), ("lds", ("ds_",)), ("vmem", ("global_", "buffer_", "flat_", "scratch_")))
per = collections.defaultdict(collections.Counter)
tot = collections.Counter()
cur = ("?", 0)
for ln in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (os.path.basename(files.get(int(m.group(1)), "?")), int(m.group(2)))
        continue
    t = ln.strip()
    for k, pre in kinds:
        if t.startswith(pre):
            per[cur][k] += 1
            tot[k] += 1
for ln in lines[end:end + 40]:
    if re.search(r"NumVgprs:|ScratchSize|Occupancy|codeLenInByte", ln):
        print(ln.strip("; "))
print("static totals:", dict(tot))
text = {}
for (f, _), _c in per.items():
    p = os.path.join(ROOT, "modelmesh_amd", "csrc", f)
    if f not in text and os.path.exists(p):
        text[f] = open(p).read().split("\n")
for (f, n), c in sorted(per.items(), key=lambda kv: -kv[1]["valu"])[:top]:
    srcline = text[f][n - 1].strip()[:100] if f in text and 0 < n <= len(text[f]) else ""
    print(f"{f}:{n:<5d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} | {srcline}")
