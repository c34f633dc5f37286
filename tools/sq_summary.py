#!/usr/bin/env python
"""Median per-launch value of every counter rocprofv3 --pmc collected for one kernel.
usage: tools/sq_summary.py <dir> <kernel-substring> [min_grid [max_grid]]"""
import collections
import csv
import glob
import json
import sys

d, kernel = sys.argv[1:3]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # only launches of at least this many work-items (a bench run mixes sizes)
max_grid = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 62
vals = collections.defaultdict(list)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and min_grid <= int(float(r.get("Grid_Size", 0) or 0)) <= max_grid:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
out["launches"] = max((len(v) for v in vals.values()), default=0)
out["min_grid"] = min_grid
if max_grid < 1 << 62:
    out["max_grid"] = max_grid
print(json.dumps(out))
