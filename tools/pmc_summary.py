#!/usr/bin/env python
"""Digest rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into a per-launch HBM traffic figure.

gfx950 corrections applied (same guide): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of
1024 B; FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly half of a wide coalesced
stream, so it is doubled ("fetch_bytes_corrected").  Both raw and corrected values are kept.

usage: tools/pmc_summary.py <fetch_dir> <write_dir> <kernel-substring> <out.json> [min_value [min_grid max_grid]]
(min_grid / max_grid: only launches of that many work-items — a whole-bench run launches a kernel at several sizes)
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_hash import kernel_source_hash  # noqa: E402


def per_launch(d, counter, kernel, min_value, grid=(0, 1 << 62)):
    vals = []
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter and grid[0] <= int(float(r.get("Grid_Size", 0) or 0)) <= grid[1]:
                v = float(r["Counter_Value"])
                if v >= min_value:
                    vals.append(v)
    vals.sort()
    return vals


def main():
    fetch_dir, write_dir, kernel, out = sys.argv[1:5]
    min_value = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    grid = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 1 << 62)
    fv = per_launch(fetch_dir, "FETCH_SIZE", kernel, min_value, grid)
    wv = per_launch(write_dir, "WRITE_SIZE", kernel, min_value, grid)
    med = lambda v: v[len(v) // 2] if v else None  # noqa: E731
    f, w = med(fv), med(wv)
    res = {
        "kernel": kernel, "kernel_source_hash": kernel_source_hash(),  # of the tree the passes ran on (tools/kernel_hash.py)
        "launches_fetch_pass": len(fv), "launches_write_pass": len(wv), "grid_range": list(grid) if len(sys.argv) > 7 else None,
        "FETCH_SIZE_median_raw": f, "WRITE_SIZE_median_raw": w,
        "fetch_bytes_raw": None if f is None else f * 1024,
        "fetch_bytes_corrected": None if f is None else 2 * f * 1024,
        "write_bytes": None if w is None else w * 1024,
        "traffic_bytes_per_launch": None if f is None or w is None else 2 * f * 1024 + w * 1024,
        "note": "FETCH_SIZE doubled per the gfx950 calibration in MI355X_MICROARCH.md §HBM; WRITE_SIZE "
                "matches the 16 B per decision result stream exactly, so it is used uncorrected",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
