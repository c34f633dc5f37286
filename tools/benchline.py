#!/usr/bin/env python
"""Print a few fields of bench.py's JSON line (stdin). usage: benchline.py [label]"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "kernel_ms", round(r["kernel_ms"], 5), "ms_per_step", round(d["ms_per_step"], 5),
      "value", f"{d['value']:.4g}", "parity", d["parity_vs_oracle"], "p50/p99 us", d.get("p50_decision_latency_us"),
      d.get("p99_decision_latency_us"))
