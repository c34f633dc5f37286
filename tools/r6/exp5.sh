#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp5
mkdir -p $OUT
timeout 300 python tools/r6/tail_clock.py 40 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_clock.txt
cd /tmp; rm -rf /tmp/prof2
SWEEP_ONLY=2 SWEEP_K=50 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -- python $GRAFT_REPO_ROOT/tools/r6/split_sweep.py 800000 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
f=$(find /tmp/prof2 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/$OUT/gaps.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "place_memo_kernel" in r["Kernel_Name"] or "place_tail_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the single-stream part: consecutive memo -> tail -> memo on one queue
import collections
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    g1, g2, dm, dt = [], [], [], []
    for a, b in zip(rs, rs[1:]):
        gap = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if "memo" in a["Kernel_Name"] and "tail" in b["Kernel_Name"]: g1.append(gap)
        if "tail" in a["Kernel_Name"] and "memo" in b["Kernel_Name"]: g2.append(gap)
    for r in rs:
        (dm if "memo" in r["Kernel_Name"] else dt).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    med = lambda v: sorted(v)[len(v)//2] if v else -1
    print(f"queue {q}: {len(rs)} kernels; memo {med(dm)} ns, tail {med(dt)} ns (median); gap memo->tail {med(g1)} ns, tail->memo {med(g2)} ns")
PY
