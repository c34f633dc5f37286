#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2g
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_s20.json.log 2> $OUT/bench_s20.err; echo "bench(20) exit $?"; python tools/benchline.py s20 < $OUT/bench_s20.json.log
tail -3 $OUT/bench_s20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2g/bench_s20.json.log").read().strip().splitlines()[-1])
for k in ("pod_axis","pod_axis_in_library_rccl","churn_pod_axis","cpu_baseline"):
    print(k, json.dumps(d.get(k))[:700])
PY
