#!/bin/bash
# the recorded long walks (LongMemo): parity, then the full-cluster launch at 100k / 800k with and without the records
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp26
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_long_memo_gpu.py tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_caller_gpu.py tests/test_place_multi_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log | cut -c1-300
for env in "X=1" "MMP_NO_LONG_MEMO=1"; do
  for cfg in "--decisions-per-step 100000" ""; do
    env $env timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --workload C3 --full-cluster --no-pod-axis --no-secondary --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$env', '$cfg', r['kernel'], 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us  parity', d['parity_vs_oracle'])"
  done
done | tee $OUT/timing.txt
