#!/bin/bash
# the full-cluster batch as one launch and split (place_long_memo_kernel + place_long_tail_kernel): parity, then launch time by batch size
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_long_split
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_long_memo_gpu.py tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_caller_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | cut -c1-300
for env in "MMP_LONG_SPLIT_FROM=0" "MMP_NO_SPLIT=1" "MMP_NO_SPLIT=1 MMP_LONG_DENSE_FROM=2000000000" "MMP_NO_LONG_MEMO=1"; do
  for n in 200000 400000 800000; do
   for st in 1 4; do
    env $env timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 --streams $st --workload C3 --full-cluster --no-pod-axis --no-secondary --no-cpu-baseline --decisions-per-step $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$env', 'n $n streams $st', 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us  ms_per_step', round(d['ms_per_step']*1e3,2), 'us  parity', d['parity_vs_oracle'])"
   done
  done
done | tee $OUT/timing.txt
