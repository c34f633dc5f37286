#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp27
mkdir -p $OUT
for e in X=1 MMP_NO_LONG_MEMO=1; do echo "== full cluster $e"; env $e MMP_PHASE_FULL=1 timeout 300 python tools/r6/wave_timeline.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/wave_timeline.txt
echo "== C3" | tee -a $OUT/wave_timeline.txt; timeout 300 python tools/r6/wave_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wave_timeline.txt
timeout 1500 python -m pytest tests/test_long_memo_gpu.py tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_caller_gpu.py tests/test_place_multi_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log | cut -c1-300
for env in "X=1" "MMP_NO_LONG_MEMO=1" "MMP_LONG_DENSE_FROM=2000000000" "MMP_LONG_DENSE_FROM=2000000000 MMP_NO_LONG_MEMO=1"; do
  for cfg in "--decisions-per-step 100000" ""; do
    env $env timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --workload C3 --full-cluster --no-pod-axis --no-secondary --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$env', '$cfg', 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us  parity', d['parity_vs_oracle'])"
  done
done | tee $OUT/timing.txt
