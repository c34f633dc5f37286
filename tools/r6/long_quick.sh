#!/bin/bash
# full-cluster kernels after a change: parity, then launch time at 100k / 800k (event pair over 200 back-to-back launches, one stream) for
# the default routing and with every launch on the barrier-free instantiation (MMP_LONG_DENSE_FROM beyond any batch)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_long_${1:-x}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for env in "X=1" "MMP_LONG_DENSE_FROM=2000000000"; do
  for cfg in "--decisions-per-step 100000" ""; do
    env $env timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --workload C3 --full-cluster --no-pod-axis --no-secondary --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$env', '$cfg', r['kernel'], 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us  parity', d['parity_vs_oracle'])"
  done
done | tee $OUT/timing.txt
