#!/bin/bash
# round 5: the full-cluster (prefix-table) kernels after a change: parity first, then launch time at 100k / 800k, then the SQ passes
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/long}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_caller_gpu.py tests/test_place_multi_gpu.py -x -q -k "full or long or every_device or caller or case_b or sparse or multi" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for n in 100000 800000; do
  timeout 300 python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --full-cluster --decisions-per-step $n > $OUT/fc_$n.log 2> $OUT/fc_$n.err
  grep "^{" $OUT/fc_$n.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('full cluster', $n, 'kernel_us', round(r.get('kernel_ms',0)*1e3,2), 'parity', d.get('parity_vs_oracle'))"
done
if [ -z "${SKIP_SQ:-}" ]; then
OUT=$OUT bash tools/r5/sq_pass.sh fc100k place_batch_long_kernel --full-cluster --decisions-per-step 100000 | tail -2
OUT=$OUT bash tools/r5/sq_pass.sh fc800k place_batch_long4_kernel --full-cluster | tail -2
fi
