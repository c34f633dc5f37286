#!/bin/bash
# (the variant libraries were built from temporary edits of the kernel sources — which stores / loads are non-temporal — that are not in the tree:
#  what was kept is store_out_streaming in place_kernel.hpp; profiles/r6/nontemporal_result_stores.txt has the numbers)
# non-temporal result stores in place_block too (every batch kernel) against the product (memo_try's only), alternating, one visit
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp31
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants
for rep in 1 2; do
 for lib in "" "$V/libmmplace_ntall.so"; do
  echo "== ${lib:-product}"
  MMP_LIB_PATH=$lib SWEEP_ONLY=0,1 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 100000 800000 2>&1 | grep "MMP_"
  for cfg in "--decisions-per-step 100000" ""; do
    MMP_LIB_PATH=$lib timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --workload C3 --full-cluster --no-pod-axis --no-secondary --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('full cluster $cfg', 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us  parity', d['parity_vs_oracle'])"
  done
 done
done 2>&1 | tee $OUT/sweep.txt
