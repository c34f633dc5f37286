#!/bin/bash
# (the variant libraries were built from temporary edits of the kernel sources — which stores / loads are non-temporal — that are not in the tree:
#  what was kept is store_out_streaming in place_kernel.hpp; profiles/r6/nontemporal_result_stores.txt has the numbers)
# non-temporal result stores in serve / gate / route kernels against the product: the bench's per-kernel leg, alternating, one visit
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp32
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants
for rep in 1 2; do
 for lib in "" "$V/libmmplace_ntsec.so"; do
  echo "== ${lib:-product}"
  MMP_LIB_PATH=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d.get('kernels', []):
    if any(k in r['kernel'] for k in ('serve_batch', 'gate_batch', 'route_batch', 'evict_batch')):
        print('  ', r['kernel'][:60].ljust(60), 'units', r['units'], 'kernel_ms', round(r['kernel_ms']*1e3,2), 'us', 'frac', None if r.get('frac_hbm_peak') is None else round(r['frac_hbm_peak'],3))"
 done
done 2>&1 | tee $OUT/kernels.txt
