#!/bin/bash
# (the variant libraries were built from temporary edits of the kernel sources — which stores / loads are non-temporal — that are not in the tree:
#  what was kept is store_out_streaming in place_kernel.hpp; profiles/r6/nontemporal_result_stores.txt has the numbers)
# non-temporal result stores (and request loads) in place_memo_kernel against the product, alternating, one visit
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp30
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants
for rep in 1 2; do
  echo "== product"; SWEEP_NOTAIL=1 SWEEP_ONLY=0,2,3 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 800000
  echo "== results stored non-temporally"; MMP_LIB_PATH=$V/libmmplace_ntout.so SWEEP_NOTAIL=1 SWEEP_ONLY=0,2,3 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 800000
  echo "== ... and requests loaded non-temporally"; MMP_LIB_PATH=$V/libmmplace_ntin.so SWEEP_NOTAIL=1 SWEEP_ONLY=0,2,3 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 800000
done 2>&1 | grep "MMP_MEMO\|==" | tee $OUT/sweep.txt
