#!/bin/bash
# (a temporary edit, not in the tree: the last 16 bytes of a request row loaded non-temporally, the first 48 as they are)
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp34
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants
for rep in 1 2; do
 for lib in "" "$V/libmmplace_ntlast.so"; do
  echo "== ${lib:-product}"
  MMP_LIB_PATH=$lib SWEEP_NOTAIL=1 SWEEP_ONLY=0,2,3 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 800000 2>&1 | grep "MMP_MEMO"
 done
done 2>&1 | tee $OUT/sweep.txt
