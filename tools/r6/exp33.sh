#!/bin/bash
# workgroup size of the batch kernels (2 / 4 / 8 wavefronts: -DMMP_PLACE_WAVES) with the split form, alternating, one visit
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp33
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants
for rep in 1 2; do
 for lib in "" "$V/libmmplace_w2.so" "$V/libmmplace_w8.so"; do
  echo "== ${lib:-product (4 wavefronts)}"
  MMP_LIB_PATH=$lib SWEEP_NOTAIL=1 SWEEP_ONLY=0,2,3 SWEEP_K=300 timeout 600 python tools/r6/split_sweep.py 800000 2>&1 | grep "MMP_MEMO"
 done
done 2>&1 | tee $OUT/sweep.txt
