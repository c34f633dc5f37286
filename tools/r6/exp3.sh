#!/bin/bash
# cost map of the first launch (variants built by tools/r6/build_variant.sh), and the split with s_setprio in the tail
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp3
mkdir -p $OUT
for v in stream nocand noown; do
  echo "== $v" | tee -a $OUT/variants.txt
  MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so SWEEP_NOTAIL=1 SWEEP_ONLY=3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 400000 800000 1600000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.txt
done
echo "== product" | tee -a $OUT/variants.txt
SWEEP_NOTAIL=1 SWEEP_ONLY=0,1,2,3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 400000 800000 1600000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.txt
