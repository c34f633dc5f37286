#!/bin/bash
# experiment: TILES x 64 requests per wavefront, one lane phase over the uncovered ones (MMP_MEMO_MT=1), library variants by TILES
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so
  timeout 300 python tools/r5/memo_sweep.py 400000 800000 1600000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; grep -v "NO_MEMO" $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
  MEMO_SWEEP_CALLER=4321 MEMO_SWEEP_FORM=c timeout 300 python tools/r5/memo_sweep.py 800000 > $OUT/sweep_c_$v.txt 2> $OUT/sweep_c_$v.err; grep -v "NO_MEMO" $OUT/sweep_c_$v.txt
done
