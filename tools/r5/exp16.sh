#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so
  timeout 300 python tools/r5/memo_sweep.py 400000 800000 1600000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; grep MEMO_FROM $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
done
