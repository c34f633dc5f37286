#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so
  timeout 300 python tools/r5/memo_sweep.py 100000 200000 400000 800000 1600000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; cat $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
done
