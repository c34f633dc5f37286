#!/bin/bash
# round 5, experiment 6: library variants of the shortlist kernel (tools/r5/build_variant.sh), launch time alone.  usage: bash tools/r5/exp6.sh out_dir variant...
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so timeout 300 python tools/r5/memo_sweep.py 800000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; cat $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
done
