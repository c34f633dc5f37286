#!/bin/bash
# experiment: place_block without the workgroup barrier (windows from global memory, per-wavefront general-path lists)
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so
  timeout 300 python tools/r5/memo_sweep.py 100000 800000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; grep NO_MEMO $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
  MEMO_SWEEP_FORM=c timeout 300 python tools/r5/memo_sweep.py 800000 > $OUT/sweep_c_$v.txt 2> $OUT/sweep_c_$v.err; cat $OUT/sweep_c_$v.txt
done
export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$1.so
timeout 600 python -m pytest tests/test_place_parity_gpu.py tests/test_shortlist_memo_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_caller_gpu.py -x -q > $OUT/pytest_$1.log 2>&1; tail -3 $OUT/pytest_$1.log
