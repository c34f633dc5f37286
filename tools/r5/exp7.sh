#!/bin/bash
# round 5, experiment 7: the wave-level shortlist kernel (place_wave_memo): parity, launch time against the ordinary kernel, library variants
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
timeout 600 python -m pytest tests/test_shortlist_memo_gpu.py -x -q > $OUT/pytest_memo.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_memo.log; tail -8 $OUT/pytest_memo.log
timeout 400 python tools/r5/memo_sweep.py 100000 200000 400000 800000 1600000 > $OUT/memo_sweep.txt 2> $OUT/memo_sweep.err; cat $OUT/memo_sweep.txt; tail -3 $OUT/memo_sweep.err | grep -v amdgpu.ids
MEMO_SWEEP_FORM=c timeout 400 python tools/r5/memo_sweep.py 800000 > $OUT/memo_sweep_c.txt 2> $OUT/memo_sweep_c.err; cat $OUT/memo_sweep_c.txt; tail -3 $OUT/memo_sweep_c.err | grep -v amdgpu.ids
for v in "$@"; do
  echo "== $v"
  MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_$v.so timeout 300 python tools/r5/memo_sweep.py 800000 > $OUT/sweep_$v.txt 2> $OUT/sweep_$v.err; grep MEMO_FROM $OUT/sweep_$v.txt; tail -2 $OUT/sweep_$v.err | grep -v amdgpu.ids
done
