#!/bin/bash
# the driver's 20-step region, repeated in one process (MMP_BENCH_REPEAT), with and without the shortlist kernels
set -u
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
for v in memo nomemo memo nomemo; do
  if [ $v = nomemo ]; then export MMP_NO_MEMO=1; else unset MMP_NO_MEMO; fi
  MMP_BENCH_REPEAT=12 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline > $OUT/b_$v.log 2> $OUT/b_$v.err
  echo "== $v"; grep "^region" $OUT/b_$v.err | awk '{print $NF, $(NF-1)}' | tr '\n' ' '; echo; python tools/benchline.py $v < $OUT/b_$v.log
done
