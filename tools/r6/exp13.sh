#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp13
mkdir -p $OUT
run() {
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=10 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only $ARGS > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | awk '{print $3 "/" $7 "/" $(NF-1)}' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
ARGS="--issue-threads 4"; run split_it4 X=1
ARGS="--issue-threads 4"; run split_it4_oldfence MMP_BENCH_FENCE=spin
ARGS="--issue-threads 5 --streams 5"; run split_it5_s5 X=1
ARGS="--issue-threads 6 --streams 6"; run split_it6_s6 X=1
ARGS=""; run split_plain X=1
ARGS=""; run nosplit_plain MMP_NO_SPLIT=1
