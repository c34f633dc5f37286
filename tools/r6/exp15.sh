#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp15
mkdir -p $OUT
run() {
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=14 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only $ARGS > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | sed 's/region [0-9]*: issue [0-9.]* us, known done //; s/ us, total /\//; s/ us//' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
ARGS="--issue-threads 4"; run split_it4 X=1
ARGS="--issue-threads 4"; run split_it4_prio MMP_BENCH_STREAM_PRIO=alt
ARGS=""; run split_plain_prio MMP_BENCH_STREAM_PRIO=alt
ARGS="--issue-threads 8 --streams 8"; run split_it8_s8 X=1
ARGS="--issue-threads 2 --streams 2"; run split_it2_s2 X=1
ARGS="--issue-threads 3 --streams 3"; run split_it3_s3 X=1
