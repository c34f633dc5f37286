#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp20
mkdir -p $OUT
run() {
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=14 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only $ARGS > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | sed 's/region [0-9]*: issue //; s/ us, known done [0-9.]*//; s/ us, total /\//; s/ us//' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
ARGS=""; run split_plain X=1
ARGS="--issue-threads 4"; run split_it4 X=1
ARGS="--issue-threads 2"; run split_it2 X=1
ARGS=""; run nosplit_plain MMP_NO_SPLIT=1
ARGS=""; run nosplit_streams MMP_NO_SPLIT=1 MMP_BENCH_FENCE=streams
