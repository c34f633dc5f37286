#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp12
mkdir -p $OUT
run() {
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=10 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only $ARGS > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | awk '{print $3, $(NF-1)}' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
ARGS="--issue-threads 4"; run split_it4 X=1
ARGS="--issue-threads 2"; run split_it2 X=1
ARGS="--issuers 2"; run split_is2 X=1
ARGS="--issuers 4"; run split_is4 X=1
ARGS=""; run split_plain X=1
ARGS="--issue-threads 4"; run nosplit_it4 MMP_NO_SPLIT=1
