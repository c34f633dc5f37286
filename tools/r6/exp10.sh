#!/bin/bash
# the driver's 20-step region repeated in one process, by stream count / hardware queues / split
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp10
mkdir -p $OUT
run() {  # label, env..., -- args
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=10 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only $ARGS > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | awk '{print $(NF-1)}' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
ARGS=""; run split_hwq8_s4 X=1
ARGS="--streams 6"; run split_hwq8_s6 X=1
ARGS="--streams 8"; run split_hwq8_s8 X=1
ARGS=""; run split_hwq4_s4 GPU_MAX_HW_QUEUES=4
ARGS="--streams 6"; run split_hwq4_s6 GPU_MAX_HW_QUEUES=4
ARGS=""; run nosplit_hwq8_s4 MMP_NO_SPLIT=1
ARGS=""; run nosplit_hwq4_s4 MMP_NO_SPLIT=1 GPU_MAX_HW_QUEUES=4
ARGS="--steps 1000 --warmup 50"; run split_hwq8_s4_1000 X=1
ARGS="--steps 1000 --warmup 50"; run nosplit_hwq4_s4_1000 MMP_NO_SPLIT=1 GPU_MAX_HW_QUEUES=4
