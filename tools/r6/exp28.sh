#!/bin/bash
# the anatomy of the driver's 20-step region: host issue time against device time, split and one-launch
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp28
mkdir -p $OUT
for env in "X=1" "MMP_NO_SPLIT=1"; do
  echo "== $env"
  env $env MMP_BENCH_REPEAT=8 timeout 600 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only 2>&1 >/dev/null | grep "^region"
done | tee $OUT/regions.txt
