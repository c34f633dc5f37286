#!/bin/bash
# the bench line on the driver's flags and on the defaults (no secondary legs for speed when BENCH_FAST=1)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_bench
mkdir -p $OUT
EXTRA=${BENCH_EXTRA:-}
timeout 1200 python bench.py --steps 20 --warmup 5 $EXTRA > $OUT/bench_steps20.json.log 2> $OUT/bench_steps20.err; echo "exit $?"
python tools/benchline.py steps20 < $OUT/bench_steps20.json.log
tail -3 $OUT/bench_steps20.err
