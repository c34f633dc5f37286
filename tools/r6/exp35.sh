#!/bin/bash
# (MMP_BENCH_ORDER was a temporary switch in bench.py for this experiment; it is not in the tree)
# which stream takes which step of the driver's 20-step region (MMP_BENCH_ORDER, an experiment switch of bench.py): round-robin, in blocks,
# a staggered start
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp35
mkdir -p $OUT
for o in rr block ramp ramp2 pairs; do
  echo "== order $o"
  MMP_BENCH_ORDER=$o MMP_BENCH_REPEAT=6 timeout 600 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only 2>&1 >$OUT/line_$o.json | grep "^region"
  python tools/benchline.py "order $o" < $OUT/line_$o.json
done | tee $OUT/orders.txt
for o in rr ramp; do
  MMP_BENCH_ORDER=$o timeout 600 python bench.py --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only 2>/dev/null | python tools/benchline.py "1000 steps, order $o"
done | tee -a $OUT/orders.txt
