#!/bin/bash
# the three bench lines again (after the PMC summaries of the same kernel sources are in profiles/: roofline.traffic is then the stamped figure)
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/bench_lines}
mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_C3_n1_steps20.json.log 2> $OUT/bench_C3_steps20.err; echo "bench(20) exit $?"
timeout 900 python bench.py > $OUT/bench_C3_n1.json.log 2> $OUT/bench_C3.err; echo "bench exit $?"
timeout 600 python bench.py --workload C4 --steps 200 --warmup 10 --no-secondary > $OUT/bench_C4_n1.json.log 2> $OUT/bench_C4.err; echo "bench C4 exit $?"
python tools/benchline.py steps20 < $OUT/bench_C3_n1_steps20.json.log; python tools/benchline.py default < $OUT/bench_C3_n1.json.log; python tools/benchline.py C4 < $OUT/bench_C4_n1.json.log
