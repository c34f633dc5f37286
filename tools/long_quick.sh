#!/bin/bash
# Quick look at the full-cluster (long) kernel after a change: parity first, then the kernel's average duration at 100k and 800k
# decisions per launch (bench.py --kernel-only, one stream) with the SQ instruction counters of the 100k launch.
# usage (repo root on the GPU box): bash tools/long_quick.sh [tag]
set -u
export TMPDIR=/tmp
tag=${1:-x}
OUT=gpurun_out/long_$tag
mkdir -p $OUT
timeout 900 python -m pytest tests/test_place_parity_gpu.py tests/test_ref_vectors_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for cfg in "100k --decisions-per-step 100000" "800k"; do
  set -- $cfg; t=$1; shift
  rm -rf /tmp/lq_$t
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lq_$t -- python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 --workload C3 --full-cluster "$@" > $OUT/bench_$t.log 2>&1
  f=$(find /tmp/lq_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$t.csv && grep "place_batch_long" $OUT/kernel_stats_$t.csv | cut -c1-200
done
rm -rf /tmp/lq_sq
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/lq_sq -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 --workload C3 --full-cluster --decisions-per-step 100000 > /dev/null 2>&1
python tools/sq_summary.py /tmp/lq_sq place_batch_long > $OUT/sq_100k.jsonl; cut -c1-400 $OUT/sq_100k.jsonl
