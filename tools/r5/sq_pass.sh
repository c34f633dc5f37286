#!/bin/bash
# SQ counter passes (two --pmc runs) + kernel stats of bench.py --kernel-only for one kernel.  usage: sq_pass.sh <tag> <kernel> [bench args]; env passes through
set -u
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/sq}
mkdir -p $OUT
tag=$1; kern=$2; shift 2
rm -rf /tmp/p_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag/stats -- python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 "$@" > $OUT/bench_$tag.log 2>&1
f=$(find /tmp/p_$tag/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && grep "$kern" $OUT/${tag}_kernel_stats.csv | cut -c1-160
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/p_$tag/sq1 -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/p_$tag/sq2 -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
(python tools/sq_summary.py /tmp/p_$tag/sq1 $kern; python tools/sq_summary.py /tmp/p_$tag/sq2 $kern) > $OUT/sq_$tag.jsonl; cat $OUT/sq_$tag.jsonl
