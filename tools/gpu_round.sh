#!/bin/bash
# One GPU-box visit: parity tests, the bench line, rocprofv3 kernel stats and the HBM PMC passes.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tests|bench|prof|pmc ...]
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
what=${*:-tests bench prof pmc}
for w in $what; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err; echo "bench exit $?"
  python tools/benchline.py bench < $OUT/bench.json.log ;;
bench2)
  # the N>1 code path on a 1-GPU box: two ranks share cuda:0 over gloo (the driver's runs use RCCL, one GPU per rank)
  MMP_BENCH_ONE_DEVICE=1 MMP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $OUT/bench2.json.log 2> $OUT/bench2.err; echo "bench2 exit $?"
  tail -1 $OUT/bench2.json.log | cut -c1-600 ;;
bench8)
  # the driver's 8-GPU control flow on a 1-GPU box: eight ranks share cuda:0 over gloo, every leg runs (the in-library group with the
  # exchange words moved by the host's gloo group), the driver's flags
  MMP_BENCH_ONE_DEVICE=1 MMP_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench8.json.log 2> $OUT/bench8.err; echo "bench8 exit $?"
  tail -1 $OUT/bench8.json.log | cut -c1-800; tail -3 $OUT/bench8.err | cut -c1-300 ;;
prof)
  # kernel stats of the bench command; --streams 1 so that every dispatch is back to back on one stream and the
  # average duration is the one bench.py reports as roofline.kernel_ms (overlapping streams stretch dispatches)
  rm -rf $OUT/prof $OUT/prof8
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 > $OUT/prof_bench.log 2>&1
  f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -3 $OUT/kernel_stats.csv | cut -c1-160
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof8 -- python bench.py --kernel-only --steps 200 --warmup 20 > $OUT/prof8_bench.log 2>&1
  f=$(find $OUT/prof8 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_16streams.csv && head -2 $OUT/kernel_stats_16streams.csv | cut -c1-160
  grep "^{" $OUT/prof_bench.log | python tools/benchline.py prof-run ;;
pmc)
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 > $OUT/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 > $OUT/pmc_write.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write place_batch_kernel $OUT/pmc_place_batch_C3.json; cat $OUT/pmc_place_batch_C3.json ;;
commit)
  # per-kernel time of a commit after 16 changed rows and of a commit from scratch (C3, C4)
  for wk in C3 C4; do for kind in delta full; do
    rm -rf $OUT/commit_$wk$kind
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/commit_$wk$kind -- python tools/commit_breakdown.py $wk $kind 40 > $OUT/commit_${wk}_$kind.log 2>&1
    f=$(find $OUT/commit_$wk$kind -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/commit_${wk}_${kind}_kernel_stats.csv
    grep "commits" $OUT/commit_${wk}_$kind.log
  done; done ;;
sq)
  # SQ counters of one kernel of an arbitrary command: SQ_KERNEL=<substring> SQ_CMD="python ..." bash tools/gpu_round.sh sq
  rm -rf $OUT/sq1 $OUT/sq2
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq1 -- $SQ_CMD > $OUT/sq1.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- $SQ_CMD > $OUT/sq2.log 2>&1
  python tools/sq_summary.py $OUT/sq1 "$SQ_KERNEL"; python tools/sq_summary.py $OUT/sq2 "$SQ_KERNEL" ;;
esac
done
