#!/bin/bash
# One GPU-box visit that produces everything profiles/rNN holds: parity tests, the bench lines (driver's flags and
# defaults; C3 and C4), rocprofv3 kernel stats (1 stream = what roofline.kernel_ms must agree with; 4 streams), the HBM
# PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs), SQ counters, the phase clock, the batch-size x stream sweep.
# usage (repo root on the GPU box): bash tools/gpu_profile.sh [out_dir]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof_r2}
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; grep -E "passed|failed|exit" $OUT/pytest_gpu.log | tail -2
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# bench: the driver's flags, then the defaults
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_C3_n1_steps20.json.log 2> $OUT/bench_C3_steps20.err; echo "bench(20) exit $?"
timeout 900 python bench.py > $OUT/bench_C3_n1.json.log 2> $OUT/bench_C3.err; echo "bench exit $?"
python tools/benchline.py steps20 < $OUT/bench_C3_n1_steps20.json.log; python tools/benchline.py default < $OUT/bench_C3_n1.json.log
for wl in ${PROFILE_WORKLOADS:-C3 C4}; do
  for ns in 1 4; do
    rm -rf /tmp/prof_${wl}_$ns
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${wl}_$ns -- python bench.py --workload $wl --kernel-only --steps 200 --warmup 20 --streams $ns > $OUT/prof_${wl}_${ns}streams_bench.log 2>&1
    f=$(find /tmp/prof_${wl}_$ns -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/place_batch_${wl}_kernel_stats_${ns}streams.csv && grep place_batch $OUT/place_batch_${wl}_kernel_stats_${ns}streams.csv | cut -c1-150
  done
  rm -rf /tmp/pmc_f /tmp/pmc_w
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python bench.py --workload $wl --kernel-only --steps 20 --warmup 2 --streams 1 > $OUT/pmc_fetch_$wl.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python bench.py --workload $wl --kernel-only --steps 20 --warmup 2 --streams 1 > $OUT/pmc_write_$wl.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_f /tmp/pmc_w place_batch_kernel $OUT/pmc_place_batch_$wl.json
  rm -rf /tmp/sq1 /tmp/sq2
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/sq1 -- python bench.py --workload $wl --kernel-only --steps 20 --warmup 2 --streams 1 > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/sq2 -- python bench.py --workload $wl --kernel-only --steps 20 --warmup 2 --streams 1 > /dev/null 2>&1
  (python tools/sq_summary.py /tmp/sq1 place_batch_kernel; python tools/sq_summary.py /tmp/sq2 place_batch_kernel) > $OUT/sq_place_batch_$wl.jsonl; cat $OUT/sq_place_batch_$wl.jsonl | cut -c1-300
  grep "^{" $OUT/prof_${wl}_1streams_bench.log | tail -1 > $OUT/bench_${wl}_kernel_only_1stream.json.log
done
# the whole bench under the kernel trace: the secondary kernels' own durations (bench.py brackets each of their launches with an event pair, which adds the gap between marker and kernel)
rm -rf /tmp/prof_full
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -- python bench.py --steps 20 --warmup 5 > $OUT/prof_full_bench.log 2>&1
f=$(find /tmp/prof_full -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_full_kernel_stats.csv && grep -E "serve_batch|gate_batch|evict_batch|cache_replay|ingest_" $OUT/bench_full_kernel_stats.csv | cut -c1-120
timeout 200 python tools/region_anatomy.py 2>&1 | grep -v amdgpu.ids > $OUT/region_anatomy.txt; grep "helpers 0 4 streams" $OUT/region_anatomy.txt
if [[ "${PROFILE_WORKLOADS:-C3 C4}" == *C4* ]]; then timeout 600 python bench.py --workload C4 --steps 200 --warmup 10 --no-secondary > $OUT/bench_C4_n1.json.log 2> $OUT/bench_C4.err; echo "bench C4 exit $?"; python tools/benchline.py C4 < $OUT/bench_C4_n1.json.log; fi
timeout 200 python tools/phase_clock.py 30 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_place_batch_C3.txt; tail -12 $OUT/phase_clock_place_batch_C3.txt
timeout 600 python tools/place_sweep.py C3 2> /dev/null > $OUT/place_sweep_C3.csv; cat $OUT/place_sweep_C3.csv
KT_GRAPH=${KT_GRAPH:-0} timeout 400 python tools/kernel_time.py C3 2>&1 | grep -v amdgpu.ids > $OUT/kernel_time_C3.txt
[ "${WITH_SYNC_COST:-0}" = 1 ] && timeout 300 python tools/sync_cost.py 2>&1 | grep -v amdgpu.ids > $OUT/sync_cost.txt
timeout 120 python tools/resident_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/resident_latency.txt; grep '1 thread' $OUT/resident_latency.txt
