#!/bin/bash
# Round 6: everything profiles/r6 holds, in one GPU-box visit (raw rocprofv3 traces stay in /tmp; these are the summaries).
# usage (repo root on the GPU box): bash tools/gpu_profile_r6.sh [out_dir]      PROFILE_SKIP_TESTS=1: without the parity suite
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof_r6}
mkdir -p $OUT
python tools/kernel_hash.py > $OUT/kernel_source_hash.txt
if [ -z "${PROFILE_SKIP_TESTS:-}" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; grep -E "passed|failed|exit" $OUT/pytest_gpu.log | tail -2; fi
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log

# one profiled configuration: kernel stats (1 stream), the two HBM PMC passes, two SQ passes
# $1 tag, $2 kernel-name substring, rest: bench.py arguments
profile_one() {
  local tag=$1 kern=$2; shift 2
  rm -rf /tmp/p_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag/stats -- python bench.py --kernel-only --steps 200 --warmup 20 --streams 1 "$@" > $OUT/bench_${tag}_kernel_only_1stream.log 2>&1
  f=$(find /tmp/p_$tag/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats_1stream.csv && grep "$kern" $OUT/${tag}_kernel_stats_1stream.csv | cut -c1-160
  grep "^{" $OUT/bench_${tag}_kernel_only_1stream.log | tail -1 > $OUT/bench_${tag}_kernel_only_1stream.json.log
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_$tag/f -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_$tag/w -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/p_$tag/f /tmp/p_$tag/w $kern $OUT/pmc_place_batch_$tag.json > /dev/null; cut -c1-400 $OUT/pmc_place_batch_$tag.json | tr '\n' ' '; echo
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/p_$tag/sq1 -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/p_$tag/sq2 -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  (python tools/sq_summary.py /tmp/p_$tag/sq1 $kern; python tools/sq_summary.py /tmp/p_$tag/sq2 $kern) > $OUT/sq_$tag.jsonl; cut -c1-260 $OUT/sq_$tag.jsonl
}
if [ -n "${PROFILE_ONLY:-}" ]; then  # one configuration again (after a script fix): PROFILE_ONLY="<tag> <kernel> <bench args...>"
  profile_one $PROFILE_ONLY
  exit 0
fi
profile_one C3_800k place_memo_kernel --workload C3   # (launches from 393 216 decisions on are split: the check's own launch is the dominant kernel)
# ... and the tail launch of the same runs
python tools/pmc_summary.py /tmp/p_C3_800k/f /tmp/p_C3_800k/w place_tail_kernel $OUT/pmc_place_tail_C3_800k.json > /dev/null; cut -c1-300 $OUT/pmc_place_tail_C3_800k.json | tr '\n' ' '; echo
(python tools/sq_summary.py /tmp/p_C3_800k/sq1 place_tail_kernel; python tools/sq_summary.py /tmp/p_C3_800k/sq2 place_tail_kernel) > $OUT/sq_C3_800k_tail.jsonl
# the one-launch form of the same batches (MMP_NO_SPLIT=1: the check in front of the lane phase of one kernel)
MMP_NO_SPLIT=1 profile_one C3_800k_one_launch place_batch_m_kernel --workload C3
profile_one C3_100k place_batch_kernel --workload C3 --decisions-per-step 100000
profile_one C3_full_cluster_100k place_batch_long_kernel --workload C3 --decisions-per-step 100000 --full-cluster
profile_one C3_full_cluster_800k place_batch_long_kernel --workload C3 --full-cluster   # (with the recorded long walks the barrier-free instantiation runs at every size)
[[ "${PROFILE_WORKLOADS:-C3 C4}" == *C4* ]] && profile_one C4 place_memo_kernel --workload C4

# the timed region's own shape — four streams, a hardware queue each — under the kernel trace (VERDICT r5 missing 6: ms_per_step below the
# one-stream kernel time is overlap; here are the stretched per-kernel durations that go with it)
rm -rf /tmp/p_4s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_4s -- python bench.py --kernel-only --steps 1000 --warmup 20 --streams 4 > $OUT/bench_C3_800k_kernel_only_4streams.log 2>&1
f=$(find /tmp/p_4s -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/C3_800k_kernel_stats_4streams.csv && head -4 $OUT/C3_800k_kernel_stats_4streams.csv | cut -c1-170
grep "^{" $OUT/bench_C3_800k_kernel_only_4streams.log | tail -1 > $OUT/bench_C3_800k_kernel_only_4streams.json.log

# the secondary kernels: the whole bench under the kernel trace, then under the two SQ passes
rm -rf /tmp/p_full
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_full/stats -- python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline > $OUT/prof_full_bench.log 2>&1
f=$(find /tmp/p_full/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_full_kernel_stats.csv && grep -E "serve_batch|gate_batch|evict_batch|cache_replay|ingest_|build_b|build_wins" $OUT/bench_full_kernel_stats.csv | cut -c1-140
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/p_full/sq1 -- python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/p_full/sq2 -- python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_full/f -- python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_full/w -- python bench.py --steps 20 --warmup 5 --no-pod-axis --no-cpu-baseline > /dev/null 2>&1
# per-kernel digests by launch size in WORK-ITEMS (the run launches these kernels for single requests, for the churn leg's mixed
# sizes, and for the `kernels` leg at 100k and 800k units): serve / gate one lane per unit, evict 8 lanes per evaluation, cache
# replay one wavefront per cache (10k caches)
digest() {  # $1 kernel, $2 file tag, $3 min work-items, $4 max work-items
  (python tools/sq_summary.py /tmp/p_full/sq1 $1 $3 $4; python tools/sq_summary.py /tmp/p_full/sq2 $1 $3 $4) > $OUT/sq_$2.jsonl
  python tools/pmc_summary.py /tmp/p_full/f /tmp/p_full/w $1 $OUT/pmc_$2.json 0 $3 $4 > /dev/null
  echo "$2: $(cut -c1-160 $OUT/sq_$2.jsonl | head -1) | $(python -c "import json; d=json.load(open('$OUT/pmc_$2.json')); print(d['launches_fetch_pass'], d['traffic_bytes_per_launch'])")"
}
digest serve_batch_kernel serve_batch_kernel 50000 400000
digest gate_batch_kernel gate_batch_kernel 50000 400000
digest evict_batch_kernel evict_batch_kernel 500000 2000000
digest cache_replay_kernel cache_replay_kernel 50000 2000000
digest serve_batch_kernel serve_batch_kernel_800k 500000 2000000
digest gate_batch_kernel gate_batch_kernel_800k 500000 2000000
digest evict_batch_kernel evict_batch_kernel_800k 3000000 1000000000
digest route_batch_kernel route_batch_kernel 50000 400000
digest route_batch_kernel route_batch_kernel_800k 500000 2000000
digest place_memo_c_kernel place_memo_c_kernel_800k 500000 2000000   # the single-caller form's first launch (bench.py: single_caller leg)

# the bench lines: the driver's flags, then the defaults; C4
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_C3_n1_steps20.json.log 2> $OUT/bench_C3_steps20.err; echo "bench(20) exit $?"
for k in 2 3 4; do timeout 900 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only 2>/dev/null | python tools/benchline.py "steps20 run $k" >> $OUT/bench_steps20_more_runs.txt; done; cat $OUT/bench_steps20_more_runs.txt
timeout 900 python bench.py > $OUT/bench_C3_n1.json.log 2> $OUT/bench_C3.err; echo "bench exit $?"
python tools/benchline.py steps20 < $OUT/bench_C3_n1_steps20.json.log; python tools/benchline.py default < $OUT/bench_C3_n1.json.log
if [[ "${PROFILE_WORKLOADS:-C3 C4}" == *C4* ]]; then timeout 600 python bench.py --workload C4 --steps 200 --warmup 10 --no-secondary > $OUT/bench_C4_n1.json.log 2> $OUT/bench_C4.err; echo "bench C4 exit $?"; python tools/benchline.py C4 < $OUT/bench_C4_n1.json.log; fi
timeout 300 python tools/full_cluster_by_type.py 2>&1 | grep -v amdgpu.ids > $OUT/full_cluster_by_type.txt; cat $OUT/full_cluster_by_type.txt
timeout 300 python tools/case_b_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/case_b_timing.txt; cat $OUT/case_b_timing.txt

# ---- round 6 ----
timeout 600 bash tools/r6/seam_tail.sh > /dev/null 2>&1; cp gpurun_out/r6_seam/seam_tail_pinning.txt $OUT/ 2>/dev/null
SWEEP_ONLY=0,1,2 SWEEP_K=200 GPU_MAX_HW_QUEUES=8 timeout 600 python tools/r6/split_sweep.py 400000 800000 1600000 2>&1 | grep -v amdgpu.ids > $OUT/split_sweep_rows.txt; cat $OUT/split_sweep_rows.txt
if [ -f modelmesh_amd/lib/libmmplace_phase.so ]; then
  timeout 300 python tools/r6/tail_clock.py 40 2>&1 | grep -v amdgpu.ids > $OUT/tail_clock.txt
  # when the wavefronts of a 100k launch start and end: the full cluster with / without the recorded long walks, C3
  (for e in X=1 MMP_NO_LONG_MEMO=1; do echo "== full cluster $e"; env $e MMP_PHASE_FULL=1 timeout 300 python tools/r6/wave_timeline.py 2>&1 | grep -v amdgpu.ids; done
   echo "== C3"; timeout 300 python tools/r6/wave_timeline.py 2>&1 | grep -v amdgpu.ids) > $OUT/wave_timeline.txt; cat $OUT/wave_timeline.txt
fi
# ---- round 5 ----
# the single-caller request form against the same decisions as 64-byte rows (launch time, C3 and the full cluster); launch time against the
# number of decisions per launch around the chip's rounds; the four n = 1 seam calls from a C++ host
timeout 300 python tools/r5/caller_timing.py 100000 800000 2>&1 | grep "^C3" > $OUT/single_caller_timing.txt; cat $OUT/single_caller_timing.txt
timeout 300 python tools/r5/nsweep.py 2>&1 | grep "^n " > $OUT/launch_size_sweep.txt; cat $OUT/launch_size_sweep.txt
g++ -O2 -std=c++17 -Iinclude tools/micro/seam_tail.cc -Lmodelmesh_amd/lib -lmmplace -Wl,-rpath,$PWD/modelmesh_amd/lib -lpthread -o /tmp/seam_tail && timeout 300 /tmp/seam_tail 20000 2>&1 | grep -v amdgpu.ids > $OUT/seam_tail_run.txt; cat $OUT/seam_tail_run.txt
# the single-caller kernel under the kernel trace + counters (its own launch: bench.py's single_caller leg runs it 220 times at 800k)
# ---- round 4 ----
# a commit after 16 changed rows / from scratch: span and wall without a profiler, then the per-kernel times under the kernel trace
for wk in C3 C4; do for kind in delta full; do
  python tools/commit_breakdown.py $wk $kind 40 2>&1 | grep commits >> $OUT/commit_timing.txt
  rm -rf /tmp/p_commit
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_commit -- python tools/commit_breakdown.py $wk $kind 40 > /dev/null 2>&1
  f=$(find /tmp/p_commit -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/commit_${wk}_${kind}_kernel_stats.csv
done; done
cat $OUT/commit_timing.txt
# the reaper's plan
for wk in C3 C4; do python tools/plan_breakdown.py $wk 2>&1 | grep "plan of" >> $OUT/plan_timing.txt; done
MMP_PLAN_SORTED=1 python tools/plan_breakdown.py C3 2>&1 | grep "plan of" | sed 's/^/sorted path (MMP_PLAN_SORTED=1): /' >> $OUT/plan_timing.txt
rm -rf /tmp/p_plan
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_plan -- python tools/plan_breakdown.py C3 > /dev/null 2>&1
f=$(find /tmp/p_plan -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/plan_C3_kernel_stats.csv
cat $OUT/plan_timing.txt
# where a wavefront's time goes (s_memtime phase clocks): the window path and the full-cluster path
if [ -f modelmesh_amd/lib/libmmplace_phase.so ]; then
  timeout 300 python tools/phase_clock.py 50 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_C3.txt
  MMP_PHASE_FULL=1 timeout 300 python tools/phase_clock.py 50 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_C3_full_cluster.txt
  tail -12 $OUT/phase_clock_C3.txt
fi
# the pod axis at one shard (the protocol's own cost), and the driver's 8-rank control flow on this one device over gloo
mkdir -p $OUT/pod_axis
timeout 300 python tools/pod_axis_timing.py C3 200 2>&1 | grep "_leg" > $OUT/pod_axis/one_shard_C3_200steps.txt; cat $OUT/pod_axis/one_shard_C3_200steps.txt
MMP_BENCH_ONE_DEVICE=1 MMP_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
  --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench_8ranks_one_device_gloo.json.log 2> $OUT/bench8.err; echo "bench8 exit $?"
tail -1 $OUT/bench_8ranks_one_device_gloo.json.log | cut -c1-300
