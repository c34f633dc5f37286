#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp9
mkdir -p $OUT
for ns in 4 6 8 12; do
  SWEEP_STREAMS=$ns SWEEP_NOTAIL=1 SWEEP_ONLY=1,2,3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 800000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/streams.txt
done
for q in 8; do
  GPU_MAX_HW_QUEUES=$q SWEEP_STREAMS=8 SWEEP_NOTAIL=1 SWEEP_ONLY=1,2,3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 800000 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq$q /" | tee -a $OUT/streams.txt
  GPU_MAX_HW_QUEUES=$q SWEEP_STREAMS=4 SWEEP_NOTAIL=1 SWEEP_ONLY=1,2,3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 800000 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq$q /" | tee -a $OUT/streams.txt
done
