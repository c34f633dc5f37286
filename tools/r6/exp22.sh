#!/bin/bash
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp22
mkdir -p $OUT
echo "== rows, C3" | tee $OUT/sweeps.txt
SWEEP_ONLY=0,1,2 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 100000 200000 300000 400000 800000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweeps.txt
echo "== one caller with room (4321), C3" | tee -a $OUT/sweeps.txt
MEMO_SWEEP_FORM=c MEMO_SWEEP_CALLER=4321 SWEEP_ONLY=0,1,2 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 100000 200000 400000 800000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweeps.txt
echo "== a full caller (17), C3" | tee -a $OUT/sweeps.txt
MEMO_SWEEP_FORM=c MEMO_SWEEP_CALLER=17 SWEEP_ONLY=1,2 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 400000 800000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweeps.txt
echo "== rows, C4" | tee -a $OUT/sweeps.txt
SWEEP_WORKLOAD=C4 SWEEP_ONLY=0,1,2 SWEEP_K=100 timeout 900 python tools/r6/split_sweep.py 1000000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweeps.txt
