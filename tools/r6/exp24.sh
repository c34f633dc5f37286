#!/bin/bash
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp24
mkdir -p $OUT
SWEEP_LDS=18432,20992,23552,27648 SWEEP_ONLY=2,3,4,5,6 SWEEP_K=200 timeout 900 python tools/r6/split_sweep.py 800000 1600000 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
