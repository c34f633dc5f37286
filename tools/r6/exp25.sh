#!/bin/bash
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp25
mkdir -p $OUT
(echo "== tail as is"; SWEEP_TAILS=32,64 SWEEP_ONLY=0,2,3,4 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 800000
 echo "== tail entries spread over the wavefronts"; MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_spread.so SWEEP_TAILS=32,64 SWEEP_ONLY=0,2,3,4 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 800000) 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
