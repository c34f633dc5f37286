#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2d
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/kernel_time.py C3 > $OUT/kernel_time_w4.txt 2>&1; grep -v amdgpu.ids $OUT/kernel_time_w4.txt | head -4
MMP_LIB_PATH=$PWD/modelmesh_amd/lib/libmmplace_w6.so timeout 300 python tools/kernel_time.py C3 > $OUT/kernel_time_w6.txt 2>&1; grep -v amdgpu.ids $OUT/kernel_time_w6.txt | head -4
MMP_LIB_PATH=$PWD/modelmesh_amd/lib/libmmplace_w6.so timeout 300 python -m pytest tests/test_place_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/phase_clock.py 30 > $OUT/phase_clock.txt 2>&1; grep -v amdgpu.ids $OUT/phase_clock.txt
export SQ_KERNEL=place_batch_kernel
export SQ_CMD="python bench.py --kernel-only --steps 20 --warmup 2 --streams 1"
bash tools/gpu_round.sh sq > $OUT/sq.txt 2>&1; tail -2 $OUT/sq.txt
