#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp8
mkdir -p $OUT
timeout 300 python tools/r6/tail_clock.py 40 2>&1 | grep -v amdgpu.ids | tee $OUT/tail_clock.txt
SWEEP_NOTAIL=1 SWEEP_ONLY=${SWEEP_ONLY:-1,2,3} SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 400000 800000 1600000 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
timeout 1500 python -m pytest tests/test_shortlist_memo_gpu.py tests/test_ref_vectors_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
