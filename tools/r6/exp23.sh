#!/bin/bash
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r6_exp23
mkdir -p $OUT
timeout 900 python -m pytest tests/test_shortlist_memo_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
SWEEP_ONLY=1,2,3 SWEEP_K=200 timeout 600 python tools/r6/split_sweep.py 400000 800000 1600000 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
run() {
  label=$1; shift
  env "$@" MMP_BENCH_REPEAT=10 timeout 300 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline --kernel-only > $OUT/b_$label.log 2> $OUT/b_$label.err
  echo "== $label"; grep "^region" $OUT/b_$label.err | sed 's/region [0-9]*: issue //; s/ us, known done [0-9.]*//; s/ us, total /\//; s/ us//' | tr '\n' ' '; echo; python tools/benchline.py $label < $OUT/b_$label.log
}
run fused X=1
run two_launches MMP_NO_FUSE=1
run one_launch MMP_NO_SPLIT=1
