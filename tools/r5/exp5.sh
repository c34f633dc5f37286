#!/bin/bash
# round 5, experiment 5: batches decided from the per-type shortlists (place_block_memo) — parity first, then the launch alone with and
# without them (tools/r5/memo_sweep.py), one GPU-box visit.   usage: bash tools/r5/exp5.sh [out_dir]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/exp5}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_shortlist_memo_gpu.py -x -q > $OUT/pytest_memo.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_memo.log; tail -15 $OUT/pytest_memo.log
timeout 400 python tools/r5/memo_sweep.py > $OUT/memo_sweep.txt 2> $OUT/memo_sweep.err; cat $OUT/memo_sweep.txt; tail -3 $OUT/memo_sweep.err
MEMO_SWEEP_FORM=c timeout 400 python tools/r5/memo_sweep.py 800000 > $OUT/memo_sweep_c.txt 2> $OUT/memo_sweep_c.err; cat $OUT/memo_sweep_c.txt; tail -3 $OUT/memo_sweep_c.err
