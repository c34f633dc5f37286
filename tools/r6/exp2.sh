#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp2
mkdir -p $OUT
SWEEP_NOTAIL=1 SWEEP_K=100 timeout 900 python tools/r6/split_sweep.py 100000 400000 800000 1600000 > $OUT/sweep_rows.txt 2>&1; cat $OUT/sweep_rows.txt
cd /tmp
rm -rf /tmp/prof1
SWEEP_ONLY=2 SWEEP_K=100 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/tools/r6/split_sweep.py 800000 > $GRAFT_REPO_ROOT/$OUT/prof_split.log 2>&1
f=$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/split_800k_kernel_stats.csv && head -4 $f | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_shortlist_memo_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_parity_gpu.py tests/test_place_caller_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
