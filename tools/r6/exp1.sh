#!/bin/bash
# round 6, first visit: the new shortlist check (in-wave and split) against the oracle / the reference text, then launch times
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp1
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_shortlist_memo_gpu.py tests/test_ref_vectors_gpu.py tests/test_place_parity_gpu.py tests/test_place_caller_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
SWEEP_TAILS=4,64 timeout 900 python tools/r6/split_sweep.py 100000 400000 800000 1600000 > $OUT/sweep_rows.txt 2>&1; cat $OUT/sweep_rows.txt
MEMO_SWEEP_FORM=c MEMO_SWEEP_CALLER=4321 timeout 600 python tools/r6/split_sweep.py 400000 800000 > $OUT/sweep_caller.txt 2>&1; cat $OUT/sweep_caller.txt
cd /tmp
rm -rf /tmp/prof1
SWEEP_ONLY=2 SWEEP_K=100 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/tools/r6/split_sweep.py 800000 > $GRAFT_REPO_ROOT/$OUT/prof_split.log 2>&1
f=$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/split_800k_kernel_stats.csv && head -5 $f | cut -c1-200
