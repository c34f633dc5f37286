#!/bin/bash
# per-kernel durations of the two-launch shortlist batch (rocprofv3 kernel trace of tools/r5/memo_sweep.py 800000)
set -u
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o memo -- python $GRAFT_REPO_ROOT/tools/r5/memo_sweep.py 800000 > $GRAFT_REPO_ROOT/$OUT/sweep.txt 2> $GRAFT_REPO_ROOT/$OUT/sweep.err
cd $GRAFT_REPO_ROOT
cat $OUT/sweep.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200
cp $f $OUT/kernel_stats.csv
rm -rf $OUT/prof
