#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp16
mkdir -p $OUT
(echo "== split"; timeout 300 python tools/r6/host_issue.py; echo "== split, no tail"; MMP_SPLIT_NOTAIL=1 timeout 300 python tools/r6/host_issue.py; echo "== one launch (MMP_NO_SPLIT=1)"; MMP_NO_SPLIT=1 timeout 300 python tools/r6/host_issue.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/host_issue.txt
