#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp18
mkdir -p $OUT
V=$PWD/modelmesh_amd/lib/variants/libmmplace_emptytail.so
(echo "== first launch + an empty kernel of 16 workgroups"; MMP_XP_EMPTYTAIL=16 MMP_LIB_PATH=$V timeout 300 python tools/r6/host_issue.py
 echo "== first launch + an empty kernel of 3125 workgroups"; MMP_XP_EMPTYTAIL=3125 MMP_LIB_PATH=$V timeout 300 python tools/r6/host_issue.py
 echo "== first launch only, 40 calls"; CALLS_X=2 MMP_SPLIT_NOTAIL=1 timeout 300 python tools/r6/host_issue.py
 echo "== one launch, 40 calls"; CALLS_X=2 MMP_NO_SPLIT=1 timeout 300 python tools/r6/host_issue.py
 echo "== split, 4 hw queues"; GPU_MAX_HW_QUEUES=4 timeout 300 python tools/r6/host_issue.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/host_issue.txt
