#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp17
mkdir -p $OUT
(echo "== split, tail at 3 wavefronts per SIMD (no scratch)"; MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_taileu3.so timeout 300 python tools/r6/host_issue.py; echo "== split"; timeout 300 python tools/r6/host_issue.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/host_issue.txt
