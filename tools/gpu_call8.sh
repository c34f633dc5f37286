#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for lib in libmmplace.so libmmplace_t8.so; do
echo "== $lib"
KT_GRAPH=0 MMP_LIB_PATH=$PWD/modelmesh_amd/lib/$lib timeout 300 python tools/kernel_time.py C3 2>&1 | grep -v amdgpu.ids | head -6
KT_GRAPH=0 KT_BATCHES=6 MMP_LIB_PATH=$PWD/modelmesh_amd/lib/$lib timeout 300 python tools/kernel_time.py C4 60 2>&1 | grep -v amdgpu.ids | head -6
done
