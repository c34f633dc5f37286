#!/bin/bash
# experiment: the small-launch full-cluster kernels without the workgroup barrier (variant nbl) against the installed library
set -u
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
for v in base nbl base nbl; do
  if [ $v = nbl ]; then export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_nbl.so; else unset MMP_LIB_PATH; fi
  timeout 300 python tools/r5/caller_timing.py 100000 2>&1 | grep "full cluster" | sed "s/^/$v: /"
done | tee $OUT/full_cluster_100k.txt
export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_nbl.so
timeout 900 python -m pytest tests/test_ref_vectors_gpu.py tests/test_place_parity_gpu.py tests/test_place_caller_gpu.py tests/test_types_gpu.py -x -q > $OUT/pytest_nbl.log 2>&1; tail -3 $OUT/pytest_nbl.log
