#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_exp21
mkdir -p $OUT
for ns in 4 5 6 8 10; do
  echo "== split, $ns streams"; NSTREAMS=$ns timeout 300 python tools/r6/host_issue.py 2>&1 | grep -E "region of|known done"
done 2>&1 | tee $OUT/streams.txt
