#!/bin/bash
# round 5, experiment 1: the software-pipelined kernel against place_batch_kernel (parity first, then kernel time alone,
# then the timed region of the bench), one GPU-box visit.   usage: bash tools/r5/exp1.sh [out_dir]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/exp1}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_stream_gpu.py -x -q > $OUT/pytest_stream.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_stream.log; tail -3 $OUT/pytest_stream.log
one() {  # $1 tag; env from the caller; rest: bench args
  local tag=$1; shift
  timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 "$@" > $OUT/$tag.log 2> $OUT/$tag.err
  grep "^{" $OUT/$tag.log | tail -1 > $OUT/$tag.json
  python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print(sys.argv[2], "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "parity", d.get("parity_vs_oracle"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
for m in 0 1 2; do MMP_STREAM=$m one k1_stream$m --streams 1; done
for sl in 512 1024; do MMP_STREAM=2 MMP_STREAM_SLOTS=$sl one k1_stream2_slots$sl --streams 1; done
for m in 0 2; do MMP_STREAM=$m one k4_stream$m --streams 4; done
for m in 0 2; do
  MMP_STREAM=$m timeout 900 python bench.py --steps 20 --warmup 5 --no-pod-axis --no-secondary --no-cpu-baseline > $OUT/bench20_stream$m.log 2> $OUT/bench20_stream$m.err
  grep "^{" $OUT/bench20_stream$m.log | tail -1 | cut -c1-600
done
