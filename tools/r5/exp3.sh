#!/bin/bash
# round 5, experiment 3: DIAGNOSTIC variants of place_batch_kernel (wrong results on purpose): what each step of a decision costs the launch
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/exp3}
mkdir -p $OUT
V=modelmesh_amd/lib/variants
one() {
  local tag=$1 lib=$2; shift 2
  MMP_LIB_PATH=$PWD/$V/libmmplace_$lib.so timeout 600 python bench.py --kernel-only --steps 200 --warmup 20 "$@" > $OUT/$tag.log 2> $OUT/$tag.err
  grep "^{" $OUT/$tag.log | tail -1 > $OUT/$tag.json
  python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print(f"{sys.argv[2]:28s} kernel_us {r.get('kernel_ms', 0) * 1e3:7.2f}  step_us {d.get('ms_per_step', 0) * 1e3:7.2f}  parity {d.get('parity_vs_oracle')}")
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
export MMP_STREAM=0
for v in base d_noclear d_noselect d_norpm d_all d_skeleton d_noreq; do one $v $v --streams 1; done
