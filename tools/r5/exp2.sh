#!/bin/bash
# round 5, experiment 2: library variants x env switches, kernel time alone (bench.py --kernel-only, one stream) and four streams.
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/exp2}
mkdir -p $OUT
V=modelmesh_amd/lib/variants
one() {  # $1 tag, $2 lib; env from the caller; rest: bench args
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
MMP_STREAM=0 one base base --streams 1
MMP_STREAM=0 MMP_XMAP=1 one base_xmap base --streams 1
MMP_STREAM=0 one nt nt --streams 1
MMP_STREAM=0 MMP_XMAP=1 one nt_xmap nt --streams 1
MMP_STREAM=0 one nohash nohash --streams 1
MMP_STREAM=0 one winmul winmul --streams 1
MMP_STREAM=0 MMP_XMAP=1 one ntwm_xmap ntwm --streams 1
MMP_STREAM=1 one s3_mode1 s3 --streams 1
MMP_STREAM=2 one s3_mode2 s3 --streams 1
MMP_STREAM=2 MMP_STREAM_SLOTS=512 one s3_mode2_512 s3 --streams 1
MMP_STREAM=0 one base_4s base --streams 4
MMP_STREAM=0 MMP_XMAP=1 one ntwm_xmap_4s ntwm --streams 4
MMP_STREAM=2 one s3_mode2_4s s3 --streams 4
