#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2f
mkdir -p $OUT
for iss in 1 2 4; do
timeout 300 python bench.py --kernel-only --steps 1000 --warmup 50 --issuers $iss --streams 16 > $OUT/bench_k_i$iss.json.log 2> $OUT/bench_k_i$iss.err; python tools/benchline.py "issuers=$iss streams=16" < $OUT/bench_k_i$iss.json.log
done
timeout 300 python bench.py --kernel-only --steps 1000 --warmup 50 --issuers 2 --streams 4 > $OUT/bench_k_i2s4.json.log 2>/dev/null; python tools/benchline.py "issuers=2 streams=4" < $OUT/bench_k_i2s4.json.log
timeout 300 python bench.py --kernel-only --steps 1000 --warmup 50 --issuers 1 --streams 4 > $OUT/bench_k_i1s4.json.log 2>/dev/null; python tools/benchline.py "issuers=1 streams=4" < $OUT/bench_k_i1s4.json.log
timeout 300 python bench.py --kernel-only --steps 20 --warmup 5 --streams 4 > $OUT/bench_k20_s4.json.log 2>/dev/null; python tools/benchline.py "steps=20 streams=4" < $OUT/bench_k20_s4.json.log
timeout 300 python bench.py --kernel-only --steps 20 --warmup 5 --streams 16 > $OUT/bench_k20_s16.json.log 2>/dev/null; python tools/benchline.py "steps=20 streams=16" < $OUT/bench_k20_s16.json.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_s20.json.log 2> $OUT/bench_s20.err; echo "bench(20) exit $?"; python tools/benchline.py s20 < $OUT/bench_s20.json.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2f/bench_s20.json.log").read().strip().splitlines()[-1])
print(json.dumps(d["roofline"])[:900]); print(json.dumps(d["cpu_baseline"])[:600]); print({k:(v if not isinstance(v,(dict,list)) else "...") for k,v in d.items()})
PY
