#!/bin/bash
# round-2 GPU visit 1: parity with and without head records, launch times, bench at the driver's and the default step counts
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2a
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
MMP_NO_HEADS=1 timeout 600 python -m pytest tests/test_place_parity_gpu.py -m gpu -x -q > $OUT/pytest_noheads.log 2>&1; echo "pytest(no heads) exit $?" >> $OUT/pytest_noheads.log; tail -2 $OUT/pytest_noheads.log
timeout 300 python tools/kernel_time.py C3 > $OUT/kernel_time_heads.txt 2>&1; cat $OUT/kernel_time_heads.txt
MMP_NO_HEADS=1 timeout 300 python tools/kernel_time.py C3 > $OUT/kernel_time_noheads.txt 2>&1; cat $OUT/kernel_time_noheads.txt
timeout 200 python tools/phase_clock.py 30 > $OUT/phase_clock.txt 2>&1; cat $OUT/phase_clock.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_s20.json.log 2> $OUT/bench_s20.err; echo "bench(20) exit $?"; python tools/benchline.py s20 < $OUT/bench_s20.json.log
timeout 300 python bench.py --kernel-only --steps 1000 --warmup 50 > $OUT/bench_s1000.json.log 2> $OUT/bench_s1000.err; echo "bench(1000) exit $?"; python tools/benchline.py s1000 < $OUT/bench_s1000.json.log
