#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2b
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/micro/floor.hip -o /tmp/floor && /tmp/floor > $OUT/floor.txt 2>&1; cat $OUT/floor.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/floorprof -- /tmp/floor > /dev/null 2>&1; cd - > /dev/null
f=$(find /tmp/floorprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/floor_kernel_stats.csv && cut -d, -f1-8 $OUT/floor_kernel_stats.csv | cut -c1-200
export SQ_KERNEL=place_batch_kernel
export SQ_CMD="python bench.py --kernel-only --steps 20 --warmup 2 --streams 1"
OUTSAVE=$OUT
bash tools/gpu_round.sh sq > $OUT/sq.txt 2>&1; cat $OUT/sq.txt | tail -20
