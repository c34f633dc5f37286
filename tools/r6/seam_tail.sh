#!/bin/bash
# the tail of a single request at the four seams: the same run unpinned, pinned to a CPU, pinned + SCHED_FIFO if granted
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_seam
mkdir -p $OUT
g++ -O2 -std=c++17 -Iinclude tools/micro/seam_tail.cc -Lmodelmesh_amd/lib -lmmplace -Wl,-rpath,$PWD/modelmesh_amd/lib -lpthread -o /tmp/seam_tail || exit 1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for run in "unpinned" "pin 3" "pin 11" "unpinned" "pin 3"; do
  echo "== $run"
  if [ "$run" = unpinned ]; then timeout 300 /tmp/seam_tail 20000; else timeout 300 /tmp/seam_tail 20000 $run; fi 2>&1 | grep -v amdgpu.ids | grep -E "calling thread|n=1|slow calls"
done | tee $OUT/seam_tail_pinning.txt
