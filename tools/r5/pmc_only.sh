#!/bin/bash
# the two HBM PMC passes (FETCH_SIZE, WRITE_SIZE) of the three launches bench.py takes roofline.traffic from, on the tree as it stands
# (the summaries are stamped with tools/kernel_hash.py).   usage: bash tools/r5/pmc_only.sh out_dir
set -u
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
python tools/kernel_hash.py > $OUT/kernel_source_hash.txt
one() {
  local tag=$1 kern=$2; shift 2
  rm -rf /tmp/p_$tag
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_$tag/f -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_$tag/w -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 "$@" > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/p_$tag/f /tmp/p_$tag/w $kern $OUT/pmc_place_batch_$tag.json > /dev/null; cut -c1-300 $OUT/pmc_place_batch_$tag.json | tr '\n' ' '; echo
}
one C3_800k place_batch_m_kernel --workload C3
one C3_100k place_batch_kernel --workload C3 --decisions-per-step 100000
one C3_full_cluster_100k place_batch_long_kernel --workload C3 --decisions-per-step 100000 --full-cluster
# one C3_full_cluster_800k place_batch_long4_kernel --workload C3 --full-cluster   (not read by bench.py)
