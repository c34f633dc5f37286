#!/bin/bash
# round 5, experiment 4: are the gathers into the resolved registry view (3.2 MB, L2 misses served by the Infinity Cache) what limits
# the 800k launch?  (a) the registry-slice order (MMP_XMAP=1): FETCH_SIZE and time; (b) every request naming one of 1024 models.
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/exp4}
mkdir -p $OUT
export MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_base.so MMP_STREAM=0
for x in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p4_$x$c
    MMP_XMAP=$x timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p4_$x$c -- python bench.py --kernel-only --steps 20 --warmup 2 --streams 1 > /dev/null 2>&1
  done
  python tools/pmc_summary.py /tmp/p4_${x}FETCH_SIZE /tmp/p4_${x}WRITE_SIZE place_batch_kernel $OUT/pmc_xmap$x.json > /dev/null; cut -c1-500 $OUT/pmc_xmap$x.json | tr '\n' ' '; echo
done
python tools/r5/nsweep.py 800000 2>&1 | grep "^n " | sed 's/^/all models:      /' | tee $OUT/models_mod.txt
NSWEEP_MODELS_MOD=1024 python tools/r5/nsweep.py 800000 2>&1 | grep "^n " | sed 's/^/1024 models only: /' | tee -a $OUT/models_mod.txt
MMP_XMAP=1 python tools/r5/nsweep.py 800000 2>&1 | grep "^n " | sed 's/^/registry slices: /' | tee -a $OUT/models_mod.txt
