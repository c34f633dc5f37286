#!/bin/bash
# Run the parity tests with an AddressSanitizer build of the CPU oracle (the checker is C: an out-of-bounds
# read there would silently weaken the parity claim).  usage: bash tools/asan_oracle.sh [pytest args]
# (default: the CPU suite, tests -q -m "not gpu"; the regular build of the oracle is restored however the run ends)
set -u
cd "$(dirname "$0")/.."
cp oracle/liboracle.so /tmp/liboracle.keep
trap 'cp /tmp/liboracle.keep oracle/liboracle.so' EXIT
[ $# -eq 0 ] && set -- tests -q -m "not gpu"
(cd oracle && gcc -O1 -g -fsanitize=address -fno-omit-frame-pointer -shared -fPIC -o liboracle.so mm_oracle.c mm_evict_oracle.c \
   mm_gates_oracle.c mm_rebalance_oracle.c mm_oracle_batch.c -lpthread)
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 python -m pytest -p no:cacheprovider "$@"
exit $?
