#!/bin/bash
# Which lines of the REFERENCE'S OWN TEXT (the method bodies oracle/ref_harness/extract.py pulls from /root/reference) do the
# committed vectors execute? Builds the harness with --coverage in a scratch directory, replays every case of
# tests/ref_fleets.py through it (make_ref_vectors.py with MMP_REF_HARNESS / MMP_REF_OUT, so the committed .npz is untouched)
# and prints executed/executable lines per extracted body plus the lines never reached.
# Needs /root/reference (this container only).    usage: tools/ref_coverage.sh [out.txt]
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/dev/stdout}; case "$out" in /*) ;; *) out="$PWD/$out";; esac
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
KEEP_GEN=1 bash "$root/oracle/ref_harness/build.sh" > /dev/null   # (re)extracts oracle/_ref/gen/*.inc and keeps them for the second compile
trap 'rm -rf "$tmp"; rm -f "$root"/oracle/_ref/gen/*.inc' EXIT
cd "$tmp"
g++ -std=c++17 -O0 -g -fwrapv --coverage -w "$root/oracle/ref_harness/harness.cc" -o ref_harness_cov
MMP_REF_HARNESS="$tmp/ref_harness_cov" MMP_REF_OUT="$tmp/out.npz" python3 "$root/oracle/ref_harness/make_ref_vectors.py" > gen.log
gcov ref_harness_cov-harness.gcda > gcov.log 2>&1
# the local cache's text (clhm + ModelCacheUnloadBufManager): the second binary, the streams of tests/golden/ref_clhm.npz
g++ -std=c++17 -O0 -g -fwrapv --coverage -w "$root/oracle/ref_harness/clhm_harness.cc" -o clhm_harness_cov
MMP_CLHM_HARNESS="$tmp/clhm_harness_cov" MMP_CLHM_OUT="$tmp/out_clhm.npz" python3 "$root/oracle/ref_harness/make_clhm_vectors.py" > gen_clhm.log
gcov clhm_harness_cov-clhm_harness.gcda >> gcov.log 2>&1
{
  echo "# reference text executed by tests/golden/ref_getnext.npz ($(tail -1 gen.log | sed 's/.*: //')) and tests/golden/ref_clhm.npz ($(tail -1 gen_clhm.log | sed 's/.*: //'))"
  echo "# body (oracle/_ref/gen/<name>.inc; source range in oracle/ref_harness/extract.py)   executed/executable lines"
  tot=0; hit=0
  for f in *.inc.gcov; do
    t=$(grep -cE '^\s+([0-9]+\*?|#####):' "$f" || true); m=$(grep -cE '^\s+#####:' "$f" || true)
    printf '%-44s %4d/%-4d\n' "${f%.inc.gcov}" $((t-m)) "$t"
    tot=$((tot+t)); hit=$((hit+t-m))
  done
  printf '%-44s %4d/%-4d\n' "TOTAL" "$hit" "$tot"
  echo
  echo "# never executed (body:line-in-body: text)"
  grep -E '^\s+#####:' *.inc.gcov | sed -E 's/\.inc\.gcov:\s+#####:\s*([0-9]+):\s*/:\1: /' | cut -c1-160
} > "$out"
