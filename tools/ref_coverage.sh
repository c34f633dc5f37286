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
for f in *.inc.gcov; do mv "$f" "h12_$f"; done   # (bodies shared with the third binary are counted where they executed most)
# the listener with type constraints + TypeConstraintManager's incremental path: the third binary, tests/golden/ref_tcm.npz
g++ -std=c++17 -O0 -g -fwrapv --coverage -w "$root/oracle/ref_harness/tcm_harness.cc" -o tcm_harness_cov
MMP_TCM_HARNESS="$tmp/tcm_harness_cov" MMP_TCM_OUT="$tmp/out_tcm.npz" python3 "$root/oracle/ref_harness/make_tcm_vectors.py" > gen_tcm.log
gcov tcm_harness_cov-tcm_harness.gcda >> gcov.log 2>&1
# a body compiled into two binaries (PLACEMENT_ORDER, isFull, InstanceSetStatsTracker, the listener's text as tcmi_listener_body):
# a line counts as executed if either binary executed it
python3 - <<'PY'
import glob, os, re
for f in glob.glob("*.inc.gcov"):
    if f.startswith("h12_"): continue
    g = "h12_" + f
    if not os.path.exists(g):
        continue
    a, b = open(f).read().split("\n"), open(g).read().split("\n")
    if len(a) != len(b):
        os.remove(f); continue   # (different instantiation: keep the first binaries' view)
    out = []
    for x, y in zip(a, b):
        out.append(y if re.match(r"^\s+#####:", x) else x)
    open(g, "w").write("\n".join(out)); os.remove(f)
for g in glob.glob("h12_*.inc.gcov"):
    os.rename(g, g[4:])
PY
{
  echo "# reference text executed by tests/golden/ref_getnext.npz ($(tail -1 gen.log | sed 's/.*: //')) tests/golden/ref_clhm.npz ($(tail -1 gen_clhm.log | sed 's/.*: //')) and tests/golden/ref_tcm.npz ($(tail -1 gen_tcm.log | sed 's/.*: //'))"
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
  cat "$root/tools/ref_coverage_residual.txt"
} > "$out"
