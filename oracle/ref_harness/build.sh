#!/bin/bash
# Builds oracle/_ref/ref_harness: the reference's own load-target / serve-target method bodies (extracted by line range
# from /root/reference at build time, never committed) compiled against the stand-ins of javastub.hpp.
# Needs the reference tree (this container; the GPU box has none and only consumes tests/golden/ref_*.npz).
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
out="$here/../_ref"
mkdir -p "$out/gen"
python3 "$here/extract.py" > "$out/gen/extract.log"
# -fwrapv: Java integer arithmetic wraps; -fno-strict-aliasing is not needed (no type punning in the stand-ins)
#: the listener's switch (MM.java:1474-1563) declares locals in one case and falls through to the next; Java
g++ -std=c++17 -O1 -g -fwrapv -Wall -Wno-unused-variable -Wno-unused-function -Wno-parentheses -Wno-unused-but-set-variable \
    "$here/harness.cc" -o "$out/ref_harness"
# the local cache (clhm + ModelCacheUnloadBufManager): a second binary, same recipe
g++ -std=c++17 -O1 -g -fwrapv -Wall -Wno-unused-variable -Wno-unused-function -Wno-parentheses -Wno-unused-but-set-variable \
    "$here/clhm_harness.cc" -o "$out/clhm_harness"
# the instance-table listener with type constraints + TypeConstraintManager's incremental path: a third binary
g++ -std=c++17 -O1 -g -fwrapv -Wall -Wno-unused-variable -Wno-unused-function -Wno-parentheses -Wno-unused-but-set-variable \
    "$here/tcm_harness.cc" -o "$out/tcm_harness"
# the extracted reference text is an intermediate of this build: it does not stay in the tree (nor travel to the GPU box);
# MANIFEST.txt (file:line ranges) and extract.log do.  KEEP_GEN=1 keeps it (tools/ref_coverage.sh compiles a second time)
if [ -z "${KEEP_GEN:-}" ]; then rm -f "$out"/gen/*.inc; fi
echo "built $out/ref_harness, $out/clhm_harness and $out/tcm_harness from: $(tr '\n' ';' < "$out/gen/MANIFEST.txt")"
