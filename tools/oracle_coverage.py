#!/usr/bin/env python
"""Branch coverage of the getNext restatement (oracle/mm_oracle.c: place_impl) under the golden fleets — gcov, not a
hand-kept tag list.  Builds an instrumented copy of the oracle (gcc --coverage -O0) in a scratch directory, replays
tests/golden/make_golden.py's FLEETS + the scenario fleets + the reference KAT fleets through it in a child process,
and reports every branch of place_impl / filter_accept / is_excluded / it_peek that was never taken.

usage: python tools/oracle_coverage.py            -> prints the report, exit 1 unless every branch was taken
       from tools.oracle_coverage import uncovered -> list of (line, text) for make_golden.py
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUNCS = ("place_impl", "filter_accept", "is_excluded", "it_peek", "it_next")
# Branches no input can take, by source text (each with its reason); everything else must be taken.
UNREACHABLE = {
    "if (ccount == 0) goto done;": "the Java's `candidates.isEmpty()` guard (:4941-4943): every path that reaches it has added "
                                   "bestIid or a preferred instance; the false edge is the only one that exists",
    "if (pos_of)": "checker plumbing, not the Java: pos_of is NULL only in orc_place_lean (covered by tests/test_lean_port.py)",
    "if (cand_out) memcpy": "checker plumbing: the optional shortlist copy-out of the single-call API",
    "for (int32_t i = 0, j = 0; i < ccount; i++) {": "the index-th survivor always exists (index < remaining), so the loop "
                                                     "of :4982-4986 leaves through its break, never through its condition",
    "if (pos_of) o->hash": "checker plumbing (audit hash)",
}

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %(root)r)
from oracle import bind as ob
ob.LIB = %(lib)r
ob._lib = None
from modelmesh_amd import workload as wl
from tests.golden.make_golden import FLEETS
for seed, profile, pods in FLEETS:
    fleet = wl.fuzz_fleet(seed, pods=pods, models=160, profile=profile)
    reqs, extra = wl.fuzz_requests(fleet, seed, 300)
    ob.OracleFleet(fleet).place(reqs, extra, fleet.now)
for name, fleet, reqs, extra in wl.scenario_fleets():
    ob.OracleFleet(fleet).place(reqs, extra, fleet.now)
for seed in range(12):
    for profile in (None, "full", "prefer"):
        fleet = wl.fuzz_fleet(seed, pods=[1, 5, 40, 64, 130][seed %% 5], models=120, profile=profile)
        reqs, extra = wl.fuzz_requests(fleet, seed, 250)
        ob.OracleFleet(fleet).place(reqs, extra, fleet.now)
ctypes.CDLL(None).__gcov_dump() if hasattr(ctypes.CDLL(None), "__gcov_dump") else None
"""


def uncovered(verbose: bool = False):
    """-> [(line number, source text, gcov note)] of branches in FUNCS never taken; [] = full branch coverage."""
    work = tempfile.mkdtemp(prefix="orc_cov_")
    try:
        srcs = ["mm_oracle.c", "mm_evict_oracle.c", "mm_oracle_batch.c", "mm_gates_oracle.c", "mm_rebalance_oracle.c", "mm_oracle.h"]
        for f in srcs:
            shutil.copy(os.path.join(ROOT, "oracle", f), work)
        lib = os.path.join(work, "liboracle_cov.so")
        subprocess.check_call(["gcc", "--coverage", "-O0", "-std=c11", "-fPIC", "-shared", "-o", lib] +
                              [f for f in srcs if f.endswith(".c")] + ["-lpthread"], cwd=work)
        subprocess.check_call([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib}], cwd=work)
        subprocess.check_call(["gcov", "-b", "-c", "liboracle_cov.so-mm_oracle.gcda"], cwd=work, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        text = open(os.path.join(work, "mm_oracle.c.gcov")).read().splitlines()
    finally:
        if not verbose:
            shutil.rmtree(work, ignore_errors=True)
    # function extents from the source itself
    src = open(os.path.join(ROOT, "oracle", "mm_oracle.c")).read().splitlines()
    spans = []
    for fn in FUNCS:
        start = next(i for i, ln in enumerate(src) if re.match(r"^(static\s+)?[\w\s\*]+\b" + fn + r"\(", ln) and not ln.rstrip().endswith(";"))
        end = next(i for i in range(start, len(src)) if src[i] == "}")
        spans.append((start + 1, end + 1))
    out, cur_line, cur_text = [], 0, ""
    for ln in text:
        m = re.match(r"\s*[-#=\d\*]+:\s*(\d+):(.*)", ln)
        if m:
            cur_line, cur_text = int(m.group(1)), m.group(2).strip()
            continue
        b = re.match(r"branch\s+\d+\s+(never executed|taken 0)", ln)
        if b and any(a <= cur_line <= z for a, z in spans) and not any(cur_text.startswith(k) for k in UNREACHABLE):
            out.append((cur_line, cur_text, ln.strip()))
    return out


if __name__ == "__main__":
    miss = uncovered()
    for line, text, note in miss:
        print(f"oracle/mm_oracle.c:{line}: {note}: {text}")
    print(f"{len(miss)} branch(es) of {', '.join(FUNCS)} never taken")
    sys.exit(1 if miss else 0)
