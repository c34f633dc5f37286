#!/usr/bin/env python3
"""Generates tests/golden/ref_tcm.npz: what the REFERENCE'S OWN TEXT (oracle/_ref/tcm_harness, built by build.sh from
/root/reference: ModelMesh.handleInstanceTableChange with type constraints, InstanceSetStatsTracker, TypeConstraintManager's
incremental path) holds after the event streams of tests/ref_tcm_cases.py — at every checkpoint: the cluster's stats, clusterState
in its order with every record's ProhibitedTypeSet, per type getCandidateInstances / getPreferredInstances / typeSetStats, the
instance partitions with their stats, instanceSetStats() of the local instance.  Build container only.
usage: python oracle/ref_harness/make_tcm_vectors.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ref_fleets as rf  # noqa: E402
from tests import ref_tcm_cases as tc  # noqa: E402

HARNESS = os.environ.get("MMP_TCM_HARNESS") or os.path.join(ROOT, "oracle", "_ref", "tcm_harness")
OUT = os.environ.get("MMP_TCM_OUT") or os.path.join(ROOT, "tests", "golden", "ref_tcm.npz")


def main():
    out, names = {}, []
    for name, case in tc.cases():
        ids = rf.string_ids(case["fleet"], 200)
        blob = tc.input_blob(case, ids)
        with tempfile.TemporaryDirectory() as td:
            fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            open(fin, "wb").write(blob)
            subprocess.run([HARNESS, fin, fout], check=True)
            words = np.frombuffer(open(fout, "rb").read(), "<i8").copy()
        out[f"{name}/words"] = words
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {case['fleet'].n_pods} instance ids, {len(case['req_bits'])} types, {len(case['events'])} events, "
              f"{int(words[0])} checkpoints, upgradeTracker added/removed {int(words[-4])}/{int(words[-3])}")
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(names)} cases")


if __name__ == "__main__":
    main()
