#!/usr/bin/env python3
"""Generates tests/golden/ref_getnext.npz: what the REFERENCE'S OWN TEXT decides (oracle/_ref/ref_harness, built by
build.sh from /root/reference) on the fleets of tests/ref_fleets.py.  Run in the build container (the GPU box has no
reference tree and only reads the committed file).

Per load-target case: the clusterState iteration order, and per request (chosen, candidates.size(), survivors of the rpm
filter, audit hash of the shortlist).  Per serve-target case: (chosen, chosenTimeStamp).  Per guard case: (MMP_GATE_* bits,
loadLocal's initial size).  Plus a digest of each case's inputs.
usage: python oracle/ref_harness/make_ref_vectors.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ref_fleets as rf  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden", "ref_getnext.npz")


def run(blob: bytes, n_place: int, n_serve: int, n_gate: int = 0):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        open(fin, "wb").write(blob)
        subprocess.run([HARNESS, fin, fout], check=True)
        raw = open(fout, "rb").read()
    n_present = int(np.frombuffer(raw, "<i8", 1)[0])
    off = 8
    order = np.frombuffer(raw, "<i4", n_present, off).copy()
    off += 4 * n_present
    place = np.frombuffer(raw, "<i4", 4 * n_place, off).reshape(n_place, 4).copy()
    off += 16 * n_place
    serve = np.frombuffer(raw, "<i8", 2 * n_serve, off).reshape(n_serve, 2).copy()
    off += 16 * n_serve
    gate = np.frombuffer(raw, "<i4", 2 * n_gate, off).reshape(n_gate, 2).copy()
    assert off + 8 * n_gate == len(raw)
    return order, place, serve, gate


def main():
    subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref_harness", "build.sh")], check=True)
    out = {}
    names = []
    for name, fleet, ids, reqs, extra in rf.place_cases():
        blob = rf.input_blob(fleet, ids, reqs, extra)
        order, place, _, _ = run(blob, len(reqs), 0)
        out[f"{name}/order"], out[f"{name}/place"] = order, place
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(order)} instances in clusterState, {len(reqs)} decisions: "
              f"{int((place[:, 0] >= 0).sum())} remote, {int((place[:, 0] == -2).sum())} self, {int((place[:, 0] == -1).sum())} none; "
              f"mean shortlist {place[:, 1].mean():.1f}")
    for name, fleet, ids, reqs, in_use, last_used, xp, xt in rf.serve_cases():
        blob = rf.input_blob(fleet, ids, serve=(reqs, in_use, last_used, xp, xt))
        _, _, serve, _ = run(blob, 0, len(reqs))
        out[f"{name}/serve"] = serve
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        print(f"{name}: {len(reqs)} serve decisions: {int((serve[:, 0] >= 0).sum())} remote, {int((serve[:, 0] == -2).sum())} self")
    from oracle import bind as ob
    for name, fleet, ids, reqs, xp, xt, expl, expiry in rf.gate_cases():
        tstats = np.ascontiguousarray(ob.type_set_stats(fleet))  # an INPUT of the guards (rows a5 / a18 are pinned separately)
        blob = rf.input_blob(fleet, ids, gates=(reqs, xp, xt, expl, expiry, tstats))
        _, _, _, gate = run(blob, 0, 0, len(reqs))
        out[f"{name}/gate"] = gate
        out[f"{name}/digest"] = np.frombuffer(rf.digest(blob).encode(), np.uint8)
        names.append(name)
        bits = gate[:, 0].astype(np.uint32)
        print(f"{name}: {len(reqs)} guard evaluations; fired: " + " ".join(f"{b}:{int(((bits >> k) & 1).sum())}" for k, b in enumerate(
            ["goLocal", "failures", "locations", "notAllowed", "churn", "earlyReject", "reload", "publish"])))
    out["names"] = np.array(names)
    out["manifest"] = np.array(open(os.path.join(ROOT, "oracle", "_ref", "gen", "MANIFEST.txt")).read())
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(names)} cases")


if __name__ == "__main__":
    main()
