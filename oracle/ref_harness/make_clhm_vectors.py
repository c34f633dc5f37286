#!/usr/bin/env python3
"""Generates tests/golden/ref_clhm.npz: what the REFERENCE'S OWN local-cache text (oracle/_ref/clhm_harness, built by build.sh
from /root/reference: clhm/ConcurrentLinkedHashMap.java, clhm/LinkedDeque.java, ModelCacheUnloadBufManager.java) does on the
operation streams of tests/ref_clhm_cases.py — per operation: result, evicted keys in listener order, unload-buffer weight,
weightedSize(), oldestTime(); per cache at the end: the eviction deque (key, weight, lastUsed), capacity, the manager's fields.
Run in the build container (the GPU box has no reference tree and only reads the committed file).
usage: python oracle/ref_harness/make_clhm_vectors.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ref_clhm_cases as rc  # noqa: E402

HARNESS = os.environ.get("MMP_CLHM_HARNESS") or os.path.join(ROOT, "oracle", "_ref", "clhm_harness")
OUT = os.environ.get("MMP_CLHM_OUT") or os.path.join(ROOT, "tests", "golden", "ref_clhm.npz")
OP_OUT = np.dtype([("result", "<i4"), ("n_evicted", "<i4"), ("evicted_off", "<i4"), ("buffer_weight", "<i4"),
                   ("weighted_size", "<i8"), ("oldest_time", "<i8")])


def main():
    out, names = {}, []
    for name, caps, reserved, ops in rc.cases():
        cut = 0
        while True:  # the domain rule: a cache's stream ends where the reference would evict the pinned unload-buffer entry
            res = run_case(caps, reserved, ops)
            outs, ev = res[0], res[1]
            bad = {}
            for i, o in enumerate(outs):
                if rc.UNLOADBUF_KEY in ev[o["evicted_off"]: o["evicted_off"] + o["n_evicted"]]:
                    bad.setdefault(int(ops[i]["cache"]), i)
            if not bad:
                break
            keep = np.array([not (int(op["cache"]) in bad and i >= bad[int(op["cache"])]) for i, op in enumerate(ops)])
            cut += int((~keep).sum())
            ops = ops[keep]
        outs, ev, hdr, keys, wts, lus = res
        assert (outs["weighted_size"] >= 0).all()
        out[f"{name}/caps"], out[f"{name}/reserved"], out[f"{name}/ops"] = caps, reserved, ops
        out[f"{name}/outs"], out[f"{name}/evicted"] = outs, ev
        out[f"{name}/final_hdr"] = hdr
        out[f"{name}/final_key"] = np.concatenate(keys) if keys else np.zeros(0, np.int32)
        out[f"{name}/final_weight"] = np.concatenate(wts) if wts else np.zeros(0, np.int32)
        out[f"{name}/final_last_used"] = np.concatenate(lus) if lus else np.zeros(0, np.int64)
        names.append(name)
        print(f"{name}: {len(caps)} caches ({int((reserved >= 0).sum())} managed), {len(ops)} operations ({cut} cut), "
              f"{int(outs['n_evicted'].sum())} evictions, {int((hdr[:, 0]).sum())} entries at the end")
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(names)} cases")


def run_case(caps, reserved, ops):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(b"MMCLHM1\0")
            f.write(np.array([len(caps), len(ops), rc.NOW], "<i8").tobytes())
            f.write(caps.astype("<i8").tobytes() + reserved.astype("<i4").tobytes() + ops.tobytes())
        subprocess.run([HARNESS, fin, fout], check=True)
        raw = open(fout, "rb").read()
    n = len(ops)
    outs = np.frombuffer(raw, OP_OUT, n).copy()
    off = 32 * n
    ne = int(np.frombuffer(raw, "<i8", 1, off)[0])
    off += 8
    ev = np.frombuffer(raw, "<i4", ne, off).copy()
    off += 4 * ne
    hdr, keys, wts, lus = [], [], [], []
    for _ in range(len(caps)):
        h = np.frombuffer(raw, "<i8", 7, off).copy()  # entries, capacity, weightedSize, totalUnloadingWeight, totalModelCacheOccupancy, cacheDeficit, data.size()
        off += 56
        k = int(h[0])
        hdr.append(h)
        keys.append(np.frombuffer(raw, "<i4", k, off).copy())
        off += 4 * k
        wts.append(np.frombuffer(raw, "<i4", k, off).copy())
        off += 4 * k
        lus.append(np.frombuffer(raw, "<i8", k, off).copy())
        off += 8 * k
    assert off == len(raw)
    return outs, ev, np.stack(hdr), keys, wts, lus


if __name__ == "__main__":
    main()
