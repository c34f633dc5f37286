"""Helpers shared by the parity tests."""
import numpy as np


def describe_mismatch(fleet, reqs, got, want, limit=8):
    """Human-readable dump of the first differing decisions (shows up in the GPU log)."""
    bad = np.nonzero((got["chosen"] != want["chosen"]) | (got["best"] != want["best"]) |
                     (got["n_candidates"] != want["n_candidates"]) | (got["hash"] != want["hash"]))[0]
    lines = [f"{len(bad)} of {len(reqs)} decisions differ"]
    for i in bad[:limit]:
        r = reqs[i]
        m = fleet.models[r["model"]]
        lines.append(
            f"  req {i}: model={r['model']} type={m['type']} k={m['n_loaded']} f={m['n_failed']} self={r['self_pod']} "
            f"flags={r['flags']} n_extra={r['n_extra']} last_used={r['last_used']} pick={r['pick']}\n"
            f"     got  chosen={got[i]['chosen']} best={got[i]['best']} n={got[i]['n_candidates']} hash={got[i]['hash']:#x}\n"
            f"     want chosen={want[i]['chosen']} best={want[i]['best']} n={want[i]['n_candidates']} hash={want[i]['hash']:#x}")
    return "\n".join(lines)


def assert_same_decisions(fleet, reqs, got, want):
    same = (np.array_equal(got["chosen"], want["chosen"]) and np.array_equal(got["best"], want["best"]) and
            np.array_equal(got["n_candidates"], want["n_candidates"]) and np.array_equal(got["hash"], want["hash"]))
    assert same, describe_mismatch(fleet, reqs, got, want)
