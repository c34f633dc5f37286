"""CPU: ranking by sampling (rank_sample.hpp) restated in numpy: S sample rows (every P/S-th), sorted; a row's range = how many
samples sort before it; its rank = rows in earlier ranges + rows of its own range that sort before it.  Must be the rank of a
sort from scratch, whatever the sample makes of the ranges (skewed tables included)."""
import numpy as np
import pytest


@pytest.mark.parametrize("seed", range(30))
def test_sampled_ranks_equal_a_sort(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(300, 6000))
    S = 256
    keys = rng.permutation(P * 3)[:P].astype(np.int64)              # distinct: a strict total order
    if seed % 3 == 0:                                              # skew: the sampled rows all large, the rest small
        stride = (np.arange(S) * P) // S
        keys[stride] += 10 * P
    want = np.empty(P, np.int64)
    want[np.argsort(keys, kind="stable")] = np.arange(P)

    samples = np.sort(keys[(np.arange(S) * P) // S])
    range_of = np.searchsorted(samples, keys, side="left")         # first sample that does NOT sort before the row
    sizes = np.bincount(range_of, minlength=S + 1)
    off = np.concatenate([[0], np.cumsum(sizes)])
    got = np.empty(P, np.int64)
    for b in range(S + 1):
        rows = np.flatnonzero(range_of == b)
        k = keys[rows]
        got[rows] = off[b] + (k[None, :] < k[:, None]).sum(axis=1)  # all pairs inside the range
    assert np.array_equal(got, want)
