"""CPU: the arithmetic of the insertion re-rank (snapshot.hpp: delta_scatter_kernel; mmplace.hip: the host's binary searches),
restated in numpy and held to a full sort.  A table under a strict total order; K rows change their keys; the unchanged rows keep
their relative order, so with  removed = the changed rows' old positions (ascending)  and  ins[k] = how many UNCHANGED rows sort
before changed row k:
    unchanged row at old position q:  j = q - #{removed < q},  rank = j + #{ins <= j}
    changed row k:                    rank = ins[k] + #{changed rows that sort before it}
must be the rank of a sort from scratch."""
import numpy as np
import pytest


@pytest.mark.parametrize("seed", range(40))
def test_insertion_ranks_equal_a_sort_from_scratch(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(2, 400))
    K = int(rng.integers(0, min(16, P) + 1))
    keys = rng.permutation(P * 4)[:P].astype(np.int64)         # distinct keys = a strict total order
    order = np.argsort(keys, kind="stable")                    # position -> row
    pos_of = np.empty(P, np.int64)
    pos_of[order] = np.arange(P)
    chg = rng.choice(P, size=K, replace=False)
    new_keys = keys.copy()
    free = np.setdiff1d(np.arange(P * 4), keys)
    new_keys[chg] = rng.choice(free, size=K, replace=False)    # still distinct
    want = np.empty(P, np.int64)
    want[np.argsort(new_keys, kind="stable")] = np.arange(P)

    removed = np.sort(pos_of[chg])
    unchanged_rows = order[~np.isin(order, chg)]               # the old order with the changed rows taken out
    unchanged_keys = new_keys[unchanged_rows]                  # (their keys did not change)
    ins = np.searchsorted(unchanged_keys, new_keys[chg])       # per changed row: unchanged rows before it (binary search)
    got = np.empty(P, np.int64)
    for k, row in enumerate(chg):
        got[row] = ins[k] + np.sum(new_keys[chg] < new_keys[row])
    for row in range(P):
        if row in chg:
            continue
        q = pos_of[row]
        j = q - np.sum(removed < q)
        got[row] = j + np.sum(ins <= j)
    assert np.array_equal(got, want)
