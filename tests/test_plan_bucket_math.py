"""CPU: the key-range maps of the reaper's bucketed plan (rebalance_kernels.hpp: plan_bucket_linear / plan_bucket_floating),
restated in Python.  What the plan relies on: both maps are MONOTONE in age = newest - lastUsed (so buckets are in TreeSet order and
equal keys share a bucket), stay below kPlanBuckets, and the floating map's buckets are at most 1/256 of their age wide."""
import numpy as np
import pytest

K_BUCKETS, LIN_BITS, MANT = 16384, 14, 8


def linear(age: int, rng_: int) -> int:
    bits = rng_.bit_length()
    return age >> (bits - LIN_BITS if bits > LIN_BITS else 0)


def floating(age: int) -> int:
    if age < (1 << MANT):
        return age
    e = age.bit_length() - 1
    return ((e - MANT + 1) << MANT) + ((age >> (e - MANT)) & ((1 << MANT) - 1))


@pytest.mark.parametrize("seed", range(20))
def test_maps_are_monotone_and_bounded(seed):
    rng = np.random.default_rng(seed)
    top = int(rng.choice([300, 1 << 20, 1 << 40, (1 << 64) - 1]))
    ages = sorted({0, 1, 255, 256, 257, top} | {int(x) for x in rng.integers(0, top, 2000, dtype=np.uint64)})
    fl = [floating(a) for a in ages]
    li = [linear(a, top) for a in ages]
    assert fl == sorted(fl) and li == sorted(li)
    assert max(fl) < K_BUCKETS and max(li) < K_BUCKETS
    for a in ages:  # the bucket of a holds only ages within a / 256 of it
        if a >= (1 << MANT):
            e = a.bit_length() - 1
            width = 1 << (e - MANT)
            assert floating(a) == floating(a - (a % width)) and floating(a - (a % width) + width - 1) == floating(a)
            assert width * 256 <= 2 * a


def test_extremes():
    assert floating((1 << 64) - 1) == ((63 - MANT + 1) << MANT) + 255 < K_BUCKETS
    assert linear((1 << 64) - 1, (1 << 64) - 1) == K_BUCKETS - 1
    assert linear(0, 0) == 0 and floating(0) == 0
