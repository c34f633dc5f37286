"""CPU: "the caller's own entry inside the shortlist" (place_kernel.hpp: memo_try) restated in numpy and held against the oracle.

The device answers a request whose CALLER stands inside its type's recorded shortlist without walking the list again — in the plain
case (no preference step, the fresh-row break off, the caller not the best instance) the caller is one more candidate of the SAME list:
same count, same audit-hash sum; what changes is favourSelf (MM.java:4931), the rpm rule's classes (:4951-4980 — the caller's entry
carries the snapshot rpm of the BEST instance's class, "the others" the fresh rpm) and that picking it means ABORT_REQUEST (:4989).
Here that derivation runs on the oracle's own observables: the list of a type is read off probe decisions (one pick per index), the
answer for a caller at rank ks of the list is computed from the list alone, and the oracle decides the same requests."""
import numpy as np
import pytest

from modelmesh_amd import workload as wl
from oracle.bind import OracleFleet

SELF, NONE = -2, -1  # MMP_SELF, MMP_NONE
INT_MAX = 2**31 - 1
GOLD = 0x9E3779B1


def rpm_limit(ago, min_rpm):
    """RpmRule::limit (place_kernel.hpp; MM.java:4957-4972) for arrays."""
    ml = np.maximum(min_rpm.astype(np.int64), 100)
    m11 = np.minimum(ml + ml // 10, INT_MAX)
    m15 = np.minimum(ml + ml // 2, INT_MAX)
    wrap = lambda v: ((v + 2**31) % 2**32) - 2**31  # Java int products wrap  # noqa: E731
    m3, m4 = wrap(ml * 3), wrap(ml * 4)
    t = np.full(len(ago), INT_MAX, np.int64)
    a1 = (ago < 5 * 24 * 3600 * 1000) & (ago < 24 * 3600 * 1000)
    t = np.where(a1, m4, t)
    a2 = a1 & (ago < 12 * 60 * 1000)
    t = np.where(a2, np.minimum(m3, t), t)
    a3 = a2 & (ago < 5000)
    t = np.where(a3, np.minimum(m15, t), t)
    a4 = a3 & (ago < -1000)
    return np.where(a4, np.minimum(m11, t), t)


def probe(fleet, model, n, now):
    """n requests for an instance-less model from a caller that is not in the table and has room (fresh-row break off)."""
    from modelmesh_amd._lib import PLACE_REQ
    r = np.zeros(n, PLACE_REQ)
    r["model"] = model
    r["self_pod"] = -1
    r["fresh_lru"] = now
    r["fresh_capacity"] = 10**12
    r["last_used"] = now - 6 * 24 * 3600 * 1000  # older than five days: the rpm rule is off
    return r


@pytest.mark.parametrize("config", ["C2", "C3"])
def test_caller_inside_the_list_decides_as_the_list_says(config):
    fleet = wl.make_fleet(config)
    orc = OracleFleet(fleet)
    now = fleet.now
    rng = np.random.default_rng(3)
    no_extra = np.zeros(0, np.int32)
    m = fleet.models
    empty = (m["n_loaded"] + m["n_failed"]) == 0
    checked = 0
    for t in range(max(fleet.n_types, 1)):
        if fleet.n_types and fleet.has_prefer is not None and fleet.has_prefer[t]:
            continue  # (a preferring type may take the preference step: not the plain case)
        model = int(np.flatnonzero(empty & (np.clip(m["type"], 0, max(fleet.n_types, 1) - 1) == t))[0])
        first = orc.place(probe(fleet, model, 1, now), no_extra, now)
        cc = int(first["n_candidates"][0])
        assert cc >= 3, (t, cc)
        # the list, index by index: pick k * 2^32 / cc (rounded up) selects index k while nothing is taken out
        p = probe(fleet, model, cc, now)
        p["pick"] = (-(-(np.arange(cc, dtype=np.int64) << 32) // cc)).astype(np.uint32)
        lst = orc.place(p, no_extra, now)
        L = lst["chosen"].astype(np.int64)
        assert len(set(L.tolist())) == cc and L[0] == first["best"][0], t
        vhash = int(first["hash"][0]) ^ ((cc * GOLD) & 0xFFFFFFFF)  # the audit-hash sum of the list, folded (the count term taken out)
        b_rpm = int(fleet.pods["rpm"][L[0]])
        # requests whose caller is candidate ks >= 1 of that list
        n = 6000
        ks = rng.integers(1, cc, n)
        r = probe(fleet, model, n, now)
        r["self_pod"] = L[ks]
        r["flags"] = (rng.random(n) < 0.25).astype(np.uint32)
        r["pick"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        r["fresh_rpm"] = rng.choice(np.array([0, 90, 150, 900, 5000, 2_000_000], np.int32), n)
        r["fresh_count"] = fleet.pods["count"][L[ks]]
        age = rng.choice(np.array([-5000, 0, 2000, 60_000, 3_600_000, 2 * 24 * 3600 * 1000, 9 * 24 * 3600 * 1000], np.int64), n)
        r["last_used"] = np.where(age == 0, 0, now - age)
        got = orc.place(r, no_extra, now, threads=8)
        # ---- the derivation (memo_try)
        favour = r["flags"] != 0
        f_rpm = r["fresh_rpm"].astype(np.int64)
        n_others = cc - 2
        mn = np.where((n_others > 0) & (f_rpm < b_rpm), f_rpm, b_rpm)
        lim = rpm_limit(age, mn)
        null0 = (b_rpm >= 100) & (b_rpm > lim)
        null_s = null0
        null_o = (n_others > 0) & (f_rpm >= 100) & (f_rpm > lim)
        remaining = cc - null0.astype(np.int64) - null_s.astype(np.int64) - np.where(null_o, n_others, 0)
        index = np.where(remaining <= 1, 0, (r["pick"].astype(np.uint64) * remaining.astype(np.uint64)) >> np.uint64(32)).astype(np.int64)
        k = np.where(null_o, np.where(index == 0, 0, ks), index + null0)
        k = np.where(~null_o & null_s & (k >= ks), k + 1, k)
        chosen = np.where(remaining >= 1, np.where(k == ks, SELF, L[np.minimum(k, cc - 1)]), NONE)
        want_chosen = np.where(favour, SELF, chosen)
        want_n = np.where(favour, 0, cc)
        want_hash = np.where(favour, 0, vhash ^ ((remaining * GOLD) & 0xFFFFFFFF))
        assert np.array_equal(got["best"], np.full(n, L[0])), t
        for name, want in (("chosen", want_chosen), ("n_candidates", want_n), ("hash", want_hash)):
            bad = np.flatnonzero(got[name].astype(np.int64) != want.astype(np.int64))
            assert len(bad) == 0, (t, name, int(bad[0]), int(got[name][bad[0]]), int(want[bad[0]]), int(ks[bad[0]]), int(age[bad[0]]),
                                   int(f_rpm[bad[0]]))
        checked += n
    assert checked >= 6000
