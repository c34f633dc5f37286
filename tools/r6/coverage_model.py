"""CPU, numpy: which requests of the bench's C3 / C4 batches the per-type recorded shortlists answer (place_kernel.hpp: memo_try),
and what the rest needs — by kind.  Restates the window walk of lane_decide_win for a request WITHOUT positions of its own on the
oracle's order, then classifies every request by where its own positions (caller, the model's instances, its own exclusions) fall.
usage: python tools/r6/coverage_model.py [C3|C4] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from modelmesh_amd import workload as wl  # noqa: E402
from oracle.bind import OracleFleet, unpack_bitmap  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xBE7C0
fleet = wl.make_fleet(name)
orc = OracleFleet(fleet)
order = np.asarray(orc.order)
P = fleet.n_pods
pos_of = np.empty(P, np.int64)
pos_of[order] = np.arange(P)
pods = fleet.pods[order]  # in rank order
rem = pods["capacity"].astype(np.int64) - pods["used"].astype(np.int64)
cnt = pods["count"].astype(np.int64)
T = max(fleet.n_types, 1)
al = np.ones((T, P), bool)
pf = np.zeros((T, P), bool)
if fleet.n_types:
    al = unpack_bitmap(fleet.allowed, P).astype(bool)
    pf = unpack_bitmap(fleet.prefer, P).astype(bool)
    for t in range(T):
        if not fleet.has_allowed[t]:
            al[t] = True
full = rem < fleet.min_space_units
rec = []
for t in range(T):
    E = al[t][order]
    has_pm = bool(fleet.n_types and fleet.has_prefer[t])
    Pm = pf[t][order] if has_pm else np.ones(P, bool)
    D = E & Pm
    best0 = int(np.flatnonzero(E)[0])
    bestpos, plain = best0, True
    if has_pm and not Pm[best0]:
        assert not full[best0], "case (b)"
        q1 = int(np.flatnonzero(D & (np.arange(P) > best0))[0])
        assert not (E & full)[best0 + 1:q1].any()
        bestpos, plain = q1, False
    b_cnt = cnt[bestpos]
    thr = b_cnt + (b_cnt >> 2)
    Tt = 10 if thr < 9 else thr + 1
    start = bestpos + 1
    bf = bool(full[bestpos])
    after = np.flatnonzero(D & (np.arange(P) >= start))
    if not bf:
        lo_t = int(np.flatnonzero((cnt >= Tt) & (np.arange(P) >= start))[0])
        end0 = int(after[after >= lo_t][0])
    else:
        lo_t, end0 = P, P
    end1 = int(after[0])
    list0 = np.concatenate([[bestpos], after[after < end0]])
    rec.append(dict(t=t, best0=best0, bestpos=bestpos, plain=plain, lo_t=lo_t, end0=end0, end1=end1, n0=len(list0), full=bf, D=D, b_rem=rem[bestpos]))
    print(f"type {t}: best0 {best0} bestpos {bestpos} plain {plain} full {bf} cnt {b_cnt} T {Tt} lo_t {lo_t} end0 {end0} n0 {len(list0)} end1 {end1}")

reqs, extra = wl.make_requests(fleet, seed=seed)
n = len(reqs)
m = fleet.models[reqs["model"]]
t = np.clip(m["type"], 0, T - 1)
lo = np.array([r["best0"] for r in rec])[t]
bp = np.array([r["bestpos"] for r in rec])[t]
lo_t = np.array([r["lo_t"] for r in rec])[t]
end0 = np.array([r["end0"] for r in rec])[t]
end1 = np.array([r["end1"] for r in rec])[t]
b_rem = np.array([r["b_rem"] for r in rec])[t]
plain = np.array([r["plain"] for r in rec])[t]
f_rem = reqs["fresh_capacity"].astype(np.int64) - reqs["fresh_used"].astype(np.int64)
nsb = (f_rem < fleet.min_space_units) | (f_rem < (b_rem >> 2))
hi = np.where(nsb, end1, end0) + 1
sp = np.where(reqs["self_pod"] >= 0, pos_of[np.maximum(reqs["self_pod"], 0)], -1)
tot = (m["n_loaded"] + m["n_failed"]).astype(np.int64)
ex = np.full((n, 10), -1, np.int64)
for j in range(6):
    p = pos_of[fleet.ent_pod[np.minimum(m["ent_off"] + j, len(fleet.ent_pod) - 1)]]
    ex[:, j] = np.where(tot > j, p, -1)
for j in range(4):
    p = pos_of[extra[np.minimum(reqs["extra_off"] + j, len(extra) - 1)]]
    ex[:, 6 + j] = np.where(reqs["n_extra"] > j, p, -1)
inr = (ex >= lo[:, None]) & (ex < hi[:, None])
self_in = (sp >= lo) & (sp < hi)
print(f"{name}: n {n}; nsb share {nsb.mean():.4f}; favour {np.mean(reqs['flags'] != 0):.3f}")
cov_r4 = ~self_in & ~inr.any(1) & (tot <= 6) & (reqs["n_extra"] <= 4)
print(f"covered (no own position in [lo, hi)): {cov_r4.mean():.5f}  uncovered {n - cov_r4.sum()}")
unc = ~cov_r4
# by kind
k_self_best = unc & (sp == lo)
k_self_bp = unc & (sp == bp) & (sp != lo)
k_self_only = unc & self_in & ~inr.any(1)
k_ex_only = unc & ~self_in & inr.any(1)
k_both = unc & self_in & inr.any(1)
print(f"  caller only {k_self_only.sum()}  exclusions only {k_ex_only.sum()}  both {k_both.sum()}  too many {(unc & ((tot > 6) | (reqs['n_extra'] > 4))).sum()}")
print(f"  caller == first eligible {k_self_best.sum()} (favour {(k_self_best & (reqs['flags'] != 0)).sum()}); caller == preferred best {k_self_bp.sum()}")
print(f"  caller in range: nsb {(unc & self_in & nsb).sum()}, !nsb plain {(unc & self_in & ~nsb & plain).sum()}, !nsb !plain {(unc & self_in & ~nsb & ~plain).sum()}")
ex_best = (ex == lo[:, None]).any(1) | (ex == bp[:, None]).any(1)
ex_inlist = ((ex > bp[:, None]) & (ex < np.where(nsb, end1, np.minimum(lo_t, end0))[:, None])).any(1)
ex_between = ((ex > lo[:, None]) & (ex < bp[:, None])).any(1)
print(f"  exclusion == best/first {(unc & ex_best).sum()}  strictly inside the list {(unc & ex_inlist & ~ex_best).sum()}  between first eligible and preferred best {(unc & ex_between).sum()}")
n_in = inr.sum(1)
print("  exclusions in range per uncovered request:", np.bincount(n_in[unc]))
# what the extended check would leave: exclusion == best0 / bestpos, caller == best without favour, caller in range with nsb (unless...), non-plain with caller in range
left = unc & (ex_best | ((sp == lo) & (reqs["flags"] == 0)) | (self_in & (sp != lo) & (nsb | ~plain)) | ex_between | (n_in > 2))
print(f"left after the extension (<= 2 exclusions inside, caller == best with favourSelf): {left.sum()} = {left.mean():.5f}")
for nm, mask in (("ex_best", unc & ex_best), ("self==best !favour", unc & (sp == lo) & (reqs["flags"] == 0)),
                 ("self in, nsb", unc & self_in & (sp != lo) & nsb), ("self in, !plain", unc & self_in & (sp != lo) & ~nsb & ~plain), ("n_in>2", unc & (n_in > 2))):
    print(f"    {nm}: {mask.sum()}")
w = np.arange(n) // 64
print(f"wavefronts with an uncovered request now: {len(np.unique(w[unc]))} of {w.max() + 1}; after: {len(np.unique(w[left]))}")
