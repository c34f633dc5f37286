"""CPU emulation of ONE pod-axis shard (TEST INFRASTRUCTURE ONLY — never imported by modelmesh_amd/).

Plays the role of `modelmesh_amd.dist.SolverShardBackend` for tests/test_dist_gloo.py: the same
`PodShardedPlacer` then drives it through commit and the six phases, with real gloo all-reduces in
between, on a machine without a GPU.  It is an independent restatement of the exchange protocol that
csrc/shard_kernels.hpp documents (what every phase publishes, what every shard derives from the
reduced vectors), written with Python sets of positions instead of 64-pod words, and checked against
the unsharded oracle (CacheMissForwardingLB.getNext, MM.java:4776-5005).
"""
from __future__ import annotations

import numpy as np

XMAX = 2**63 - 1
NOPOS = 0x7FFFFFFF
M64 = (1 << 64) - 1
SLOTS = {1: 2, 2: 9, 3: 6, 4: 3, 6: 1}
NONE, SELF = -1, -2


def _w64(v):  # Java long wrap
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def _jdiv(a, b):  # Java division truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _age(t, now):
    return 0 if t == 0 else _w64(now - t)


def _audit_mul(word):
    """csrc/wave.hpp audit_mul: the audit hash is H = sum over words of bits * audit_mul(word) mod 2^64 (linear in the bits)."""
    x = ((word + 1) * 0x9E3779B97F4A7C15) & M64
    x ^= x >> 29
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 32
    return x | 1


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >> 31 else v


class _Rule:  # MM.java:4951-4972
    def __init__(self, ago, min_rpm):
        self.ago = ago
        self.active = ago < 5 * 24 * 3600 * 1000
        ml = max(min_rpm, 100)
        self.m11 = min(int(1.1 * ml), 2**31 - 1)
        self.m15 = min(int(1.5 * ml), 2**31 - 1)
        self.m3, self.m4 = _i32(ml * 3), _i32(ml * 4)

    def nulls(self, rpm):
        a = self.ago
        return bool(self.active and rpm >= 100 and ((a < -1000 and rpm > self.m11) or (a < 5000 and rpm > self.m15) or
                                                   (a < 12 * 60 * 1000 and rpm > self.m3) or
                                                   (a < 24 * 3600 * 1000 and rpm > self.m4)))


class EmulShardBackend:
    def __init__(self, fleet, order_full: np.ndarray, shard: int, n_shards: int):
        """order_full: position -> pod index over ALL pod slots (present rows in PLACEMENT_ORDER, absent
        rows after them), e.g. from oracle.bind.OracleFleet."""
        import torch
        self.f, self.shard, self.n_shards = fleet, shard, n_shards
        self.device = torch.device("cpu")
        self.n_pods = fleet.n_pods
        self._order_full = np.asarray(order_full)

    def xchg_slots(self, ph):
        return 1 + self.n_shards if ph == 5 else SLOTS[ph]

    # ---- commit ----------------------------------------------------------------------------
    def rank_partial(self):
        import torch
        P, G = self.n_pods, self.n_shards
        rank_of = np.zeros(max(P, 1), np.int64)
        rank_of[self._order_full] = np.arange(P)
        per = -(-P // G) if P else 0
        lo, hi = min(P, self.shard * per), min(P, self.shard * per + per)
        r = np.zeros(max(P, 1), np.int32)
        r[lo:hi] = rank_of[lo:hi]
        return torch.from_numpy(r)

    def commit(self, rank):
        f, P, G = self.f, self.n_pods, self.n_shards
        self.pos_of = rank.numpy().astype(np.int64)[:P]
        W = max(-(-P // 64), 1)
        self.Wl = -(-W // G)
        w_lo = min(W, self.shard * self.Wl)
        self.lo, self.hi = w_lo * 64, min(W, w_lo + self.Wl) * 64
        self.orig = {int(self.pos_of[p]): p for p in range(P) if self.lo <= self.pos_of[p] < self.hi}
        pods = f.pods
        present = (pods["flags"] & 5) == 0
        live = (pods["flags"] & 2) != 0
        rs_bad = np.isin(pods["replica_set"], f.replaced_rs) & (pods["replica_set"] >= 0) if len(f.replaced_rs) else \
            np.zeros(P, bool)
        T = max(f.n_types, 1)

        def bit(bm, t, p):
            return bool((int(bm[t][p >> 6]) >> (p & 63)) & 1)
        self.elig, self.elig_nors, self.pref, self.has_pref = [], [], [], []
        for t in range(T):
            ha = f.n_types > 0 and f.has_allowed is not None and f.has_allowed[t]
            hp = f.n_types > 0 and f.has_prefer is not None and f.has_prefer[t]
            en = {pos for pos, p in self.orig.items() if present[p] and live[p] and (not ha or bit(f.allowed, t, p))}
            self.elig_nors.append(en)
            self.elig.append({pos for pos in en if not rs_bad[self.orig[pos]]})
            self.pref.append({pos for pos, p in self.orig.items() if hp and bit(f.prefer, t, p)})
            self.has_pref.append(bool(hp))
        self.any_rs = len(f.replaced_rs) > 0
        rem = np.maximum(pods["capacity"] - pods["used"], 0)
        self.row = {pos: (int(pods["lru_time"][p]), int(rem[p]), int(pods["count"][p]), int(pods["rpm"][p]))
                    for pos, p in self.orig.items()}
        self.full = {pos for pos in self.orig if self.row[pos][1] < f.min_space_units}

    def owns(self, pos):
        return self.lo <= pos < self.hi

    def owner_of(self, pos):
        return (pos >> 6) // self.Wl

    # ---- per-decision ---------------------------------------------------------------------------
    def _staged(self, rq, extra, nors):
        f = self.f
        m = f.models[rq["model"]]
        t = m["type"] if 0 <= m["type"] < max(f.n_types, 1) else 0
        ex = list(f.ent_pod[m["ent_off"]: m["ent_off"] + m["n_loaded"] + m["n_failed"]]) + \
            list(extra[rq["extra_off"]: rq["extra_off"] + rq["n_extra"]])
        expos = {int(self.pos_of[p]) for p in ex if 0 <= p < self.n_pods}
        base = self.elig_nors[t] if nors else self.elig[t]
        return sorted(base - expos), int(t)

    @staticmethod
    def _first(seq, start, pred=None, skip=None):
        for p in seq:
            if p >= start and p != skip and (pred is None or pred(p)):
                return p
        return XMAX

    def _derive(self, rq, now, X, d, upto, has_pm):
        """Everything a shard derives from the reduced vectors of phases < upto (mirrors derive0..5)."""
        f, D = self.f, {"has_pm": has_pm}
        D["selfpos"] = int(self.pos_of[rq["self_pod"]]) if 0 <= rq["self_pod"] < self.n_pods else -1
        D["favour"] = bool(rq["flags"] & 1)
        D["f_lru"], D["f_rem"] = int(rq["fresh_lru"]), max(int(rq["fresh_capacity"]) - int(rq["fresh_used"]), 0)
        D["f_rpm"], D["f_cnt"] = int(rq["fresh_rpm"]), int(rq["fresh_count"])
        D["ago"] = _age(int(rq["last_used"]), now)
        D["exit"] = None
        x1 = X[0][d]
        fe, fn = (min(int(v), NOPOS) for v in x1)
        D["use_nors"] = fe == NOPOS and self.any_rs
        D["best0"] = fe if fe != NOPOS else (fn if self.any_rs else NOPOS)
        D["none"] = D["best0"] == NOPOS
        if D["none"]:
            D["exit"] = (NONE, -1)
            return D
        if upto <= 2:
            return D
        x2 = [int(v) for v in X[1][d]]
        e_lru, e_rem, e_cnt, e_rpm, e_orig = x2[0], x2[1], _i32(x2[2]), _i32(x2[3]), _i32(x2[4])
        D.update(e_lru=e_lru, e_rem=e_rem, e_cnt=e_cnt, e_rpm=e_rpm, e_orig=e_orig, e_pref=x2[5] == 1,
                 q1=min(x2[6], NOPOS), q2=min(x2[7], NOPOS))
        sb = 0 if x2[8] == XMAX else x2[8]
        D["self_in_ew"], D["self_in_pm"] = bool(sb & 1), bool(sb & 2)
        us = D["best0"] == D["selfpos"]
        D["b"] = (D["f_lru"], D["f_rem"], D["f_cnt"], D["f_rpm"]) if us else (e_lru, e_rem, e_cnt, e_rpm)
        D["us"], D["b_orig"] = us, e_orig
        D["best_is_full"] = D["b"][1] < f.min_space_units
        D["bestpos"], D["use_dm"], D["limit"], D["mode_b"] = D["best0"], D["has_pm"], self.n_pods, False
        D["case_a"] = D["case_b"] = False
        if D["has_pm"] and not D["e_pref"]:
            if not D["best_is_full"]:
                if D["q1"] != NOPOS and D["q1"] <= D["q2"]:
                    D["case_a"] = True
                else:
                    D["use_dm"], D["limit"] = False, min(D["q2"], self.n_pods)
            else:
                D["case_b"] = True
        if upto <= 3:
            return D
        x3 = [int(v) for v in X[2][d]]
        if D["case_a"]:
            D["bestpos"] = D["q1"]
            D["b"] = (x3[0], x3[1], _i32(x3[2]), _i32(x3[3]))
            D["b_orig"] = _i32(x3[4])
            D["us"] = D["q1"] == D["selfpos"]
        elif D["case_b"]:
            lim = min(min(x3[5], NOPOS), self.n_pods)
            D["limit"] = lim
            if D["q1"] < lim:
                D["mode_b"] = True
            else:
                D["use_dm"] = False
        b_lru, b_rem, b_cnt, b_rpm = D["b"]
        sp = D["selfpos"]
        if D["mode_b"]:
            D["start"] = D["best0"] + 1
            if D["start"] <= sp < D["limit"] and D["self_in_ew"] and D["self_in_pm"] and D["favour"]:
                D["exit"] = (NONE, e_orig)
            D["ns_break"] = D["self_break"] = False
            D["thr"] = 0
        else:
            if D["us"] and D["favour"]:
                D["exit"] = (SELF, D["b_orig"])
            if D["best_is_full"]:
                rel = _jdiv(_age(b_lru, now), 10)
                d1, d2 = _w64(D["f_lru"] - b_lru), _w64(e_lru - b_lru)
                D["ns_break"], D["self_break"] = d1 > 45000 and d1 > rel, d2 > 45000 and d2 > rel
            else:
                q = b_rem >> 2
                D["ns_break"] = D["f_rem"] < f.min_space_units or D["f_rem"] < q
                D["self_break"] = e_rem < f.min_space_units or e_rem < q
            D["start"] = D["bestpos"] + 1
            D["thr"] = _i32(b_cnt + (b_cnt >> 2))
        if upto <= 4:
            return D
        x4 = [int(v) for v in X[3][d]]
        D["end"], D["self_in_d"], D["self_in_c"] = D["limit"], False, False
        D["mn_b"] = 2**31 - 1 if x4[2] == XMAX else _i32(x4[2])
        if not D["mode_b"]:
            D["self_in_d"] = D["start"] <= sp < D["limit"] and D["self_in_ew"] and (not D["use_dm"] or D["self_in_pm"])
            if D["ns_break"]:
                D["end"] = min(D["end"], min(x4[0], NOPOS))
            if D["self_in_d"] and D["self_break"]:
                D["end"] = min(D["end"], sp)
            if not D["best_is_full"]:
                D["end"] = min(D["end"], min(x4[1], NOPOS))
            D["self_in_c"] = D["self_in_d"] and sp < D["end"]
            if D["self_in_c"] and D["favour"] and D["exit"] is None:
                D["exit"] = (SELF, D["b_orig"])
        if upto <= 5 or D["exit"] is not None:
            return D
        x5 = [int(v) & M64 for v in X[4][d]]
        D["hsum"] = x5[0]
        cc = [v & 0xFFFFFFFF for v in x5[1:]]
        nn = [v >> 32 for v in x5[1:]]
        ccount = sum(cc)
        D["ccount"] = D["remaining"] = ccount
        D["null0"] = D["null_s"] = D["null_o"] = D["apply_b"] = False
        own_b = self.owner_of(D["bestpos"])
        own_s = self.owner_of(sp) if D["self_in_c"] else -1
        if ccount >= 2:
            if D["mode_b"]:
                D["rule"] = _Rule(D["ago"], D["mn_b"])
                if D["rule"].active:
                    D["apply_b"], D["remaining"] = True, sum(nn)
            else:
                n_others = ccount - 1 - (1 if D["self_in_c"] else 0)
                mn = b_rpm
                if D["self_in_c"] and e_rpm < mn:
                    mn = e_rpm
                if n_others > 0 and D["f_rpm"] < mn:
                    mn = D["f_rpm"]
                r = D["rule"] = _Rule(D["ago"], mn)
                D["null0"], D["null_s"] = r.nulls(b_rpm), D["self_in_c"] and r.nulls(e_rpm)
                D["null_o"] = n_others > 0 and r.nulls(D["f_rpm"])
                D["remaining"] = ccount - D["null0"] - D["null_s"] - (n_others if D["null_o"] else 0)
        rem = D["remaining"]
        D["index"] = 0 if rem <= 1 else (int(rq["pick"]) * rem) >> 32
        D["sel_shard"], D["sel_prefix"] = -1, 0
        run = 0
        for g in range(self.n_shards):
            if D["mode_b"]:
                r_g = nn[g] if D["apply_b"] else cc[g]
            else:
                spc = (g == own_b) + (g == own_s)
                r_g = cc[g] - (D["null0"] and g == own_b) - (D["null_s"] and g == own_s) - ((cc[g] - spc) if D["null_o"] else 0)
            if rem >= 1 and D["index"] < run + r_g:
                D["sel_shard"], D["sel_prefix"] = g, run
                break
            run += r_g
        return D

    def _cands(self, D, ew, pm):
        """this shard's candidate positions before the rpm filter, in order"""
        if D["mode_b"]:
            return [p for p in ew if D["start"] <= p < D["limit"] and p in pm]
        c = [p for p in ew if D["start"] <= p < D["end"] and (not D["use_dm"] or p in pm)]
        if self.owns(D["bestpos"]):
            c = sorted(set(c) | {D["bestpos"]})
        return c

    # ---- the speculative single-exchange form (csrc/shard_kernels.hpp "speculative") -----------------
    # Emulated as what it means: this shard plays the WHOLE protocol alone on its own slice (private
    # exchange vectors, no reduction), and then judges whether that was legitimate.
    FAST_SLOTS = 2   # x[0] = shard<<56 | incomplete<<55 | (chosen+2)<<28 | (best+1);  x[1] = shard<<56 | n_candidates<<32 | hash

    def fast_slots(self):
        return self.FAST_SLOTS

    def fast(self, reqs, n, extra, now, xf):
        import torch
        from modelmesh_amd._lib import PLACE_OUT
        priv = [torch.full((max(n, 1) * self.xchg_slots(ph),), XMAX, dtype=torch.int64) for ph in range(1, 7)]
        outs = np.zeros(n, dtype=PLACE_OUT)
        any_rs, self.any_rs = self.any_rs, False  # alone, only the replica-set-filtered set may be used:
        try:                                       # whether the retry applies is a global question
            for ph in range(1, 8):
                self.phase(ph, reqs, n, extra, now, priv, outs)
            X = [x.numpy().reshape(n, -1) if n else x.numpy().reshape(0, 1) for x in
                 [priv[k][: n * self.xchg_slots(k + 1)] for k in range(6)]]
            xv = xf.numpy().reshape(n, self.FAST_SLOTS) if n else xf.numpy().reshape(0, self.FAST_SLOTS)
            last = self.hi >= self.n_pods
            for d in range(n):
                rq = reqs[d]
                if not (0 <= rq["model"] < self.f.n_models):
                    complete, here = True, True  # unknown model: null, every shard says so
                else:
                    t = self.f.models[rq["model"]]["type"]
                    t = int(t) if 0 <= t < max(self.f.n_types, 1) else 0
                    D = self._derive(rq, now, X, d, 5, self.has_pref[t])
                    here = not D["none"]
                    complete = False
                    if here:
                        special = D["has_pm"] and not D["e_pref"]
                        if special and not D["case_a"]:
                            complete = False  # case (b) / no preferred pod before a full one: general protocol
                        elif D["exit"] is not None and D.get("end") is None:
                            complete = True
                        else:
                            early_self = D["us"] and D["favour"]  # decided by the best entry alone (:4891-4895)
                            complete = early_self or last or D["end"] < min(self.hi, self.n_pods)
                if not here:
                    xv[d] = XMAX
                    continue
                o = outs[d]
                key = self.shard << 56
                xv[d] = (key | ((0 if complete else 1) << 55) | (((int(o["chosen"]) + 2) & 0x7FFFFFF) << 28) |
                         ((int(o["best"]) + 1) & 0xFFFFFFF),
                         key | ((int(o["n_candidates"]) & 0xFFFFFF) << 32) | (int(o["hash"]) & 0xFFFFFFFF))
        finally:
            self.any_rs = any_rs

    def fast_finish(self, reqs, n, xf, outs):
        xv = xf.numpy().reshape(n, self.FAST_SLOTS) if n else xf.numpy().reshape(0, self.FAST_SLOTS)
        rest = []
        for d in range(n):
            k0 = int(xv[d][0])
            o = outs[d]
            o["chosen"], o["best"], o["n_candidates"], o["hash"] = NONE, -1, 0, 0
            if k0 == XMAX:
                if self.any_rs:
                    rest.append(d)
            elif (k0 >> 55) & 1:
                rest.append(d)
            else:
                k1 = int(xv[d][1])
                o["chosen"] = ((k0 >> 28) & 0x7FFFFFF) - 2
                o["best"] = (k0 & 0xFFFFFFF) - 1
                o["n_candidates"] = (k1 >> 32) & 0xFFFFFF
                o["hash"] = k1 & 0xFFFFFFFF
        self._rest_idx = np.asarray(rest, np.int64)
        return len(rest), reqs[self._rest_idx].copy(), np.zeros(len(rest), dtype=outs.dtype)

    def fast_scatter(self, n_rest, rest_outs, outs):
        outs[self._rest_idx] = rest_outs

    def phase(self, ph, reqs, n, extra, now, xchg, outs):
        X = [x.numpy().reshape(n, -1) if n else x.numpy().reshape(0, 1) for x in
             [xchg[k][: n * self.xchg_slots(k + 1)] for k in range(6)]]
        f = self.f
        for d in range(n):
            rq = reqs[d]
            bad = not (0 <= rq["model"] < f.n_models)
            if ph == 1:
                fe = fn = XMAX
                if not bad:
                    ew, _ = self._staged(rq, extra, False)
                    fe = ew[0] if ew else XMAX
                    if self.any_rs:
                        ewn, _ = self._staged(rq, extra, True)
                        fn = ewn[0] if ewn else XMAX
                X[0][d] = (fe, fn)
                continue
            if bad:
                D, t = {"none": True, "exit": (NONE, -1)}, 0
            else:
                t = f.models[rq["model"]]["type"]
                t = int(t) if 0 <= t < max(f.n_types, 1) else 0
                D = self._derive(rq, now, X, d, ph, self.has_pref[t])
            if ph == 7:
                o = outs[d]
                o["chosen"], o["best"], o["n_candidates"], o["hash"] = NONE, -1, 0, 0
                if D["exit"] is not None:
                    o["chosen"], o["best"] = D["exit"]
                else:
                    o["best"] = D["b_orig"]
                    if D["ccount"] > 0:
                        c = int(X[5][d][0])
                        o["chosen"] = NONE if c == XMAX else c
                        o["n_candidates"] = D["ccount"]
                        h = D["hsum"]
                        o["hash"] = ((h ^ (h >> 32)) & 0xFFFFFFFF) ^ ((D["remaining"] * 0x9E3779B1) & 0xFFFFFFFF)
                continue
            ew, pm = [], set()
            if not D["none"]:
                ew, _ = self._staged(rq, extra, D["use_nors"])
                pm = self.pref[t]
            ews = set(ew)
            if ph == 2:
                v = [XMAX] * 9
                if not D["none"]:
                    b0 = D["best0"]
                    if self.owns(b0):
                        v[0:4] = self.row[b0]
                        v[4] = self.orig[b0]
                        v[5] = 1 if (self.has_pref[t] and b0 in pm) else 0
                    if self.has_pref[t]:
                        v[6] = self._first(ew, b0 + 1, lambda p: p in pm)
                    v[7] = self._first(ew, b0 + 1, lambda p: p in self.full)
                    sp = D["selfpos"]
                    if sp >= 0 and self.owns(sp):
                        v[8] = (1 if sp in ews else 0) | (2 if (self.has_pref[t] and sp in pm) else 0)
                X[1][d] = v
            elif ph == 3:
                v = [XMAX] * 6
                if not D["none"]:
                    if D["case_a"] and self.owns(D["q1"]):
                        v[0:4] = self.row[D["q1"]]
                        v[4] = self.orig[D["q1"]]
                    if D["case_b"]:
                        oldest = D["b"][0]
                        rel = _jdiv(_age(oldest, now), 4)

                        def brk(p):
                            df = _w64(self.row[p][0] - oldest)
                            return df > 120000 and df > rel
                        v[5] = self._first(ew, D["best0"] + 1, brk)
                X[2][d] = v
            elif ph == 4:
                p1 = pc = mnb = XMAX
                if D["exit"] is None:
                    if D["mode_b"]:
                        c = self._cands(D, ew, pm)
                        if c:
                            mnb = min(self.row[p][3] for p in c)
                    else:
                        dm = (lambda p: p in pm) if D["use_dm"] else None
                        if D["ns_break"]:
                            p1 = self._first(ew, D["start"], dm, skip=D["selfpos"])
                        if not D["best_is_full"]:
                            thr = D["thr"]
                            pc = self._first([p for p in ew if p < D["limit"]], D["start"],
                                             lambda p: (dm is None or dm(p)) and self.row[p][2] >= 10 and self.row[p][2] > thr)
                X[3][d] = (p1, pc, mnb)
            elif ph == 5:
                cc = nn = h = 0
                if D["exit"] is None:
                    c = self._cands(D, ew, pm)
                    cc = len(c)
                    words = {}
                    for p in c:
                        words[p >> 6] = words.get(p >> 6, 0) | (1 << (p & 63))
                    for w, bits in words.items():
                        h = (h + bits * _audit_mul(w)) & M64
                    if D["mode_b"]:
                        rb = _Rule(D["ago"], D["mn_b"])
                        nn = sum(1 for p in c if not rb.nulls(self.row[p][3]))
                v = [0] * (1 + self.n_shards)
                v[0] = _w64(h)
                v[1 + self.shard] = _w64((nn << 32) | cc)
                X[4][d] = v
            elif ph == 6:
                chosen = XMAX
                if D["exit"] is None and D["ccount"] > 0 and D["sel_shard"] == self.shard:
                    c = self._cands(D, ew, pm)
                    if D["mode_b"]:
                        if D["apply_b"]:
                            c = [p for p in c if not D["rule"].nulls(self.row[p][3])]
                    else:
                        sp = D["selfpos"] if D["self_in_c"] else None
                        keep = []
                        for p in c:
                            if p == D["bestpos"]:
                                if not D["null0"]:
                                    keep.append(p)
                            elif p == sp:
                                if not D["null_s"]:
                                    keep.append(p)
                            elif not D["null_o"]:
                                keep.append(p)
                        c = keep
                    k = D["index"] - D["sel_prefix"]
                    if 0 <= k < len(c):
                        chosen = self.orig[c[k]]
                        if not D["favour"] and c[k] == D["selfpos"]:
                            chosen = SELF
                X[5][d] = (chosen,)
