// shard_kernels.hpp — load-target selection with the POD AXIS sharded across GPUs.
//
// SURVEY.md §8(e)(2) / BASELINE.json north_star: shard g owns a contiguous range of
// PLACEMENT_ORDER positions (rank space), i.e. the words [w_lo, w_lo+Wn) of every
// rank-ordered bitmap and the matching slice of every per-pod column.  Every
// shard sees the whole decision batch.  One CacheMissForwardingLB.getNext
// (MM.java:4776-5005) then becomes six local scans separated by all-reduces of
// small per-decision vectors (the host does the collective between two phase
// launches — RCCL all-reduce over xGMI, see modelmesh_amd/dist.py):
//
//   phase 1  first eligible position, with / without excludeReplicaSets   -> MIN
//   phase 2  bestEntry's row (owner), first preferred / first full after it,
//            self's eligibility bits (owner)                              -> MIN
//   phase 3  the re-designated preferred best's row (case (a), :4836-4841)
//            or the first pod outside the LRU window (case (b), :4862-4866) -> MIN
//   phase 4  first break position of the shortlist loop (:4905-4928), or
//            the minimum rpm of the preferred candidates (case (b))       -> MIN
//   phase 5  per-shard candidate counts + shortlist hash                  -> SUM
//   phase 6  the shard that holds the index-th survivor picks it (:4981-4986) -> MIN
//   phase 7  every shard writes the (identical) result rows
//
// Each phase re-derives the decision state from the reduced vectors of the
// earlier phases (derive1..derive5 below are the single source of that logic and
// mirror place_one() in place_kernel.hpp clause by clause), then does its own
// local scan.  Results are bit-identical to the single-GPU kernel (tests/test_shard_gpu.py).
#pragma once
#include "place_kernel.hpp"

namespace mmp {

constexpr int64_t kXMax = INT64_MAX;  // identity of the MIN all-reduce
constexpr int kX1 = 2, kX2 = 9, kX3 = 6, kX4 = 3, kX6 = 1;
__host__ __device__ constexpr int x5_slots(int n_shards) { return 1 + n_shards; }
constexpr int kMaxShards = 64;

struct ShardSnap {
    int32_t P, W, T, any_rs;
    int64_t min_space;
    int32_t shard, n_shards;
    int32_t Wl;    // words per shard: owner(pos) = (pos >> 6) / Wl
    int32_t w_lo;  // first owned word
    int32_t Wn;    // owned words (0 if this shard is beyond the table)
    // local slices, indexed by (pos - w_lo*64)
    const int64_t *lru, *rem;
    const int32_t *cnt, *rpm, *orig;
    const int32_t *pos_of;  // full: pod index -> global position
    const uint64_t *elig, *elig_nors, *pref;  // [T][max(Wn,1)]
    const uint8_t *has_pref;
    const uint64_t *fullw;  // [Wn]
};

struct XchgPtrs {
    int64_t *x1, *x2, *x3, *x4, *x5, *x6;
};

__device__ __forceinline__ int xpos(int64_t v) { return v >= (int64_t)kNoPos ? kNoPos : (int)v; }
__device__ __forceinline__ int64_t gpos(const ShardSnap &S, int local) { return local == kNoPos ? kXMax : (int64_t)(local + S.w_lo * 64); }
// global position -> local position clamped into [0, Wn*64]
__device__ __forceinline__ int lpos(const ShardSnap &S, int64_t g)
{
    const int64_t l = g - (int64_t)S.w_lo * 64;
    const int64_t hi = (int64_t)S.Wn * 64;
    return (int)(l < 0 ? 0 : (l > hi ? hi : l));
}
__device__ __forceinline__ bool owns(const ShardSnap &S, int g) { return g >= S.w_lo * 64 && g < (S.w_lo + S.Wn) * 64; }
__device__ __forceinline__ int owner_of(const ShardSnap &S, int g) { return (g >> 6) / S.Wl; }

// The decision state every shard re-derives identically from the reduced vectors.
struct DState {
    // request
    int selfpos;
    bool favour;
    int64_t f_lru, f_rem;
    int32_t f_rpm, f_cnt;
    int64_t ago;
    bool has_pm;
    // after X1
    bool none, use_nors;
    int best0;
    // after X2
    int64_t e_lru, e_rem;
    int32_t e_cnt, e_rpm, e_orig;
    bool e_pref, self_in_ew, self_in_pm;
    int q1, q2;
    bool us, best_is_full, case_a, case_b;
    int64_t b_lru, b_rem;
    int32_t b_cnt, b_rpm, b_orig;
    int bestpos, limit;
    bool use_dm, mode_b;
    // after X3
    int exit_chosen;  // 0: keep going; else MMP_NONE / MMP_SELF early return with best = exit_best
    int exit_best;
    int start;
    bool ns_break, self_break;
    int32_t thr;
    // after X4
    int end;
    bool self_in_d, self_in_c;
    int32_t mn_b;
    // after X5
    int ccount, remaining, index;
    uint64_t hsum;
    bool null0, null_s, null_o, apply_b;
    RpmRule rule;
    int sel_shard, sel_prefix;
};

__device__ __forceinline__ void derive0(const ShardSnap &S, const PlaceArgs &A, const mmp_place_req &rq, int type, DState &D)
{
    D.selfpos = (rq.self_pod >= 0 && rq.self_pod < S.P) ? S.pos_of[rq.self_pod] : -1;
    D.favour = (rq.flags & MMP_REQ_FAVOUR_SELF) != 0;
    D.f_lru = rq.fresh_lru;
    D.f_rem = remaining_of(rq.fresh_capacity, rq.fresh_used);
    D.f_rpm = rq.fresh_rpm;
    D.f_cnt = rq.fresh_count;
    D.ago = age_of(rq.last_used, A.now);
    D.has_pm = S.has_pref[type] != 0;
    D.exit_chosen = 0;
    D.exit_best = -1;
}

// MM.java:4793-4805
__device__ __forceinline__ void derive1(const ShardSnap &S, const int64_t *x1, DState &D)
{
    const int fe = xpos(x1[0]), fn = xpos(x1[1]);
    D.use_nors = (fe == kNoPos) && S.any_rs;
    D.best0 = fe != kNoPos ? fe : (S.any_rs ? fn : kNoPos);
    D.none = D.best0 == kNoPos;
    if (D.none) {
        D.exit_chosen = MMP_NONE;
        D.exit_best = -1;
    }
}

// MM.java:4806-4823 and the split into case (a) / case (b)
__device__ __forceinline__ void derive2(const ShardSnap &S, const int64_t *x2, DState &D)
{
    D.e_lru = x2[0];
    D.e_rem = x2[1];
    D.e_cnt = (int32_t)x2[2];
    D.e_rpm = (int32_t)x2[3];
    D.e_orig = (int32_t)x2[4];
    D.e_pref = x2[5] == 1;
    D.q1 = xpos(x2[6]);
    D.q2 = xpos(x2[7]);
    const int64_t sb = x2[8] == kXMax ? 0 : x2[8];
    D.self_in_ew = sb & 1;
    D.self_in_pm = (sb >> 1) & 1;
    D.us = D.best0 == D.selfpos;  // :4808
    D.b_lru = D.us ? D.f_lru : D.e_lru;
    D.b_rem = D.us ? D.f_rem : D.e_rem;
    D.b_cnt = D.us ? D.f_cnt : D.e_cnt;
    D.b_rpm = D.us ? D.f_rpm : D.e_rpm;
    D.b_orig = D.e_orig;
    D.best_is_full = D.b_rem < S.min_space;  // :4811 (never recomputed, quirk B#14)
    D.bestpos = D.best0;
    D.use_dm = D.has_pm;
    D.limit = S.P;
    D.mode_b = false;
    D.case_a = D.case_b = false;
    if (D.has_pm && !D.e_pref) {  // !simpleCase
        if (!D.best_is_full) {
            if (D.q1 != kNoPos && D.q1 <= D.q2)
                D.case_a = true;  // a preferred pod before the first full one: its row comes in X3
            else {
                D.use_dm = false;
                D.limit = D.q2 < S.P ? D.q2 : S.P;
            }
        } else
            D.case_b = true;  // the LRU-window bound comes in X3
    }
}

__device__ __forceinline__ void derive3(const ShardSnap &S, const PlaceArgs &A, const int64_t *x3, DState &D)
{
    if (D.case_a) {  // :4836-4841
        D.bestpos = D.q1;
        D.b_lru = x3[0];
        D.b_rem = x3[1];
        D.b_cnt = (int32_t)x3[2];
        D.b_rpm = (int32_t)x3[3];
        D.b_orig = (int32_t)x3[4];
        D.us = D.q1 == D.selfpos;
    } else if (D.case_b) {  // :4853-4887
        const int q3 = xpos(x3[5]);
        const int lim = q3 < S.P ? q3 : S.P;
        D.limit = lim;
        if (D.q1 < lim)
            D.mode_b = true;
        else
            D.use_dm = false;
    }
    if (D.mode_b) {
        D.start = D.best0 + 1;
        if (D.selfpos >= D.start && D.selfpos < D.limit && D.self_in_ew && D.self_in_pm && D.favour) {
            D.exit_chosen = MMP_NONE;  // :4871-4873
            D.exit_best = D.e_orig;
        }
        D.ns_break = D.self_break = false;
        D.thr = 0;
        return;
    }
    if (D.us && D.favour) {  // :4891-4895
        D.exit_chosen = MMP_SELF;
        D.exit_best = D.b_orig;
    }
    const int64_t oldest = D.b_lru;
    if (D.best_is_full) {
        const int64_t rel = age_of(oldest, A.now) / 10;
        const int64_t d1 = jsub64(D.f_lru, oldest), d2 = jsub64(D.e_lru, oldest);
        D.ns_break = d1 > 45000LL && d1 > rel;  // :4913-4917
        D.self_break = d2 > 45000LL && d2 > rel;
    } else {
        const int64_t q = D.b_rem >> 2;  // :4922
        D.ns_break = D.f_rem < S.min_space || D.f_rem < q;
        D.self_break = D.e_rem < S.min_space || D.e_rem < q;
    }
    D.start = D.bestpos + 1;
    D.thr = (int32_t)((uint32_t)D.b_cnt + (uint32_t)(D.b_cnt >> 2));  // :4926
}

__device__ __forceinline__ void derive4(const ShardSnap &S, const int64_t *x4, DState &D)
{
    D.end = D.limit;
    D.self_in_d = D.self_in_c = false;
    D.mn_b = (int32_t)(x4[2] == kXMax ? INT32_MAX : x4[2]);
    if (D.mode_b) return;
    D.self_in_d = D.selfpos >= D.start && D.selfpos < D.limit && D.self_in_ew && (!D.use_dm || D.self_in_pm);
    if (D.ns_break) {
        const int p1 = xpos(x4[0]);
        D.end = p1 < D.end ? p1 : D.end;
    }
    if (D.self_in_d && D.self_break) D.end = D.selfpos < D.end ? D.selfpos : D.end;
    if (!D.best_is_full) {
        const int pc = xpos(x4[1]);
        D.end = pc < D.end ? pc : D.end;
    }
    D.self_in_c = D.self_in_d && D.selfpos < D.end;
    if (D.self_in_c && D.favour && D.exit_chosen == 0) {  // :4931-4933
        D.exit_chosen = MMP_SELF;
        D.exit_best = D.b_orig;
    }
}

// rpm filter (:4951-4980) and the owner of the index-th survivor
__device__ __forceinline__ void derive5(const ShardSnap &S, const mmp_place_req &rq, const int64_t *x5, DState &D)
{
    D.hsum = (uint64_t)x5[0];
    int cc = 0, nn = 0;
    for (int g = 0; g < S.n_shards; g++) {
        cc += (int)(uint32_t)((uint64_t)x5[1 + g] & 0xffffffffull);
        nn += (int)(uint32_t)((uint64_t)x5[1 + g] >> 32);
    }
    D.ccount = cc;
    D.remaining = cc;
    D.null0 = D.null_s = D.null_o = D.apply_b = false;
    const int own_b = owner_of(S, D.bestpos), own_s = D.self_in_c ? owner_of(S, D.selfpos) : -1;
    if (cc >= 2) {
        if (D.mode_b) {
            D.rule.init(D.ago, D.mn_b);
            if (D.rule.active) {
                D.apply_b = true;
                D.remaining = nn;
            }
        } else {
            const int n_others = cc - 1 - (D.self_in_c ? 1 : 0);
            int32_t mn = D.b_rpm;
            if (D.self_in_c && D.e_rpm < mn) mn = D.e_rpm;
            if (n_others > 0 && D.f_rpm < mn) mn = D.f_rpm;
            D.rule.init(D.ago, mn);
            D.null0 = D.rule.nulls(D.b_rpm);
            D.null_s = D.self_in_c && D.rule.nulls(D.e_rpm);
            D.null_o = n_others > 0 && D.rule.nulls(D.f_rpm);
            D.remaining = cc - (D.null0 ? 1 : 0) - (D.null_s ? 1 : 0) - (D.null_o ? n_others : 0);
        }
    }
    D.index = D.remaining <= 1 ? 0 : (int)(((uint64_t)rq.pick * (uint64_t)(uint32_t)D.remaining) >> 32);
    D.sel_shard = -1;
    D.sel_prefix = 0;
    if (D.remaining >= 1) {
        int run = 0;
        for (int g = 0; g < S.n_shards; g++) {
            const int c = (int)(uint32_t)((uint64_t)x5[1 + g] & 0xffffffffull);
            int r;
            if (D.mode_b)
                r = D.apply_b ? (int)(uint32_t)((uint64_t)x5[1 + g] >> 32) : c;
            else {
                const int sp = (g == own_b ? 1 : 0) + (g == own_s ? 1 : 0);
                r = c - ((D.null0 && g == own_b) ? 1 : 0) - ((D.null_s && g == own_s) ? 1 : 0) - (D.null_o ? c - sp : 0);
            }
            if (D.index < run + r) {
                D.sel_shard = g;
                D.sel_prefix = run;
                break;
            }
            run += r;
        }
    }
}

// local staging of the filter() result for this shard's words
__device__ __forceinline__ void stage_local(const ShardSnap &S, const uint64_t *src, uint64_t *ew, const int32_t *ents,
                                            int32_t n_ents, const int32_t *extra, int32_t n_extra)
{
    const int lane = lane_id();
    for (int w = lane; w < S.Wn; w += 64) ew[w] = src[w];
    wave_sync();
    const int nex = n_ents + n_extra;
    for (int i = lane; i < nex; i += 64) {
        const int32_t pod = i < n_ents ? ents[i] : extra[i - n_ents];
        if (pod >= 0 && pod < S.P) {
            const int pos = S.pos_of[pod];
            if (owns(S, pos)) {
                const int lp = pos - S.w_lo * 64;
                atomicAnd((unsigned long long *)&ew[lp >> 6], ~(1ull << (lp & 63)));
            }
        }
    }
    wave_sync();
}

// candidate word lw of this shard (before the rpm filter)
__device__ __forceinline__ uint64_t cand_word(const ShardSnap &S, const DState &D, const uint64_t *ew, const uint64_t *Pm, int lw)
{
    const int ls = lpos(S, D.start), le = lpos(S, D.mode_b ? D.limit : D.end);
    uint64_t v = ew[lw];
    if (D.mode_b || D.use_dm) v &= Pm[lw];
    v = ls < le ? clip_word(v, lw, ls, le) : 0ull;
    if (!D.mode_b && owns(S, D.bestpos)) {
        const int lb = D.bestpos - S.w_lo * 64;
        if ((lb >> 6) == lw) v |= 1ull << (lb & 63);
    }
    return v;
}

__device__ __forceinline__ void store_lane0(int64_t *p, int64_t v)
{
    if (lane_id() == 0) *p = v;
}

template <int PH>
__device__ __forceinline__ void shard_phase(const ShardSnap &S, const PlaceArgs &A, const XchgPtrs &X, int d, uint64_t *ew, uint64_t *fw)
{
    const int lane = lane_id();
    const int G = S.n_shards;
    int64_t *x1 = X.x1 + (size_t)d * kX1, *x2 = X.x2 + (size_t)d * kX2, *x3 = X.x3 + (size_t)d * kX3;
    int64_t *x4 = X.x4 + (size_t)d * kX4, *x5 = X.x5 + (size_t)d * x5_slots(G), *x6 = X.x6 + (size_t)d * kX6;
    const mmp_place_req rq = A.reqs[d];
    const bool bad_model = rq.model < 0 || rq.model >= A.n_models;
    mmp_model_row m{};
    if (!bad_model) m = A.models[rq.model];
    int type = m.type;
    if (type < 0 || type >= S.T) type = 0;
    const int32_t *ents = A.ent_pod + m.ent_off;
    const int32_t n_ents = bad_model ? 0 : m.n_loaded + m.n_failed;
    const int32_t *extra = A.extra + rq.extra_off;
    const int Wn1 = S.Wn > 0 ? S.Wn : 1;
    const uint64_t *Pm = S.pref + (size_t)type * Wn1;

    if (PH == 1) {
        int64_t fe = kXMax, fn = kXMax;
        if (!bad_model) {
            stage_local(S, S.elig + (size_t)type * Wn1, ew, ents, n_ents, extra, rq.n_extra);
            fe = gpos(S, first_set_from(ew, nullptr, 0, S.Wn));
            if (S.any_rs) {
                wave_sync();
                stage_local(S, S.elig_nors + (size_t)type * Wn1, ew, ents, n_ents, extra, rq.n_extra);
                fn = gpos(S, first_set_from(ew, nullptr, 0, S.Wn));
            }
        }
        store_lane0(&x1[0], fe);
        store_lane0(&x1[1], fn);
        return;
    }

    DState D;
    derive0(S, A, rq, type, D);
    derive1(S, x1, D);
    if (!D.none) stage_local(S, (D.use_nors ? S.elig_nors : S.elig) + (size_t)type * Wn1, ew, ents, n_ents, extra, rq.n_extra);

    if (PH == 2) {
        int64_t v[kX2];
        for (int i = 0; i < kX2; i++) v[i] = kXMax;
        if (!D.none) {
            if (owns(S, D.best0)) {
                const int lb = D.best0 - S.w_lo * 64;
                v[0] = S.lru[lb];
                v[1] = S.rem[lb];
                v[2] = S.cnt[lb];
                v[3] = S.rpm[lb];
                v[4] = S.orig[lb];
                v[5] = (D.has_pm && test_bit(Pm, lb)) ? 1 : 0;
            }
            const int lnext = lpos(S, (int64_t)D.best0 + 1);
            if (D.has_pm) v[6] = gpos(S, first_set_from(ew, Pm, lnext, S.Wn));
            v[7] = gpos(S, first_set_from(ew, S.fullw, lnext, S.Wn));
            if (D.selfpos >= 0 && owns(S, D.selfpos)) {
                const int lsp = D.selfpos - S.w_lo * 64;
                v[8] = (test_bit(ew, lsp) ? 1 : 0) | ((D.has_pm && test_bit(Pm, lsp)) ? 2 : 0);
            }
        }
        if (lane < kX2) {
            int64_t mine = kXMax;
            for (int i = 0; i < kX2; i++)
                if (lane == i) mine = v[i];
            x2[lane] = mine;
        }
        return;
    }
    if (!D.none) derive2(S, x2, D);

    if (PH == 3) {
        int64_t v[kX3];
        for (int i = 0; i < kX3; i++) v[i] = kXMax;
        if (!D.none) {
            if (D.case_a && owns(S, D.q1)) {
                const int lq = D.q1 - S.w_lo * 64;
                v[0] = S.lru[lq];
                v[1] = S.rem[lq];
                v[2] = S.cnt[lq];
                v[3] = S.rpm[lq];
                v[4] = S.orig[lq];
            }
            if (D.case_b)
                v[5] = gpos(S, first_lru_break(ew, lpos(S, (int64_t)D.best0 + 1), S.Wn, S.lru, D.b_lru, 120000LL,
                                               age_of(D.b_lru, A.now) / 4));
        }
        if (lane < kX3) {
            int64_t mine = kXMax;
            for (int i = 0; i < kX3; i++)
                if (lane == i) mine = v[i];
            x3[lane] = mine;
        }
        return;
    }
    if (!D.none) derive3(S, A, x3, D);

    if (PH == 4) {
        int64_t p1 = kXMax, pc = kXMax, mnb = kXMax;
        if (D.exit_chosen == 0) {
            const int ls = lpos(S, D.start), ll = lpos(S, D.limit);
            if (D.mode_b) {
                int mn = INT32_MAX;
                for (int base = 0; base < S.Wn; base += 64) {
                    const int w = base + lane;
                    if (w < S.Wn) {
                        const uint64_t v = cand_word(S, D, ew, Pm, w);
                        for (uint64_t t = v; t; t &= t - 1) {
                            const int32_t r = S.rpm[w * 64 + (__ffsll((unsigned long long)t) - 1)];
                            mn = r < mn ? r : mn;
                        }
                    }
                }
                mn = wave_min_i32(mn);
                if (mn != INT32_MAX) mnb = mn;
            } else {
                const uint64_t *Dm = D.use_dm ? Pm : nullptr;
                if (D.ns_break) {
                    int l1 = first_set_from(ew, Dm, ls, S.Wn);
                    if (l1 != kNoPos && l1 + S.w_lo * 64 == D.selfpos) l1 = first_set_from(ew, Dm, l1 + 1, S.Wn);
                    p1 = gpos(S, l1);
                }
                if (!D.best_is_full) pc = gpos(S, first_count_break(ew, Dm, ls, ll, S.cnt, D.thr));
            }
        }
        store_lane0(&x4[0], p1);
        store_lane0(&x4[1], pc);
        store_lane0(&x4[2], mnb);
        return;
    }
    if (!D.none) derive4(S, x4, D);

    if (PH == 5) {
        int cc = 0, nn = 0;
        uint64_t h = 0;
        if (D.exit_chosen == 0) {
            RpmRule rb;
            rb.init(D.ago, D.mn_b);
            for (int base = 0; base < S.Wn; base += 64) {
                const int w = base + lane;
                if (w < S.Wn) {
                    const uint64_t v = cand_word(S, D, ew, Pm, w);
                    cc += __popcll((unsigned long long)v);
                    h += audit_term(v, (uint64_t)(S.w_lo + w));
                    if (D.mode_b) {
                        for (uint64_t t = v; t; t &= t - 1)
                            if (!rb.nulls(S.rpm[w * 64 + (__ffsll((unsigned long long)t) - 1)])) nn++;
                    }
                }
            }
            cc = wave_sum_i32(cc);
            nn = wave_sum_i32(nn);
            h = wave_sum_u64(h);
        }
        for (int g = lane; g < G; g += 64)
            x5[1 + g] = g == S.shard ? (int64_t)(((uint64_t)(uint32_t)nn << 32) | (uint64_t)(uint32_t)cc) : 0;
        store_lane0(&x5[0], (int64_t)h);
        return;
    }
    if (D.exit_chosen == 0) derive5(S, rq, x5, D);

    if (PH == 6) {
        int64_t chosen = kXMax;
        if (D.exit_chosen == 0 && D.ccount > 0 && D.sel_shard == S.shard) {
            // rebuild this shard's surviving candidate words
            const int lb = owns(S, D.bestpos) ? D.bestpos - S.w_lo * 64 : -1;
            const int lsf = (D.self_in_c && owns(S, D.selfpos)) ? D.selfpos - S.w_lo * 64 : -1;
            for (int base = 0; base < S.Wn; base += 64) {
                const int w = base + lane;
                if (w < S.Wn) {
                    uint64_t v = cand_word(S, D, ew, Pm, w);
                    if (D.mode_b) {
                        if (D.apply_b) {
                            uint64_t keep = v;
                            for (uint64_t t = v; t; t &= t - 1) {
                                const int bit = __ffsll((unsigned long long)t) - 1;
                                if (D.rule.nulls(S.rpm[w * 64 + bit])) keep &= ~(1ull << bit);
                            }
                            v = keep;
                        }
                    } else {
                        uint64_t special = 0;
                        if (lb >= 0 && (lb >> 6) == w) special |= 1ull << (lb & 63);
                        if (lsf >= 0 && (lsf >> 6) == w) special |= 1ull << (lsf & 63);
                        if (D.null_o) v &= special;
                        if (D.null0 && lb >= 0 && (lb >> 6) == w) v &= ~(1ull << (lb & 63));
                        if (D.null_s && lsf >= 0 && (lsf >> 6) == w) v &= ~(1ull << (lsf & 63));
                    }
                    fw[w] = v;
                }
            }
            wave_sync();
            const int cl = S.Wn > 0 ? select_in_range(fw, 0, S.Wn - 1, D.index - D.sel_prefix) : kNoPos;
            if (cl != kNoPos) {
                chosen = S.orig[cl];
                if (!D.favour && cl + S.w_lo * 64 == D.selfpos) chosen = MMP_SELF;  // :4989-4991
            }
        }
        store_lane0(&x6[0], chosen);
        return;
    }
}

// LDS per workgroup: kPlaceWaves × 2 bitmaps × wpad words (wpad covers the owned words).
template <int PH>
__global__ __launch_bounds__(kPlaceWaves * 64) void place_shard_kernel(ShardSnap S, PlaceArgs A, XchgPtrs X, int32_t wpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t *ew = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * 2 * wpad;
    uint64_t *fw = ew + wpad;
    // the rest sub-batch of the speculative form is launched for its CAPACITY; the number of real rows is on the device
    const int n = A.n_dev ? min(A.n, __builtin_amdgcn_readfirstlane(*A.n_dev)) : A.n;
    for (int d = blockIdx.x * kPlaceWaves + wave; d < n; d += gridDim.x * kPlaceWaves) {
        shard_phase<PH>(S, A, X, d, ew, fw);
        wave_sync();
    }
}

// phase 7: one thread per decision writes the result row (identical on every shard)
__global__ void place_shard_finish_kernel(ShardSnap S, PlaceArgs A, XchgPtrs X)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= A.n || (A.n_dev && d >= *A.n_dev)) return;
    const int G = S.n_shards;
    const mmp_place_req rq = A.reqs[d];
    mmp_place_out o;
    o.chosen = MMP_NONE;
    o.best = -1;
    o.n_candidates = 0;
    o.hash = 0;
    if (rq.model >= 0 && rq.model < A.n_models) {
        const mmp_model_row m = A.models[rq.model];
        int type = m.type;
        if (type < 0 || type >= S.T) type = 0;
        DState D;
        derive0(S, A, rq, type, D);
        derive1(S, X.x1 + (size_t)d * kX1, D);
        if (!D.none) {
            derive2(S, X.x2 + (size_t)d * kX2, D);
            derive3(S, A, X.x3 + (size_t)d * kX3, D);
            derive4(S, X.x4 + (size_t)d * kX4, D);
            if (D.exit_chosen == 0) {
                derive5(S, rq, X.x5 + (size_t)d * x5_slots(G), D);
                o.best = D.b_orig;
                if (D.ccount > 0) {  // :4941-4943 otherwise
                    const int64_t c = X.x6[(size_t)d * kX6];
                    o.chosen = c == kXMax ? MMP_NONE : (int32_t)c;
                    o.n_candidates = D.ccount;
                    o.hash = (uint32_t)(D.hsum ^ (D.hsum >> 32)) ^ ((uint32_t)D.remaining * 0x9E3779B1u);
                }
            } else {
                o.chosen = D.exit_chosen;
                o.best = D.exit_best;
            }
        }
    }
    A.outs[d] = o;
}

// ---- the speculative single-exchange form ---------------------------------------------------------
// The head of PLACEMENT_ORDER decides almost every request: the first eligible pod and its whole
// shortlist normally sit inside one shard's slice, and then that shard can run the complete
// lane-per-decision getNext (place_kernel.hpp: lane_decide on a view of its slice) on its own.  So before
// the six-exchange protocol above, every shard publishes per decision either "no eligible pod here"
// (kXMax) or its local result keyed by its shard number, with an "incomplete" bit when a scan ran off the
// end of its slice or the decision left the lane path's shape.  ONE all-reduce(MIN) of kXF (= 2) int64 per
// decision picks the result of the lowest shard that holds an eligible pod — which is the shard the
// global walk would have started in.  Only decisions whose winner is incomplete (or that found no
// eligible pod anywhere while replica sets are excluded: the retry of MM.java:4797-4804) go through the
// general protocol, on a compacted request list that every shard builds identically (flags -> exclusive
// scan -> gather, so the order is by decision index on every shard).
// Two int64 per decision, both led by the shard number so that MIN takes every field from the same (lowest) shard:
//   x[0] = shard << 56 | incomplete << 55 | (chosen + 2) << 28 | (best + 1)      (27 + 28 bits)
//   x[1] = shard << 56 | n_candidates << 32 | hash                                (24 + 32 bits)
// which bounds a pod-axis table at 2^24 instances (mmp_shard_configure checks) and the node at 127 shards.
constexpr int kXF = 2;
constexpr int kShardMaxPods = 1 << 24;

// The kernel is place_block's lane phase on the view: the slice's per-type head windows (built by the sharded commit over the
// view) staged in LDS beside the request fetch, the model's entries from the resolved rows (global rank positions, translated
// into the view), lane_decide_win<true>, and lane_decide_r<true> for what the window cannot answer.  Dynamic LDS:
// place_lane_lds(T).  (Round 2 ran lane_decide_r alone here, through models -> ent_pod -> pos_of: 6.8x the unsharded launch at
// one shard.)
// the slice's answer for decision d as the two exchange words (kXMax, kXMax: no eligible pod here)
__device__ __forceinline__ void shard_fast_words(const Snap &V, const PlaceArgs &A, int32_t shard, unsigned char *smem, int d, int64_t &k0, int64_t &k1)
{
    TypeWin *s_wins = reinterpret_cast<TypeWin *>(smem);
    uint64_t *s_scr = reinterpret_cast<uint64_t *>(smem + win_lds_bytes(V.T));
    const bool use_wins = A.wins != nullptr;  // wave-uniform
    if (use_wins) {
        constexpr int kWinBytes = (int)sizeof(TypeWin);
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int chunks = ((V.T < kWinLds ? V.T : kWinLds) * kWinBytes + 1023) >> 10;
        const char *src = reinterpret_cast<const char *>(A.wins);
        char *dst = reinterpret_cast<char *>(s_wins);
        for (int c = wave; c < chunks; c += kPlaceWaves)
            __builtin_amdgcn_global_load_lds(src + (size_t)c * 1024 + lane_id() * 16,
                                             (__attribute__((address_space(3))) void *)(dst + c * 1024), 16, 0, 0);
    }
    mmp_place_req rq{};
    if (d < A.n) rq = A.reqs[d];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    k0 = k1 = kXMax;
    if (d >= A.n) return;
    ResolvedReq r = resolve_req<true, true>(V, A, rq);
    mmp_place_out o;
    int code = kLaneHeadMiss;
    if (use_wins) code = lane_decide_win<true>(V, A, r, s_wins, s_scr + threadIdx.x, o);
    if (code == kLaneHeadMiss) {
        merge_late_extras(r);
        code = lane_decide_r<true>(V, A, r, o);
    }
    if (code != kLaneNoneHere) {
        const int64_t key = (int64_t)shard << 56;
        k0 = key | ((int64_t)(code != kLaneDone) << 55) | ((int64_t)(uint32_t)(o.chosen + 2) << 28) | (int64_t)(uint32_t)(o.best + 1);
        k1 = key | ((int64_t)(uint32_t)o.n_candidates << 32) | (int64_t)o.hash;
    }
}

__global__ __launch_bounds__(kPlaceBlock) void place_shard_fast_kernel(Snap V, PlaceArgs A, int32_t shard, int64_t *__restrict__ xf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = blockIdx.x * kPlaceBlock + threadIdx.x;
    int64_t k0, k1;
    shard_fast_words(V, A, shard, smem, d, k0, k1);
    if (d >= A.n) return;
    int64_t *x = xf + (size_t)d * kXF;
    x[0] = k0;
    x[1] = k1;
}

// after the all-reduce: write the decided rows, flag the rest
// cnt: one device word, zero between launches: finished workgroups << 32 | flagged decisions.  Every workgroup adds its own
// pair with ONE relaxed returning atomic (no fence: a device-scope release per wavefront made this kernel 34 us per 100k
// decisions — on eight XCDs it is an L2 write-back each); the workgroup that sees the others' count complete stores
// seq << 32 | flagged into the pinned word `done`, where the host finds it after synchronising the stream (no device-to-host
// copy); it runs scan + gather + the general protocol only when the count is not zero (the usual batch has none).
// one decision's reduced words -> its result row or its "rest" flag
__device__ __forceinline__ int32_t shard_finish_row(int64_t k0, int64_t k1, int32_t any_rs, int d, mmp_place_out *__restrict__ outs,
                                                    int32_t *__restrict__ flags)
{
    int32_t rest = 0;
    mmp_place_out o;
    o.chosen = MMP_NONE;
    o.best = -1;
    o.n_candidates = 0;
    o.hash = 0;
    if (k0 == kXMax)
        rest = any_rs ? 1 : 0;  // nowhere eligible: null, unless the excludeReplicaSets retry has to run
    else if ((k0 >> 55) & 1)
        rest = 1;
    else {
        o.chosen = (int32_t)((k0 >> 28) & 0x7ffffffll) - 2;
        o.best = (int32_t)(k0 & 0xfffffffll) - 1;
        o.n_candidates = (int32_t)((k1 >> 32) & 0xffffffll);
        o.hash = (uint32_t)(k1 & 0xffffffffll);
    }
    flags[d] = rest;
    if (!rest) outs[d] = o;
    return rest;
}
// several buffers zeroed by ONE launch (a commit cleared a dozen small buffers with a dozen hipMemsetAsync calls: ~3 us of host
// time and ~2.5 us of stream time each)
constexpr int kZeroRegions = 16;
struct ZeroList {
    int32_t n;
    int32_t pad_;
    void *p[kZeroRegions];
    uint64_t bytes[kZeroRegions];  // multiples of 4
};
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroList Z)
{
    for (int k = 0; k < Z.n; k++) {
        uint32_t *q = static_cast<uint32_t *>(Z.p[k]);
        const size_t words = Z.bytes[k] >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) q[i] = 0u;
    }
}

// a shard that ranked by sorting the WHOLE table (cheaper than its slice against all rows from ~67M pairs on) keeps only its own
// slice's ranks: the other rows stay zero, so the SUM all-reduce over the shards still assembles the one rank vector
__global__ void rank_from_order_range_kernel(const RankRow *__restrict__ sorted, int32_t P, int32_t p_lo, int32_t p_hi,
                                             int32_t *__restrict__ rank)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= P) return;
    const int32_t pod = (int32_t)sorted[pos].pad0;
    if (pod >= p_lo && pod < p_hi) rank[pod] = pos;
}

// the workgroup's flagged decisions into the launch's count; the last workgroup publishes seq << 32 | total in pinned memory.
// Relaxed on purpose: the word tells the host HOW MANY decisions are left, nothing about the rows — the host only enqueues further
// work on the same stream with it, and result rows are read after the stream is synchronised (mmp_shard_wait / the synchronous
// calls do that).  A release per workgroup (agent scope = an L2 write-back per workgroup on this multi-XCD part) was measured:
// 12.6 -> 17.5 us per 100k-decision batch.
__device__ __forceinline__ void shard_finish_count(int32_t rest, uint32_t *s_rest, unsigned long long *__restrict__ cnt,
                                                   uint64_t *__restrict__ done, uint32_t seq)
{
    const uint64_t m = __ballot(rest != 0);
    if (m && (int)__builtin_ctzll(m) == lane_id()) atomicAdd(s_rest, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long mine = (1ull << 32) | *s_rest;
        const unsigned long long prev = __hip_atomic_fetch_add(cnt, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(prev >> 32) == gridDim.x - 1) {
            const uint32_t total = (uint32_t)prev + *s_rest;
            __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, ((uint64_t)seq << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(256) void place_shard_fast_finish_kernel(const int64_t *__restrict__ xf, int32_t n, int32_t any_rs,
                                                                      mmp_place_out *__restrict__ outs, int32_t *__restrict__ flags,
                                                                      unsigned long long *__restrict__ cnt, uint64_t *__restrict__ done, uint32_t seq)
{
    __shared__ uint32_t s_rest;
    if (threadIdx.x == 0) s_rest = 0;
    __syncthreads();
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t rest = 0;
    if (d == n) flags[n] = 0;  // the scan runs over n + 1 items: offs[n] = number of flagged decisions
    if (d < n) {
        const int64_t *x = xf + (size_t)d * kXF;
        rest = shard_finish_row(x[0], x[1], any_rs, d, outs, flags);
    }
    shard_finish_count(rest, &s_rest, cnt, done, seq);
}

// A group of ONE shard: there is nothing to exchange, so the slice's answer is the answer — the fast kernel writes the result
// rows, the rest flags and the count itself (no exchange words, no all-reduce, no finish kernel: one launch per batch, as the
// unsharded context has).
__global__ __launch_bounds__(kPlaceBlock) void place_shard_fast_direct_kernel(Snap V, PlaceArgs A, int32_t shard, int32_t any_rs,
                                                                             int32_t *__restrict__ flags, unsigned long long *__restrict__ cnt,
                                                                             uint64_t *__restrict__ done, uint32_t seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_rest;
    if (threadIdx.x == 0) s_rest = 0;
    const int d = blockIdx.x * kPlaceBlock + threadIdx.x;
    int64_t k0, k1;
    shard_fast_words(V, A, shard, smem, d, k0, k1);  // (its barrier orders the s_rest store)
    int32_t rest = 0;
    if (d == A.n) flags[A.n] = 0;
    if (d < A.n) rest = shard_finish_row(k0, k1, any_rs, d, A.outs, flags);
    shard_finish_count(rest, &s_rest, cnt, done, seq);
}

__global__ void place_shard_gather_kernel(const mmp_place_req *__restrict__ reqs, int32_t n, const int32_t *__restrict__ flags,
                                          const int32_t *__restrict__ offs, mmp_place_req *__restrict__ rest_reqs,
                                          int32_t *__restrict__ rest_idx)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n || !flags[d]) return;
    rest_reqs[offs[d]] = reqs[d];
    rest_idx[offs[d]] = d;
}

__global__ void place_shard_scatter_kernel(const mmp_place_out *__restrict__ rest_outs, const int32_t *__restrict__ rest_idx,
                                           int32_t n_rest, mmp_place_out *__restrict__ outs, const int32_t *__restrict__ n_dev = nullptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rest && (!n_dev || i < *n_dev)) outs[rest_idx[i]] = rest_outs[i];
}

// ---- sharded commit ---------------------------------------------------------------------------
// After the SUM all-reduce every shard holds the full rank[] (4 B per pod — the only replicated
// per-pod state besides the raw input rows).  Each shard keeps the columns of the positions it owns.
__global__ void scatter_shard_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, const int32_t *__restrict__ rank,
                                     int32_t *__restrict__ occupancy, int32_t pos_lo, int32_t pos_hi,
                                     int64_t *__restrict__ lru, int64_t *__restrict__ rem, int32_t *__restrict__ cnt,
                                     int32_t *__restrict__ rpm, int32_t *__restrict__ orig, int32_t *__restrict__ pos_of,
                                     int32_t *__restrict__ err)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int pos = rank[p];
    if (pos < 0 || pos >= P || atomicAdd(&occupancy[pos], 1) != 0) {
        atomicExch(err, 1);
        return;
    }
    pos_of[p] = pos;
    if (pos < pos_lo || pos >= pos_hi) return;
    const mmp_pod_row r = pods[p];
    const int l = pos - pos_lo;
    lru[l] = r.lru_time;
    rem[l] = remaining_of(r.capacity, r.used);
    cnt[l] = r.count;
    rpm[l] = r.rpm;
    orig[l] = p;
}

// One wave per (bitmap row, owned word); `orig` is the local position -> pod slice.
__global__ void build_masks_shard_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t Wfull, int32_t w_lo,
                                         int32_t Wn, int32_t T, int64_t min_space, const int32_t *__restrict__ orig,
                                         const uint64_t *__restrict__ allowed, const uint8_t *__restrict__ has_allowed,
                                         const uint64_t *__restrict__ prefer, const uint8_t *__restrict__ has_prefer,
                                         const uint8_t *__restrict__ rs_bad, uint64_t *__restrict__ elig,
                                         uint64_t *__restrict__ elig_nors, uint64_t *__restrict__ pref,
                                         uint64_t *__restrict__ fullw)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= T * Wn) return;
    const int t = wave / Wn, w = wave - t * Wn;
    const int pos = (w_lo + w) * 64 + lane;
    bool e = false, en = false, pf = false, fl = false;
    if (pos < P) {
        const int p = orig[w * 64 + lane];
        const mmp_pod_row r = pods[p];
        const bool present = (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) == 0;
        const bool live = (r.flags & MMP_POD_LIVE) != 0;
        bool al = true;
        if (has_allowed && has_allowed[t]) al = (allowed[(size_t)t * Wfull + (p >> 6)] >> (p & 63)) & 1ull;
        en = present && live && al;
        e = en && !(rs_bad && rs_bad[p]);
        if (has_prefer && has_prefer[t]) pf = (prefer[(size_t)t * Wfull + (p >> 6)] >> (p & 63)) & 1ull;
        fl = remaining_of(r.capacity, r.used) < min_space;
    }
    const uint64_t be = __ballot(e), ben = __ballot(en), bp = __ballot(pf), bf = __ballot(fl);
    if (lane == 0) {
        elig[(size_t)t * Wn + w] = be;
        elig_nors[(size_t)t * Wn + w] = ben;
        pref[(size_t)t * Wn + w] = bp;
        if (t == 0) fullw[w] = bf;
    }
}

}  // namespace mmp
