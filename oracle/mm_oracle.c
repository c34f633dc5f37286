/*
 * mm_oracle.c — see mm_oracle.h.  TEST INFRASTRUCTURE ONLY (parity checker).
 *
 * The code deliberately keeps the control flow of the Java (iterators, replay
 * lists, sequential breaks, the `us` toggle) instead of a "clever" form, so it
 * can be reviewed line against line with the cited reference ranges.
 */
#include "mm_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- Java integer semantics helpers ------------------------------------ */
static inline int64_t jsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int32_t jadd32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t jsub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t jmul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int cmp64(int64_t a, int64_t b) { return a < b ? -1 : (a > b ? 1 : 0); }
static inline int cmp32(int32_t a, int32_t b) { return a < b ? -1 : (a > b ? 1 : 0); }
static inline int cmpu32(uint32_t a, uint32_t b) { return a < b ? -1 : (a > b ? 1 : 0); }
/* (int)(double) narrowing conversion, JLS 5.1.3 */
static inline int32_t jd2i(double d)
{
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}

/* MM.java:4162-4164: 0 means "now" */
static inline int64_t age(int64_t t, int64_t now) { return t == 0 ? 0 : jsub64(now, t); }

/* MM.java:765-771 */
int64_t orc_min_space_units(int32_t dflt, int32_t threads, int64_t cap_units, int have_unload)
{
    int32_t min = jmul32(dflt, (have_unload || threads <= 1) ? 1 : 2);
    int32_t a = jmul32(dflt, threads);
    int32_t b = (int32_t)(cap_units / 20); /* (int) narrowing of a long */
    int32_t target = a < b ? a : b;
    return min > target ? min : target;
}

/* InstanceRecord.java:203-205 */
int64_t orc_remaining(const orc_pod *p)
{
    int64_t d = jsub64(p->capacity, p->used);
    return d > 0 ? d : 0;
}

/* MM.java:4640-4642 */
int orc_is_full(int64_t avail, int64_t min_space_units) { return avail < min_space_units; }

/* MM.java:4646-4703 */
int orc_placement_compare(const orc_pod *ir1, const orc_pod *ir2, int64_t min_space,
                          int64_t min_churn)
{
    if (ir1 == ir2) return 0; /* :4650 same record ⇒ key compare ⇒ equal */
    int sd1 = ir1->shutting_down != 0, sd2 = ir2->shutting_down != 0;
    if (sd1 ^ sd2) return sd1 ? 1 : -1; /* :4653-4656 */
    int64_t vers1 = ir1->version, vers2 = ir2->version;
    int64_t rem1 = orc_remaining(ir1), rem2 = orc_remaining(ir2);
    int full1 = orc_is_full(rem1, min_space), full2 = orc_is_full(rem2, min_space);
    if (vers1 != vers2) { /* :4660-4666 (lruTime vs a duration: quirk B#1) */
        int64_t thr = (int64_t)((uint64_t)min_churn * 2u);
        if (vers1 > vers2) {
            if (!full1 || ir1->lru_time > thr) return -1;
        } else if (!full2 || ir2->lru_time > thr) return 1;
    }
    if (full1 ^ full2) return full1 ? 1 : -1; /* :4669 */
    if (full1) {                              /* :4670-4674 */
        int d = cmp64(ir1->lru_time, ir2->lru_time);
        if (d != 0) return d;
    }
    int32_t count_diff = jsub32(ir1->count, ir2->count); /* :4676-4677 */
    if (count_diff != 0) return count_diff < 0 ? -1 : 1;
    int rem_diff = cmp64(rem2, rem1); /* :4679-4680 */
    if (rem_diff != 0) return rem_diff;
    if (!full1) { /* :4681-4685 */
        int d = cmp64(ir1->lru_time, ir2->lru_time);
        if (d != 0) return d;
    }
    /* :4691-4701 ComparisonChain */
    int32_t lip1 = ir1->loading_in_progress, lip2 = ir2->loading_in_progress;
    int c;
    if ((c = cmp32(jsub32(ir2->loading_threads, lip2), jsub32(ir1->loading_threads, lip1)))) return c;
    if ((c = cmp32(lip1, lip2))) return c;
    if ((c = cmp64(ir2->capacity, ir1->capacity))) return c;
    if ((c = cmp32(ir1->rpm, ir2->rpm))) return c;
    return cmpu32(ir1->id_order, ir2->id_order);
}

/* Stable binary-insertion order = what inserting every row into a
 * ConcurrentSkipListSet yields when the comparator is a total order. */
typedef struct {
    const orc_pod *pods;
    int64_t ms, mc;
} sort_ctx;

static void merge_sort(int32_t *a, int32_t *tmp, int32_t n, const sort_ctx *c)
{
    if (n < 2) return;
    int32_t h = n / 2;
    merge_sort(a, tmp, h, c);
    merge_sort(a + h, tmp, n - h, c);
    int32_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (orc_placement_compare(&c->pods[a[j]], &c->pods[a[i]], c->ms, c->mc) < 0)
            tmp[k++] = a[j++];
        else
            tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(int32_t));
}

int orc_sort_pods(const orc_pod *pods, int32_t n, int64_t min_space, int64_t min_churn,
                  int32_t *order_out)
{
    /* Precondition for PLACEMENT_ORDER to be transitive: the
     * version clause must be decisive whenever versions differ, i.e. no row
     * may be both full and have lruTime <= 2*minChurnAgeMs unless all
     * versions are equal. */
    int multi_version = 0, weak = 0;
    int64_t thr = (int64_t)((uint64_t)min_churn * 2u);
    for (int32_t i = 0; i < n; i++) {
        if (pods[i].version != pods[0].version) multi_version = 1;
        if (orc_is_full(orc_remaining(&pods[i]), min_space) && !(pods[i].lru_time > thr)) weak = 1;
    }
    for (int32_t i = 0; i < n; i++) order_out[i] = i;
    int32_t *tmp = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    sort_ctx c = {pods, min_space, min_churn};
    merge_sort(order_out, tmp, n, &c);
    free(tmp);
    return (multi_version && weak) ? -1 : 0;
}

/* ---------------------------------------------------------------------- */
/* CacheMissExcludeSet.isExcluded, MM.java:4740-4743 */
static int is_excluded(const orc_place_req *r, int32_t pod)
{
    for (int32_t i = 0; i < r->n_extra; i++)
        if (r->extra[i] == pod) return 1;
    for (int32_t i = 0; i < r->n_loaded; i++)
        if (r->loaded[i] == pod) return 1;
    for (int32_t i = 0; i < r->n_failed; i++)
        if (r->failed[i] == pod) return 1;
    return 0;
}

/* The Guava filtered iterator of MM.java:4760-4772, or (after a rewind) an
 * iterator over clusterStateReplay. */
typedef struct {
    const orc_snapshot *s;
    const orc_place_req *r;
    int use_rs;            /* excludeReplicaSets non-empty for this pass */
    int32_t cursor;        /* next position in s->order */
    const int32_t *replay; /* non-NULL ⇒ iterate this list instead */
    int32_t n_replay, replay_cur;
} pod_iter;

static int filter_accept(const pod_iter *it, int32_t pod)
{
    const orc_snapshot *s = it->s;
    const uint8_t *constrain = s->allowed ? s->allowed[it->r->type] : NULL;
    if ((constrain != NULL && !constrain[pod]) || is_excluded(it->r, pod) || !s->live[pod]) return 0;
    if (!it->use_rs) return 1; /* excludeReplicaSets.isEmpty() */
    int32_t rs = s->pods[pod].replica_set;
    if (rs < 0) return 1; /* iid.length() < 7 */
    for (int32_t i = 0; i < s->n_replaced_rs; i++)
        if (s->replaced_rs[i] == rs) return 0;
    return 1;
}

static int it_peek(pod_iter *it)
{
    if (it->replay) return it->replay_cur < it->n_replay;
    while (it->cursor < it->s->n_pods) {
        if (filter_accept(it, it->s->order[it->cursor])) return 1;
        it->cursor++;
    }
    return 0;
}

static int32_t it_next(pod_iter *it)
{
    if (it->replay) return it->replay[it->replay_cur++];
    return it->s->order[it->cursor++]; /* caller must have seen it_peek()==1 */
}

static void it_init(pod_iter *it, const orc_snapshot *s, const orc_place_req *r, int use_rs)
{
    memset(it, 0, sizeof *it);
    it->s = s;
    it->r = r;
    it->use_rs = use_rs;
}

/* The audit hash (csrc/wave.hpp): H = sum over 64-position words of bits(word) * audit_mul(word index) mod 2^64.
 * Linear in the candidate bits — the candidate at rank position p contributes audit_mul(p >> 6) << (p & 63). */
static uint64_t audit_mul(uint64_t word)
{
    uint64_t x = (word + 1ull) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return x | 1ull;
}

uint32_t orc_shortlist_hash(const int32_t *pos_of, const int32_t *cand, int32_t n,
                            int32_t n_remaining, int32_t n_pods)
{
    (void)n_pods;
    uint64_t h = 0;
    for (int32_t i = 0; i < n; i++) { /* the candidates are distinct pods: one term each */
        int32_t p = pos_of[cand[i]];
        h += audit_mul((uint64_t)(p >> 6)) << (p & 63);
    }
    return (uint32_t)(h ^ (h >> 32)) ^ ((uint32_t)n_remaining * 0x9E3779B1u);
}

#define TWELVE_MIN_MS (12LL * 60 * 1000)
#define ONE_DAY_MS (24LL * 3600 * 1000)
#define FIVE_DAYS_MS (5 * ONE_DAY_MS)

/* CacheMissForwardingLB.getNext, MM.java:4776-5005.
 * One body, two callers: orc_place (the CHECKER: allocates its lists per call, rebuilds pos_of and
 * computes the audit hash of the shortlist — machinery the reference does not have) and
 * orc_place_lean (the algorithm alone, on caller-owned scratch: what bench.py times as cpu_baseline).
 * `sc` holds the four lists of P + 1 ints; `pos_of` non-NULL = compute the audit hash. */
static int place_impl(const orc_snapshot *s, const orc_place_req *r, orc_place_out *o, int32_t *cand_out,
                      int32_t *const sc[4], int32_t *pos_of)
{
    const int32_t P = s->n_pods;
    o->chosen = ORC_NONE;
    o->best = -1;
    o->n_candidates = 0;
    o->n_remaining = 0;
    o->hash = 0;

    int32_t *candidates = sc[0], *inst_req_load = sc[1], *replay = sc[2];
    int32_t ccount = 0, n_replay = 0;
    int32_t best_iid = -1; /* bestIid; reported in o->best on every path once known */
    int rc = 0;

    const int exclude_self = r->self >= 0 ? is_excluded(r, r->self) : 0; /* :4779 */
    const int favour_self = r->favour_self != 0;                          /* :4781 */

    pod_iter it;
    it_init(&it, s, r, s->n_replaced_rs > 0); /* :4793 */
    if (!it_peek(&it)) {
        if (s->n_replaced_rs == 0) goto done; /* :4795-4797 return null */
        it_init(&it, s, r, 0);                 /* :4801 */
        if (!it_peek(&it)) goto done;          /* :4802-4804 */
    }
    const int32_t best_entry = it_next(&it); /* :4806 */
    best_iid = best_entry;
    int us = !exclude_self && r->self == best_iid; /* :4808 */
    const orc_pod *best_inst = us ? &r->fresh : &s->pods[best_entry]; /* :4810 */
    const int best_is_full = orc_is_full(orc_remaining(best_inst), s->min_space_units); /* :4811 */

    const uint8_t *prefer = s->prefer ? s->prefer[r->type] : NULL; /* :4817-4818 */

    int simple_case = prefer == NULL || prefer[best_iid]; /* :4822 */
    if (!simple_case) {
        int have_replay = 1; /* clusterStateReplay != null */
        if (!best_is_full) { /* case (a) :4828-4852 */
            int found = 0;
            while (it_peek(&it)) {
                int32_t ent = it_next(&it);
                if (prefer[ent]) {
                    found = 1;
                    best_iid = ent;
                    best_inst = &s->pods[ent];
                    us = !us && !exclude_self && r->self == best_iid;
                    break;
                }
                if (orc_is_full(orc_remaining(&s->pods[ent]), s->min_space_units)) break;
                replay[n_replay++] = ent;
            }
            if (!found) {
                it.replay = replay;
                it.n_replay = n_replay;
                it.replay_cur = 0;
                prefer = NULL;
            }
            simple_case = 1;
        } else { /* case (b) :4853-4887 */
            int64_t oldest = best_inst->lru_time;
            while (it_peek(&it)) {
                int32_t ent = it_next(&it);
                const orc_pod *cur_inst = &s->pods[ent];
                int64_t diff = jsub64(cur_inst->lru_time, oldest);
                if (diff > 120000LL && diff > age(oldest, r->now) / 4) break; /* :4864 */
                if (prefer[ent]) {
                    us = !us && !exclude_self && r->self == ent;
                    if (us && favour_self) goto done; /* :4871-4873 return null */
                    have_replay = 0;
                    candidates[ccount] = ent;
                    inst_req_load[ccount++] = cur_inst->rpm;
                } else if (have_replay) {
                    replay[n_replay++] = ent;
                }
            }
            if (have_replay) { /* :4881-4886 */
                it.replay = replay;
                it.n_replay = n_replay;
                it.replay_cur = 0;
                prefer = NULL;
                simple_case = 1;
            }
        }
    }

    if (simple_case) { /* :4890-4938 */
        if (us && favour_self) {
            o->chosen = ORC_SELF; /* :4894 ABORT_REQUEST */
            goto done;
        }
        candidates[ccount] = best_iid;
        inst_req_load[ccount++] = best_inst->rpm;

        const int64_t oldest = best_inst->lru_time;
        while (it_peek(&it)) {
            int32_t ent = it_next(&it);
            if (prefer != NULL && !prefer[ent]) continue; /* :4905-4907 */
            us = !us && !exclude_self && r->self == ent;
            /* :4909 — quirk B#2: bestEntry's row for self, the CALLER's fresh row otherwise */
            const orc_pod *cur_inst = us ? &s->pods[best_entry] : &r->fresh;

            if (best_is_full) {
                int64_t diff = jsub64(cur_inst->lru_time, oldest);
                if (diff > 45000LL && diff > age(oldest, r->now) / 10) break; /* :4915 */
            } else {
                int64_t rem = orc_remaining(cur_inst);
                if (orc_is_full(rem, s->min_space_units) || rem < (orc_remaining(best_inst) >> 2))
                    break; /* :4922 */
                int32_t count = s->pods[ent].count, first_count = best_inst->count;
                if (count >= 10 && count > jadd32(first_count, first_count >> 2)) break; /* :4926 */
            }
            if (us && favour_self) {
                o->chosen = ORC_SELF; /* :4931-4933 */
                goto done;
            }
            candidates[ccount] = ent;
            inst_req_load[ccount++] = cur_inst->rpm;
        }
    }

    if (ccount == 0) goto done; /* :4941-4943 */

    if (pos_of)
        for (int32_t i = 0; i < P; i++) pos_of[s->order[i]] = i;
    if (cand_out) memcpy(cand_out, candidates, (size_t)ccount * sizeof(int32_t));
    o->n_candidates = ccount;

    const int64_t last_used_ago = age(r->last_used, r->now); /* :4951 */
    int32_t chosen = -1;
    int32_t remaining = ccount;
    if (ccount == 1) {
        chosen = candidates[0];
    } else {
        int32_t *cand = sc[3];
        memcpy(cand, candidates, (size_t)ccount * sizeof(int32_t));
        if (last_used_ago < FIVE_DAYS_MS) { /* :4956 */
            int32_t mn = inst_req_load[0];
            for (int32_t i = 1; i < ccount; i++)
                if (inst_req_load[i] < mn) mn = inst_req_load[i];
            int32_t min_load = mn > 100 ? mn : 100;
            int32_t m11 = jd2i(1.1 * (double)min_load), m15 = jd2i(1.5 * (double)min_load);
            for (int32_t i = 0; i < ccount; i++) {
                int32_t rpm = inst_req_load[i];
                if (rpm >= 100 &&
                    ((last_used_ago < -1000LL && rpm > m11) || (last_used_ago < 5000LL && rpm > m15) ||
                     (last_used_ago < TWELVE_MIN_MS && rpm > jmul32(min_load, 3)) ||
                     (last_used_ago < ONE_DAY_MS && rpm > jmul32(min_load, 4)))) {
                    cand[i] = -1; /* candidates.set(i, null) */
                    remaining--;
                    if (remaining == 1) break;
                }
            }
        }
        /* :4981 ThreadLocalRandom.nextInt(remaining) → explicit pick */
        int32_t index = remaining == 1 ? 0 : (int32_t)(((uint64_t)r->pick * (uint64_t)(uint32_t)remaining) >> 32);
        for (int32_t i = 0, j = 0; i < ccount; i++) { /* :4982-4986 */
            chosen = cand[i];
            if (chosen != -1 && index == j++) break;
        }
    }
    o->n_remaining = remaining;
    if (pos_of) o->hash = orc_shortlist_hash(pos_of, candidates, ccount, remaining, P);
    if (!favour_self && r->self >= 0 && r->self == chosen) { /* :4989-4991 */
        o->chosen = ORC_SELF;
        goto done;
    }
    o->chosen = chosen;

done:
    o->best = best_iid;
    return rc;
}

int orc_place(const orc_snapshot *s, const orc_place_req *r, orc_place_out *o, int32_t *cand_out)
{
    const int32_t P = s->n_pods;
    const size_t list = (size_t)(P + 1), rows = (size_t)((s->n_rows > P ? s->n_rows : P) + 1);
    int32_t *mem = (int32_t *)malloc((4 * list + rows) * sizeof(int32_t));
    int32_t *const sc[4] = {mem, mem + list, mem + 2 * list, mem + 3 * list};
    const int rc = place_impl(s, r, o, cand_out, sc, mem + 4 * list);
    free(mem);
    return rc;
}

/* scratch: 4 * (n_pods + 1) ints owned by the caller (one per thread); o->hash stays 0 */
int orc_place_lean(const orc_snapshot *s, const orc_place_req *r, orc_place_out *o, int32_t *scratch)
{
    const size_t list = (size_t)(s->n_pods + 1);
    int32_t *const sc[4] = {scratch, scratch + list, scratch + 2 * list, scratch + 3 * list};
    return place_impl(s, r, o, NULL, sc, NULL);
}

/* ---------------------------------------------------------------------- */
/* ForwardingLB.getNext, MM.java:4315-4392 */
int32_t orc_serve(const orc_serve_req *r, const uint8_t *live, const int32_t *in_use,
                  const int64_t *last_used, int64_t *chosen_ts)
{
    if (r->n_copies <= 0) return ORC_NONE; /* :4318-4321 */
    int seen_self = 0;
    int32_t chosen = -1;
    int64_t chosen_time_stamp = 0;
    int32_t min = INT32_MAX;
    int64_t lru = INT64_MAX, first_started = INT64_MAX;
    int64_t still_loading_cutoff = -1;
    for (int32_t e = 0; e < r->n_copies; e++) {
        const int32_t iid = r->copy_pod[e];
        int us = 0;
        if (!seen_self && iid == r->self) { /* :4334-4342 */
            seen_self = 1;
            if (r->exclude_self) continue;
            us = 1;
        }
        if (iid < 0 || !live[iid]) continue; /* sii == null :4343-4347 */
        const int64_t load_started = r->copy_loaded[e];
        if (still_loading_cutoff == -1) still_loading_cutoff = jsub64(r->now, r->assume_completed_ms);
        if (load_started < still_loading_cutoff) { /* :4352-4367 */
            const int32_t inuse = us ? r->local_in_flight : in_use[iid];
            if (inuse > min) continue;
            const int64_t nlu = us ? (r->prefer_self ? 0 : r->last_invoke_time) : last_used[iid];
            if (inuse < min)
                min = inuse;
            else if (nlu >= lru)
                continue;
            chosen = iid;
            chosen_time_stamp = load_started;
            lru = nlu;
        } else if (min == INT32_MAX && load_started < first_started) { /* :4369-4376 */
            chosen = iid;
            chosen_time_stamp = load_started;
            first_started = load_started;
        }
    }
    if (chosen_ts) *chosen_ts = chosen_time_stamp;
    if (chosen >= 0) {
        if (!r->exclude_self && chosen == r->self) return ORC_SELF; /* :4381-4385 */
        return chosen;
    }
    return ORC_NONE;
}

/* ---------------------------------------------------------------------- */
/* InstanceSetStatsTracker.java:53-92 applied to a whole table; shutting-down
 * rows are treated as deleted (MM.java:1462-1464). */
void orc_cluster_stats_of(const orc_pod *pods, int32_t n, int64_t min_space, orc_cluster_stats *out)
{
    memset(out, 0, sizeof *out);
    out->global_lru = INT64_MAX; /* resetLru */
    for (int32_t i = 0; i < n; i++) {
        const orc_pod *ir = &pods[i];
        if (ir->shutting_down) continue;
        out->instance_count++;
        out->model_copy_count = jadd32(out->model_copy_count, ir->count);
        out->total_capacity = (int64_t)((uint64_t)out->total_capacity + (uint64_t)ir->capacity);
        int64_t avail = orc_remaining(ir);
        if (!orc_is_full(avail, min_space))
            out->total_free = (int64_t)((uint64_t)out->total_free + (uint64_t)avail);
        if (ir->lru_time > 0 && ir->lru_time < out->global_lru) out->global_lru = ir->lru_time; /* :57-61 */
    }
}
