/*
 * mm_rebalance_oracle.c — CPU restatement of the batch rebalancers that generate bursts of
 * load-target decisions.  TEST INFRASTRUCTURE ONLY.
 * Pinning: a15 (rate-tracking task), a16 (janitor scale-down) and a17 (reaper) are held to the reference's own text
 * (oracle/ref_harness -> tests/golden/ref_getnext.npz, tests/test_ref_vectors.py); C.3's second-copy timing window is the
 * reference's own test; a21 (preShutdown's migration loop) is held to the text too.
 */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

static inline int64_t jsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t age(int64_t t, int64_t now) { return t == 0 ? 0 : jsub64(now, t); }

/* a17 — leader "reaper" proactive loading: the candidate rule of pruneModelRegistry (MM.java:6459-6462,
 * :6574-6577, always on the cluster-wide stats `global`) and triggerProactiveLoadsForInstanceSubset
 * :6616-6747 for one instance subset: `stats` = the subset's ClusterStats, in_subset[p] (NULL = all) =
 * excludeTypes.equals(ir.prohibitedTypes), prohibited (NULL = none) = excludeTypes as a bitset over the
 * n_types configured type rows, skip[m] (NULL = none) = allCandidates entries nulled by an earlier subset
 * of the same run (:6724).  models are in registry iteration order.  out_model/out_last_used receive the
 * models the Java would call ensureLoadedInternal for, in call order (most recently used first). */
int32_t orc_proactive_plan(const orc_pod *pods, int32_t n_pods, const orc_cluster_stats *global,
                           const orc_cluster_stats *stats, const uint8_t *in_subset, const uint64_t *prohibited,
                           int32_t n_types, const uint8_t *skip, const orc_flat_model *models, int32_t n_models,
                           int32_t default_model_size_units, int64_t now, int32_t *out_model, int64_t *out_last_used,
                           int32_t max_out, orc_proactive_info *info)
{
    memset(info, 0, sizeof *info);
    int32_t free_count = 0, total_count = 0;
    if (stats->total_capacity > 0 && stats->total_free > 0) { /* :6621 */
        int32_t size_estimate;
        if (stats->model_copy_count < 3) {
            size_estimate = default_model_size_units;
        } else {
            /* (int)(totalCapacity - totalFree) / modelCopyCount — cast binds first */
            int32_t narrowed = (int32_t)(uint32_t)(uint64_t)jsub64(stats->total_capacity, stats->total_free);
            int32_t average = narrowed / stats->model_copy_count;
            size_estimate = stats->model_copy_count > 10
                                ? average
                                : (int32_t)((uint32_t)average + (uint32_t)default_model_size_units) / 2;
        }
        info->size_estimate = size_estimate;
        if (size_estimate == 0) { /* Java would throw ArithmeticException at :6651 */
            info->error = 1;
            return 0;
        }
        int64_t space = 0;
        for (int32_t i = 0; i < n_pods; i++) { /* :6635-6649, clusterState holds only present rows */
            const orc_pod *ir = &pods[i];
            if (ir->shutting_down) continue;
            if (in_subset && !in_subset[i]) continue; /* :6635 */
            int32_t max_loads = (int32_t)((uint32_t)ir->loading_threads * 50u - (uint32_t)ir->loading_in_progress);
            if (max_loads <= 0) continue;
            int64_t reserve = ir->capacity / 8, avail = jsub64(orc_remaining(ir), reserve);
            if (avail > 0) {
                int64_t cap_by_loads = (int64_t)(int32_t)((uint32_t)max_loads * (uint32_t)size_estimate);
                space = (int64_t)((uint64_t)space + (uint64_t)(avail < cap_by_loads ? avail : cap_by_loads));
            }
        }
        space /= 2;
        info->space_to_fill = space;
        free_count = (int32_t)(space / size_estimate);
        int32_t by_cap = (int32_t)(stats->total_capacity / (20LL * size_estimate));
        total_count = free_count > by_cap ? free_count : by_cap;
    }
    info->free_count = free_count;
    info->total_count = total_count;
    const int64_t cutoff = stats->global_lru == INT64_MAX
                               ? 0
                               : (int64_t)((uint64_t)stats->global_lru +
                                           (uint64_t)(age(stats->global_lru, now) / 3 > 1200000 ? age(stats->global_lru, now) / 3
                                                                                                  : 1200000));
    info->cutoff = cutoff;
    /* proactiveLoadCandidates exists only if globalStats.totalCapacity > 0; globalLru == 0 <=> free space (:6459-6462) */
    const int cand_enabled = global->total_capacity > 0;
    const int64_t global_lru = global->total_free > 0 ? 0 : global->global_lru;

    /* NavigableSet<ModelToLoad> toLoad: a TreeSet ordered by lastUsed DESC whose compareTo looks at
     * lastUsed only, so an element with an equal lastUsed is a duplicate and add() is a no-op. */
    int32_t cap = total_count > 0 ? total_count + 1 : 1;
    int64_t *set_lu = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    int32_t *set_model = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t size = 0, n_cand = 0;
    for (int32_t i = 0; i < n_models; i++) {
        const orc_flat_model *mr = &models[i];
        /* proactiveLoadCandidates rule, :6574-6577 */
        if (!(cand_enabled && mr->n_loaded == 0 && mr->n_failed < 2 && (global_lru == 0 || mr->last_used > global_lru))) continue;
        n_cand++;
        if (skip && skip[i]) continue; /* ent == null, :6679 */
        if (prohibited && mr->type >= 0 && mr->type < n_types && ((prohibited[mr->type >> 6] >> (mr->type & 63)) & 1)) continue; /* :6682 */
        int64_t last_used = mr->last_used;
        if (total_count > 0 && (free_count > 0 || last_used > cutoff)) { /* :6683-6685 */
            if (size < total_count || set_lu[size - 1] < last_used) {
                /* TreeSet.add: find position in descending order; equal ⇒ not added */
                int32_t pos = 0, dup = 0;
                while (pos < size && set_lu[pos] > last_used) pos++;
                if (pos < size && set_lu[pos] == last_used) dup = 1;
                if (!dup) {
                    memmove(&set_lu[pos + 1], &set_lu[pos], (size_t)(size - pos) * sizeof(int64_t));
                    memmove(&set_model[pos + 1], &set_model[pos], (size_t)(size - pos) * sizeof(int32_t));
                    set_lu[pos] = last_used;
                    set_model[pos] = i;
                    size++;
                }
                if (size > total_count) size--; /* pollLast */
            }
        }
    }
    info->n_candidates = n_cand;
    /* :6709-6734 — most recently used first; free space first, then only those newer than the cutoff */
    int32_t count = 0, fs = free_count;
    for (int32_t k = 0; k < size; k++) {
        if (fs > 0)
            fs--;
        else if (set_lu[k] < cutoff)
            break;
        if (count < max_out) {
            out_model[count] = set_model[k];
            out_last_used[count] = set_lu[k];
        }
        count++;
    }
    info->n_selected = count;
    free(set_lu);
    free(set_model);
    return count;
}

/* ======================================================================== */
/* a15 — rateTrackingTask, MM.java:5636-5832 (limitModelConcurrency == false, typeSetStats ==
 * clusterStats), getExcludeSet :5835-5856, loadedSince :5860-5871.  entries = usedSinceLastRun in
 * its iteration order.  overloaded_out[p] = 1 for members of getExcludeSet().  Returns 0, or 1 if
 * the task returns before looking at any entry (:5646-5648, :5658-5660). */
static int loaded_since(const int32_t *pods, const int64_t *times, int32_t n, int64_t cutoff, int32_t ignore)
{
    for (int32_t i = 0; i < n; i++) {
        if (ignore >= 0 && pods[i] == ignore) continue;
        if (times[i] > cutoff) return 1;
    }
    return 0;
}

/* MaxConcCacheEntry.getRpmScaleThreshold(andReset), MM.java:2766-2796.  Java arithmetic: `>>>` on the long, int additions and
 * long products wrap, `/` truncates toward zero, (int) of a long keeps the low 32 bits.  o: what a reset stores (:2771-2773). */
int32_t orc_rpm_scale_threshold(const orc_conc_entry *m, int and_reset, int32_t scale_up_rpm_threshold, int64_t dyn_const, orc_conc_out *o)
{
    const uint64_t count_mask = ((uint64_t)1 << ORC_CONC_COUNT_BITS) - 1; /* :2760 */
    const uint64_t cur_val = (uint64_t)m->count_and_time_sum;             /* :2767 */
    int64_t time_sum;
    int32_t count = (int32_t)(cur_val & count_mask);
    if (count >= 64) { /* :2769 */
        time_sum = (int64_t)(cur_val >> ORC_CONC_COUNT_BITS);
        if (and_reset && o) { /* sumThenReset(): priorSum = timeSum, priorCount = count */
            o->reset = 1;
            o->new_prior_sum = time_sum;
            o->new_prior_count = count;
        }
    } else { /* supplement with prior count, :2778-2785 */
        const int32_t pc = m->prior_count;
        if (pc <= 0 && count < 8) return scale_up_rpm_threshold;
        time_sum = (int64_t)((uint64_t)m->prior_sum + (count > 0 ? (cur_val >> ORC_CONC_COUNT_BITS) : 0));
        count = (int32_t)((uint32_t)count + (uint32_t)pc);
    }
    if (time_sum == 0) return INT32_MAX; /* :2787 */
    const int64_t num = (int64_t)((uint64_t)(int64_t)m->max_conc * ((uint64_t)(int64_t)count * (uint64_t)dyn_const)); /* :2795 */
    if (num == INT64_MIN && time_sum == -1) return 0; /* Long.MIN_VALUE / -1 == Long.MIN_VALUE in Java */
    return (int32_t)(uint32_t)(uint64_t)(num / time_sum);
}

/* (int) of a double, JLS 5.1.3 */
static int32_t jd2i(double d) { return d != d ? 0 : d >= 2147483647.0 ? INT32_MAX : d <= -2147483648.0 ? INT32_MIN : (int32_t)d; }

static int scaleup_plan_impl(const orc_pod *pods, int32_t n_pods, const int32_t *order, int32_t n_order,
                             const orc_cluster_stats *stats, const orc_cluster_stats *type_stats, int32_t t_rows, int has_tc,
                             const orc_flat_model *models, const int32_t *ent_pod,
                             const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                             const orc_scaleup_params *p, const orc_conc_params *cp, orc_scaleup_out *outs, orc_conc_out *conc_outs,
                             uint8_t *overloaded_out, orc_conc_result *result)
{
    const int latency_based = cp != NULL; /* boolean latencyBased = limitModelConcurrency, :5677 */
    if (latency_based) {
        memset(result, 0, sizeof *result);
        result->average_model_parallelism = cp->average_model_parallelism;
        for (int32_t i = 0; i < n; i++) {
            memset(&conc_outs[i], 0, sizeof conc_outs[i]);
            conc_outs[i].new_prior_sum = conc[i].prior_sum;
            conc_outs[i].new_prior_count = conc[i].prior_count;
        }
    }
    memset(outs, 0, (size_t)n * sizeof *outs);
    memset(overloaded_out, 0, (size_t)n_pods);
    for (int32_t i = 0; i < n; i++) { /* untouched entries keep their iteration markers */
        outs[i].new_i1 = entries[i].earlier_use_iteration;
        outs[i].new_i2 = entries[i].last_used_iteration;
    }
    const int64_t last_time = p->last_check_time, now = p->now;
    const int64_t time_delta = jsub64(now, last_time);
    if (time_delta * 5 < p->rate_check_interval_ms * 3) return 1; /* :5646 */
    const int32_t lower = p->iteration_counter - p->second_copy_max_age_iters;
    const int32_t upper = p->iteration_counter - p->second_copy_min_age_iters;
    const int32_t inst_count = stats->instance_count;
    if (inst_count < 2) return 1; /* :5658 */
    if (n == 0) return 1;         /* :5667 */
    const int64_t new_copies_ts = now + 20000; /* :5675 */
    int32_t scale_up_rpms = 0, heavy_rpms = 0;
    if (!latency_based) { /* :5679-5682 */
        scale_up_rpms = p->scale_up_rpm_threshold;
        heavy_rpms = (int32_t)((uint32_t)scale_up_rpms * 3u) / 4;
    }
    int have_exclude_set = 0;
    int32_t excluded_count = 0;
    int32_t model_parallelism_sum = 0;
    for (int32_t e = 0; e < n; e++) {
        const orc_cache_entry *ce = &entries[e];
        orc_scaleup_out *o = &outs[e];
        const int64_t count = ce->interval_count;
        /* clusterStats = typeSetStats(ce.modelInfo.serviceType) (:5691); an entry without a registry record
         * has no type here and is evaluated against the cluster-wide stats */
        const orc_cluster_stats *cst = stats;
        if (has_tc && ce->model >= 0) {
            const int32_t ty = models[ce->model].type;
            cst = &type_stats[(ty < 0 || ty >= t_rows) ? 0 : ty];
        }
        int32_t suitable = inst_count;
        if (has_tc) { /* :5693-5700 */
            suitable = cst->instance_count;
            if (suitable < 2) continue;
        }
        if (latency_based) { /* :5702-5707 */
            scale_up_rpms = orc_rpm_scale_threshold(&conc[e], 1, p->scale_up_rpm_threshold, cp->dynamic_rpm_scale_constant, &conc_outs[e]);
            conc_outs[e].threshold = scale_up_rpms;
            heavy_rpms = (int32_t)((uint32_t)scale_up_rpms * 3u) / 4;
            model_parallelism_sum = (int32_t)((uint32_t)model_parallelism_sum + (uint32_t)conc[e].max_conc);
        }
        const int32_t rpm = (int32_t)((count * 60000) / time_delta);
        o->rpm = rpm;
        if (rpm > heavy_rpms) o->heavy = 1; /* ce.setLastHeavyTime(now) */
        if (ce->model < 0) continue;        /* mr == null */
        const orc_flat_model *mr = &models[ce->model];
        const int32_t loaded_count = mr->n_loaded;
        if (loaded_count == 0) continue;
        const int32_t failed_count = mr->n_failed;
        int32_t candidate_count = suitable - (loaded_count + failed_count);
        if (candidate_count <= 0) continue;
        const int32_t *lp = ent_pod + mr->ent_off;
        const int64_t *lt = ent_time + mr->ent_off;
        if (loaded_count == 1) { /* :5726-5758 */
            const int32_t i1 = ce->earlier_use_iteration, i2 = ce->last_used_iteration;
            int i1in = 0, i2in = 0;
            if (i2 >= lower && i1 <= upper) {
                i1in = i1 >= lower;
                i2in = i2 <= upper;
            }
            if (i2in || !i1in) o->new_i1 = i2;
            o->new_i2 = p->iteration_counter;
            if (i1in || i2in) {
                if (cst->total_capacity == 0) continue; /* ArithmeticException, caught at :5810 */
                if ((10 * cst->total_free) / cst->total_capacity >= 1 ||
                    jsub64(now, cst->global_lru) > p->second_copy_lru_threshold_ms) {
                    o->action = 1; /* ensureLoadedInternalAsync(modelId, lastTime, weight, excludeThisInstance, 0) */
                    o->timestamp = last_time;
                    o->copies = 1;
                    continue;
                }
            }
        }
        if (rpm < scale_up_rpms) continue; /* :5762 */
        if (scale_up_rpms == 0) continue;  /* rpm / scaleUpRpms would throw; caught at :5810 */
        const int64_t recent_cutoff = jsub64(now, time_delta + p->rate_check_interval_ms + 2 * p->assume_completed_ms);
        if (loaded_since(lp, lt, loaded_count, recent_cutoff, p->self_pod)) continue; /* :5769 */
        if (!have_exclude_set) { /* getExcludeSet(), :5835-5856 */
            have_exclude_set = 1;
            /* getExcludeSet has its OWN scaleUpRpms (:5836): the configured threshold, or 900 x the task's averageModelParallelism */
            const int32_t x_rpms = latency_based ? jd2i(900.0 * cp->average_model_parallelism) : p->scale_up_rpm_threshold;
            if (latency_based) result->exclude_set_rpms = x_rpms;
            const int32_t a = (int32_t)((uint32_t)x_rpms * 4u);
            const int32_t b = (int32_t)((uint32_t)p->our_rpm - 2u * (uint32_t)x_rpms);
            const int32_t max_rpm = a > b ? a : b;
            for (int32_t k = 0; k < n_order; k++) {
                const int32_t iid = order[k];
                if (iid == p->self_pod) continue;
                if (pods[iid].rpm > max_rpm) {
                    overloaded_out[iid] = 1;
                    excluded_count++;
                }
            }
        }
        if (excluded_count != 0) { /* :5776-5787 (non-member excluded pods are subtracted twice) */
            for (int32_t iid = 0; iid < n_pods; iid++) {
                if (!overloaded_out[iid]) continue;
                int in = 0;
                for (int32_t k = 0; k < loaded_count + failed_count; k++)
                    if (lp[k] == iid) in = 1;
                if (!in) candidate_count--;
            }
            candidate_count -= excluded_count;
            if (candidate_count <= 0) continue;
        }
        int32_t copies = rpm / scale_up_rpms < candidate_count ? rpm / scale_up_rpms : candidate_count; /* :5792 */
        if (copies > 2) copies = copies < suitable / 3 ? copies : suitable / 3;
        o->action = 2;
        o->copies = copies;
        o->timestamp = new_copies_ts;
    }
    if (latency_based) { /* :5815-5818 */
        const double q = ((double)model_parallelism_sum) / n;
        result->average_model_parallelism = 1.0 >= q ? 1.0 : q; /* Math.max(1.0, q) */
        result->model_parallelism_sum = model_parallelism_sum;
        if (!have_exclude_set) result->exclude_set_rpms = jd2i(900.0 * cp->average_model_parallelism); /* (what it WOULD use: informative) */
    }
    return 0;
}

int orc_scaleup_plan(const orc_pod *pods, int32_t n_pods, const int32_t *order, int32_t n_order,
                     const orc_cluster_stats *stats, const orc_cluster_stats *type_stats, int32_t t_rows, int has_tc,
                     const orc_flat_model *models, const int32_t *ent_pod,
                     const int64_t *ent_time, const orc_cache_entry *entries, int32_t n, const orc_scaleup_params *p,
                     orc_scaleup_out *outs, uint8_t *overloaded_out)
{
    return scaleup_plan_impl(pods, n_pods, order, n_order, stats, type_stats, t_rows, has_tc, models, ent_pod, ent_time, entries, NULL, n, p,
                             NULL, outs, NULL, overloaded_out, NULL);
}

int orc_scaleup_plan_conc(const orc_pod *pods, int32_t n_pods, const int32_t *order, int32_t n_order,
                          const orc_cluster_stats *stats, const orc_cluster_stats *type_stats, int32_t t_rows, int has_tc,
                          const orc_flat_model *models, const int32_t *ent_pod,
                          const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                          const orc_scaleup_params *p, const orc_conc_params *cp, orc_scaleup_out *outs, orc_conc_out *conc_outs,
                          uint8_t *overloaded_out, orc_conc_result *result)
{
    return scaleup_plan_impl(pods, n_pods, order, n_order, stats, type_stats, t_rows, has_tc, models, ent_pod, ent_time, entries, conc, n, p,
                             cp, outs, conc_outs, overloaded_out, result);
}

/* a16 — janitor scale-down, MM.java:6110-6145 with removeModelCopies :6197-6310 and
 * removeSecondModelCopy :6314-6335 (mcce == null).  entries = scaleCopiesCandidates, oldest first;
 * entry.interval_count carries ce.getRpm(timeSinceLastCheck) inputs (the count). in_table[p] =
 * instanceInfo.get(iid) != null. pos_of = PLACEMENT_ORDER position of every present pod. */
static void scaledown_plan_impl(const orc_pod *pods, const int32_t *pos_of, const uint8_t *in_table,
                                const orc_cluster_stats *stats, const orc_flat_model *models, const int32_t *ent_pod,
                                const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                                const orc_scaledown_params *p, int64_t dyn_const, uint8_t *removed_out)
{
    memset(removed_out, 0, (size_t)n);
    if (p->shutting_down) return; /* :6111 */
    int64_t max_weight = p->adjusted_cache_capacity / 20;
    int32_t removed_count = 0;
    const int64_t now = p->now;
    for (int32_t e = 0; e < n; e++) {
        const orc_cache_entry *ce = &entries[e];
        const int32_t weight = ce->weight;
        const int can_remove = removed_count == 0 || weight <= max_weight;
        int removed = 0;
        do {
            const int64_t last_used = ce->last_used;
            if (last_used == 0) break; /* :6199 */
            if (ce->model < 0) break;  /* no ModelRecord: never entered scaleCopiesCandidates (MM.java:6076-6090) */
            const orc_flat_model *mr = &models[ce->model];
            const int32_t num = mr->n_loaded;
            if (!can_remove || num < 2) break;
            if (stats->total_capacity == 0 || stats->total_free * 100 / stats->total_capacity > 5) break; /* :6229 */
            const int32_t *lp = ent_pod + mr->ent_off;
            const int64_t *lt = ent_time + mr->ent_off;
            int32_t other = -1;
            for (int32_t k = 0; k < num; k++) { /* :6234-6243 */
                const int32_t iid = lp[k];
                if (iid != p->self_pod && in_table[iid] && !pods[iid].shutting_down) { other = iid; break; }
            }
            if (other < 0) break;
            if (num == 2) { /* :6250-6261 */
                const int64_t last_heavy = ce->last_heavy_time, cache_age = jsub64(now, stats->global_lru);
                int64_t scale_down_age = cache_age / 10;
                if (last_heavy == 0 || jsub64(now, last_heavy) < cache_age / 5)
                    scale_down_age = 36000000LL < scale_down_age ? 36000000LL : scale_down_age;
                if (jsub64(now, last_used) > scale_down_age) {
                    /* removeSecondModelCopy :6314-6335 */
                    if (!(p->self_pod >= 0 && in_table[p->self_pod] && !pods[p->self_pod].shutting_down)) break;
                    if (pos_of[other] > pos_of[p->self_pod]) break; /* PLACEMENT_ORDER.compare(other, this) > 0 */
                    removed = 1;
                }
            } else { /* :6263-6307 */
                const int64_t last_unload = ce->last_unload_time;
                if (last_unload > 0 && jsub64(now, last_unload) < 8 * p->rate_check_interval_ms) break;
                if (loaded_since(lp, lt, num, now - 1800000, -1)) break;
                int64_t min_age = (int64_t)(((uint64_t)3 * (uint64_t)stats->global_lru + 10400000ull)) / 100; /* quirk B#13 */
                if (min_age < 600000) min_age = 600000;
                else if (min_age > 18000000) min_age = 18000000;
                if (jsub64(now, ce->last_heavy_time) < min_age) break;
                const int64_t since = jsub64(now, p->last_check_time);
                if (since < p->rate_check_interval_ms / 10) break;
                const int64_t rpm = ce->interval_count == 0 ? 0 : (60000 * ce->interval_count) / since;
                /* :6294-6295: mcce == null ? scaleUpRpmThreshold : mcce.getRpmScaleThreshold(false) */
                const int64_t threshold = conc ? orc_rpm_scale_threshold(&conc[e], 0, p->scale_up_rpm_threshold, dyn_const, NULL)
                                               : p->scale_up_rpm_threshold;
                if (rpm > (threshold * 2) / 3) break;
                if (conc && conc[e].queued_requests > 1) break; /* :6303 */
                removed = 1;
            }
        } while (0);
        if (removed) {
            removed_out[e] = 1;
            removed_count++;
            max_weight -= weight;
        }
    }
}

void orc_scaledown_plan(const orc_pod *pods, const int32_t *pos_of, const uint8_t *in_table,
                        const orc_cluster_stats *stats, const orc_flat_model *models, const int32_t *ent_pod,
                        const int64_t *ent_time, const orc_cache_entry *entries, int32_t n,
                        const orc_scaledown_params *p, uint8_t *removed_out)
{
    scaledown_plan_impl(pods, pos_of, in_table, stats, models, ent_pod, ent_time, entries, NULL, n, p, 0, removed_out);
}

void orc_scaledown_plan_conc(const orc_pod *pods, const int32_t *pos_of, const uint8_t *in_table,
                             const orc_cluster_stats *stats, const orc_flat_model *models, const int32_t *ent_pod,
                             const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                             const orc_scaledown_params *p, int64_t dyn_const, uint8_t *removed_out)
{
    scaledown_plan_impl(pods, pos_of, in_table, stats, models, ent_pod, ent_time, entries, conc, n, p, dyn_const, removed_out);
}

/* a21 — preShutdown migration order, MM.java:7000-7040 with triggerNewModelCopyElsewhere
 * :6913-6928.  entries = runtimeCache.descendingLruMap() (most recently used first).
 * action_out: 1 = triggerNewModelCopyElsewhere(modelId, mr, lruTime, weight) is called;
 * wait_out: 1 = the shutdown waits for that copy (lruTime >= now - CUTOFF_AGE_MS). */
void orc_migration_plan(const orc_flat_model *models, const int32_t *ent_pod, const orc_cache_entry *entries,
                        int32_t n, int32_t self_pod, int64_t now, int64_t cutoff_age_ms, uint8_t *action_out,
                        uint8_t *wait_out)
{
    const int64_t cutoff = jsub64(now, cutoff_age_ms);
    for (int32_t e = 0; e < n; e++) {
        action_out[e] = wait_out[e] = 0;
        const orc_cache_entry *ce = &entries[e];
        if (ce->model < 0) continue; /* mr == null */
        const orc_flat_model *mr = &models[ce->model];
        int has_us = 0;
        for (int32_t k = 0; k < mr->n_loaded; k++)
            if (ent_pod[mr->ent_off + k] == self_pod) has_us = 1;
        if (!has_us) continue;          /* :7008-7010 */
        if (ce->flags & 1u) continue;   /* ce == null || ce.isFailed() */
        const int64_t lru_time = ce->last_used;
        if (lru_time > 0) {
            action_out[e] = 1;
            wait_out[e] = lru_time >= cutoff; /* when the status comes back LOADING */
        }
    }
}
