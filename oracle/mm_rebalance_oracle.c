/*
 * mm_rebalance_oracle.c — CPU restatement of the batch rebalancers that generate bursts of
 * load-target decisions.  TEST INFRASTRUCTURE ONLY.
 * Parity unpinned by the reference's own tests (only C.3's second-copy timing window is).
 */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

static inline int64_t jsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t age(int64_t t, int64_t now) { return t == 0 ? 0 : jsub64(now, t); }

/* a17 — leader "reaper" proactive loading, MM.java:6574-6577 (candidate rule) and
 * triggerProactiveLoadsForInstanceSubset :6616-6747 with excludeTypes == null.
 * models are in registry iteration order.  out_model/out_last_used receive the models the Java
 * would call ensureLoadedInternal for, in call order (most recently used first). */
int32_t orc_proactive_plan(const orc_pod *pods, int32_t n_pods, const orc_cluster_stats *stats,
                           const orc_flat_model *models, int32_t n_models, int32_t default_model_size_units,
                           int64_t now, int32_t *out_model, int64_t *out_last_used, int32_t max_out,
                           orc_proactive_info *info)
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
    const int64_t global_lru = stats->global_lru;

    /* NavigableSet<ModelToLoad> toLoad: a TreeSet ordered by lastUsed DESC whose compareTo looks at
     * lastUsed only, so an element with an equal lastUsed is a duplicate and add() is a no-op. */
    int32_t cap = total_count > 0 ? total_count + 1 : 1;
    int64_t *set_lu = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    int32_t *set_model = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t size = 0, n_cand = 0;
    for (int32_t i = 0; i < n_models; i++) {
        const orc_flat_model *mr = &models[i];
        /* proactiveLoadCandidates rule, :6574-6577 */
        if (!(mr->n_loaded == 0 && mr->n_failed < 2 && (global_lru == 0 || mr->last_used > global_lru))) continue;
        n_cand++;
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
