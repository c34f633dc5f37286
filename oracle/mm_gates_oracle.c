/*
 * mm_gates_oracle.c — CPU restatement of the request-level guards around instance selection
 * (SURVEY.md §8 rows a10, a11, a14, a20).  TEST INFRASTRUCTURE ONLY (parity checker).
 * Pinning: held to the reference's own text (oracle/ref_harness -> tests/golden/ref_getnext.npz, tests/test_ref_vectors.py:
 * 12 000 guarded requests); ModelMeshLoadFailureTest:432-492 pins MAX_LOAD_FAILURES; cross-checked by tests/test_gates.py.
 */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

static inline int64_t jsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t jabs64(int64_t a) { return a < 0 ? (int64_t)(0u - (uint64_t)a) : a; }
static inline int32_t jabs32(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }
static inline int64_t age(int64_t t, int64_t now) { return t == 0 ? 0 : jsub64(now, t); }

#define MAX_LOAD_FAILURES 3  /* MM.java:222 */
#define MAX_LOAD_LOCATIONS 5 /* MM.java:224 */

/* goLocal, MM.java:3598-3626.  copies = filteredInstances (already filtered), in map order. */
int orc_go_local(const int32_t *copy_pod, const int64_t *copy_loaded, int32_t n, int32_t self,
                 int favour_self_for_hits, int have_cache_entry, int entry_done, int64_t now)
{
    if (n <= 0) return 0;
    int has_local = 0;
    int64_t local_loaded = 0;
    for (int32_t i = 0; i < n; i++)
        if (copy_pod[i] == self) { has_local = 1; local_loaded = copy_loaded[i]; break; }
    int go_local = 0;
    if (has_local) {
        go_local = n == 1;
        if (!go_local && favour_self_for_hits) {
            if (have_cache_entry) {
                if (entry_done) {
                    go_local = 1;
                } else {
                    int64_t oldest = copy_loaded[0]; /* Collections.min(values), :4166 */
                    for (int32_t i = 1; i < n; i++)
                        if (copy_loaded[i] < oldest) oldest = copy_loaded[i];
                    if (oldest == local_loaded || age(oldest, now) < 1500) go_local = 1;
                }
            }
        }
    }
    return go_local;
}

/* checkLoadFailureCount, MM.java:4607-4627 */
int orc_load_failures_breached(const int64_t *fail_time, int32_t n, int64_t now, int64_t in_use_expiry_ms)
{
    if (n == 0) return 0;
    int count = 0;
    const int64_t cutoff = jsub64(now, in_use_expiry_ms);
    for (int32_t i = 0; i < n; i++) {
        if (fail_time[i] > cutoff) count++;
        if (count >= MAX_LOAD_FAILURES) return 1;
    }
    return 0;
}

/* checkLoadLocationCount, MM.java:4590-4604.  in_table[p] = instanceInfo.contains(id) */
int orc_load_locations_breached(const int32_t *loaded_pod, int32_t n, const int32_t *explicit_excl,
                                int32_t n_explicit, const uint8_t *in_table)
{
    int count = 0;
    for (int32_t i = 0; i < n; i++) {
        int excl = 0;
        for (int32_t j = 0; j < n_explicit; j++)
            if (explicit_excl[j] == loaded_pod[i]) excl = 1;
        if (!excl && in_table[loaded_pod[i]])
            if (++count >= MAX_LOAD_LOCATIONS) return 1;
    }
    return 0;
}

/* churn guard, MM.java:3870-3884 */
int orc_churn_reject(int64_t min_churn_age_ms, int64_t min_space_units, int64_t cache_capacity,
                     int64_t cache_weighted_size, int64_t cache_oldest_time, int64_t now)
{
    if (min_churn_age_ms > 0) {
        int64_t remaining = jsub64(cache_capacity, cache_weighted_size);
        if (orc_is_full(remaining, min_space_units)) {
            int64_t lru = cache_oldest_time;
            if (lru >= 0 && lru != INT64_MAX && age(lru, now) < min_churn_age_ms) return 1;
        }
    }
    return 0;
}

/* loadLocal size prediction + early reject, MM.java:5158-5197.  Returns the signed initialSize;
 * *reject = the "abort early" branch. */
int32_t orc_load_local_initial_size(int have_size_hint, int32_t size_hint, int32_t loading_count,
                                    int32_t weight_predict_cutoff, int32_t loader_predicted,
                                    const orc_cluster_stats *stats, int we_created_entry, int64_t last_used_time,
                                    int64_t cache_capacity, int64_t cache_weighted_size, int64_t cache_oldest_time,
                                    int *reject)
{
    int32_t initial = 0;
    if (have_size_hint) {
        initial = size_hint;
    } else if (loading_count > weight_predict_cutoff) {
        int32_t copy_count = stats->model_copy_count;
        if (copy_count >= 10) {
            /* -(1 + (int)(totalCapacity - totalFree) / copyCount): the cast binds before the divide (quirk B#7) */
            int32_t narrowed = (int32_t)(uint32_t)(uint64_t)jsub64(stats->total_capacity, stats->total_free);
            int32_t q = (narrowed == INT32_MIN && copy_count == -1) ? INT32_MIN : narrowed / copy_count;
            initial = (int32_t)(0u - (1u + (uint32_t)q));
        }
    }
    if (initial == 0) initial = loader_predicted;
    int32_t abs_size = jabs32(initial);
    *reject = 0;
    if (we_created_entry) {
        if ((int64_t)abs_size > cache_capacity ||
            (last_used_time > 0 && (int64_t)abs_size > jsub64(cache_capacity, cache_weighted_size) &&
             last_used_time < cache_oldest_time))
            *reject = 1;
    }
    return initial;
}

/* onEviction reload rule, MM.java:2886-2920 */
int orc_reload_elsewhere(int entry_failed, int64_t loaded_time /* <0: not in registry */, int64_t load_timeout_ms,
                         int64_t now, const orc_cluster_stats *stats)
{
    int attempt = 0;
    if (!entry_failed && loaded_time >= 0) attempt = jsub64(now, loaded_time) > 2 * load_timeout_ms;
    if (!attempt) return 0;
    return stats->total_capacity > 0 && stats->instance_count > 1 &&
           (20 * stats->total_free) / stats->total_capacity >= 1;
}

/* loadingChange / loadChange, MM.java:5536-5550 */
static int loading_change(int32_t cur_in_prog, int32_t cur_threads, int32_t in_prog)
{
    if (in_prog == cur_in_prog) return 0;
    if ((in_prog == 0) ^ (cur_in_prog == 0)) return 1;
    if ((in_prog <= cur_threads) ^ (cur_in_prog <= cur_threads)) return 1;
    return jabs32(in_prog - cur_in_prog) >= 3;
}
static int load_change(int32_t cur_rpm, int32_t rpm)
{
    int32_t diff = jabs32(cur_rpm - rpm);
    return diff >= 100 || (cur_rpm == 0 ? rpm != 0 : (100 * diff) / cur_rpm > 10);
}

/* The "should this InstanceRecord be re-published" decision of publishInstanceRecord,
 * MM.java:5397-5468 (lock / KV CAS plumbing excluded).  cur = the record currently in the table,
 * fresh = getFreshInstanceRecord() + rpm.  Returns 1 if an update must be written. */
int orc_should_publish(const orc_pod *cur, const orc_pod *fresh, int64_t now, int64_t last_published,
                       int force, int pre_shutdown, int64_t min_space_units)
{
    const int64_t FREQ = 40000, MINP = 2000; /* MM.java:231-232 */
    int64_t last_done = jsub64(now, last_published);
    if (!pre_shutdown && (last_done < MINP || (!force && last_done < FREQ - 1000))) return 0;
    int old = last_done > FREQ * 4;
    if (!cur) return fresh->shutting_down ? 0 : 1; /* :5432-5436: no record and shutting down -> return, else create it */
    int64_t cap = fresh->capacity, used = fresh->used, oldest = fresh->lru_time;
    if (oldest == -1) oldest = INT64_MAX; /* :5423-5425: runtimeCache.oldestTime() of an empty cache */
    int32_t count = fresh->count;
    if (!old) {
        int64_t diff;
        int64_t fr = jsub64(cap, used);
        if (fr < 0) fr = 0;
        if ((cur->shutting_down != 0) == (fresh->shutting_down != 0) &&
            jabs64(jsub64(cur->capacity, cap)) < cap / 50 &&
            (diff = jabs64(jsub64(cur->lru_time, oldest))) < 20000 &&
            (cur->lru_time == INT64_MAX || diff < jsub64(now, cur->lru_time) / 16) &&
            (diff = jabs32(cur->count - count)) < 10 &&
            (cur->count == 0 ? count == 0 : (diff * 100) / cur->count < 15) &&
            (cur->used == 0 ? used == 0 : (jabs64(jsub64(cur->used, used)) * 100) / cur->used < 20) &&
            orc_is_full(orc_remaining(cur), min_space_units) == orc_is_full(fr, min_space_units) &&
            cur->loading_threads == fresh->loading_threads &&
            !loading_change(cur->loading_in_progress, cur->loading_threads, fresh->loading_in_progress) &&
            !load_change(cur->rpm, fresh->rpm))
            return 0;
    } else if ((cur->shutting_down != 0) == (fresh->shutting_down != 0) && cur->capacity == cap &&
               cur->count == count && cur->lru_time == oldest && cur->used == used &&
               cur->loading_threads == fresh->loading_threads &&
               cur->loading_in_progress == fresh->loading_in_progress && cur->rpm == fresh->rpm) {
        return 0;
    }
    return 1;
}
