// rebalance_kernels.hpp — the batch rebalancers that generate bursts of load-target decisions
// (SURVEY.md §8 rows a15-a17, a21): "select a set on the device, then K × place_batch".
//
// a17 (leader reaper, MM.java:6616-6747): a capacity budget from a reduction over the instance
// table, then a bounded top-K of the unloaded models by lastUsed (the Java's TreeSet whose
// comparator only looks at lastUsed, so equal timestamps collapse to the first one seen).
// On the device: order-preserving compaction (block counts → scan → ballot scatter), one stable
// descending radix sort of (lastUsed, model) pairs, a second compaction that keeps run starts.
#pragma once
#include "snapshot.hpp"

namespace mmp {

struct PlanScalars {  // mirrors mmp_proactive_info + work counters
    int32_t size_estimate, free_count, total_count, n_candidates, n_selected, error;
    int64_t space_to_fill, cutoff;
    unsigned long long space_acc;  // Σ min(avail, maxLoads*sizeEstimate) before the /2
    int32_t n_qualified;           // candidates that pass the :6683-6685 test (sort input size)
    int32_t n_distinct;            // distinct lastUsed values among them
    int32_t n_ge_cutoff;           // selected entries with lastUsed >= cutoff
    int32_t cand_enabled;          // proactiveLoadCandidates != null (globalStats.totalCapacity > 0, MM.java:6459)
    int64_t cand_glru;             // the reaper's globalLru: 0 <=> the cluster has free space (:6462)
};

// The instance subset a plan is made for (triggerProactiveLoadsForInstanceSubset, MM.java:6616: one call per
// ProhibitedTypeSet partition when type constraints exist, :6473-6488).  pts < 0: the whole cluster.
struct PlanSubset {
    const StatsAcc *global;   // clusterStats: the candidate rule of pruneModelRegistry uses it (:6459-6462, :6574-6577)
    const StatsAcc *stats;    // the subset's stats (== global for the whole cluster)
    const int32_t *pod_pts;   // pod -> partition
    const uint64_t *prohib;   // the partition's prohibited type rows (excludeTypes), null for the whole cluster
    const uint8_t *skip;      // per model: already triggered for an earlier partition (allCandidates.set(i, null)), or null
    int32_t pts, n_types;
};

__device__ __forceinline__ bool plan_excluded(const PlanSubset &U, int32_t i, int32_t type)
{
    if (U.skip && U.skip[i]) return true;
    return U.prohib && type >= 0 && type < U.n_types && ((U.prohib[type >> 6] >> (type & 63)) & 1ull);
}

// sizeEstimate, MM.java:6622-6629
__device__ __forceinline__ int32_t size_estimate_of(const StatsAcc *st, int32_t default_units)
{
    if (st->model_copy_count < 3) return default_units;
    const int32_t narrowed = (int32_t)(uint32_t)(st->total_capacity - st->total_free);  // (int) binds first
    const int32_t average = narrowed / st->model_copy_count;
    return st->model_copy_count > 10 ? average : (int32_t)((uint32_t)average + (uint32_t)default_units) / 2;
}

// spaceToFill accumulation, MM.java:6633-6649
__global__ void proactive_space_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, PlanSubset U,
                                       int32_t default_units, PlanScalars *ps)
{
    const StatsAcc *st = U.stats;
    const bool active = (int64_t)st->total_capacity > 0 && (int64_t)st->total_free > 0;
    int64_t sum = 0;
    if (active) {
        const int32_t se = size_estimate_of(st, default_units);
        if (se != 0) {
            for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
                const mmp_pod_row r = pods[p];
                if (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) continue;
                if (U.pts >= 0 && U.pod_pts[p] != U.pts) continue;  // !excludeTypes.equals(ir.prohibitedTypes), :6635
                const int32_t max_loads = (int32_t)((uint32_t)r.loading_threads * 50u - (uint32_t)r.loading_in_progress);
                if (max_loads <= 0) continue;
                const int64_t avail = jsub64(remaining_of(r.capacity, r.used), r.capacity / 8);
                if (avail > 0) {
                    const int64_t by_loads = (int64_t)(int32_t)((uint32_t)max_loads * (uint32_t)se);
                    sum = (int64_t)((uint64_t)sum + (uint64_t)(avail < by_loads ? avail : by_loads));
                }
            }
        }
    }
    sum = wave_sum_i64(sum);
    if (lane_id() == 0 && sum != 0) atomicAdd(&ps->space_acc, (unsigned long long)sum);
}

// the scalar part of :6621-6664, one lane
__global__ void proactive_scalars_kernel(PlanSubset U, int32_t default_units, int64_t now, PlanScalars *ps)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const StatsAcc *st = U.stats;
    ps->cand_enabled = (int64_t)U.global->total_capacity > 0 ? 1 : 0;
    ps->cand_glru = (int64_t)U.global->total_free > 0 ? 0 : U.global->global_lru;
    int32_t free_count = 0, total_count = 0;
    ps->error = 0;
    ps->size_estimate = 0;
    ps->space_to_fill = 0;
    if ((int64_t)st->total_capacity > 0 && (int64_t)st->total_free > 0) {
        const int32_t se = size_estimate_of(st, default_units);
        ps->size_estimate = se;
        if (se == 0) {
            ps->error = 1;  // the Java throws ArithmeticException at :6651
        } else {
            const int64_t space = (int64_t)ps->space_acc / 2;
            ps->space_to_fill = space;
            free_count = (int32_t)(space / se);
            const int32_t by_cap = (int32_t)((int64_t)st->total_capacity / (20LL * se));
            total_count = free_count > by_cap ? free_count : by_cap;
        }
    }
    ps->free_count = free_count;
    ps->total_count = total_count;
    const int64_t glru = st->global_lru;
    int64_t cutoff = 0;
    if (glru != INT64_MAX) {
        const int64_t third = age_of(glru, now) / 3;
        cutoff = (int64_t)((uint64_t)glru + (uint64_t)(third > 1200000 ? third : 1200000));
    }
    ps->cutoff = cutoff;
}

// candidate predicate = registry rule :6574-6577 ∧ per-candidate test :6683-6685
__device__ __forceinline__ bool proactive_candidate(const mmp_model_row &m, const PlanScalars *ps)
{
    return ps->cand_enabled && m.n_loaded == 0 && m.n_failed < 2 && (ps->cand_glru == 0 || m.last_used > ps->cand_glru);
}

constexpr int kCompactBlock = 256;

// pass 1: per-block counts of (candidates, qualified)
__global__ __launch_bounds__(kCompactBlock) void proactive_count_kernel(const mmp_model_row *__restrict__ models,
                                                                        int32_t M, PlanSubset U,
                                                                        const PlanScalars *ps,
                                                                        int32_t *__restrict__ block_counts,
                                                                        int32_t *__restrict__ n_candidates)
{
    __shared__ int32_t wsum[kCompactBlock / 64], csum[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    bool cand = false, q = false;
    if (i < M) {
        const mmp_model_row m = models[i];
        cand = proactive_candidate(m, ps);
        q = cand && !plan_excluded(U, i, m.type) && ps->total_count > 0 && (ps->free_count > 0 || m.last_used > ps->cutoff);
    }
    const int nq = __popcll(__ballot(q)), nc = __popcll(__ballot(cand));
    if (lane_id() == 0) {
        wsum[threadIdx.x >> 6] = nq;
        csum[threadIdx.x >> 6] = nc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t a = 0, b = 0;
        for (int w = 0; w < kCompactBlock / 64; w++) {
            a += wsum[w];
            b += csum[w];
        }
        block_counts[blockIdx.x] = a;
        if (b) atomicAdd(n_candidates, b);
    }
}

// exclusive scan of up to a few thousand block counts by ONE workgroup; writes the total
__global__ __launch_bounds__(256) void block_scan_kernel(int32_t *__restrict__ counts, int32_t n, int32_t *total)
{
    __shared__ int32_t carry_s, wtot[4];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        const int32_t v = i < n ? counts[i] : 0;
        const int32_t incl = wave_incl_scan_i32(v);
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        int32_t before = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wtot[w];
        if (i < n) counts[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// pass 2: order-preserving scatter of the qualified (lastUsed, model) pairs
__global__ __launch_bounds__(kCompactBlock) void proactive_scatter_kernel(const mmp_model_row *__restrict__ models,
                                                                          int32_t M, PlanSubset U,
                                                                          const PlanScalars *ps,
                                                                          const int32_t *__restrict__ block_off,
                                                                          int64_t *__restrict__ keys,
                                                                          int32_t *__restrict__ vals)
{
    __shared__ int32_t wsum[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    bool q = false;
    int64_t lu = 0;
    if (i < M) {
        const mmp_model_row m = models[i];
        lu = m.last_used;
        q = proactive_candidate(m, ps) && !plan_excluded(U, i, m.type) && ps->total_count > 0 &&
            (ps->free_count > 0 || m.last_used > ps->cutoff);
    }
    const uint64_t b = __ballot(q);
    const int lane = lane_id();
    if (lane == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    int32_t off = block_off[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += wsum[w];
    if (q) {
        const int dst = off + __popcll(b & ((1ull << lane) - 1ull));
        keys[dst] = lu;
        vals[dst] = i;
    }
}

// after the stable descending sort: run starts = distinct lastUsed values (first model wins)
__global__ __launch_bounds__(kCompactBlock) void distinct_count_kernel(const int64_t *__restrict__ keys, int32_t n,
                                                                       int32_t *__restrict__ block_counts)
{
    __shared__ int32_t wsum[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    const bool start = i < n && (i == 0 || keys[i] != keys[i - 1]);
    const int c = __popcll(__ballot(start));
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t a = 0;
        for (int w = 0; w < kCompactBlock / 64; w++) a += wsum[w];
        block_counts[blockIdx.x] = a;
    }
}

__global__ __launch_bounds__(kCompactBlock) void distinct_scatter_kernel(const int64_t *__restrict__ keys,
                                                                         const int32_t *__restrict__ vals, int32_t n,
                                                                         const int32_t *__restrict__ block_off,
                                                                         PlanScalars *ps, int32_t max_out,
                                                                         int32_t *__restrict__ out_model,
                                                                         int64_t *__restrict__ out_lu)
{
    __shared__ int32_t wsum[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    const bool start = i < n && (i == 0 || keys[i] != keys[i - 1]);
    const uint64_t b = __ballot(start);
    const int lane = lane_id();
    if (lane == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    int32_t off = block_off[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += wsum[w];
    bool ge = false;
    if (start) {
        const int dst = off + __popcll(b & ((1ull << lane) - 1ull));
        if (dst < ps->total_count) {  // toLoad keeps the totalProactiveLoadCount largest values
            if (dst < max_out) {
                out_model[dst] = vals[i];
                out_lu[dst] = keys[i];
            }
            ge = keys[i] >= ps->cutoff;
        }
    }
    const int nge = __popcll(__ballot(ge));
    if (lane == 0 && nge) atomicAdd(&ps->n_ge_cutoff, nge);
}

// :6709-6734 — free space first, then only entries at or above the cutoff (a prefix, the list is descending)
__global__ void proactive_final_kernel(PlanScalars *ps)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t n_sel = ps->n_distinct < ps->total_count ? ps->n_distinct : ps->total_count;
    const int32_t by_free = ps->free_count < n_sel ? (ps->free_count > 0 ? ps->free_count : 0) : n_sel;
    const int32_t by_cut = ps->n_ge_cutoff < n_sel ? ps->n_ge_cutoff : n_sel;
    ps->n_selected = by_free > by_cut ? by_free : by_cut;
}

}  // namespace mmp

namespace mmp {

// ---- a15 ---------------------------------------------------------------------------------------
// getExcludeSet(), MM.java:5835-5856: pods of clusterState (present rows) other than self whose
// published rpm exceeds max(4*threshold, ourRpm - 2*threshold)
__global__ void overloaded_pods_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t self_pod,
                                       int32_t max_rpm, uint8_t *__restrict__ overloaded, int32_t *__restrict__ count)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool ov = false;
    if (p < P) {
        const mmp_pod_row r = pods[p];
        ov = !(r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) && p != self_pod && r.rpm > max_rpm;
        overloaded[p] = ov ? 1 : 0;
    }
    const int c = __popcll(__ballot(ov));
    if (lane_id() == 0 && c) atomicAdd(count, c);
}

struct ScaleupArgs {
    const mmp_cache_entry *entries;
    const mmp_model_row *models;
    const int32_t *ent_pod;
    const int64_t *ent_time;
    const StatsAcc *stats;   // cluster-wide
    const StatsAcc *tstats;  // [T_rows] typeSetStats(type), MM.java:5691
    const uint8_t *overloaded;
    const int32_t *excluded_count;
    mmp_scaleup_out *outs;
    mmp_scaleup_params p;
    int32_t n, n_models, P;
    int32_t T_rows, has_tc;  // has_tc: typeConstraints != null
};

// loadedSince, MM.java:5860-5871
__device__ __forceinline__ bool loaded_since(const int32_t *pods, const int64_t *times, int n, int64_t cutoff,
                                             int32_t ignore)
{
    for (int i = 0; i < n; i++) {
        if (ignore >= 0 && pods[i] == ignore) continue;
        if (times[i] > cutoff) return true;
    }
    return false;
}

// rateTrackingTask body for one used-since-last-run entry, MM.java:5687-5806
__global__ void scaleup_plan_kernel(ScaleupArgs A)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n) return;
    const mmp_cache_entry ce = A.entries[e];
    const mmp_scaleup_params &p = A.p;
    mmp_scaleup_out o;
    o.action = MMP_SCALE_NONE;
    o.copies = 0;
    o.timestamp = 0;
    o.new_i1 = ce.earlier_use_iteration;
    o.new_i2 = ce.last_used_iteration;
    o.heavy = 0;
    const int64_t time_delta = jsub64(p.now, p.last_check_time);
    const int32_t lower = p.iteration_counter - p.second_copy_max_age_iters;
    const int32_t upper = p.iteration_counter - p.second_copy_min_age_iters;
    const bool have_model = ce.model >= 0 && ce.model < A.n_models;
    mmp_model_row mr{};
    if (have_model) mr = A.models[ce.model];
    // typeSetStats(ce.modelInfo.serviceType), MM.java:5691 (cluster-wide for an entry without a registry record)
    const StatsAcc *st = (A.has_tc && have_model) ? &A.tstats[(mr.type < 0 || mr.type >= A.T_rows) ? 0 : mr.type] : A.stats;
    int32_t suitable = A.stats->instance_count;  // instCount, :5692
    if (A.has_tc) {                              // :5693-5700: a type confined to one instance is skipped outright
        suitable = st->instance_count;
        if (suitable < 2) {
            o.rpm = 0;
            A.outs[e] = o;
            return;
        }
    }
    const int32_t scale_up = p.scale_up_rpm_threshold;
    const int32_t heavy = (int32_t)((uint32_t)scale_up * 3u) / 4;
    const int32_t rpm = (int32_t)((ce.interval_count * 60000) / time_delta);
    o.rpm = rpm;
    if (rpm > heavy) o.heavy = 1;
    do {
        if (!have_model) break;
        const int32_t loaded = mr.n_loaded, failed = mr.n_failed;
        if (loaded == 0) break;
        int32_t cand = suitable - (loaded + failed);
        if (cand <= 0) break;
        const int32_t *lp = A.ent_pod + mr.ent_off;
        const int64_t *lt = A.ent_time + mr.ent_off;
        if (loaded == 1) {  // :5726-5758
            const int32_t i1 = ce.earlier_use_iteration, i2 = ce.last_used_iteration;
            bool i1in = false, i2in = false;
            if (i2 >= lower && i1 <= upper) {
                i1in = i1 >= lower;
                i2in = i2 <= upper;
            }
            if (i2in || !i1in) o.new_i1 = i2;
            o.new_i2 = p.iteration_counter;
            if (i1in || i2in) {
                const int64_t tc = (int64_t)st->total_capacity, tf = (int64_t)st->total_free;
                if (tc == 0) break;  // the Java throws ArithmeticException here; caught, entry skipped
                if ((10 * tf) / tc >= 1 || jsub64(p.now, st->global_lru) > p.second_copy_lru_threshold_ms) {
                    o.action = MMP_SCALE_SECOND_COPY;
                    o.timestamp = p.last_check_time;
                    o.copies = 1;
                    break;
                }
            }
        }
        if (rpm < scale_up) break;
        if (scale_up == 0) break;
        const int64_t recent = jsub64(p.now, time_delta + p.rate_check_interval_ms + 2 * p.assume_completed_ms);
        if (loaded_since(lp, lt, loaded, recent, p.self_pod)) break;
        const int32_t excluded = *A.excluded_count;
        if (excluded != 0) {  // :5776-5787
            int32_t members = 0;  // excluded pods that ARE in loaded ∪ failed
            for (int k = 0; k < loaded + failed; k++) {
                const int32_t iid = lp[k];
                if (iid >= 0 && iid < A.P && A.overloaded[iid]) members++;
            }
            cand -= (excluded - members);
            cand -= excluded;
            if (cand <= 0) break;
        }
        int32_t copies = rpm / scale_up < cand ? rpm / scale_up : cand;
        if (copies > 2) copies = copies < suitable / 3 ? copies : suitable / 3;
        o.action = MMP_SCALE_UP;
        o.copies = copies;
        o.timestamp = p.now + 20000;
    } while (false);
    A.outs[e] = o;
}

// ---- a16 ---------------------------------------------------------------------------------------
struct ScaledownArgs {
    const mmp_cache_entry *entries;
    const mmp_model_row *models;
    const int32_t *ent_pod;
    const int64_t *ent_time;
    const mmp_pod_row *pods;
    const int32_t *pos_of;
    const StatsAcc *stats;  // instanceSetStats() (MM.java:1446-1448, :6228): this instance's partition, or cluster-wide
    uint8_t *decide;   // per entry: removeModelCopies would remove if canRemove
    uint8_t *removed;  // final
    mmp_scaledown_params p;
    int32_t n, n_models, P;
};

// removeModelCopies with canRemove == true, MM.java:6197-6310 — everything except the running budget
__global__ void scaledown_decide_kernel(ScaledownArgs A)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n) return;
    const mmp_cache_entry ce = A.entries[e];
    const mmp_scaledown_params &p = A.p;
    bool removed = false;
    do {
        if (ce.last_used == 0 || ce.model < 0 || ce.model >= A.n_models) break;
        const mmp_model_row mr = A.models[ce.model];
        const int32_t num = mr.n_loaded;
        if (num < 2) break;
        const int64_t tc = (int64_t)A.stats->total_capacity, tf = (int64_t)A.stats->total_free;
        if (tc == 0 || tf * 100 / tc > 5) break;  // :6229
        const int32_t *lp = A.ent_pod + mr.ent_off;
        const int64_t *lt = A.ent_time + mr.ent_off;
        int32_t other = -1;
        for (int k = 0; k < num; k++) {
            const int32_t iid = lp[k];
            if (iid != p.self_pod && iid >= 0 && iid < A.P &&
                !(A.pods[iid].flags & (MMP_POD_TOMBSTONE | MMP_POD_SHUTTING_DOWN))) {
                other = iid;
                break;
            }
        }
        if (other < 0) break;
        const int64_t glru = A.stats->global_lru;
        if (num == 2) {
            const int64_t cache_age = jsub64(p.now, glru);
            int64_t sda = cache_age / 10;
            if (ce.last_heavy_time == 0 || jsub64(p.now, ce.last_heavy_time) < cache_age / 5)
                sda = 36000000LL < sda ? 36000000LL : sda;  // SECOND_COPY_REMOVE_MAX_AGE_MS, MM.java:257
            if (jsub64(p.now, ce.last_used) > sda) {
                const bool self_ok = p.self_pod >= 0 && p.self_pod < A.P &&
                                     !(A.pods[p.self_pod].flags & (MMP_POD_TOMBSTONE | MMP_POD_SHUTTING_DOWN));
                if (!self_ok) break;
                if (A.pos_of[other] > A.pos_of[p.self_pod]) break;  // PLACEMENT_ORDER.compare(other, this) > 0
                removed = true;
            }
        } else {
            if (ce.last_unload_time > 0 && jsub64(p.now, ce.last_unload_time) < 8 * p.rate_check_interval_ms) break;
            if (loaded_since(lp, lt, num, p.now - 1800000, -1)) break;
            int64_t min_age = (int64_t)(3ull * (uint64_t)glru + 10400000ull) / 100;  // absolute timestamp: quirk B#13
            min_age = min_age < 600000 ? 600000 : (min_age > 18000000 ? 18000000 : min_age);
            if (jsub64(p.now, ce.last_heavy_time) < min_age) break;
            const int64_t since = jsub64(p.now, p.last_check_time);
            if (since < p.rate_check_interval_ms / 10) break;
            const int64_t rpm = ce.interval_count == 0 ? 0 : (60000 * ce.interval_count) / since;
            if (rpm > ((int64_t)p.scale_up_rpm_threshold * 2) / 3) break;
            removed = true;
        }
    } while (false);
    A.decide[e] = removed ? 1 : 0;
}

// the janitor's running allowance, MM.java:6117-6136: a serial recurrence over the (short) list
__global__ void scaledown_budget_kernel(ScaledownArgs A)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int64_t max_weight = A.p.adjusted_cache_capacity / 20;
    int32_t removed_count = 0;
    for (int e = 0; e < A.n; e++) {
        const int32_t w = A.entries[e].weight;
        const bool can = removed_count == 0 || w <= max_weight;
        const bool rem = !A.p.shutting_down && can && A.decide[e];
        A.removed[e] = rem ? 1 : 0;
        if (rem) {
            removed_count++;
            max_weight -= w;
        }
    }
}

// ---- a21 ---------------------------------------------------------------------------------------
__global__ void migration_plan_kernel(const mmp_cache_entry *__restrict__ entries, int32_t n,
                                      const mmp_model_row *__restrict__ models, int32_t n_models,
                                      const int32_t *__restrict__ ent_pod, int32_t self_pod, int64_t cutoff,
                                      uint8_t *__restrict__ action, uint8_t *__restrict__ wait)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const mmp_cache_entry ce = entries[e];
    uint8_t a = 0, w = 0;
    if (ce.model >= 0 && ce.model < n_models && !(ce.flags & MMP_CE_FAILED)) {
        const mmp_model_row mr = models[ce.model];
        bool has_us = false;
        for (int k = 0; k < mr.n_loaded; k++)
            if (ent_pod[mr.ent_off + k] == self_pod) has_us = true;  // :7008
        if (has_us && ce.last_used > 0) {
            a = 1;
            w = ce.last_used >= cutoff ? 1 : 0;
        }
    }
    action[e] = a;
    wait[e] = w;
}

}  // namespace mmp
