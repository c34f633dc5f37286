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
    int32_t pad;
};

// sizeEstimate, MM.java:6622-6629
__device__ __forceinline__ int32_t size_estimate_of(const StatsAcc *st, int32_t default_units)
{
    if (st->model_copy_count < 3) return default_units;
    const int32_t narrowed = (int32_t)(uint32_t)(st->total_capacity - st->total_free);  // (int) binds first
    const int32_t average = narrowed / st->model_copy_count;
    return st->model_copy_count > 10 ? average : (int32_t)((uint32_t)average + (uint32_t)default_units) / 2;
}

// spaceToFill accumulation, MM.java:6633-6649
__global__ void proactive_space_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, const StatsAcc *st,
                                       int32_t default_units, PlanScalars *ps)
{
    const bool active = (int64_t)st->total_capacity > 0 && (int64_t)st->total_free > 0;
    int64_t sum = 0;
    if (active) {
        const int32_t se = size_estimate_of(st, default_units);
        if (se != 0) {
            for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
                const mmp_pod_row r = pods[p];
                if (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) continue;
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
__global__ void proactive_scalars_kernel(const StatsAcc *st, int32_t default_units, int64_t now, PlanScalars *ps)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
__device__ __forceinline__ bool proactive_candidate(const mmp_model_row &m, int64_t glru)
{
    return m.n_loaded == 0 && m.n_failed < 2 && (glru == 0 || m.last_used > glru);
}

constexpr int kCompactBlock = 256;

// pass 1: per-block counts of (candidates, qualified)
__global__ __launch_bounds__(kCompactBlock) void proactive_count_kernel(const mmp_model_row *__restrict__ models,
                                                                        int32_t M, const StatsAcc *st,
                                                                        const PlanScalars *ps,
                                                                        int32_t *__restrict__ block_counts,
                                                                        int32_t *__restrict__ n_candidates)
{
    __shared__ int32_t wsum[kCompactBlock / 64], csum[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    bool cand = false, q = false;
    if (i < M) {
        const mmp_model_row m = models[i];
        cand = proactive_candidate(m, st->global_lru);
        q = cand && ps->total_count > 0 && (ps->free_count > 0 || m.last_used > ps->cutoff);
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
                                                                          int32_t M, const StatsAcc *st,
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
        q = proactive_candidate(m, st->global_lru) && ps->total_count > 0 &&
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
