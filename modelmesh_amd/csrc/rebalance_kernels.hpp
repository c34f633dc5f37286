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
    // the bucketed plan (no host round trip): key range of the qualified candidates, launch tickets, the fallback flag
    long long kmin, kmax;
    unsigned int ticket[4];
    int32_t overflow;              // a bucket holds more than kPlanBucketMax candidates: the host takes the sorted path
    int32_t bucket_map;            // 0: linear in age; m >= 8: floating (exponent + m mantissa bits of age)
    long long t_phase[16];         // -DMMP_PLAN_CLOCK builds: the 100 MHz clock at the one-launch plan's phase boundaries (workgroup 0)
    // the one-launch plan's barrier (plan_grid_barrier): the arrivals and the flag that lets the workgroups go, a cache line each
    alignas(128) unsigned int bar_count[32];
    alignas(128) unsigned int bar_flag[32];
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
__device__ __forceinline__ void proactive_scalars(const PlanSubset &U, int32_t default_units, int64_t now, PlanScalars *ps,
                                                  unsigned long long space_acc)
{
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
            const int64_t space = (int64_t)space_acc / 2;
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

__global__ void proactive_scalars_kernel(PlanSubset U, int32_t default_units, int64_t now, PlanScalars *ps)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) proactive_scalars(U, default_units, now, ps, ps->space_acc);
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

// ---- the same plan WITHOUT the sort and without the host reading the qualified count in the middle ---------------------------
// The TreeSet keeps, per distinct lastUsed, the first model seen, and toLoad is its totalProactiveLoadCount largest values.  A
// model's place in that order is (number of distinct qualified lastUsed values above its own), and equal values are neighbours
// under ANY monotone partition of the key range: so the qualified candidates are binned into kPlanBuckets key-range buckets
// (bucket 0 = the newest), each bucket resolves its own ties and local ranks by counting (all pairs, in LDS, one wavefront per
// bucket), and a scan of the buckets' distinct counts turns local ranks into global ones.  lastUsed values are timestamps: dense
// near `now`, a long tail behind — so two monotone maps of age = newest - lastUsed are histogrammed side by side, a LINEAR one
// (age >> shift) and a FLOATING one (exponent + 8 mantissa bits of age: buckets 1/256 of their age wide), and the launch
// that scans the histograms picks the map with the smaller fullest bucket.  Six dependent launches, every size read on the device:
//   space (+ scalars) -> qualify: count, key range -> histograms (+ choice, scan) -> bin -> per-bucket rank (+ scan) -> emit (+ final)
// A bucket above kPlanBucketMax under both maps (thousands of models on one millisecond) raises `overflow`: the sorted path runs.
constexpr int kPlanBuckets = 16384, kPlanLinBits = 14, kPlanMantBits = 8, kPlanBucketMax = 1024, kPlanLdsBucket = 256;

// the floating map's mantissa bits: as many as keep the bucket of the oldest key (age = range) inside the table — 8 always do
// ((64 - 8 + 1) << 8 = 14592); a key range of 2^34 ms takes 9 (buckets half as wide, half as full)
__device__ __forceinline__ int plan_mantissa_bits(uint64_t range)
{
    const int e_max = range ? 63 - __builtin_clzll(range) : 0;
    int m = kPlanMantBits;
    while (m < 13 && ((e_max - (m + 1) + 2) << (m + 1)) <= kPlanBuckets && e_max >= m + 1) m++;
    return m;
}

// true in exactly one workgroup of the launch: the one that arrives last (its view includes every other workgroup's writes)
__device__ __forceinline__ bool last_workgroup(unsigned int *ticket)
{
    __shared__ unsigned int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) __threadfence();
    return s_last != 0;
}

__global__ __launch_bounds__(256) void proactive_space_scalars_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, PlanSubset U,
                                                                      int32_t default_units, int64_t now, PlanScalars *ps)
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
                if (U.pts >= 0 && U.pod_pts[p] != U.pts) continue;
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
    if (last_workgroup(&ps->ticket[0]) && threadIdx.x == 0) {
        const unsigned long long acc = __hip_atomic_load(&ps->space_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        proactive_scalars(U, default_units, now, ps, acc);
    }
}

__device__ __forceinline__ bool proactive_qualifies(const mmp_model_row &m, int i, const PlanSubset &U, const PlanScalars *ps)
{
    return proactive_candidate(m, ps) && !plan_excluded(U, i, m.type) && ps->total_count > 0 &&
           (ps->free_count > 0 || m.last_used > ps->cutoff);
}

__device__ __forceinline__ int64_t wave_max_i64(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t t = (int64_t)shfl_u64((uint64_t)v, lane_id() ^ o);
        v = t > v ? t : v;
    }
    return v;
}

// Per-workgroup partials of pass 1 (no atomics: a few thousand wavefronts adding into one cache line serialise at ~7 ns each)
struct PlanPartial {
    int32_t n_cand, n_qual;
    long long kmin, kmax;
    double key_sum;  // of the qualified keys (only to choose the bucket map)
};

// pass 1 over the registry: how many candidates / qualified, and the qualified keys' range — per workgroup
__global__ __launch_bounds__(kCompactBlock) void proactive_qualify_kernel(const mmp_model_row *__restrict__ models, int32_t M,
                                                                          PlanSubset U, const PlanScalars *ps, PlanPartial *__restrict__ part,
                                                                          int32_t *__restrict__ hist)
{
    __shared__ PlanPartial s_w[kCompactBlock / 64];
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    for (int k = i; k < kPlanBuckets; k += gridDim.x * kCompactBlock) hist[k] = 0;  // (the next launch's histogram)
    bool cand = false, q = false;
    int64_t lu = 0;
    if (i < M) {
        const mmp_model_row m = models[i];
        lu = m.last_used;
        cand = proactive_candidate(m, ps);
        q = cand && proactive_qualifies(m, i, U, ps);
    }
    const int nq = __popcll(__ballot(q)), nc = __popcll(__ballot(cand));
    const int64_t lo = wave_min_i64(q ? lu : INT64_MAX), hi = wave_max_i64(q ? lu : INT64_MIN);
    double sum = q ? (double)lu : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = PlanPartial{nc, nq, lo, hi, sum};
    __syncthreads();
    if (threadIdx.x == 0) {
        PlanPartial t = s_w[0];
        for (int w = 1; w < kCompactBlock / 64; w++) {
            t.n_cand += s_w[w].n_cand;
            t.n_qual += s_w[w].n_qual;
            t.kmin = s_w[w].kmin < t.kmin ? s_w[w].kmin : t.kmin;
            t.kmax = s_w[w].kmax > t.kmax ? s_w[w].kmax : t.kmax;
            t.key_sum += s_w[w].key_sum;
        }
        part[blockIdx.x] = t;
    }
}

// Every workgroup of the next launch folds the partials itself (a few hundred rows from L2: cheaper than a launch of its own);
// workgroup 0 also stores the totals and the map choice in `ps`.  The map: timestamps dense near the newest with a long tail
// (mean age far below half the range) -> floating; spread over the range -> linear.
struct PlanRange {
    int64_t kmin, kmax;
    int32_t n_qual, map;
};
__device__ __forceinline__ PlanRange plan_fold(const PlanPartial *__restrict__ part, int nb, PlanScalars *ps, bool store)
{
    __shared__ PlanPartial s_f[kCompactBlock / 64];
    __shared__ PlanRange s_r;
    PlanPartial t{0, 0, INT64_MAX, INT64_MIN, 0.0};
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        const PlanPartial v = part[k];
        t.n_cand += v.n_cand;
        t.n_qual += v.n_qual;
        t.kmin = v.kmin < t.kmin ? v.kmin : t.kmin;
        t.kmax = v.kmax > t.kmax ? v.kmax : t.kmax;
        t.key_sum += v.key_sum;
    }
    t.n_cand = wave_sum_i32(t.n_cand);
    t.n_qual = wave_sum_i32(t.n_qual);
    t.kmin = wave_min_i64(t.kmin);
    t.kmax = wave_max_i64(t.kmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t.key_sum += __shfl_xor(t.key_sum, o, 64);
    if (lane_id() == 0) s_f[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            t.n_cand += s_f[w].n_cand;
            t.n_qual += s_f[w].n_qual;
            t.kmin = s_f[w].kmin < t.kmin ? s_f[w].kmin : t.kmin;
            t.kmax = s_f[w].kmax > t.kmax ? s_f[w].kmax : t.kmax;
            t.key_sum += s_f[w].key_sum;
        }
        int map = 0;
        if (t.n_qual > 0) {
            const double range = (double)((uint64_t)t.kmax - (uint64_t)t.kmin);
            const double mean_age = (double)t.kmax - t.key_sum / (double)t.n_qual;
            map = mean_age * 8.0 < range ? plan_mantissa_bits((uint64_t)t.kmax - (uint64_t)t.kmin) : 0;
        }
        s_r = PlanRange{t.kmin, t.kmax, t.n_qual, map};
        if (store) {
            ps->n_candidates = t.n_cand;
            ps->n_qualified = t.n_qual;
            ps->kmin = t.kmin;
            ps->kmax = t.kmax;
            ps->bucket_map = map;
        }
    }
    __syncthreads();
    return s_r;
}

// buckets of a qualified key, 0 = the newest; both monotone in the key, so equal keys share a bucket and the buckets are in
// TreeSet order
__device__ __forceinline__ int plan_bucket_linear(uint64_t age, uint64_t range)
{
    const int bits = range ? 64 - __builtin_clzll(range) : 0;
    return (int)(age >> (bits > kPlanLinBits ? bits - kPlanLinBits : 0));
}
__device__ __forceinline__ int plan_bucket_floating(uint64_t age, int m)
{
    if (age < (1ull << m)) return (int)age;
    const int e = 63 - __builtin_clzll(age);  // >= m
    return ((e - m + 1) << m) + (int)((age >> (e - m)) & ((1u << m) - 1));  // < (e_max - m + 2) << m
}
__device__ __forceinline__ int plan_bucket_of(int64_t key, int64_t kmin, int64_t kmax, int map)
{
    const uint64_t age = (uint64_t)kmax - (uint64_t)key;
    return map ? plan_bucket_floating(age, map) : plan_bucket_linear(age, (uint64_t)kmax - (uint64_t)kmin);
}
__device__ __forceinline__ int plan_bucket(int64_t key, const PlanScalars *ps) { return plan_bucket_of(key, ps->kmin, ps->kmax, ps->bucket_map); }

// hist[b] += 1 for every lane with `on`, one atomic per DISTINCT bucket of the wavefront
__device__ __forceinline__ void wave_hist_add(int32_t *__restrict__ hist, int b, bool on)
{
    uint64_t todo = __ballot(on);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int b0 = __shfl(b, leader, 64);
        const uint64_t same = __ballot(on && b == b0) & todo;
        if (lane_id() == leader) atomicAdd(&hist[b0], (int32_t)__popcll(same));
        todo &= ~same;
    }
}

// pass 2: the histogram of the chosen map
__global__ __launch_bounds__(kCompactBlock) void proactive_hist_kernel(const mmp_model_row *__restrict__ models, int32_t M, PlanSubset U,
                                                                       PlanScalars *ps, const PlanPartial *__restrict__ part, int nb,
                                                                       int32_t *__restrict__ hist)
{
    const PlanRange R = plan_fold(part, nb, ps, blockIdx.x == 0);
    if (R.n_qual <= 0) return;
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    bool q = false;
    int b = 0;
    if (i < M) {
        const mmp_model_row m = models[i];
        q = proactive_qualifies(m, i, U, ps);
        b = q ? plan_bucket_of(m.last_used, R.kmin, R.kmax, R.map) : 0;
    }
    wave_hist_add(hist, b, q);
}

// exclusive scan of kPlanBuckets counts by ONE workgroup of 1024; off[kPlanBuckets] = total, returned in every thread;
// *max_out = the fullest bucket
constexpr int kPlanScanBlock = 1024;
template <int BLOCK = kPlanScanBlock>
__device__ __forceinline__ int32_t bucket_scan(const int32_t *cnt, int32_t *off, int32_t *off2, int32_t *max_out)
{
    __shared__ int32_t wtot[BLOCK / 64], wmax[BLOCK / 64];
    constexpr int per = kPlanBuckets / BLOCK;
    static_assert(per % 4 == 0, "a thread's counts are read as int4");
    int32_t v[per], mine = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < per; k += 4) {
        const int4 x = reinterpret_cast<const int4 *>(cnt)[(threadIdx.x * per + k) >> 2];
        v[k] = x.x;
        v[k + 1] = x.y;
        v[k + 2] = x.z;
        v[k + 3] = x.w;
    }
#pragma unroll
    for (int k = 0; k < per; k++) {
        mine += v[k];
        mx = v[k] > mx ? v[k] : mx;
    }
    const int32_t incl = wave_incl_scan_i32(mine);
    mx = (int32_t)wave_max_i64(mx);
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    if (lane_id() == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    int32_t before = incl - mine, total = 0, m2 = 0;
    for (int w = 0; w < BLOCK / 64; w++) {
        if (w < (int)(threadIdx.x >> 6)) before += wtot[w];
        total += wtot[w];
        m2 = wmax[w] > m2 ? wmax[w] : m2;
    }
#pragma unroll
    for (int k = 0; k < per; k += 4) {
        int4 o;
        o.x = before;
        o.y = o.x + v[k];
        o.z = o.y + v[k + 1];
        o.w = o.z + v[k + 2];
        before = o.w + v[k + 3];
        reinterpret_cast<int4 *>(off)[(threadIdx.x * per + k) >> 2] = o;
        if (off2) reinterpret_cast<int4 *>(off2)[(threadIdx.x * per + k) >> 2] = o;
    }
    if (threadIdx.x == BLOCK - 1) off[kPlanBuckets] = total;
    if (max_out) *max_out = m2;
    return total;
}
__global__ __launch_bounds__(kPlanScanBlock) void proactive_scan_kernel(PlanScalars *ps, const int32_t *__restrict__ hist,
                                                                        int32_t *__restrict__ off, int32_t *__restrict__ cur)
{
    int32_t fullest = 0;
    bucket_scan(hist, off, cur, &fullest);
    if (threadIdx.x == 0 && fullest > kPlanBucketMax) ps->overflow = 1;  // the launches behind this one return at once
}

// pass 3: the qualified (lastUsed, model) pairs into their buckets (any order inside a bucket: ranks are counted, not sorted)
__global__ __launch_bounds__(kCompactBlock) void proactive_bin_kernel(const mmp_model_row *__restrict__ models, int32_t M, PlanSubset U,
                                                                      const PlanScalars *ps, int32_t *__restrict__ cur,
                                                                      int64_t *__restrict__ keys, int32_t *__restrict__ vals)
{
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    if (ps->n_qualified <= 0 || ps->overflow || i >= M) return;
    const mmp_model_row m = models[i];
    if (!proactive_qualifies(m, i, U, ps)) return;
    const int dst = atomicAdd(&cur[plan_bucket(m.last_used, ps)], 1);
    keys[dst] = m.last_used;
    vals[dst] = i;
}

// a bucket above the LDS tile: one wavefront, the pairs read from global memory (L2) on every pass; rank[] carries the
// run-start flags between the two passes (-1 = duplicate; a run start's rank is >= 0 before and after it is counted)
__device__ __forceinline__ void rank_heavy_bucket(const int64_t *__restrict__ keys, const int32_t *__restrict__ vals, int cnt, int64_t cutoff,
                                                  int32_t *rank, int32_t *__restrict__ dcnt, int32_t *__restrict__ dge)
{
    const int lane = lane_id();
    int32_t starts = 0, ge = 0;
    for (int e = lane; e < cnt; e += 64) {
        const int64_t key = keys[e];
        const int32_t val = vals[e];
        bool first = true;
        for (int j = 0; j < cnt; j++) first &= !(keys[j] == key && vals[j] < val);
        rank[e] = first ? 0 : -1;
        starts += first ? 1 : 0;
        ge += (first && key >= cutoff) ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int e = lane; e < cnt; e += 64) {
        if (__hip_atomic_load(&rank[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 0) continue;
        const int64_t key = keys[e];
        int32_t r = 0;
        for (int j = 0; j < cnt; j++)
            r += (keys[j] > key && __hip_atomic_load(&rank[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 0) ? 1 : 0;
        __hip_atomic_store(&rank[e], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    starts = wave_sum_i32(starts);
    ge = wave_sum_i32(ge);
    if (lane == 0) {
        *dcnt = starts;
        *dge = ge;
    }
}

// one WAVEFRONT per bucket (four buckets per workgroup): an entry is a run start if no equal key has a lower model index (the
// TreeSet's first one seen); its local rank = run starts of the bucket with a larger key.  rank[pos] = local rank, -1 = duplicate.
// dcnt[b] = run starts of the bucket, dge[b] = those of them at or above the cutoff.
__device__ __forceinline__ void plan_rank_bucket(int b, int64_t *s_key, int32_t *s_val, const int64_t *keys, const int32_t *vals,
                                                 const int32_t *off, int64_t cutoff, int32_t *rank, int32_t *dcnt, int32_t *dge)
{
    const int lane = lane_id();
    const int lo = off[b], cnt = off[b + 1] - lo;
    if (cnt > kPlanLdsBucket && cnt <= kPlanBucketMax) {  // a heavy bucket (rare): the same counting, the bucket read from L2
        rank_heavy_bucket(keys + lo, vals + lo, cnt, cutoff, rank + lo, dcnt + b, dge + b);
        return;
    }
    int32_t starts = 0, ge = 0;
    if (cnt == 1) {
        if (lane == 0) {
            rank[lo] = 0;
            starts = 1;
            ge = keys[lo] >= cutoff;
        }
    } else if (cnt > 1 && cnt <= kPlanLdsBucket) {
        for (int j = lane; j < cnt; j += 64) {
            s_key[j] = keys[lo + j];
            s_val[j] = vals[lo + j];
        }
        wave_sync();
        constexpr int per = kPlanLdsBucket / 64;
        uint32_t first = 0;
        for (int k = 0; k < per && k * 64 < cnt; k++) {
            const int e = lane + k * 64;
            if (e < cnt) {
                const int64_t key = s_key[e];
                const int32_t val = s_val[e];
                bool f = true;
                for (int j = 0; j < cnt; j++) f &= !(s_key[j] == key && s_val[j] < val);
                first |= (uint32_t)f << k;
            }
        }
        wave_sync();
        for (int k = 0; k < per && k * 64 < cnt; k++) {  // a duplicate leaves the counting: its index becomes -1
            const int e = lane + k * 64;
            if (e < cnt && !((first >> k) & 1)) s_val[e] = -1;
        }
        wave_sync();
        for (int k = 0; k < per && k * 64 < cnt; k++) {
            const int e = lane + k * 64;
            if (e >= cnt) continue;
            int32_t r = -1;
            if ((first >> k) & 1) {
                const int64_t key = s_key[e];
                r = 0;
                for (int j = 0; j < cnt; j++) r += (s_val[j] >= 0 && s_key[j] > key) ? 1 : 0;
                starts++;
                ge += key >= cutoff;
            }
            rank[lo + e] = r;
        }
        wave_sync();  // (the tile is the wavefront's again for its next bucket)
    }
    starts = wave_sum_i32(starts);
    ge = wave_sum_i32(ge);
    if (lane == 0) {
        dcnt[b] = starts;
        dge[b] = ge;
    }
}
__global__ __launch_bounds__(256) void proactive_bucket_rank_kernel(const int64_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                                                    const int32_t *__restrict__ off, const PlanScalars *ps,
                                                                    int32_t *__restrict__ rank, int32_t *__restrict__ dcnt,
                                                                    int32_t *__restrict__ dge)
{
    __shared__ int64_t s_key_all[4][kPlanLdsBucket];
    __shared__ int32_t s_val_all[4][kPlanLdsBucket];
    if (ps->overflow) return;
    const int wv = threadIdx.x >> 6;
    plan_rank_bucket(blockIdx.x * 4 + wv, s_key_all[wv], s_val_all[wv], keys, vals, off, ps->cutoff, rank, dcnt, dge);
}

// distinct counts -> global rank offsets, and :6709-6734: free space first, then only entries at or above the cutoff (the list is
// descending, so those are a prefix: min(totalCount, distinct values >= cutoff))
__global__ __launch_bounds__(kPlanScanBlock) void proactive_scan2_kernel(PlanScalars *ps, const int32_t *__restrict__ dcnt,
                                                                         const int32_t *__restrict__ dge, int32_t *__restrict__ doff)
{
    __shared__ int32_t s_ge[kPlanScanBlock / 64];
    if (ps->overflow) return;
    const int32_t total = bucket_scan(dcnt, doff, nullptr, nullptr);
    constexpr int per = kPlanBuckets / kPlanScanBlock;
    int32_t g = 0;
#pragma unroll
    for (int k = 0; k < per; k += 4) {
        const int4 x = reinterpret_cast<const int4 *>(dge)[(threadIdx.x * per + k) >> 2];
        g += x.x + x.y + x.z + x.w;
    }
    g = wave_sum_i32(g);
    __syncthreads();
    if (lane_id() == 0) s_ge[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t n_ge = 0;
        for (int w = 0; w < kPlanScanBlock / 64; w++) n_ge += s_ge[w];
        ps->n_distinct = total;
        const int32_t n_sel = total < ps->total_count ? total : ps->total_count;
        const int32_t by_free = ps->free_count < n_sel ? (ps->free_count > 0 ? ps->free_count : 0) : n_sel;
        const int32_t by_cut = n_ge < n_sel ? n_ge : n_sel;
        ps->n_ge_cutoff = by_cut;
        ps->n_selected = by_free > by_cut ? by_free : by_cut;
    }
}

// global rank = distinct values in newer buckets + local rank; the first totalProactiveLoadCount of them are toLoad
__global__ __launch_bounds__(kCompactBlock) void proactive_emit_kernel(const int64_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                                                       const int32_t *__restrict__ rank, const int32_t *__restrict__ doff,
                                                                       const PlanScalars *ps, int32_t max_out, int32_t *__restrict__ out_model,
                                                                       int64_t *__restrict__ out_lu)
{
    const int i = blockIdx.x * kCompactBlock + threadIdx.x;
    if (i >= ps->n_qualified || ps->overflow) return;
    const int32_t lr = rank[i];
    if (lr < 0) return;
    const int64_t key = keys[i];
    const int32_t dst = doff[plan_bucket(key, ps)] + lr;
    if (dst < ps->total_count && dst < max_out) {
        out_model[dst] = vals[i];
        out_lu[dst] = key;
    }
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

// ---- the plan as ONE launch (round 5) ----------------------------------------------------------------
// The eight launches above each wait for the one before: 58 us on C3, of which the kernels' own work is a third.  Here the same
// steps are phases of ONE launch whose workgroups are all on the chip (at most two per compute unit: a grid-wide barrier cannot
// wait for a workgroup that has not started), separated by four barriers.
//
// What a barrier costs decides the design.  With ordinary loads and stores every workgroup needs a device-scope release and
// acquire fence at every barrier — on gfx950 a write-back and an invalidation of its XCD's whole L2 each, 400 workgroups x 2 of
// them per barrier: 16-36 us per barrier, measured (tools/plan_clock.py).  So everything one workgroup writes and another reads
// inside this launch (histograms, offsets, the binned pairs, ranks, the partials) goes through DEVICE-SCOPE relaxed atomics —
// `sc1` loads and stores, coherent at the memory side without any cache maintenance — and a barrier is: every lane waits for its
// own stores (s_waitcnt vmcnt(0)), the workgroup barrier, one counter increment, one flag.  The registry and the instance table
// (inputs: never written here) are read the ordinary, cached way.
//   A  space budget over the pods; candidate / qualified counts and key ranges over the registry for BOTH forms the qualify rule
//      can take (freeCount > 0: every candidate; else only lastUsed > cutoff, :6683-6685) — freeCount is known after the barrier
//                                                                                               | barrier 1
//   B  scalars (:6621-6664) and the fold of the partials in every workgroup; histogram of the chosen map, each qualified model
//      keeping the slot its increment returned                                                  | barrier 2; the last workgroup to
//                                                                                                 arrive scans the buckets first
//   C  (lastUsed, model) pairs to offset + slot                                                  | barrier 3
//   D  per ENTRY (work follows the qualified count, not the bucket count): a bucket that lies inside one wavefront's 64 entries is
//      ranked through lane shuffles; the one bucket a wavefront's range can cut is ranked by the wavefront where it starts, from
//      memory (plan_rank_bucket)                                                                 | barrier 4; the last workgroup
//                                                                                                 scans the rank offsets, :6709-6734
//   E  emit
// 64 workgroups of 1024 lanes rather than 400 of 256: what a barrier costs grows with the workgroups that arrive at it (their
// increments of one counter are applied one after the other, ~25 ns each)
constexpr int kPlanFusedBlock = 1024, kPlanFusedGrid = 64, kPlanPartWords = 8;

template <typename T>
__device__ __forceinline__ T dld(const T *p)
{
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void dst(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Barrier number `gen` (1-based) of the launch: one counter, one flag (in cache lines of their own: polls in the line the arrivals
// are counted in would queue in front of them).  1.6-2.4 us with 64 workgroups.  Measured and dropped: a flag per workgroup, the first
// wavefront of each reading all of them (no counter to queue at) — 2.9-3.2 us.
// The launch ASSUMES its workgroups are all on the chip (the host sizes the grid by the occupancy query and never asks for more than
// one per compute unit) — which another tenant's resident kernels could still deny it.  A workgroup that has waited kPlanBarrierTicks
// (20 ms; an ordinary wait is microseconds) gives up instead of hanging the queue: it raises PlanScalars::overflow to 2, every
// workgroup that sees that leaves, and the host runs the plan as separate launches (the path it takes for an overflowing bucket).
// false: leave the kernel.
constexpr long long kPlanBarrierTicks = 2'000'000;  // of the 100 MHz wall clock
__device__ __forceinline__ bool plan_grid_barrier(PlanScalars *ps, unsigned int gen)
{
    __shared__ int s_abort;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's sc1 stores and atomics are done at the memory side
    __syncthreads();
    if (threadIdx.x == 0) {
        int ab = 0;
        if (__hip_atomic_fetch_add(&ps->bar_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen * gridDim.x - 1)
            __hip_atomic_store(&ps->bar_flag[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            const long long t0 = wall_clock64();
            unsigned int spins = 0;
            while (__hip_atomic_load(&ps->bar_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 255u) == 0 && wall_clock64() - t0 > kPlanBarrierTicks) {
                    ab = 1;
                    break;
                }
            }
        }
        if (ab) __hip_atomic_store(&ps->overflow, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(&ps->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) ab = 1;  // somebody gave up before
        s_abort = ab;
    }
    __syncthreads();
    return s_abort == 0;
}
#ifdef MMP_PLAN_CLOCK
#define PLAN_CLOCK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) dst(&ps->t_phase[k], (long long)wall_clock64()); } while (0)
#define PLAN_CLOCK_ANY(k) do { if (threadIdx.x == 0) dst(&ps->t_phase[k], (long long)wall_clock64()); } while (0)  // (the last workgroup)
#else
#define PLAN_CLOCK(k) do { } while (0)
#define PLAN_CLOCK_ANY(k) do { } while (0)
#endif

// Wavefront totals through DPP moves (row-local permutes, then the two row broadcasts of gfx9): seven VALU steps per value and
// no trip through the LDS crossbar.  The eight totals of a PlanPartial2 by ds_bpermute shuffles were 4 us of phase A, and again of B.
// Every lane of the wavefront must be active.  The result is the same in every lane.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_mov32(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint64_t dpp_mov64(uint64_t old, uint64_t v)
{
    return (uint64_t)dpp_mov32<CTRL, ROW_MASK>((uint32_t)old, (uint32_t)v) |
           ((uint64_t)dpp_mov32<CTRL, ROW_MASK>((uint32_t)(old >> 32), (uint32_t)(v >> 32)) << 32);
}
template <typename Op>
__device__ __forceinline__ uint64_t wave_reduce_dpp64(uint64_t v, uint64_t ident, Op op)
{
    v = op(v, dpp_mov64<0xB1>(v, v));           // quad_perm [1,0,3,2]
    v = op(v, dpp_mov64<0x4E>(v, v));           // quad_perm [2,3,0,1]
    v = op(v, dpp_mov64<0x141>(v, v));          // row_half_mirror
    v = op(v, dpp_mov64<0x140>(v, v));          // row_mirror: every lane of a row of 16 has the row's total
    v = op(v, dpp_mov64<0x142, 0xA>(ident, v));  // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_mov64<0x143, 0xC>(ident, v));  // row_bcast:31 into rows 2 and 3: lane 63 has the total
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32);
}
__device__ __forceinline__ int32_t wave_sum_dpp_i32(int32_t x)
{
    uint32_t v = (uint32_t)x;
    v += dpp_mov32<0xB1>(v, v);
    v += dpp_mov32<0x4E>(v, v);
    v += dpp_mov32<0x141>(v, v);
    v += dpp_mov32<0x140>(v, v);
    v += dpp_mov32<0x142, 0xA>(0u, v);
    v += dpp_mov32<0x143, 0xC>(0u, v);
    return __builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int64_t wave_min_dpp_i64(int64_t x)
{
    return (int64_t)wave_reduce_dpp64((uint64_t)x, (uint64_t)INT64_MAX, [](uint64_t a, uint64_t b) { return (int64_t)a < (int64_t)b ? a : b; });
}
__device__ __forceinline__ int64_t wave_max_dpp_i64(int64_t x)
{
    return (int64_t)wave_reduce_dpp64((uint64_t)x, (uint64_t)INT64_MIN, [](uint64_t a, uint64_t b) { return (int64_t)a > (int64_t)b ? a : b; });
}
__device__ __forceinline__ double wave_sum_f64(double x)
{
    return __longlong_as_double((long long)wave_reduce_dpp64((uint64_t)__double_as_longlong(x), 0ull, [](uint64_t a, uint64_t b) {
        return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    }));
}

struct PlanPartial2 {
    int32_t n_cand, nq_a, nq_b;
    int64_t kmin_a, kmin_b, kmax;
    double sum_a, sum_b;
};
__device__ __forceinline__ void plan_partial_add(PlanPartial2 &t, const PlanPartial2 &v)
{
    t.n_cand += v.n_cand;
    t.nq_a += v.nq_a;
    t.nq_b += v.nq_b;
    t.kmin_a = v.kmin_a < t.kmin_a ? v.kmin_a : t.kmin_a;
    t.kmin_b = v.kmin_b < t.kmin_b ? v.kmin_b : t.kmin_b;
    t.kmax = v.kmax > t.kmax ? v.kmax : t.kmax;
    t.sum_a += v.sum_a;
    t.sum_b += v.sum_b;
}
// the workgroup's total in thread 0: wavefront totals by shuffles, then one LDS atomic per field and wavefront (a loop of thread 0
// over sixteen rows of eight fields was 3 us of phase A).  The order of the double adds is not fixed — the sums only choose the bucket
// map, and every workgroup folds the same published partials in the same order.
struct PlanPartialLds {
    int32_t n_cand, nq_a, nq_b, pad;
    long long kmin_a, kmin_b, kmax;
    double sum_a, sum_b;
};
__device__ __forceinline__ void plan_partial_reduce(PlanPartial2 &t, PlanPartialLds *s_acc)
{
    if (threadIdx.x == 0) *s_acc = PlanPartialLds{0, 0, 0, 0, INT64_MAX, INT64_MAX, INT64_MIN, 0.0, 0.0};
    t.n_cand = wave_sum_dpp_i32(t.n_cand);
    t.nq_a = wave_sum_dpp_i32(t.nq_a);
    t.nq_b = wave_sum_dpp_i32(t.nq_b);
    t.kmin_a = wave_min_dpp_i64(t.kmin_a);
    t.kmin_b = wave_min_dpp_i64(t.kmin_b);
    t.kmax = wave_max_dpp_i64(t.kmax);
    t.sum_a = wave_sum_f64(t.sum_a);
    t.sum_b = wave_sum_f64(t.sum_b);
    __syncthreads();
    if (lane_id() == 0) {
        atomicAdd(&s_acc->n_cand, t.n_cand);
        atomicAdd(&s_acc->nq_a, t.nq_a);
        atomicAdd(&s_acc->nq_b, t.nq_b);
        atomicMin(&s_acc->kmin_a, (long long)t.kmin_a);
        atomicMin(&s_acc->kmin_b, (long long)t.kmin_b);
        atomicMax(&s_acc->kmax, (long long)t.kmax);
        atomicAdd(&s_acc->sum_a, t.sum_a);
        atomicAdd(&s_acc->sum_b, t.sum_b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t.n_cand = s_acc->n_cand;
        t.nq_a = s_acc->nq_a;
        t.nq_b = s_acc->nq_b;
        t.kmin_a = s_acc->kmin_a;
        t.kmin_b = s_acc->kmin_b;
        t.kmax = s_acc->kmax;
        t.sum_a = s_acc->sum_a;
        t.sum_b = s_acc->sum_b;
    }
}

// A bucket too long for the workgroup's LDS window (rare: hundreds of models on one bucket width, and the window cut through it):
// ranked by ONE wavefront with every pair fetched from memory on every pass; rank[] carries the run-start flags between the passes.
// Returns (in every lane) the run starts of the bucket; *ge_out those at or above the cutoff.  rank[e] = base + local rank, -1 = duplicate.
__device__ __forceinline__ int32_t plan_rank_bucket_mem(int lo, int cnt, const int64_t *keys, const int32_t *vals, int64_t cutoff, int32_t base,
                                                        int32_t *rank, int32_t *ge_out)
{
    const int lane = lane_id();
    int32_t starts = 0, ge = 0;
    for (int e = lane; e < cnt; e += 64) {
        const int64_t key = dld(&keys[lo + e]);
        const int32_t val = dld(&vals[lo + e]);
        bool first = true;
        for (int j = 0; j < cnt; j++) first &= !(dld(&keys[lo + j]) == key && dld(&vals[lo + j]) < val);
        dst(&rank[lo + e], first ? 0 : -1);
        starts += first ? 1 : 0;
        ge += (first && key >= cutoff) ? 1 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    // second pass into registers first: a rank written while other lanes still read the flags could be taken for a duplicate's -1
    // only if it were negative — base + r is not
    for (int e = lane; e < cnt; e += 64) {
        if (dld(&rank[lo + e]) < 0) continue;
        const int64_t key = dld(&keys[lo + e]);
        int32_t r = 0;
        for (int j = 0; j < cnt; j++) r += (dld(&keys[lo + j]) > key && dld(&rank[lo + j]) >= 0) ? 1 : 0;
        dst(&rank[lo + e], base + r);
    }
    *ge_out = wave_sum_i32(ge);
    return wave_sum_i32(starts);
}

constexpr int kPlanChunk = kPlanFusedBlock, kPlanAhead = kPlanLdsBucket, kPlanWindow = kPlanChunk + kPlanAhead;
constexpr size_t kPlanFusedLds = (size_t)(kPlanBuckets + 1) * sizeof(int32_t);  // dynamic: the offset table
constexpr uint64_t kPlanChunkValid = 1ull << 63;

// Exclusive scan of the finished histogram into the workgroup's LDS table (s_off[kPlanBuckets] = the total); returns the fullest
// bucket.  Lane t takes the int4 at k * 1024 + t of each quarter k of the table: neighbouring lanes read and write neighbouring
// 16 bytes (a lane scanning 16 buckets of its own reads and writes at a stride of 64 bytes: 3.9 us against this form's).
__device__ __forceinline__ int32_t plan_scan_to_lds(const int32_t *hist, int32_t *s_off)
{
    static_assert(kPlanBuckets == 4 * 4 * kPlanFusedBlock, "four int4 per lane");
    __shared__ int32_t s_wtot[4][kPlanFusedBlock / 64], s_wmax[kPlanFusedBlock / 64];
    const int tid = threadIdx.x, wv = tid >> 6;
    int4 x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = reinterpret_cast<const int4 *>(hist)[k * kPlanFusedBlock + tid];
    int32_t sum[4], incl[4], mx = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sum[k] = x[k].x + x[k].y + x[k].z + x[k].w;
        const int32_t a = x[k].x > x[k].y ? x[k].x : x[k].y, b = x[k].z > x[k].w ? x[k].z : x[k].w;
        mx = a > mx ? a : mx;
        mx = b > mx ? b : mx;
        incl[k] = wave_incl_scan_i32(sum[k]);
    }
    mx = (int32_t)wave_max_dpp_i64(mx);
    if (lane_id() == 63) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_wtot[k][wv] = incl[k];
        s_wmax[wv] = mx;
    }
    __syncthreads();
    int32_t base = 0, fullest = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int32_t before = 0, tot = 0;
        for (int w = 0; w < kPlanFusedBlock / 64; w++) {
            const int32_t v = s_wtot[k][w];
            before += w < wv ? v : 0;
            tot += v;
        }
        int4 o;
        o.x = base + before + incl[k] - sum[k];
        o.y = o.x + x[k].x;
        o.z = o.y + x[k].y;
        o.w = o.z + x[k].z;
        reinterpret_cast<int4 *>(s_off)[k * kPlanFusedBlock + tid] = o;
        base += tot;
    }
    for (int w = 0; w < kPlanFusedBlock / 64; w++) fullest = s_wmax[w] > fullest ? s_wmax[w] : fullest;
    if (tid == 0) s_off[kPlanBuckets] = base;
    return fullest;
}

__global__ __launch_bounds__(kPlanFusedBlock) void proactive_plan_fused_kernel(
    const mmp_pod_row *__restrict__ pods, int32_t P, const mmp_model_row *__restrict__ models, int32_t M, PlanSubset U,
    int32_t default_units, int64_t now, PlanScalars *ps, uint64_t *part, int32_t *hist, int32_t *slot, int64_t *keys,
    int32_t *vals, int32_t *rank, uint64_t *chunk_tot, int32_t max_out, int32_t *__restrict__ out_model, int64_t *__restrict__ out_lu)
{
    extern __shared__ int32_t s_off[];  // kPlanBuckets + 1 bucket offsets: every workgroup scans the histogram itself (dynamic LDS)
    __shared__ PlanScalars s_ps;  // this workgroup's copy of the scalars (every workgroup computes the same ones)
    __shared__ PlanPartialLds s_acc;
    __shared__ int64_t s_key[kPlanWindow];
    __shared__ int32_t s_val[kPlanWindow], s_pre[kPlanWindow + 1];
    __shared__ uint8_t s_first[kPlanWindow];
    __shared__ int32_t s_wcnt[kPlanWindow / 64], s_misc[8];
    const int tid = threadIdx.x, G = gridDim.x, nthreads = G * kPlanFusedBlock, gtid = blockIdx.x * kPlanFusedBlock + tid;
    const int nb = (M + kPlanFusedBlock - 1) / kPlanFusedBlock;
    const StatsAcc *st = U.stats;

    // ---- A ----
    PLAN_CLOCK(0);
    // the lane's first two registry rows: in flight while the rest of the phase's inputs arrive
    const int ia = blockIdx.x * kPlanFusedBlock + tid, ib = (blockIdx.x + G) * kPlanFusedBlock + tid;
    const mmp_model_row ra = models[ia < M ? ia : 0], rb = models[ib < M ? ib : 0];
    // ... and its first instance row (the workgroups share the instance table evenly)
    const int per_wg = (P + G - 1) / G, p_end = (blockIdx.x + 1) * per_wg < P ? (blockIdx.x + 1) * per_wg : P;
    const int pa = blockIdx.x * per_wg + tid;
    const mmp_pod_row rp = pods[pa < P ? pa : 0];
    const int32_t pts_a = U.pts >= 0 ? U.pod_pts[pa < P ? pa : 0] : 0;
    for (int k = gtid; k < kPlanBuckets; k += nthreads) dst(&hist[k], 0);
    for (int k = gtid; k < nb; k += nthreads) dst(&chunk_tot[k], (uint64_t)0);  // (chunks of qualified entries: at most nb of them)
    // what of the scalars does not need the space budget: the candidate rule's inputs and the cutoff (one lane, while the others
    // add up the instances)
    if (tid == kPlanFusedBlock - 1) {
        s_ps.cand_enabled = (int64_t)U.global->total_capacity > 0 ? 1 : 0;
        s_ps.cand_glru = (int64_t)U.global->total_free > 0 ? 0 : U.global->global_lru;
        const int64_t glru = st->global_lru;
        int64_t cutoff = 0;
        if (glru != INT64_MAX) {
            const int64_t third = age_of(glru, now) / 3;
            cutoff = (int64_t)((uint64_t)glru + (uint64_t)(third > 1200000 ? third : 1200000));
        }
        s_ps.cutoff = cutoff;
    }
    {
        int64_t sum = 0;
        if ((int64_t)st->total_capacity > 0 && (int64_t)st->total_free > 0) {
            const int32_t se = size_estimate_of(st, default_units);
            if (se != 0) {
                auto add = [&](const mmp_pod_row &r, int32_t pts) {  // :6633-6649
                    if (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) return;
                    if (U.pts >= 0 && pts != U.pts) return;
                    const int32_t max_loads = (int32_t)((uint32_t)r.loading_threads * 50u - (uint32_t)r.loading_in_progress);
                    if (max_loads <= 0) return;
                    const int64_t avail = jsub64(remaining_of(r.capacity, r.used), r.capacity / 8);
                    if (avail > 0) {
                        const int64_t by_loads = (int64_t)(int32_t)((uint32_t)max_loads * (uint32_t)se);
                        sum = (int64_t)((uint64_t)sum + (uint64_t)(avail < by_loads ? avail : by_loads));
                    }
                };
                if (pa < p_end) add(rp, pts_a);
                for (int p = pa + kPlanFusedBlock; p < p_end; p += kPlanFusedBlock) add(pods[p], U.pts >= 0 ? U.pod_pts[p] : 0);
            }
        }
        sum = wave_sum_i64(sum);
        if (lane_id() == 0 && sum != 0) atomicAdd(&ps->space_acc, (unsigned long long)sum);
    }
    PLAN_CLOCK(13);
    __syncthreads();
    {
        PlanPartial2 t{0, 0, 0, INT64_MAX, INT64_MAX, INT64_MIN, 0.0, 0.0};
        const int64_t cutoff = s_ps.cutoff;
        auto take = [&](const mmp_model_row &m, int i) {
            if (!proactive_candidate(m, &s_ps)) return;
            t.n_cand++;
            if (plan_excluded(U, i, m.type)) return;
            const int64_t lu = m.last_used;
            t.nq_a++;
            t.kmin_a = lu < t.kmin_a ? lu : t.kmin_a;
            t.kmax = lu > t.kmax ? lu : t.kmax;
            t.sum_a += (double)lu;
            if (lu > cutoff) {
                t.nq_b++;
                t.kmin_b = lu < t.kmin_b ? lu : t.kmin_b;
                t.sum_b += (double)lu;
            }
        };
        if (ia < M) take(ra, ia);
        if (ib < M) take(rb, ib);
        for (int vb = blockIdx.x + 2 * G; vb < nb; vb += 2 * G) {  // two rows per lane and turn, both loads in flight
            const int i0 = vb * kPlanFusedBlock + tid, i1 = (vb + G) * kPlanFusedBlock + tid;
            const bool h0 = i0 < M, h1 = i1 < M;
            const mmp_model_row m0 = models[h0 ? i0 : 0], m1 = models[h1 ? i1 : 0];
            if (h0) take(m0, i0);
            if (h1) take(m1, i1);
        }
        PLAN_CLOCK(14);
        plan_partial_reduce(t, &s_acc);
        PLAN_CLOCK(15);
        if (tid == 0) {
            uint64_t *o = part + (size_t)blockIdx.x * kPlanPartWords;
            dst(&o[0], (uint64_t)(uint32_t)t.n_cand | ((uint64_t)(uint32_t)t.nq_a << 32));
            dst(&o[1], (uint64_t)(uint32_t)t.nq_b);
            dst(&o[2], (uint64_t)t.kmin_a);
            dst(&o[3], (uint64_t)t.kmin_b);
            dst(&o[4], (uint64_t)t.kmax);
            dst(&o[5], (uint64_t)__double_as_longlong(t.sum_a));
            dst(&o[6], (uint64_t)__double_as_longlong(t.sum_b));
        }
    }
    PLAN_CLOCK(1);
    if (!plan_grid_barrier(ps, 1)) return;
    PLAN_CLOCK(2);

    // ---- B ----
    if (tid < 64) {  // the fold of the partials (one per workgroup, at most 64: one wavefront) and the scalars, :6621-6664
        static_assert(kPlanFusedGrid <= 64, "one lane per workgroup's partial");
        PlanPartial2 t{0, 0, 0, INT64_MAX, INT64_MAX, INT64_MIN, 0.0, 0.0};
        const unsigned long long acc = __hip_atomic_load(&ps->space_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < G) {
            const uint64_t *o = part + (size_t)tid * kPlanPartWords;
            const uint64_t w0 = dld(&o[0]);
            t.n_cand = (int32_t)(uint32_t)w0;
            t.nq_a = (int32_t)(uint32_t)(w0 >> 32);
            t.nq_b = (int32_t)(uint32_t)dld(&o[1]);
            t.kmin_a = (int64_t)dld(&o[2]);
            t.kmin_b = (int64_t)dld(&o[3]);
            t.kmax = (int64_t)dld(&o[4]);
            t.sum_a = __longlong_as_double((long long)dld(&o[5]));
            t.sum_b = __longlong_as_double((long long)dld(&o[6]));
        }
        t.n_cand = wave_sum_dpp_i32(t.n_cand);
        t.nq_a = wave_sum_dpp_i32(t.nq_a);
        t.nq_b = wave_sum_dpp_i32(t.nq_b);
        t.kmin_a = wave_min_dpp_i64(t.kmin_a);
        t.kmin_b = wave_min_dpp_i64(t.kmin_b);
        t.kmax = wave_max_dpp_i64(t.kmax);
        t.sum_a = wave_sum_f64(t.sum_a);
        t.sum_b = wave_sum_f64(t.sum_b);
        if (tid == 0) {
            proactive_scalars(U, default_units, now, &s_ps, acc);
            const bool all = s_ps.free_count > 0;  // :6683-6685: with free space every candidate qualifies
            const int32_t nq = s_ps.total_count > 0 ? (all ? t.nq_a : t.nq_b) : 0;
            const int64_t kmin = all ? t.kmin_a : t.kmin_b;
            const double ksum = all ? t.sum_a : t.sum_b;
            int map = 0;
            if (nq > 0) {
                const double range = (double)((uint64_t)t.kmax - (uint64_t)kmin);
                const double mean_age = (double)t.kmax - ksum / (double)nq;
                map = mean_age * 8.0 < range ? plan_mantissa_bits((uint64_t)t.kmax - (uint64_t)kmin) : 0;
            }
            s_ps.n_candidates = t.n_cand;
            s_ps.n_qualified = nq;
            s_ps.kmin = kmin;
            s_ps.kmax = t.kmax;
            s_ps.bucket_map = map;
            if (blockIdx.x == 0) {  // (device-scope stores: the row also holds the barrier's counters)
                dst(&ps->size_estimate, s_ps.size_estimate);
                dst(&ps->free_count, s_ps.free_count);
                dst(&ps->total_count, s_ps.total_count);
                dst(&ps->error, s_ps.error);
                dst(&ps->space_to_fill, s_ps.space_to_fill);
                dst(&ps->cutoff, s_ps.cutoff);
                dst(&ps->cand_enabled, s_ps.cand_enabled);
                dst(&ps->cand_glru, s_ps.cand_glru);
                dst(&ps->n_candidates, t.n_cand);
                dst(&ps->n_qualified, nq);
                dst(&ps->kmin, (long long)kmin);
                dst(&ps->kmax, (long long)t.kmax);
                dst(&ps->bucket_map, map);
            }
        }
    }
    __syncthreads();
    const int32_t nq = s_ps.n_qualified;
    if (nq <= 0) return;  // (every workgroup: the same scalars)
    for (int vb = blockIdx.x; vb < nb; vb += 2 * G) {
        const int i0 = vb * kPlanFusedBlock + tid, i1 = (vb + G) * kPlanFusedBlock + tid;
        const bool h0 = i0 < M, h1 = i1 < M, again = vb != (int)blockIdx.x;  // (the first two rows are still in registers)
        const mmp_model_row m0 = again ? models[h0 ? i0 : 0] : ra, m1 = again ? models[h1 ? i1 : 0] : rb;
        const bool q0 = h0 && proactive_qualifies(m0, i0, U, &s_ps), q1 = h1 && proactive_qualifies(m1, i1, U, &s_ps);
        // hist[b] += 1; the lane's slot in its bucket = what the add returned (one add per lane, all in flight together: an add per
        // DISTINCT bucket of the wavefront, each waiting for the one before, cost 24 us here)
        const int32_t my0 = q0 ? atomicAdd(&hist[plan_bucket(m0.last_used, &s_ps)], 1) : -1;
        const int32_t my1 = q1 ? atomicAdd(&hist[plan_bucket(m1.last_used, &s_ps)], 1) : -1;
        if (h0) slot[i0] = my0;  // read back by this lane in phase C
        if (h1) slot[i1] = my1;
    }
    PLAN_CLOCK(3);
    if (!plan_grid_barrier(ps, 2)) return;
    // The bucket offsets: EVERY workgroup scans the finished histogram into its own LDS (64 KB read from memory per workgroup,
    // 2 us) — one workgroup scanning for all, a write-back, a flag and 64 x 3 device-scope offset loads per lane later cost more.
    // The adds went to the memory side; one invalidation so that the ordinary int4 loads here do not find an older line in this L2.
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    {
        PLAN_CLOCK(10);
        const int32_t fullest = plan_scan_to_lds(hist, s_off);
        __syncthreads();
        PLAN_CLOCK(11);
        if (fullest > kPlanBucketMax) {  // (every workgroup finds the same) the host takes the sorted path
            if (blockIdx.x == 0 && tid == 0) dst(&ps->overflow, 1);
            return;
        }
    }

    // ---- C ----
    PLAN_CLOCK(4);
    for (int vb = blockIdx.x; vb < nb; vb += 2 * G) {
        const int i0 = vb * kPlanFusedBlock + tid, i1 = (vb + G) * kPlanFusedBlock + tid;
        const bool h0 = i0 < M, h1 = i1 < M;
        const int32_t my0 = h0 ? slot[i0] : -1, my1 = h1 ? slot[i1] : -1;
        const int64_t lu0 = models[h0 ? i0 : 0].last_used, lu1 = models[h1 ? i1 : 0].last_used;
        const int32_t o0 = my0 >= 0 ? s_off[plan_bucket(lu0, &s_ps)] : 0, o1 = my1 >= 0 ? s_off[plan_bucket(lu1, &s_ps)] : 0;
        if (my0 >= 0) {
            dst(&keys[o0 + my0], lu0);
            dst(&vals[o0 + my0], (int32_t)i0);
        }
        if (my1 >= 0) {
            dst(&keys[o1 + my1], lu1);
            dst(&vals[o1 + my1], (int32_t)i1);
        }
    }
    PLAN_CLOCK(5);
    if (!plan_grid_barrier(ps, 3)) return;
    PLAN_CLOCK(6);

    // ---- D ----
    // A workgroup takes kPlanChunk entries at a time plus the kPlanAhead behind them into LDS and ranks every bucket that STARTS in
    // the chunk (a bucket is whole in the window unless it is longer than the lookahead and cut by the window's end: that one
    // goes through memory).  Per entry: a run start = no equal key with a lower model index (the TreeSet's first one seen); its
    // rank inside the chunk = run starts of the chunk's earlier buckets + run starts of its bucket with a larger key.  The chunk
    // publishes its run starts; an entry's final place is then the run starts of all earlier chunks + that rank (phase E).
    const int64_t cutoff = s_ps.cutoff;
    const int n_chunks = (nq + kPlanChunk - 1) / kPlanChunk;
    for (int c = blockIdx.x; c < n_chunks; c += G) {
        const int E0 = c * kPlanChunk, n_here = nq - E0 < kPlanWindow ? nq - E0 : kPlanWindow;
        __syncthreads();  // (the window of this workgroup's last chunk is done with)
        for (int j = tid; j < n_here; j += kPlanFusedBlock) {
            s_key[j] = dld(&keys[E0 + j]);
            s_val[j] = dld(&vals[E0 + j]);
        }
        // the lane's entries: j0 = tid (the chunk), j1 = kPlanChunk + tid (the lookahead; lanes below kPlanAhead)
        int lo[2], cn[2];
        bool mine[2];  // the entry's bucket starts in this chunk and is whole in the window
        int cut_lo = -1, cut_cnt = 0;  // the bucket the window's end cuts (the lane that holds its first entry knows)
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = h * kPlanChunk + tid;
            lo[h] = cn[h] = 0;
            mine[h] = false;
            if (j < n_here && (h == 0 || tid < kPlanAhead)) {
                const int b = plan_bucket(s_key[j], &s_ps);
                const int l = s_off[b] - E0, hi = s_off[b + 1] - E0;
                lo[h] = l;
                cn[h] = hi - l;
                const bool starts_here = l >= 0 && l < kPlanChunk;
                mine[h] = starts_here && hi <= kPlanWindow;
                if (starts_here && hi > kPlanWindow && j == l) {
                    cut_lo = l;
                    cut_cnt = hi - l;
                }
            }
        }
        bool first[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = h * kPlanChunk + tid;
            first[h] = mine[h];
            if (mine[h]) {
                const int64_t key = s_key[j];
                const int32_t val = s_val[j];
                for (int t = 0; t < cn[h]; t++) first[h] &= !(s_key[lo[h] + t] == key && s_val[lo[h] + t] < val);
            }
            if (j < kPlanWindow) s_first[j] = first[h] ? 1 : 0;
        }
        __syncthreads();
        // run starts in front of every window position (exclusive), s_pre[kPlanWindow] = all of them
        {
            const int wv = tid >> 6, lane = lane_id();
            const uint64_t m0 = __ballot(first[0]), m1 = __ballot(first[1]);
            if (lane == 0) {
                s_wcnt[wv] = (int32_t)__popcll(m0);
                if (wv < kPlanAhead / 64) s_wcnt[kPlanChunk / 64 + wv] = (int32_t)__popcll(m1);
            }
            __syncthreads();
            int32_t before0 = 0, before1 = 0;
            for (int w = 0; w < kPlanWindow / 64; w++) {
                const int32_t n = s_wcnt[w];
                before0 += w < wv ? n : 0;
                before1 += w < kPlanChunk / 64 + wv ? n : 0;
            }
            s_pre[tid] = before0 + (int32_t)__popcll(m0 & ((1ull << lane) - 1));
            if (tid < kPlanAhead) s_pre[kPlanChunk + tid] = before1 + (int32_t)__popcll(m1 & ((1ull << lane) - 1));
            if (tid == 0) {
                int32_t all = 0;
                for (int w = 0; w < kPlanWindow / 64; w++) all += s_wcnt[w];
                s_pre[kPlanWindow] = all;
            }
        }
        __syncthreads();
        int32_t ge = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = h * kPlanChunk + tid;
            if (!mine[h]) continue;
            int32_t r = -1;
            if (first[h]) {
                const int64_t key = s_key[j];
                r = s_pre[lo[h]];
                for (int t = 0; t < cn[h]; t++) r += (s_first[lo[h] + t] && s_key[lo[h] + t] > key) ? 1 : 0;
                ge += key >= cutoff;
            }
            dst(&rank[E0 + j], r);
        }
        // the cut bucket: one wavefront, from memory; its entries rank behind every run start of the window
        {
            const uint64_t has = __ballot(cut_lo >= 0);
            if (tid < 64) s_misc[0] = -1;
            __syncthreads();
            if (cut_lo >= 0) {
                s_misc[0] = cut_lo;
                s_misc[1] = cut_cnt;
            }
            (void)has;
            __syncthreads();
        }
        int32_t cut_starts = 0, cut_ge = 0;
        if (s_misc[0] >= 0 && tid < 64)
            cut_starts = plan_rank_bucket_mem(E0 + s_misc[0], s_misc[1], keys, vals, cutoff, s_pre[kPlanWindow], rank, &cut_ge);
        // the chunk's totals: run starts, and those of them at or above the cutoff — published when every lane's ranks are written
        ge = wave_sum_i32(ge);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (s_wcnt is read above)
        if (lane_id() == 0) s_wcnt[tid >> 6] = ge;
        __syncthreads();
        if (tid == 0) {
            int32_t g = cut_ge;
            for (int w = 0; w < kPlanFusedBlock / 64; w++) g += s_wcnt[w];
            dst(&chunk_tot[c], kPlanChunkValid | ((uint64_t)(uint32_t)g << 31) | (uint64_t)(uint32_t)(s_pre[kPlanWindow] + cut_starts));
        }
    }
    PLAN_CLOCK(7);

    // ---- E ----  (no barrier: a chunk waits for the totals of the chunks in front of it, each in a word of its own)
    PLAN_CLOCK(8);
    for (int c = blockIdx.x; c < n_chunks; c += G) {
        const int E0 = c * kPlanChunk, e = E0 + tid;
        // this lane's entry first (the loads are in flight while the totals arrive); its rank — which the chunk in front of this one
        // may have written — after they have
        int32_t r = -1, val = 0, lo = 0;
        int64_t key = 0;
        if (e < nq) {
            key = dld(&keys[e]);
            val = dld(&vals[e]);
            lo = s_off[plan_bucket(key, &s_ps)];
        }
        // run starts (and those at or above the cutoff) of the chunks 0 .. c-1, and of chunk c-1 alone
        int32_t sum = 0, sum_ge = 0, prev = 0;
        for (int k = tid; k < c; k += kPlanFusedBlock) {
            uint64_t w;
            for (unsigned int spins = 0; !((w = dld(&chunk_tot[k])) & kPlanChunkValid); spins++) {  // (its owner passed barrier 3 with this workgroup)
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 1023u) == 1023u && dld(&ps->overflow) == 2) break;  // ... unless it gave up there (plan_grid_barrier): nothing of this launch is used
            }
            sum += (int32_t)(w & 0x7fffffffu);
            sum_ge += (int32_t)((w >> 31) & 0x7fffffffu);
            if (k == c - 1) prev = (int32_t)(w & 0x7fffffffu);
        }
        sum = wave_sum_i32(sum);
        sum_ge = wave_sum_i32(sum_ge);
        prev = wave_sum_i32(prev);
        __syncthreads();
        if (lane_id() == 0) {
            s_wcnt[tid >> 6] = sum;
            s_pre[tid >> 6] = sum_ge;
            s_pre[64 + (tid >> 6)] = prev;
        }
        __syncthreads();
        if (e < nq) r = dld(&rank[e]);
        int32_t base = 0, base_ge = 0, last_one = 0;
        for (int w = 0; w < kPlanFusedBlock / 64; w++) {
            base += s_wcnt[w];
            base_ge += s_pre[w];
            last_one += s_pre[64 + w];
        }
        if (r >= 0) {
            // the entry's bucket starts in this chunk or in the one before (a bucket is no longer than a chunk)
            const int32_t d = (lo >= E0 ? base : base - last_one) + r;
            if (d < s_ps.total_count && d < max_out) {
                out_model[d] = val;
                out_lu[d] = key;
            }
        }
        if (c == n_chunks - 1 && tid == 0) {  // the last chunk knows every total: :6709-6734
            uint64_t w;
            for (unsigned int spins = 0; !((w = dld(&chunk_tot[c])) & kPlanChunkValid); spins++) {
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 1023u) == 1023u && dld(&ps->overflow) == 2) break;
            }
            const int32_t total = base + (int32_t)(w & 0x7fffffffu), n_ge = base_ge + (int32_t)((w >> 31) & 0x7fffffffu);
            dst(&ps->n_distinct, total);
            const int32_t n_sel = total < s_ps.total_count ? total : s_ps.total_count;
            const int32_t by_free = s_ps.free_count < n_sel ? (s_ps.free_count > 0 ? s_ps.free_count : 0) : n_sel;
            const int32_t by_cut = n_ge < n_sel ? n_ge : n_sel;
            dst(&ps->n_ge_cutoff, by_cut);
            dst(&ps->n_selected, by_free > by_cut ? by_free : by_cut);
            PLAN_CLOCK_ANY(12);
        }
    }
    PLAN_CLOCK(9);
}

}  // namespace mmp

namespace mmp {

// ---- a15 ---------------------------------------------------------------------------------------
// getExcludeSet(), MM.java:5835-5856: pods of clusterState (present rows) other than self whose
// published rpm exceeds max(4*threshold, ourRpm - 2*threshold)
// ---- limitModelConcurrency == true ("latency-based" scaling): MaxConcCacheEntry, MM.java:2641-2797 ------------------
// mcce.getRpmScaleThreshold(andReset), :2766-2796, on one mmp_conc_entry row.  Java long / int arithmetic (wrapping products,
// `>>>`, division truncating toward zero, (int) of a long = its low 32 bits).  o (may be null): the reset the call makes.
__device__ __forceinline__ int32_t rpm_scale_threshold(const mmp_conc_entry &m, bool and_reset, int32_t scale_up_rpm_threshold,
                                                       int64_t dyn_const, mmp_conc_out *o)
{
    constexpr uint64_t kMask = (1ull << MMP_CONC_COUNT_BITS) - 1;
    const uint64_t cur = (uint64_t)m.count_and_time_sum;
    int64_t time_sum;
    int32_t count = (int32_t)(cur & kMask);
    if (count >= 64) {
        time_sum = (int64_t)(cur >> MMP_CONC_COUNT_BITS);
        if (and_reset && o) {  // sumThenReset(): priorSum = timeSum, priorCount = count
            o->reset = 1;
            o->new_prior_sum = time_sum;
            o->new_prior_count = count;
        }
    } else {
        const int32_t pc = m.prior_count;
        if (pc <= 0 && count < 8) return scale_up_rpm_threshold;
        time_sum = (int64_t)((uint64_t)m.prior_sum + (count > 0 ? (cur >> MMP_CONC_COUNT_BITS) : 0ull));
        count = (int32_t)((uint32_t)count + (uint32_t)pc);
    }
    if (time_sum == 0) return INT32_MAX;
    const int64_t num = (int64_t)((uint64_t)(int64_t)m.max_conc * ((uint64_t)(int64_t)count * (uint64_t)dyn_const));
    if (num == INT64_MIN && time_sum == -1) return 0;  // Java: Long.MIN_VALUE / -1 == Long.MIN_VALUE, whose low 32 bits are 0
    return (int32_t)(uint32_t)(uint64_t)(num / time_sum);
}

// the scalars of a latency-based run: getExcludeSet's threshold from the task's averageModelParallelism (:5836) -> the most
// an instance may serve before it is excluded (:5841); one thread
__global__ void conc_exclude_rpms_kernel(double average_model_parallelism, int32_t our_rpm, int32_t *__restrict__ max_rpm,
                                         mmp_conc_result *__restrict__ res)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double d = 900.0 * average_model_parallelism;
    // (int) of a double, JLS 5.1.3: NaN -> 0, saturating
    const int32_t rpms = d != d ? 0 : d >= 2147483647.0 ? INT32_MAX : d <= -2147483648.0 ? INT32_MIN : (int32_t)d;
    const int32_t a = (int32_t)((uint32_t)rpms * 4u), b = (int32_t)((uint32_t)our_rpm - 2u * (uint32_t)rpms);
    *max_rpm = a > b ? a : b;
    res->exclude_set_rpms = rpms;
    res->model_parallelism_sum = 0;
    res->average_model_parallelism = average_model_parallelism;
}
// averageModelParallelism = Math.max(1.0, ((double) modelParallelismSum) / usedSinceLastRun.size()), :5815-5818; one thread
__global__ void conc_average_kernel(int32_t n_entries, mmp_conc_result *__restrict__ res)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double q = (double)res->model_parallelism_sum / (double)n_entries;
    res->average_model_parallelism = 1.0 >= q ? 1.0 : q;  // Math.max(1.0, q) (q is never NaN: n_entries > 0)
}

// max_rpm_dev (latency-based runs): the threshold conc_exclude_rpms_kernel computed, else max_rpm
__global__ void overloaded_pods_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t self_pod,
                                       int32_t max_rpm, const int32_t *__restrict__ max_rpm_dev, uint8_t *__restrict__ overloaded,
                                       int32_t *__restrict__ count)
{
    if (max_rpm_dev) max_rpm = *max_rpm_dev;
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
    // limitModelConcurrency == true (null otherwise): the entries' MaxConcCacheEntry rows, what getRpmScaleThreshold(true) did to
    // them, modelParallelismSum
    const mmp_conc_entry *conc;
    mmp_conc_out *conc_outs;
    mmp_conc_result *conc_res;
    int64_t dyn_const;
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
    mmp_conc_out co{};
    if (A.conc) {  // what an untouched row keeps
        co.new_prior_sum = A.conc[e].prior_sum;
        co.new_prior_count = A.conc[e].prior_count;
    }
    if (A.has_tc) {                              // :5693-5700: a type confined to one instance is skipped outright
        suitable = st->instance_count;
        if (suitable < 2) {
            o.rpm = 0;
            A.outs[e] = o;
            if (A.conc) A.conc_outs[e] = co;
            return;
        }
    }
    int32_t scale_up = p.scale_up_rpm_threshold;
    if (A.conc) {  // latencyBased, :5702-5707
        const mmp_conc_entry m = A.conc[e];
        scale_up = rpm_scale_threshold(m, true, p.scale_up_rpm_threshold, A.dyn_const, &co);
        co.threshold = scale_up;
        A.conc_outs[e] = co;
        atomicAdd(&A.conc_res->model_parallelism_sum, m.max_conc);  // (int addition wraps: any order gives the Java's sum)
    }
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
    const mmp_conc_entry *conc;  // MaxConcCacheEntry rows (mcce != null, :6294), null otherwise
    int64_t dyn_const;
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
            int64_t threshold = p.scale_up_rpm_threshold;  // :6295
            if (A.conc) threshold = rpm_scale_threshold(A.conc[e], false, p.scale_up_rpm_threshold, A.dyn_const, nullptr);
            if (rpm > (threshold * 2) / 3) break;
            if (A.conc && A.conc[e].queued_requests > 1) break;  // :6303
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
