// rank_sample.hpp — ranking a table from scratch without a comparison SORT (round 4).
// The literal PLACEMENT_ORDER comparator (snapshot.hpp: placement_less, MM.java:4646-4703) is a strict total order on the table
// whenever the host's O(P) check passes (snapshot.hpp "ranking by sorting").  rocprim::merge_sort with it takes 120 us for 10k rows
// (a block sort of 64-byte keys + five merge passes, each a launch); the all-pairs kernel P^2 comparator calls (172 us).  A rank
// is a COUNT, though, and counting splits: S sample rows (every P/S-th row), ranked among themselves in one workgroup, cut the order
// into S + 1 contiguous ranges; a row finds its range by binary search over the sorted samples (log2 S comparator calls), and its
// rank is (rows in earlier ranges) + (rows of its own range that sort before it) — all pairs inside a range of ~40 rows.
// P (log2 S + P/S) comparator calls in five dependent launches instead of P log^2 P in seven, every one of them a lane per pair or
// a lane per row.  A range of any size is ranked correctly; an unlucky sample only costs time.
#pragma once
#include "snapshot.hpp"

namespace mmp {

constexpr int kSampleMax = 512;   // splitters (256 below 32k rows)
constexpr int kCtrStride = 16;    // the range counters one per 64-byte line: P atomics into a few dozen lines serialise (~7 ns each)

// the S sample rows (every P/S-th row of the table) + cleared range counters
__global__ __launch_bounds__(256) void sample_gather_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t S, int64_t min_space,
                                                            RankRow *__restrict__ srows, int32_t *__restrict__ hist, int32_t *__restrict__ cur)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < S) {
        const int32_t p = (int32_t)(((int64_t)i * P) / S);
        RankRow r = make_rank_row(pods[p], min_space);
        r.pad0 = (uint32_t)p;
        srows[i] = r;
    }
    if (i <= S) hist[i * kCtrStride] = cur[i * kCtrStride] = 0;
}

// one workgroup per sample, one LANE per pair: its rank among the samples = how many sort before it; written in place.
// Two samples that compare equal (rows with one id_order and every other field the same: the table is not strictly ordered) are
// told apart by their sample index, so that every slot of `split` is written exactly once — the splitters stay sorted, the equal
// rows land in one range, and the scatter's occupancy check reports them (MMP_EORDER) instead of a stale splitter from an earlier
// commit deciding where rows go.
__global__ __launch_bounds__(256) void sample_sort_kernel(const RankRow *__restrict__ srows, int32_t S, int64_t churn2, RankRow *__restrict__ split)
{
    __shared__ int32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const RankRow me = srows[blockIdx.x];
    int32_t c = 0;
    for (int j = threadIdx.x; j < S; j += 256) {
        const RankRow o = srows[j];
        c += (placement_less(o, me, churn2) || (j < (int)blockIdx.x && !placement_less(me, o, churn2))) ? 1 : 0;
    }
    c = wave_sum_i32(c);
    if (lane_id() == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) split[s_cnt] = me;
}

// range of every row = how many samples sort before it; range sizes
__global__ __launch_bounds__(256) void sample_range_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t S, int64_t min_space, int64_t churn2,
                                                           const RankRow *__restrict__ split, int32_t *__restrict__ range_of,
                                                           int32_t *__restrict__ hist)
{
    __shared__ RankRow s_split[kSampleMax];
    for (int i = threadIdx.x; i < S; i += 256) s_split[i] = split[i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const RankRow me = make_rank_row(pods[p], min_space);
    int lo = 0, hi = S;  // first sample that does NOT sort before me
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (placement_less(s_split[mid], me, churn2))
            lo = mid + 1;
        else
            hi = mid;
    }
    range_of[p] = lo;
    atomicAdd(&hist[lo * kCtrStride], 1);
}

// exclusive scan of the S + 1 range sizes by every workgroup for itself (513 integers: cheaper than a launch of its own)
__device__ __forceinline__ void range_offsets(const int32_t *__restrict__ hist, int32_t S, int32_t *s_off /*[kSampleMax + 2]*/)
{
    __shared__ int32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base <= S; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int32_t v = i <= S ? hist[i * kCtrStride] : 0;
        // (blockDim.x is a multiple of 64: wave scans + per-wave totals through LDS)
        __shared__ int32_t s_wt[16];
        const int32_t incl = wave_incl_scan_i32(v);
        if (lane_id() == 63) s_wt[threadIdx.x >> 6] = incl;
        __syncthreads();
        int32_t before = s_carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_wt[w];
        if (i <= S) s_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) s_off[S + 1] = s_carry;
    __syncthreads();
}

// rows into range order (any order inside a range); workgroup 0 also leaves the offsets in global memory for the last launch
__global__ __launch_bounds__(256) void sample_scatter_kernel(int32_t P, int32_t S, const int32_t *__restrict__ range_of, const int32_t *__restrict__ hist,
                                                             int32_t *__restrict__ cur, int32_t *__restrict__ off, int32_t *__restrict__ idx)
{
    __shared__ int32_t s_off[kSampleMax + 2];
    range_offsets(hist, S, s_off);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i <= S + 1; i += 256) off[i] = s_off[i];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int b = range_of[p];
    idx[s_off[b] + atomicAdd(&cur[b * kCtrStride], 1)] = p;
}

// one WAVEFRONT per row: rank = rows in earlier ranges + rows of its own range that sort before it (one lane per pair, the range's
// rows gathered 64 at a time) — the cost of a row is its range's size, spread over the chip, whatever the sample made of the ranges.
// Rows are taken in RANGE order, sixteen per workgroup: neighbours gather the same rows, from the CU's L1 instead of L2.
__global__ __launch_bounds__(1024) void sample_rank_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space, int64_t churn2,
                                                           const int32_t *__restrict__ range_of, const int32_t *__restrict__ off,
                                                           const int32_t *__restrict__ idx, int32_t p_lo, int32_t p_hi, int32_t *__restrict__ rank)
{
    const int pos = blockIdx.x * 16 + (int)(threadIdx.x >> 6), lane = lane_id();
    if (pos >= P) return;
    const int p = idx[pos];
    if (p < p_lo || p >= p_hi) return;  // (a pod-axis shard keeps its own slice's ranks)
    const int b = range_of[p], lo = off[b], n = off[b + 1] - lo;
    const RankRow me = make_rank_row(pods[p], min_space);
    int32_t cnt = 0;
    for (int t = 0; t < n; t += 64) {
        bool less = false;
        if (t + lane < n) less = placement_less(make_rank_row(pods[idx[lo + t + lane]], min_space), me, churn2);
        cnt += __popcll(__ballot(less));
    }
    if (lane == 0) rank[p] = lo + cnt;
}

}  // namespace mmp
