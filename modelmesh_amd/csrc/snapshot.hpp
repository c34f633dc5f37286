// snapshot.hpp — the immutable device snapshot the decision kernels read, and
// the kernels that build it from raw InstanceRecord rows.
//
// Layout in HBM.  All per-pod columns are stored in
// PLACEMENT_ORDER (MM.java:4646-4703) so that "position" == rank and every
// ordered construct of CacheMissForwardingLB.getNext (first eligible, break at
// the first violator, index-th survivor) becomes a bit-scan / prefix-popcount
// over 64-pod words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmplace.h"
#include "wave.hpp"

namespace mmp {

struct Snap {
    int32_t P;       // pod slots (including tombstones / shutting-down rows)
    int32_t W;       // ceil(P/64) words per bitmap
    int32_t T;       // bitmap rows (>=1; row 0 is "no type table")
    int32_t any_rs;  // excludeReplicaSets non-empty (MM.java:4792)
    int64_t min_space;
    // rank-ordered columns
    const int64_t *lru;
    const int64_t *rem;
    const int32_t *cnt;
    const int32_t *rpm;
    const int32_t *orig;    // position -> pod index
    const int32_t *pos_of;  // pod index -> position
    // rank-ordered bitmaps, [T][W] unless noted
    const uint64_t *elig;       // allowed ∧ live ∧ present ∧ ¬replaced-replica-set
    const uint64_t *elig_nors;  // allowed ∧ live ∧ present
    const uint64_t *pref;       // preferred instances
    const uint8_t *has_pref;    // [T] getPreferredInstances(type) != null
    const uint64_t *fullw;      // [W] isFull(row.remaining)
    const uint64_t *ge;         // [kGeRows][W] row r: count >= kGeBase + r (the count break of MM.java:4925)
    // A pod-axis shard's view of its own slice (shard_kernels.hpp: place_shard_fast_kernel): positions and
    // words above are LOCAL, pos_of stays global.  All zero for the ordinary single-device snapshot.
    int32_t pos_base;    // global position of local position 0
    int32_t w_base;      // global word of local word 0
    int32_t more_after;  // positions beyond this view exist (the view is not the tail of the order)
    // Prefix tables over the words of every type's candidate bitmap, [2][T][W+1] (variant 0: elig, variant 1:
    // elig & pref): pc[w] = set bits in words [0, w), ph[w] = sum of the audit-hash terms of words [0, w).
    // They let one lane count / hash / select in a shortlist that spans the table (place_kernel.hpp,
    // lane_decide<…, LONG>).  Null on shard views.
    // nz[w] = first word index >= w with a bit set (W if none; nz[W] = W): a scan steps over the empty stretches
    // of a type that few instances may host with one load.
    const int32_t *pc;
    const uint64_t *ph;
    const int32_t *nz;
    // Where the count break (MM.java:4925-4926) can first fire, without a scan: counts do not decrease along the head of the order
    // (non-full instances of one version stand in count order, PLACEMENT_ORDER :4676) — ctpos[kGeRows] = the end of that
    // non-decreasing head, ctpos[r] = its first position with count >= kGeBase + r (the end if none).  Null on shard views.
    const int32_t *ctpos;
    // The inverse of pc, [2][T][W * 64] each (round 5; null on shard views): sel[k] = position of the k-th candidate bit of the
    // row (k counted over the whole table: pc's numbering), rk[pos] = that k for a position whose bit is set, -1 otherwise.  The
    // long path's index-th survivor is then ONE lookup instead of a binary search over pc plus a select inside the word, and an
    // excluded candidate's place in the shortlist one lookup instead of its word, its preference word and its prefix count.
    const int32_t *sel;
    const int32_t *rk;
    // audit_mul(w) for every 64-position word w of the table (round 6; null on shard views): the long path takes an excluded candidate's
    // term off the audit hash with one lookup (a 64-bit multiply-mix per exclusion slot was a tenth of its instructions)
    const uint64_t *amul;
    // The per-type shortlists of place_kernel.hpp: TypeMemo (round 5; null on shard views and when the head windows are off):
    // memo[kWinLds], memo_cand[kWinLds][kMemoCand] = list 0's pod indices in candidate order, memo_rk[kWinLds][kMemoCand] = what
    // each position of the type's window is to list 0 (candidate number, the best instance, nothing: kRk*).
    const struct TypeMemo *memo;
    const int32_t *memo_cand;
    const int16_t *memo_rk;
    // The recorded walks of the long shortlists (round 6; place_kernel.hpp: LongMemo), a row per type; null: not built (no inverse
    // tables, shard views, MMP_NO_LONG_MEMO=1)
    const struct LongMemo *lmemo;
};

// count >= 10 is a fixed clause of the shortlist's count break (MM.java:4925-4926); the other clause,
// count > bestCount + bestCount/4, makes the effective threshold max(10, that + 1).
constexpr int kGeBase = 10;
constexpr int kGeRows = 54;  // thresholds 10..63; a larger one falls back to the count column scan

// Row staged in LDS by the rank kernel: exactly the fields PLACEMENT_ORDER reads.
struct __attribute__((aligned(16))) RankRow {
    int64_t vers, rem, lru, cap;
    int32_t count, lt_free, lip, rpm;
    uint32_t id_order, flags;  // flags: bit0 absent (shutting down / tombstone), bit1 full
    uint32_t pad0, pad1;
};
static_assert(sizeof(RankRow) == 64, "RankRow is 64 bytes");

// Java long subtraction (two's-complement wrap) and ModelMesh.age(), MM.java:4162-4164 (0 means "now")
__device__ __forceinline__ int64_t jsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t age_of(int64_t t, int64_t now) { return t == 0 ? 0 : jsub64(now, t); }

__host__ __device__ __forceinline__ int64_t remaining_of(int64_t cap, int64_t used)
{
    int64_t d = (int64_t)((uint64_t)cap - (uint64_t)used);
    return d > 0 ? d : 0;  // InstanceRecord.java:203-205
}

__host__ __device__ __forceinline__ RankRow make_rank_row(const mmp_pod_row &r, int64_t min_space)
{
    RankRow o;
    o.vers = r.version;
    o.rem = remaining_of(r.capacity, r.used);
    o.lru = r.lru_time;
    o.cap = r.capacity;
    o.count = r.count;
    o.lt_free = (int32_t)((uint32_t)r.loading_threads - (uint32_t)r.loading_in_progress);
    o.lip = r.loading_in_progress;
    o.rpm = r.rpm;
    o.id_order = r.id_order;
    const bool absent = (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) != 0;
    o.flags = (absent ? 1u : 0u) | ((o.rem < min_space) ? 2u : 0u);
    o.pad0 = o.pad1 = 0;
    return o;
}

// PLACEMENT_ORDER.compare(a,b) < 0, transcribed clause by clause from
// MM.java:4646-4703.  Absent rows play the shuttingDown role (sorted last).
__host__ __device__ __forceinline__ bool placement_less(const RankRow &a, const RankRow &b, int64_t churn2)
{
    const bool sd1 = a.flags & 1u, sd2 = b.flags & 1u;
    if (sd1 != sd2) return !sd1;  // :4653-4656
    const bool full1 = a.flags & 2u, full2 = b.flags & 2u;
    if (a.vers != b.vers) {  // :4660-4666
        if (a.vers > b.vers) {
            if (!full1 || a.lru > churn2) return true;
        } else if (!full2 || b.lru > churn2)
            return false;
    }
    if (full1 != full2) return !full1;  // :4669
    if (full1 && a.lru != b.lru) return a.lru < b.lru;  // :4670-4674
    const int32_t cd = (int32_t)((uint32_t)a.count - (uint32_t)b.count);  // :4676 int subtraction
    if (cd != 0) return cd < 0;
    if (a.rem != b.rem) return a.rem > b.rem;            // :4679-4680
    if (!full1 && a.lru != b.lru) return a.lru < b.lru;  // :4681-4685
    if (a.lt_free != b.lt_free) return a.lt_free > b.lt_free;  // :4692
    if (a.lip != b.lip) return a.lip < b.lip;                  // :4693
    if (a.cap != b.cap) return a.cap > b.cap;                  // :4694
    if (a.rpm != b.rpm) return a.rpm < b.rpm;                  // :4695
    return a.id_order < b.id_order;                            // :4696
}

// rank[p] += #{q in this block's slice : q sorts before p}.  All-pairs: P is at
// most tens of thousands, the row tile is broadcast from LDS, and the literal
// comparator (not a derived key) is what gets evaluated.
constexpr int kRankBlock = 256;
__global__ __launch_bounds__(kRankBlock) void rank_pods_kernel(const mmp_pod_row *__restrict__ pods,
                                                               int32_t P, int64_t min_space,
                                                               int64_t churn2, int32_t slices,
                                                               int32_t p_lo, int32_t p_hi,
                                                               int32_t *__restrict__ rank)
{
    // ranks the pods [p_lo, p_hi) against ALL pods (a pod-axis shard ranks only its own slice,
    // the slices are then summed by an all-reduce; unsharded: [0, P))
    __shared__ RankRow tile[kRankBlock];
    const int p = p_lo + blockIdx.x * kRankBlock + threadIdx.x;
    RankRow me;
    if (p < p_hi) me = make_rank_row(pods[p], min_space);
    // this block compares against q in [q0, q1)
    const int per = (P + slices - 1) / slices;
    const int q0 = blockIdx.y * per;
    const int q1 = min(P, q0 + per);
    int32_t before = 0;
    for (int base = q0; base < q1; base += kRankBlock) {
        const int q = base + threadIdx.x;
        __syncthreads();
        if (q < q1) tile[threadIdx.x] = make_rank_row(pods[q], min_space);
        __syncthreads();
        const int nq = min(kRankBlock, q1 - base);
        if (p < p_hi) {
            for (int j = 0; j < nq; j++) {
                if (base + j != p && placement_less(tile[j], me, churn2)) before++;
            }
        }
    }
    if (p < p_hi && before) atomicAdd(&rank[p], before);
}

// ---- ranking by sorting --------------------------------------------------------------------------
// PLACEMENT_ORDER's version clause (:4660-4666) is the only place where the comparator can fail to be
// transitive: "newer version first unless it is full with lruTime <= 2*minChurnAgeMs".  When no full
// row has such an lruTime, or all rows carry one instanceVersion, the clause reduces to a plain key and
// the literal comparator is a strict total order (ids are unique), so ANY comparison sort with that
// same literal comparator yields the one and only order: O(P log P) comparator evaluations
// (rocprim::merge_sort over pod indices) instead of the P^2 of rank_pods_kernel.  The host checks the
// condition while it stages the rows; otherwise the all-pairs kernel runs and its duplicate-rank test
// reports a genuinely cyclic table (MMP_EORDER).
__global__ void rank_rows_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                 RankRow *__restrict__ rows)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    RankRow r = make_rank_row(pods[p], min_space);
    r.pad0 = (uint32_t)p;  // the row carries its pod index through the sort
    rows[p] = r;
}

// The rows themselves are the sort keys (64 B each): the comparator then reads nothing but its two
// arguments.  (Sorting pod indices with a comparator that fetches the rows was measured first: the block
// sort spent 126 us of dependent global loads on 10k pods.)
struct PlacementRowLess {
    int64_t churn2;
    __host__ __device__ bool operator()(const RankRow &a, const RankRow &b) const { return placement_less(a, b, churn2); }
};

__global__ void rank_from_order_kernel(const RankRow *__restrict__ sorted, int32_t P, int32_t *__restrict__ rank)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < P) rank[sorted[pos].pad0] = pos;
}

// Scatter rows into rank order; detect a non-total order (two rows with the
// same rank) through the occupancy counters.
__global__ void scatter_pods_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                    const int32_t *__restrict__ rank, int32_t *__restrict__ occupancy,
                                    int64_t *__restrict__ lru, int64_t *__restrict__ rem,
                                    int32_t *__restrict__ cnt, int32_t *__restrict__ rpm,
                                    int32_t *__restrict__ orig, int32_t *__restrict__ pos_of,
                                    int32_t *__restrict__ err)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int pos = rank[p];
    if (pos < 0 || pos >= P || atomicAdd(&occupancy[pos], 1) != 0) {
        atomicExch(err, 1);
        return;
    }
    const mmp_pod_row r = pods[p];
    lru[pos] = r.lru_time;
    rem[pos] = remaining_of(r.capacity, r.used);
    cnt[pos] = r.count;
    rpm[pos] = r.rpm;
    orig[pos] = p;
    pos_of[p] = pos;
}

// the rank-ordered columns are padded to whole 64-position words: positions [P, padded) read as an absent row
__global__ __launch_bounds__(64) void zero_tails_kernel(int64_t *__restrict__ lru, int64_t *__restrict__ rem, int32_t *__restrict__ cnt,
                                                        int32_t *__restrict__ rpm, int32_t *__restrict__ orig, int32_t *__restrict__ pos_of,
                                                        int32_t P, int32_t padded)
{
    const int i = P + (int)threadIdx.x;
    if (i < padded) {
        lru[i] = 0;
        rem[i] = 0;
        cnt[i] = 0;
        rpm[i] = 0;
        orig[i] = 0;
        pos_of[i] = 0;
    }
}

// ---- a commit whose table differs from the published one in a few rows (handleInstanceTableChange: one InstanceRecord per
// event, MM.java:1455-1568; the rate task republishes this instance's own row, :232) ------------------------------------------
// PLACEMENT_ORDER compares two rows by their own fields, so the rows that did not change keep their relative order: the new
// order = the old one with the K changed rows taken out and put back where the literal comparator places them.  The HOST does
// the K binary searches against its mirror of the order (K * log2 P comparator calls, no dependent device loads) and hands
// the kernel, by value: the old positions taken out (ascending), the insertion points among the unchanged rows, the changed
// rows and their final ranks.  The kernel is the sort AND the scatter of a full commit in one pass over the table:
//   unchanged row at old position q:  j = q - #{removed < q},  rank = j + #{insertion points <= j}
constexpr int kDeltaRows = 16;
struct DeltaRows {
    int32_t K;
    int32_t pod[kDeltaRows];      // the changed rows
    int32_t removed[kDeltaRows];  // their old positions, ascending
    int32_t ins[kDeltaRows];      // per changed row: how many UNCHANGED rows sort before it
    int32_t newrank[kDeltaRows];  // per changed row: its position in the new order
    mmp_pod_row rows[kDeltaRows];
};

__global__ __launch_bounds__(256) void delta_scatter_kernel(const mmp_pod_row *__restrict__ old_pods, const int32_t *__restrict__ old_pos_of,
                                                            int32_t P, DeltaRows D, mmp_pod_row *__restrict__ pods,
                                                            int32_t *__restrict__ rank, int64_t *__restrict__ lru,
                                                            int64_t *__restrict__ rem, int32_t *__restrict__ cnt, int32_t *__restrict__ rpm,
                                                            int32_t *__restrict__ orig, int32_t *__restrict__ pos_of)
{
    __shared__ int32_t s_pod[kDeltaRows], s_removed[kDeltaRows], s_ins[kDeltaRows];
    if (threadIdx.x < kDeltaRows) {
        const int k = threadIdx.x;
        s_pod[k] = k < D.K ? D.pod[k] : -1;
        s_removed[k] = k < D.K ? D.removed[k] : INT32_MAX;  // (never below a position)
        s_ins[k] = k < D.K ? D.ins[k] : INT32_MAX;          // (never at or before a row)
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    mmp_pod_row r = old_pods[p];
    const int q = old_pos_of[p];
    int below = 0, mine = -1;
#pragma unroll
    for (int k = 0; k < kDeltaRows; k++) {
        below += s_removed[k] < q ? 1 : 0;
        mine = s_pod[k] == p ? k : mine;
    }
    int pos;
    if (mine >= 0) {  // (a handful of threads of the launch)
        for (int k = 0; k < D.K; k++)
            if (k == mine) {
                r = D.rows[k];
                pos = D.newrank[k];
            }
    } else {
        const int j = q - below;
        pos = j;
#pragma unroll
        for (int k = 0; k < kDeltaRows; k++) pos += s_ins[k] <= j ? 1 : 0;
    }
    pods[p] = r;
    rank[p] = pos;
    lru[pos] = r.lru_time;
    rem[pos] = remaining_of(r.capacity, r.used);
    cnt[pos] = r.count;
    rpm[pos] = r.rpm;
    orig[pos] = p;
    pos_of[p] = pos;
}

// One wave per (bitmap row, word): gather 64 per-pod predicates through the
// rank permutation and ballot them into one rank-ordered word.
//   allowed/prefer: [T][W] over pod index (or null = all / none)
//   rs_bad: per pod, 1 if its replica set is "likely replaced"
__device__ __forceinline__ void build_masks_kernel_body(int bid, int nblk, const mmp_pod_row *__restrict__ pods, int32_t P, int32_t W, int32_t T,
                                   int64_t min_space, const int32_t *__restrict__ orig,
                                   const uint64_t *__restrict__ allowed,
                                   const uint8_t *__restrict__ has_allowed,
                                   const uint64_t *__restrict__ prefer,
                                   const uint8_t *__restrict__ has_prefer,
                                   const uint8_t *__restrict__ rs_bad, uint64_t *__restrict__ elig,
                                   uint64_t *__restrict__ elig_nors, uint64_t *__restrict__ pref,
                                   uint64_t *__restrict__ fullw)
{
    const int wave = (bid * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= T * W) return;
    const int t = wave / W, w = wave - t * W;
    const int pos = w * 64 + lane;
    bool e = false, en = false, pf = false, fl = false;
    if (pos < P) {
        const int p = orig[pos];
        const mmp_pod_row r = pods[p];
        const bool present = (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) == 0;
        const bool live = (r.flags & MMP_POD_LIVE) != 0;
        bool al = true;
        if (has_allowed && has_allowed[t]) al = (allowed[(size_t)t * W + (p >> 6)] >> (p & 63)) & 1ull;
        en = present && live && al;
        e = en && !(rs_bad && rs_bad[p]);
        if (has_prefer && has_prefer[t]) pf = (prefer[(size_t)t * W + (p >> 6)] >> (p & 63)) & 1ull;
        fl = remaining_of(r.capacity, r.used) < min_space;
    }
    const uint64_t be = __ballot(e), ben = __ballot(en), bp = __ballot(pf), bf = __ballot(fl);
    if (lane == 0) {
        elig[(size_t)t * W + w] = be;
        elig_nors[(size_t)t * W + w] = ben;
        pref[(size_t)t * W + w] = bp;
        if (t == 0) fullw[w] = bf;
    }
}
__global__ void build_masks_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int32_t W, int32_t T,
                                   int64_t min_space, const int32_t *__restrict__ orig,
                                   const uint64_t *__restrict__ allowed,
                                   const uint8_t *__restrict__ has_allowed,
                                   const uint64_t *__restrict__ prefer,
                                   const uint8_t *__restrict__ has_prefer,
                                   const uint8_t *__restrict__ rs_bad, uint64_t *__restrict__ elig,
                                   uint64_t *__restrict__ elig_nors, uint64_t *__restrict__ pref,
                                   uint64_t *__restrict__ fullw)
{
    build_masks_kernel_body((int)blockIdx.x, (int)gridDim.x, pods, P, W, T, min_space, orig, allowed, has_allowed, prefer, has_prefer, rs_bad, elig, elig_nors, pref, fullw);
}

// ge[r][w] = ballot over the 64 rank positions of word w of (count >= kGeBase + r): one wave per (r, w)
__device__ __forceinline__ void build_ge_kernel_body(int bid, int nblk, const int32_t *__restrict__ cnt, int32_t P, int32_t W, uint64_t *__restrict__ ge)
{
    const int wave = (bid * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= kGeRows * W) return;
    const int r = wave / W, w = wave - r * W;
    const int pos = w * 64 + lane;
    const uint64_t b = __ballot(pos < P && cnt[pos] >= kGeBase + r);
    if (lane == 0) ge[(size_t)r * W + w] = b;
}
__global__ void build_ge_kernel(const int32_t *__restrict__ cnt, int32_t P, int32_t W, uint64_t *__restrict__ ge)
{
    build_ge_kernel_body((int)blockIdx.x, (int)gridDim.x, cnt, P, W, ge);
}

// Snap::ctpos: one workgroup.  The end of the non-decreasing head of the count column (first descent: a minimum over all
// positions, every chunk of the column read at once), then per threshold row a binary search in it (thread r: kGeBase + r).
constexpr int kCtposBlock = 1024;
__device__ __forceinline__ void build_ctpos_block(const int32_t *__restrict__ cnt, int32_t P, int32_t *__restrict__ ctpos)
{
    __shared__ int32_t s_end;
    if (threadIdx.x == 0) s_end = P;
    __syncthreads();
    int32_t first = P;  // (no early exit: a data-dependent break serialises the loads — 195 trips of an L2 latency each at 50k rows)
    for (int p = 1 + (int)threadIdx.x; p < P; p += blockDim.x) {
        const bool viol = cnt[p] < cnt[p - 1];
        first = (viol && p < first) ? p : first;
    }
    if (first < P) atomicMin(&s_end, first);
    __syncthreads();
    const int mono_end = s_end;
    if (threadIdx.x < kGeRows) {
        const int32_t thr = kGeBase + threadIdx.x;
        int lo = 0, hi = mono_end;  // first p in [0, mono_end) with cnt[p] >= thr
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cnt[mid] >= thr)
                hi = mid;
            else
                lo = mid + 1;
        }
        ctpos[threadIdx.x] = lo;
    }
    if (threadIdx.x == 0) ctpos[kGeRows] = mono_end;
}
__global__ __launch_bounds__(kCtposBlock) void build_ctpos_kernel(const int32_t *__restrict__ cnt, int32_t P, int32_t *__restrict__ ctpos)
{
    build_ctpos_block(cnt, P, ctpos);
}

// rs_bad[p] = pod p's replica set is in the replaced list (MM.java:4769-4770)
__global__ void mark_replaced_kernel(const mmp_pod_row *__restrict__ pods, int32_t P,
                                     const int32_t *__restrict__ rs, int32_t n_rs,
                                     uint8_t *__restrict__ rs_bad)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int32_t mine = pods[p].replica_set;
    uint8_t bad = 0;
    if (mine >= 0)
        for (int i = 0; i < n_rs; i++)
            if (rs[i] == mine) bad = 1;
    rs_bad[p] = bad;
}

// ClusterStats (InstanceSetStatsTracker.java:53-92): sums and min over present rows.
struct StatsAcc {
    unsigned long long total_capacity, total_free;
    long long global_lru;
    int32_t instance_count, model_copy_count;
    int32_t sparse_types;  // set by build_prefix_kernel: some type's eligible instances lie further apart than a lane scan reaches
};

__device__ __forceinline__ void cluster_stats_kernel_body(int bid, int nblk, const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                     StatsAcc *__restrict__ acc)
{
    int64_t cap = 0, fre = 0, lru = INT64_MAX;
    int32_t n = 0, mc = 0;
    for (int p = bid * blockDim.x + threadIdx.x; p < P; p += nblk * blockDim.x) {
        const mmp_pod_row r = pods[p];
        if (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) continue;
        n++;
        mc = (int32_t)((uint32_t)mc + (uint32_t)r.count);
        cap = (int64_t)((uint64_t)cap + (uint64_t)r.capacity);
        const int64_t avail = remaining_of(r.capacity, r.used);
        if (!(avail < min_space)) fre = (int64_t)((uint64_t)fre + (uint64_t)avail);
        if (r.lru_time > 0 && r.lru_time < lru) lru = r.lru_time;  // addLru ignores <= 0
    }
    cap = wave_sum_i64(cap);
    fre = wave_sum_i64(fre);
    lru = wave_min_i64(lru);
    n = wave_sum_i32(n);
    mc = wave_sum_i32(mc);
    if (lane_id() == 0) {
        atomicAdd(&acc->total_capacity, (unsigned long long)cap);
        atomicAdd(&acc->total_free, (unsigned long long)fre);
        atomicMin(&acc->global_lru, (long long)lru);
        atomicAdd(&acc->instance_count, n);
        atomicAdd(&acc->model_copy_count, mc);
    }
}
__global__ void cluster_stats_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                     StatsAcc *__restrict__ acc)
{
    cluster_stats_kernel_body((int)blockIdx.x, (int)gridDim.x, pods, P, min_space, acc);
}

// ---- instance partitions and per-type subset stats (TypeConstraintManager) ---------------------------
// With type constraints configured, ModelMesh does not use the cluster-wide ClusterStats everywhere:
// instances are partitioned by their ProhibitedTypeSet — the constrained types they cannot host
// (TypeConstraintManager.java:553-578) — every partition keeps its own InstanceSetStatsTracker, a model
// type's stats are the sum over the partitions that can host it (candidateSubsetStats, :356-377;
// typeSetStats, MM.java:1432-1439) and an instance's own stats are its partition's (instanceSetStats,
// MM.java:1446-1448).  pod_pts[p] = partition of pod p (interned on the host at commit from the allowed
// bitmaps), prohib[k] = the partition's prohibited types as a bitset over type rows.
// Quirk (MM.java:1515-1542): a partition's lru is reset and re-accumulated over
// ALL instances of clusterState on every event that touches the partition, so it is the cluster-wide
// minimum as of that event; rebuilt from a table snapshot it is the cluster-wide minimum.
// Accumulated in LDS per workgroup first (a few partitions receive every pod: global atomics on the same
// four words from 10k lanes were measured at 221 us), then one global atomic per touched partition and field.
constexpr int kPtsLds = 512;  // partitions accumulated in LDS; beyond that, global atomics
__device__ __forceinline__ void partition_stats_kernel_body(int bid, int nblk, const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                                              const int32_t *__restrict__ pod_pts, int32_t NP,
                                                              StatsAcc *__restrict__ pstats)
{
    __shared__ unsigned long long cap_s[kPtsLds], free_s[kPtsLds];
    __shared__ int32_t cnt_s[kPtsLds], mc_s[kPtsLds];
    const bool lds = NP <= kPtsLds;
    if (lds) {
        for (int k = threadIdx.x; k < NP; k += blockDim.x) {
            cap_s[k] = free_s[k] = 0;
            cnt_s[k] = mc_s[k] = 0;
        }
        __syncthreads();
    }
    for (int p = bid * blockDim.x + threadIdx.x; p < P; p += nblk * blockDim.x) {
        const mmp_pod_row r = pods[p];
        if (r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) continue;
        const int k = pod_pts[p];
        const int64_t rem = remaining_of(r.capacity, r.used);
        const unsigned long long fr = !(rem < min_space) ? (unsigned long long)rem : 0ull;
        if (lds) {
            atomicAdd(&cap_s[k], (unsigned long long)r.capacity);
            if (fr) atomicAdd(&free_s[k], fr);
            atomicAdd(&cnt_s[k], 1);
            atomicAdd(&mc_s[k], r.count);
        } else {
            StatsAcc *a = &pstats[k];
            atomicAdd(&a->total_capacity, (unsigned long long)r.capacity);
            if (fr) atomicAdd(&a->total_free, fr);
            atomicAdd(&a->instance_count, 1);
            atomicAdd(&a->model_copy_count, r.count);
        }
    }
    if (lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < NP; k += blockDim.x) {
            if (cnt_s[k] == 0) continue;
            StatsAcc *a = &pstats[k];
            atomicAdd(&a->total_capacity, cap_s[k]);
            if (free_s[k]) atomicAdd(&a->total_free, free_s[k]);
            atomicAdd(&a->instance_count, cnt_s[k]);
            atomicAdd(&a->model_copy_count, mc_s[k]);
        }
    }
}
__global__ __launch_bounds__(256) void partition_stats_kernel(const mmp_pod_row *__restrict__ pods, int32_t P, int64_t min_space,
                                                              const int32_t *__restrict__ pod_pts, int32_t NP,
                                                              StatsAcc *__restrict__ pstats)
{
    partition_stats_kernel_body((int)blockIdx.x, (int)gridDim.x, pods, P, min_space, pod_pts, NP, pstats);
}

// one thread per partition / per type row; global = the cluster-wide stats of the same snapshot
__device__ __forceinline__ void subset_stats_finish_kernel_body(int bid, int nblk, const StatsAcc *__restrict__ global, StatsAcc *__restrict__ pstats, int32_t NP,
                                           const uint64_t *__restrict__ prohib, int32_t Tw, int32_t T,
                                           const uint8_t *__restrict__ has_allowed, StatsAcc *__restrict__ tstats)
{
    const int i = bid * blockDim.x + threadIdx.x;
    if (i < NP) pstats[i].global_lru = pstats[i].instance_count > 0 ? global->global_lru : INT64_MAX;
    if (i < T) {
        StatsAcc t = *global;
        if (NP > 0 && has_allowed && has_allowed[i]) {  // a constrained type: the partitions that may host it
            t.total_capacity = t.total_free = 0;
            t.instance_count = t.model_copy_count = 0;
            for (int k = 0; k < NP; k++) {
                if ((prohib[(size_t)k * Tw + (i >> 6)] >> (i & 63)) & 1ull) continue;
                t.total_capacity += pstats[k].total_capacity;
                t.total_free += pstats[k].total_free;
                t.instance_count += pstats[k].instance_count;
                t.model_copy_count += pstats[k].model_copy_count;
            }
            t.global_lru = t.instance_count > 0 ? global->global_lru : INT64_MAX;
        }
        tstats[i] = t;
    }
}
__global__ void subset_stats_finish_kernel(const StatsAcc *__restrict__ global, StatsAcc *__restrict__ pstats, int32_t NP,
                                           const uint64_t *__restrict__ prohib, int32_t Tw, int32_t T,
                                           const uint8_t *__restrict__ has_allowed, StatsAcc *__restrict__ tstats)
{
    subset_stats_finish_kernel_body((int)blockIdx.x, (int)gridDim.x, global, pstats, NP, prohib, Tw, T, has_allowed, tstats);
}

// pc / ph of Snap: one wavefront per (variant, type row), 64 words per step
__device__ __forceinline__ void build_prefix_kernel_body(int bid, int nblk, const uint64_t *__restrict__ elig, const uint64_t *__restrict__ pref,
                                                          int32_t T, int32_t W, int32_t *__restrict__ pc,
                                                          uint64_t *__restrict__ ph, int32_t *__restrict__ nz,
                                                          StatsAcc *__restrict__ acc)
{
    const int v = bid / T, t = bid - v * T, lane = threadIdx.x;
    const uint64_t *E = elig + (size_t)t * W, *Pm = pref + (size_t)t * W;
    int32_t *PC = pc + ((size_t)v * T + t) * (W + 1);
    uint64_t *PH = ph + ((size_t)v * T + t) * (W + 1);
    int32_t carry_c = 0;
    uint64_t carry_h = 0;
    if (lane == 0) {
        PC[0] = 0;
        PH[0] = 0;
    }
    for (int base = 0; base < W; base += 64) {
        const int w = base + lane;
        uint64_t word = 0;
        if (w < W) word = v ? (E[w] & Pm[w]) : E[w];
        int32_t c = __popcll((unsigned long long)word);
        uint64_t h = audit_term(word, (uint64_t)w);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {  // inclusive scans across the wavefront
            const int32_t tc = __shfl_up(c, o, 64);
            const uint64_t th = shfl_u64(h, lane >= o ? lane - o : lane);
            if (lane >= o) {
                c += tc;
                h += th;
            }
        }
        if (w < W) {
            PC[w + 1] = carry_c + c;
            PH[w + 1] = carry_h + h;
        }
        carry_c += readlane_i32(c, 63);
        carry_h += readlane_u64(h, 63);
    }
    // fewer candidates than words: a lane scan of kLaneSpan words sees only a few of them (or none)
    if (v == 0 && lane == 0 && carry_c > 0 && carry_c < W) atomicOr(&acc->sparse_types, 1);
    // nz: the same words back to front, a suffix minimum of the indices of the non-empty ones
    int32_t *NZ = nz + ((size_t)v * T + t) * (W + 1);
    if (lane == 0) NZ[W] = W;
    int32_t carry_n = W;
    for (int base = ((W - 1) / 64) * 64; base >= 0; base -= 64) {
        const int w = base + lane;
        uint64_t word = 0;
        if (w < W) word = v ? (E[w] & Pm[w]) : E[w];
        int32_t m = word ? w : W;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t tm = __shfl_down(m, o, 64);
            if (lane + o < 64 && tm < m) m = tm;
        }
        if (carry_n < m) m = carry_n;
        if (w < W) NZ[w] = m;
        carry_n = readlane_i32(m, 0);
    }
}
__global__ __launch_bounds__(64) void build_prefix_kernel(const uint64_t *__restrict__ elig, const uint64_t *__restrict__ pref,
                                                          int32_t T, int32_t W, int32_t *__restrict__ pc,
                                                          uint64_t *__restrict__ ph, int32_t *__restrict__ nz,
                                                          StatsAcc *__restrict__ acc)
{
    build_prefix_kernel_body((int)blockIdx.x, (int)gridDim.x, elig, pref, T, W, pc, ph, nz, acc);
}

}  // namespace mmp
