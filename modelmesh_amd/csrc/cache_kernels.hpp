// cache_kernels.hpp — stateful per-pod model caches on the device: the timestamp-ordered weighted
// LRU (clhm/ConcurrentLinkedHashMap.java, clhm/LinkedDeque.java) and the unload-buffer accounting
// on top of it (ModelCacheUnloadBufManager.java), replayed one wavefront per cache.
//
// A call brings an ordered list of operations per cache (put / get / weight update / remove and the
// manager's insertNewEntry, adjustNewEntrySpaceRequest, claimRequestedSpaceIfReady,
// adjustWeightAfterLoad, unloadComplete, removeEntry, discardFailedEntry,
// insertFailedPlaceholderEntry).  The wave stages the cache's evictionDeque into LDS (lastUsed, weight,
// key columns), applies the operations in order — lanes cooperate on the O(E) parts: key lookup
// (ballot), the tail walk of LinkedDeque.insert (ballot + clz), the shifts — and writes the deque
// back.  Caches are independent, so the grid is one wave per cache with operations.
//
// The Java evicts under the cache's eviction lock and calls ModelMesh.onEviction -> entryRemoved for
// every victim while still holding it (MM.java:2876-2878); entryRemoved grows the unload buffer entry,
// which can evict again.  That recursion is run here with an explicit LIFO of pending victims, which
// visits them in the same depth-first order.
#pragma once
#include "snapshot.hpp"

namespace mmp {

constexpr int kCacheTile = 2048;  // deque slots staged per cache (entries + inserts of one call)
#define MMP_UNLOADBUF_KEY_C (-1000000)

struct CacheStore {
    // deque storage with per-cache regions [off[c], off[c+1]); live entries are the first n[c]
    const int32_t *off;
    int64_t *lu;
    int32_t *wt;
    int32_t *key;
    int32_t *n;
};

struct ReplayArgs {
    CacheStore src, dst;       // the call re-lays the store out (room for this call's inserts)
    int64_t *capacity;         // [n_caches] clhm capacity (unloadComplete(failure) shrinks it)
    int64_t *weighted_size;    // [n_caches]
    int64_t *oldest;           // [n_caches] the map's `oldestTime` FIELD (clhm :1120): refreshed by afterWrite / the read drain only
    mmp_ubm_state *ubm;        // [n_caches]
    const mmp_cache_op *ops;   // all operations of the call GROUPED BY CACHE (caller order within a cache): one load per operation
    const int32_t *op_order;   // the caller's index of each grouped operation (where its result goes)
    const int32_t *op_off;     // [n_caches+1] ranges of op_order
    mmp_cache_op_out *outs;    // by caller operation index
    int32_t *evicted;          // evicted keys; cache c writes from ev_off[c]
    const int32_t *ev_off;     // [n_caches+1]
    int32_t n_caches;
    int32_t tile;              // LDS slots per column
    int64_t now;
};

// A cache is replayed by a TEAM of TW lanes (TW = 64: the whole wavefront; 8 or 16: several small caches per wavefront, each
// with its own slice of the LDS tile).  Control flow is uniform inside a team, so a ballot taken anywhere below has all of the
// team's lanes active; lanes of other teams may be elsewhere and simply do not contribute.
template <int TW>
__device__ __forceinline__ int team_lane() { return lane_id() & (TW - 1); }
template <int TW>
__device__ __forceinline__ uint64_t team_ballot(bool p)
{
    const uint64_t b = __ballot(p);
    if (TW == 64) return b;
    return (b >> (lane_id() & ~(TW - 1))) & ((1ull << (TW & 63)) - 1ull);
}

struct Deque {
    int64_t *lu;
    int32_t *wt, *key;
    int32_t *stk_key, *stk_wt;  // pending victims (LIFO)
    int head, n, sp;
    int64_t wsize, cap;
    int64_t oldest;  // clhm `oldestTime`: what updateOldestTime() last stored (afterWrite :444, tryToDrainBuffers :463) — setCapacity's
                     // own evict (:305-316) does not store it, so it can be stale after a failed unload (unloadComplete :318-338)
    // manager
    int32_t reserved, tu, deficit;
    int64_t occ;
    // eviction record
    int32_t *ev;
    int nev;
};

template <int TW>
__device__ __forceinline__ int dq_find(const Deque &D, int32_t k)
{
    const int lane = team_lane<TW>();
    for (int base = 0; base < D.n; base += TW) {
        const int i = base + lane;
        const uint64_t b = team_ballot<TW>(i < D.n && D.key[D.head + i] == k);
        if (b) return base + (__ffsll((unsigned long long)b) - 1);
    }
    return -1;
}

// LinkedDeque.insert, LinkedDeque.java:259-288: walk from the tail to the first node with
// lastUsed <= ts and link the new node after it.  Returns the position.
template <int TW>
__device__ __forceinline__ int dq_insert(Deque &D, int64_t ts, int32_t w, int32_t k)
{
    const int lane = team_lane<TW>();
    int l = -1;
    for (int base = D.n > 0 ? (D.n - 1) & ~(TW - 1) : -1; base >= 0; base -= TW) {
        const int i = base + lane;
        const uint64_t b = team_ballot<TW>(i < D.n && D.lu[D.head + i] <= ts);
        if (b) {
            l = base + 63 - __clzll((unsigned long long)b);
            break;
        }
    }
    const int pos = l + 1;
    for (int hi = D.n; hi > pos; hi -= TW) {  // shift [pos, n) one slot towards the tail, tail chunk first
        const int lo = hi - TW > pos ? hi - TW : pos;
        const int i = lo + lane;
        const bool v = i < hi;
        int64_t a = 0;
        int32_t b = 0, c = 0;
        if (v) {
            a = D.lu[D.head + i];
            b = D.wt[D.head + i];
            c = D.key[D.head + i];
        }
        wave_sync();
        if (v) {
            D.lu[D.head + i + 1] = a;
            D.wt[D.head + i + 1] = b;
            D.key[D.head + i + 1] = c;
        }
        wave_sync();
    }
    if (lane == 0) {
        D.lu[D.head + pos] = ts;
        D.wt[D.head + pos] = w;
        D.key[D.head + pos] = k;
    }
    wave_sync();
    D.n++;
    return pos;
}

// `pop`: the head leaves by moving the window (eviction, removal: the slot is never needed again).  A node that is unlinked to be
// re-inserted (reposition) must NOT move the window: head + n would grow by one per repositioned head and walk off the tile the
// host sized as entries + inserts
template <int TW>
__device__ __forceinline__ void dq_unlink(Deque &D, int i, int64_t &ts, int32_t &w, int32_t &k, bool pop = true)
{
    const int lane = team_lane<TW>();
    ts = D.lu[D.head + i];
    w = D.wt[D.head + i];
    k = D.key[D.head + i];
    wave_sync();
    if (i == 0 && pop) {
        D.head++;
        D.n--;
        return;
    }
    for (int lo = i + 1; lo < D.n; lo += TW) {  // shift (i, n) one slot towards the head, head chunk first
        const int idx = lo + lane;
        const bool v = idx < D.n;
        int64_t a = 0;
        int32_t b = 0, c = 0;
        if (v) {
            a = D.lu[D.head + idx];
            b = D.wt[D.head + idx];
            c = D.key[D.head + idx];
        }
        wave_sync();
        if (v) {
            D.lu[D.head + idx - 1] = a;
            D.wt[D.head + idx - 1] = b;
            D.key[D.head + idx - 1] = c;
        }
        wave_sync();
    }
    D.n--;
}

// Node.touch, clhm :1357-1360
template <int TW>
__device__ __forceinline__ void dq_touch(Deque &D, int i, int64_t time, int64_t now)
{
    const int64_t old = D.lu[D.head + i];
    const int64_t nv = time == 0 ? now : (old > time ? old : time);
    wave_sync();
    if (team_lane<TW>() == 0) D.lu[D.head + i] = nv;
    wave_sync();
}

// LinkedDeque.reposition, LinkedDeque.java:243-256
template <int TW>
__device__ __forceinline__ void dq_reposition(Deque &D, int i)
{
    const int64_t lu = D.lu[D.head + i];
    if (i == 0 || D.lu[D.head + i - 1] <= lu) {
        if (i == D.n - 1 || D.lu[D.head + i + 1] >= lu) return;
    }
    int64_t ts;
    int32_t w, k;
    dq_unlink<TW>(D, i, ts, w, k, false);
    dq_insert<TW>(D, ts, w, k);
}

template <int TW>
__device__ __forceinline__ void dq_set_wt(Deque &D, int i, int32_t w)
{
    wave_sync();
    if (team_lane<TW>() == 0) D.wt[D.head + i] = w;
    wave_sync();
}

// updateOldestTime, clhm :1129-1133
__device__ __forceinline__ void dq_refresh_oldest(Deque &D) { D.oldest = D.n ? D.lu[D.head] : -1; }

template <int TW>
__device__ __forceinline__ void record_evicted(Deque &D, int32_t k)
{
    if (team_lane<TW>() == 0) D.ev[D.nev] = k;
    D.nev++;
}

// evict(), clhm :329-352 (makeDead subtracts |weight|, :566-575).  Plain caches record the victims
// directly; with a manager they go onto the pending LIFO, oldest on top.
template <int TW>
__device__ __forceinline__ void dq_evict(Deque &D, bool managed)
{
    const int first = D.sp;
    while (D.wsize > D.cap && D.n > 0) {
        int64_t ts;
        int32_t w, k;
        dq_unlink<TW>(D, 0, ts, w, k);
        D.wsize -= w < 0 ? -(int64_t)w : (int64_t)w;
        if (managed) {
            if (team_lane<TW>() == 0) {
                D.stk_key[D.sp] = k;
                D.stk_wt[D.sp] = w;
            }
            D.sp++;
        } else
            record_evicted<TW>(D, k);
    }
    if (managed && D.sp - first > 1) {  // reverse the new run so the oldest victim is popped first
        wave_sync();
        const int cnt = D.sp - first;
        for (int base = 0; base < cnt / 2; base += TW) {
            const int i = base + team_lane<TW>();
            const bool v = i < cnt / 2;
            int32_t k1 = 0, w1 = 0, k2 = 0, w2 = 0;
            if (v) {
                k1 = D.stk_key[first + i];
                w1 = D.stk_wt[first + i];
                k2 = D.stk_key[D.sp - 1 - i];
                w2 = D.stk_wt[D.sp - 1 - i];
            }
            wave_sync();
            if (v) {
                D.stk_key[first + i] = k2;
                D.stk_wt[first + i] = w2;
                D.stk_key[D.sp - 1 - i] = k1;
                D.stk_wt[D.sp - 1 - i] = w1;
            }
            wave_sync();
        }
    }
    wave_sync();
}

// CacheEntry.updateWeightLocked -> replaceQuietly -> UpdateTask(quiet), without running the listener
template <int TW>
__device__ __forceinline__ void ubm_set_weight_nodrain(Deque &D, int32_t k, int32_t w)
{
    const int i = dq_find<TW>(D, k);
    if (i < 0) return;
    const int32_t diff = (int32_t)((uint32_t)w - (uint32_t)D.wt[D.head + i]);
    if (diff == 0) return;
    dq_set_wt<TW>(D, i, w);
    D.wsize += diff;
    dq_evict<TW>(D, true);
    dq_refresh_oldest(D);  // afterWrite(UpdateTask)
}

// adjustAggregateUnloadingWeight, ModelCacheUnloadBufManager.java:375-392 (listener not yet run)
template <int TW>
__device__ __forceinline__ void ubm_adjust_agg_nodrain(Deque &D, int32_t delta)
{
    if (delta == 0) return;
    D.tu = (int32_t)((uint32_t)D.tu + (uint32_t)delta);
    int32_t nw = D.tu;
    if (nw <= D.reserved)
        nw = D.reserved;
    else {
        const int32_t cap = D.cap > INT32_MAX ? INT32_MAX : (int32_t)D.cap;
        if (cap < nw) nw = cap;
    }
    ubm_set_weight_nodrain<TW>(D, MMP_UNLOADBUF_KEY_C, nw);
}

// the eviction listener: ModelMesh.onEviction -> entryRemoved (:311-316) per victim, depth first
template <int TW>
__device__ __forceinline__ void ubm_drain(Deque &D)
{
    while (D.sp > 0) {
        D.sp--;
        const int32_t k = D.stk_key[D.sp], w = D.stk_wt[D.sp];
        wave_sync();
        record_evicted<TW>(D, k);
        D.occ -= w;
        ubm_adjust_agg_nodrain<TW>(D, w);
    }
}

template <int TW>
__device__ __forceinline__ void ubm_adjust_agg(Deque &D, int32_t delta)
{
    ubm_adjust_agg_nodrain<TW>(D, delta);
    ubm_drain<TW>(D);
}

template <int TW>
__device__ __forceinline__ void ubm_set_weight(Deque &D, int32_t k, int32_t w)
{
    ubm_set_weight_nodrain<TW>(D, k, w);
    ubm_drain<TW>(D);
}

// cacheSpaceIsReady, :395-402
__device__ __forceinline__ bool ubm_space_ready(const Deque &D, int32_t required)
{
    const int32_t ntu = (int32_t)((uint32_t)D.tu + (uint32_t)required);
    if (ntu <= D.reserved) return true;
    return (int64_t)ntu + D.occ <= D.cap;
}

// payDownDeficitAndNotifyWaiters, :351-366
template <int TW>
__device__ __forceinline__ void ubm_pay_down(Deque &D, int32_t weight, bool release)
{
    const int32_t reduction = weight < D.deficit ? weight : D.deficit;
    if (reduction != 0) {
        D.deficit -= reduction;
        weight -= reduction;
    }
    ubm_adjust_agg<TW>(D, release ? -weight : reduction);
}

// cacheRemaining, :340-342
__device__ __forceinline__ int32_t ubm_cache_remaining(const Deque &D)
{
    const int64_t r = D.cap - D.wsize;
    return r > INT32_MAX ? INT32_MAX : (int32_t)r;
}

template <int TW>
__device__ __forceinline__ int32_t apply_op(Deque &D, const mmp_cache_op &o, int64_t now)
{
    switch (o.op) {
    case MMP_COP_PUT_IF_ABSENT: {  // clhm :804-834, AddTask :590-611
        const int i = dq_find<TW>(D, o.key);
        if (i >= 0) {
            dq_touch<TW>(D, i, o.time, now);
            dq_reposition<TW>(D, i);
            dq_refresh_oldest(D);  // the reader drains its own read buffer: tryToDrainBuffers :458-469
            return 0;
        }
        D.wsize += o.arg;
        dq_insert<TW>(D, o.time == 0 ? now : o.time, o.arg, o.key);
        dq_evict<TW>(D, false);
        dq_refresh_oldest(D);
        return 1;
    }
    case MMP_COP_GET: {  // clhm :726-733, applyRead :503-521
        const int i = dq_find<TW>(D, o.key);
        if (i < 0) return 0;
        dq_touch<TW>(D, i, o.time, now);
        dq_reposition<TW>(D, i);
        dq_refresh_oldest(D);
        return 1;
    }
    case MMP_COP_UPDATE_WEIGHT: {  // clhm :902-985, UpdateTask :629-652; time -1 = quiet
        const int i = dq_find<TW>(D, o.key);
        if (i < 0) return 0;
        const int32_t diff = (int32_t)((uint32_t)o.arg - (uint32_t)D.wt[D.head + i]);
        dq_set_wt<TW>(D, i, o.arg);
        if (diff == 0) {
            if (o.time >= 0) {
                dq_touch<TW>(D, i, o.time, now);
                dq_reposition<TW>(D, i);
                dq_refresh_oldest(D);
            }
            return 1;
        }
        D.wsize += diff;
        if (o.time >= 0 && o.time != D.lu[D.head + i]) {
            dq_touch<TW>(D, i, o.time, now);
            dq_reposition<TW>(D, i);
        }
        dq_evict<TW>(D, false);
        dq_refresh_oldest(D);
        return 1;
    }
    case MMP_COP_REMOVE: {  // clhm :861-870, RemovalTask :614-627
        const int i = dq_find<TW>(D, o.key);
        if (i < 0) return 0;
        int64_t ts;
        int32_t w, k;
        dq_unlink<TW>(D, i, ts, w, k);
        D.wsize -= w < 0 ? -(int64_t)w : (int64_t)w;
        dq_refresh_oldest(D);
        return 1;
    }
    case MMP_COP_UBM_INSERT_NEW_ENTRY: {  // :130-145
        ubm_adjust_agg<TW>(D, -o.arg);
        const int i = dq_find<TW>(D, o.key);
        if (i >= 0) {
            dq_touch<TW>(D, i, o.time, now);
            dq_reposition<TW>(D, i);
            dq_refresh_oldest(D);
            ubm_adjust_agg<TW>(D, o.arg);
            return 0;
        }
        D.wsize += o.arg;
        dq_insert<TW>(D, o.time == 0 ? now : o.time, o.arg, o.key);
        D.occ += o.arg;
        dq_evict<TW>(D, true);
        dq_refresh_oldest(D);
        ubm_drain<TW>(D);
        return 1;
    }
    case MMP_COP_UBM_ADJUST_SPACE_REQUEST: {  // adjustNewEntrySpaceRequest, :152-166
        const int i = dq_find<TW>(D, o.key);
        if (i < 0) return 0;
        const int32_t nw = (int32_t)((uint32_t)D.wt[D.head + i] + (uint32_t)o.arg);
        D.occ += o.arg;
        ubm_adjust_agg<TW>(D, -o.arg);
        ubm_set_weight<TW>(D, o.key, nw);
        return 1;
    }
    case MMP_COP_UBM_SPACE_IS_READY: return ubm_space_ready(D, o.arg) ? 1 : 0;
    case MMP_COP_UBM_CLAIM_SPACE: {  // claimRequestedSpaceIfReady, :190-202
        if (!ubm_space_ready(D, o.arg)) return 0;
        ubm_adjust_agg<TW>(D, o.arg);
        return 1;
    }
    case MMP_COP_UBM_ADJUST_AFTER_LOAD: {  // adjustWeightAfterLoad, :224-246
        const int32_t delta = o.arg;
        if (delta == 0) return 1;
        if (delta > 0) {
            const int32_t deficit = (int32_t)((uint32_t)delta - (uint32_t)ubm_cache_remaining(D));
            if (deficit > 0) {
                ubm_adjust_agg<TW>(D, -deficit);
                D.deficit += deficit;
            }
        }
        D.occ += delta;
        const int i = dq_find<TW>(D, o.key);
        if (i >= 0) ubm_set_weight<TW>(D, o.key, (int32_t)((uint32_t)D.wt[D.head + i] + (uint32_t)delta));
        if (delta < 0) ubm_pay_down<TW>(D, -delta, false);
        return i >= 0 ? 1 : 0;
    }
    case MMP_COP_UBM_UNLOAD_COMPLETE: {  // :318-338
        if (o.flag) {
            ubm_pay_down<TW>(D, o.arg, true);
            return 1;
        }
        const int64_t cap = D.cap;
        ubm_adjust_agg<TW>(D, -o.arg);
        D.cap = cap - o.arg > 1 ? cap - o.arg : 1;
        dq_evict<TW>(D, true);  // setCapacity evicts and notifies under the lock, clhm :305-316 — and does NOT updateOldestTime()
        ubm_drain<TW>(D);
        return 0;
    }
    case MMP_COP_UBM_REMOVE_ENTRY: {  // removeEntry :281-298 + entryRemoved :311-316; result = weight or -1
        const int i = dq_find<TW>(D, o.key);
        if (i < 0) return -1;
        int64_t ts;
        int32_t w, k;
        dq_unlink<TW>(D, i, ts, w, k);
        D.wsize -= w < 0 ? -(int64_t)w : (int64_t)w;
        dq_refresh_oldest(D);
        D.occ -= w;
        ubm_adjust_agg<TW>(D, w);
        return w;
    }
    case MMP_COP_UBM_DISCARD_FAILED: {  // discardFailedEntry, :343-349
        D.occ -= o.arg;
        ubm_pay_down<TW>(D, o.arg, false);
        return 1;
    }
    case MMP_COP_UBM_INSERT_FAILED_PLACEHOLDER: {  // insertFailedPlaceholderEntry, :250-274
        const int32_t deficit = (int32_t)((uint32_t)o.arg - (uint32_t)ubm_cache_remaining(D));
        if (deficit > 0) ubm_adjust_agg<TW>(D, -deficit);
        const int i = dq_find<TW>(D, o.key);
        if (i >= 0) {
            dq_touch<TW>(D, i, o.time, now);
            dq_reposition<TW>(D, i);
            dq_refresh_oldest(D);
            if (deficit > 0) ubm_adjust_agg<TW>(D, deficit);
            return 0;
        }
        D.wsize += o.arg;
        dq_insert<TW>(D, o.time == 0 ? now : o.time, o.arg, o.key);
        dq_evict<TW>(D, true);
        dq_refresh_oldest(D);
        ubm_drain<TW>(D);
        D.occ += o.arg;
        if (deficit > 0) D.deficit += deficit;
        return 1;
    }
    default: return MMP_EINVAL;
    }
}

// One launch per team width: `ids` lists the caches this launch replays (the host sorts them by the deque slots they need:
// small caches to 8- or 16-lane teams, 8 or 4 caches per wavefront; caches above kTeam16Slots to a wavefront each).
constexpr int kTeam8Slots = 48, kTeam16Slots = 160;

template <int TW>
__global__ __launch_bounds__(64) void cache_replay_kernel(ReplayArgs A, const int32_t *__restrict__ ids, int32_t n_ids)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kTeams = 64 / TW;
    const int team = threadIdx.x / TW, which = blockIdx.x * kTeams + team;
    if (which >= n_ids) return;
    const int c = ids ? ids[which] : which;  // (null: every cache of the store, in order)
    const int lane = team_lane<TW>();
    Deque D;
    D.lu = reinterpret_cast<int64_t *>(smem) + (size_t)team * A.tile;  // columns of all teams side by side: 8-byte column first
    D.wt = reinterpret_cast<int32_t *>(reinterpret_cast<int64_t *>(smem) + (size_t)kTeams * A.tile) + (size_t)team * 4 * A.tile;
    D.key = D.wt + A.tile;
    D.stk_key = D.key + A.tile;
    D.stk_wt = D.stk_key + A.tile;
    D.head = 0;
    D.sp = 0;
    D.n = A.src.n[c];
    const int so = A.src.off[c], dn = A.dst.off[c];
    const int o0 = A.op_off[c], o1 = A.op_off[c + 1];
    if (o0 == o1) {  // untouched cache: just move it into the new layout
        for (int i = lane; i < D.n; i += TW) {
            A.dst.lu[dn + i] = A.src.lu[so + i];
            A.dst.wt[dn + i] = A.src.wt[so + i];
            A.dst.key[dn + i] = A.src.key[so + i];
        }
        if (lane == 0) A.dst.n[c] = D.n;
        return;
    }
    for (int i = lane; i < D.n; i += TW) {
        D.lu[i] = A.src.lu[so + i];
        D.wt[i] = A.src.wt[so + i];
        D.key[i] = A.src.key[so + i];
    }
    wave_sync();
    D.wsize = A.weighted_size[c];
    D.oldest = A.oldest[c];
    D.cap = A.capacity[c];
    const mmp_ubm_state u = A.ubm[c];
    D.reserved = u.reserved;
    D.tu = u.total_unloading;
    D.occ = u.total_occupancy;
    D.deficit = u.cache_deficit;
    D.ev = A.evicted + A.ev_off[c];
    D.nev = 0;
    mmp_cache_op o = A.ops[o0];
    int oi = A.op_order[o0];
    for (int q = o0; q < o1; q++) {
        // the next operation is fetched while this one runs on the LDS tile (a replay is a chain of dependent steps per cache:
        // what can be taken off the chain is the global loads)
        mmp_cache_op o_next = o;
        int oi_next = oi;
        if (q + 1 < o1) {
            o_next = A.ops[q + 1];
            oi_next = A.op_order[q + 1];
        }
        const int ev0 = D.nev;
        const int32_t res = apply_op<TW>(D, o, A.now);
        const int ubi = u.reserved >= 0 ? dq_find<TW>(D, MMP_UNLOADBUF_KEY_C) : -1;
        if (lane == 0) {
            mmp_cache_op_out r;
            r.result = res;
            r.n_evicted = D.nev - ev0;
            r.evicted_off = A.ev_off[c] + ev0;
            r.buffer_weight = ubi >= 0 ? D.wt[D.head + ubi] : 0;
            r.weighted_size = D.wsize;
            r.oldest_time = D.oldest;
            A.outs[oi] = r;
        }
        wave_sync();
        o = o_next;
        oi = oi_next;
    }
    for (int i = lane; i < D.n; i += TW) {
        A.dst.lu[dn + i] = D.lu[D.head + i];
        A.dst.wt[dn + i] = D.wt[D.head + i];
        A.dst.key[dn + i] = D.key[D.head + i];
    }
    if (lane == 0) {
        A.dst.n[c] = D.n;
        A.weighted_size[c] = D.wsize;
        A.oldest[c] = D.oldest;
        A.capacity[c] = D.cap;
        mmp_ubm_state v = u;
        v.total_unloading = D.tu;
        v.total_occupancy = D.occ;
        v.cache_deficit = D.deficit;
        A.ubm[c] = v;
    }
}

}  // namespace mmp
