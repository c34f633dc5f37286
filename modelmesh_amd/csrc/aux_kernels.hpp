// aux_kernels.hpp — serve-target selection and eviction-victim kernels.
#pragma once
#include "snapshot.hpp"

namespace mmp {

#define MMP_ANY_TIME INT64_MIN  // excl_time wildcard: MapFilteringSet.keyExcludes (MM.java:4282)

struct ServeArgs {
    const mmp_serve_req *reqs;
    const mmp_model_row *models;
    const int32_t *ent_pod;
    const int64_t *ent_time;
    const mmp_serve_counter *counters;  // the requests' (instance, inUse, lastUsed) entries: the copies litelinks lists
    const int32_t *excl_pod;
    const int64_t *excl_time;
    mmp_serve_out *outs;
    int32_t n, n_models, P;
    int64_t now;
    DoneFlag done;  // latency path (wave.hpp); {nullptr} otherwise
};

// ForwardingLB.getNext (MM.java:4315-4392): k is tiny (1-3 copies), so one
// lane runs the loop exactly as written; lanes = independent requests.
// A launch lasts as long as its chain of dependent fetches, and the request names everything but the copies themselves:
// request -> { model row, the request's counters, its exclusions } -> copies.  The first kServePre copies / counters and
// kServeExcl exclusions are fetched level by level, everything of a level in flight together, before the loop runs on
// registers; what lies beyond is fetched in the loop.  The per-instance counters come WITH the request (one entry per copy
// that litelinks lists, mmp_serve_req): nothing is indexed by the instance table (round 2 took two P-sized arrays per call).
constexpr int kServePre = 4;
constexpr int kServeExcl = 4;

// one request (see serve_batch_kernel)
__device__ __forceinline__ mmp_serve_out serve_eval(const ServeArgs &A, const mmp_serve_req &r)
{
    mmp_serve_out o;
    o.chosen = MMP_NONE;
    o.pad = 0;
    o.chosen_load_start = 0;
    if (r.model >= 0 && r.model < A.n_models) {
        // level 2: the model row, the request's first counters and exclusions
        const mmp_model_row m = A.models[r.model];
        mmp_serve_counter c_pre[kServePre];
#pragma unroll
        for (int j = 0; j < kServePre; j++) {
            c_pre[j].pod = -1;
            c_pre[j].in_use = 0;
            c_pre[j].last_used = 0;
            if (j < r.n_cnt) c_pre[j] = A.counters[r.cnt_off + j];
        }
        int32_t x_pod[kServeExcl];
        int64_t x_time[kServeExcl];
#pragma unroll
        for (int x = 0; x < kServeExcl; x++) {
            x_pod[x] = x < r.n_excl ? A.excl_pod[r.excl_off + x] : -1;
            x_time[x] = x < r.n_excl ? A.excl_time[r.excl_off + x] : 0;
        }
        // level 3: the first copies
        int32_t p_iid[kServePre];
        int64_t p_ts[kServePre];
#pragma unroll
        for (int e = 0; e < kServePre; e++) {
            p_iid[e] = e < m.n_loaded ? A.ent_pod[m.ent_off + e] : -1;
            p_ts[e] = e < m.n_loaded ? A.ent_time[m.ent_off + e] : 0;
        }
        const bool exclude_self = r.flags & MMP_SERVE_EXCLUDE_SELF, prefer_self = r.flags & MMP_SERVE_PREFER_SELF;
        bool seen_self = false;
        int32_t chosen = -1;
        int64_t chosen_ts = 0;
        int32_t mn = INT32_MAX;
        int64_t lru = INT64_MAX, first_started = INT64_MAX;
        const int64_t cutoff = (int64_t)((uint64_t)A.now - (uint64_t)r.assume_completed_ms);  // :4350
        // one copy of the loop body
        auto visit = [&](int32_t iid, int64_t load_started) {
            // MapFilteringSet.apply (MM.java:4279-4283)
            bool filtered = false;
#pragma unroll
            for (int x = 0; x < kServeExcl; x++)
                if (x < r.n_excl && x_pod[x] == iid && (x_time[x] == MMP_ANY_TIME || x_time[x] == load_started)) filtered = true;
            for (int x = kServeExcl; x < r.n_excl; x++) {
                const int32_t xp = A.excl_pod[r.excl_off + x];
                const int64_t xt = A.excl_time[r.excl_off + x];
                if (xp == iid && (xt == MMP_ANY_TIME || xt == load_started)) filtered = true;
            }
            if (filtered) return;
            bool us = false;
            if (!seen_self && iid == r.self_pod) {  // :4334-4342
                seen_self = true;
                if (exclude_self) return;
                us = true;
            }
            // siMap.get(iid), :4343: the request's counter entry of this instance
            bool listed = false;
            int32_t pod_in_use = 0;
            int64_t pod_last_used = 0;
#pragma unroll
            for (int j = 0; j < kServePre; j++)
                if (j < r.n_cnt && c_pre[j].pod == iid && !listed) {
                    listed = true;
                    pod_in_use = c_pre[j].in_use;
                    pod_last_used = c_pre[j].last_used;
                }
            for (int j = kServePre; j < r.n_cnt && !listed; j++) {
                const mmp_serve_counter cj = A.counters[r.cnt_off + j];
                if (cj.pod == iid) {
                    listed = true;
                    pod_in_use = cj.in_use;
                    pod_last_used = cj.last_used;
                }
            }
            if (!listed || iid < 0) return;  // sii == null: litelinks does not list the instance
            if (load_started < cutoff) {  // :4352-4367
                const int32_t inuse = us ? r.local_in_flight : pod_in_use;
                if (inuse > mn) return;
                const int64_t nlu = us ? (prefer_self ? 0 : r.last_invoke_time) : pod_last_used;
                if (inuse < mn)
                    mn = inuse;
                else if (nlu >= lru)
                    return;
                chosen = iid;
                chosen_ts = load_started;
                lru = nlu;
            } else if (mn == INT32_MAX && load_started < first_started) {  // :4369-4376
                chosen = iid;
                chosen_ts = load_started;
                first_started = load_started;
            }
        };
#pragma unroll
        for (int e = 0; e < kServePre; e++)
            if (e < m.n_loaded) visit(p_iid[e], p_ts[e]);
        for (int e = kServePre; e < m.n_loaded; e++) visit(A.ent_pod[m.ent_off + e], A.ent_time[m.ent_off + e]);
        if (chosen >= 0) {
            o.chosen = (!exclude_self && chosen == r.self_pod) ? MMP_SELF : chosen;  // :4381-4385
            o.chosen_load_start = chosen_ts;
        }
    }
    return o;
}

__global__ void serve_batch_kernel(ServeArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) {
        const mmp_serve_req r = A.reqs[i];
        A.outs[i] = serve_eval(A, r);
    }
    announce_done(A.done);
}

struct EvictArgs {
    const mmp_evict_req *reqs;
    const int32_t *seg_off;   // [n_caches+1]
    const int64_t *last_used;  // deque order, oldest first
    const int32_t *weight;
    const int64_t *capacity;   // [n_caches]
    mmp_evict_out *outs;
    int32_t n, n_caches;
    int64_t now;
    DoneFlag done;  // latency path (wave.hpp); {nullptr} otherwise
};

// AddTask.run → evictionDeque.insert → evict()
// (clhm/ConcurrentLinkedHashMap.java:590-611,329-352; clhm/LinkedDeque.java:259-288).
// A pod's deque holds ~2M/P entries (20 at C3), so a whole wavefront per evaluation leaves two thirds of
// the lanes idle (measured: 39 us per 100k evaluations, VALU-issue bound).  Several evaluations share a
// wavefront instead, TEAM lanes each: the team's slice of a ballot finds the insertion point, a TEAM-lane
// prefix sum of the merged weight sequence finds the victim count.  TEAM = 16 (four evaluations per wavefront)
// or 8 (eight): the host picks 8 while the mean deque is short (<= 24 entries: three trips of 8 instead of two
// of 16, for twice the evaluations per wavefront and a scan of three steps instead of four).
constexpr int kEvBlock = 256;

template <int TEAM>
__device__ __forceinline__ int64_t team_sum_i64(int64_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = TEAM / 2; o > 0; o >>= 1) v += (int64_t)shfl_u64((uint64_t)v, lane ^ o);
    return v;
}

template <int TEAM>
__global__ __launch_bounds__(kEvBlock) void evict_batch_kernel(EvictArgs A)
{
    constexpr int kEvTeam = TEAM;
    constexpr uint32_t kTeamMask = (1u << TEAM) - 1u;
    const int lane = lane_id();
    const int team = lane / kEvTeam, tl = lane & (kEvTeam - 1), tbase = team * kEvTeam;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / kEvTeam;
    const bool in_range = i < A.n;
    mmp_evict_req r;
    r.cache = -1;
    r.weight = 0;
    r.last_used = 0;
    if (in_range) r = A.reqs[i];
    const bool valid = in_range && r.cache >= 0 && r.cache < A.n_caches;
    int s0 = 0, E = 0;
    int64_t cap = 0;
    if (valid) {
        s0 = A.seg_off[r.cache];
        E = A.seg_off[r.cache + 1] - s0;
        cap = A.capacity[r.cache];
    }
    const int64_t *lu = A.last_used + s0;
    const int32_t *wt = A.weight + s0;
    const int64_t ts = r.last_used == 0 ? A.now : r.last_used;  // Node ctor / touch, clhm :1357-1360
    int emax = E;  // the four teams loop together
#pragma unroll
    for (int o = 32; o >= kEvTeam; o >>= 1) {
        const int t = __shfl_xor(emax, o, 64);
        emax = t > emax ? t : emax;
    }

    // insert(): walk from the tail to the first node with lastUsed <= ts, link after it
    int pos = 0;
    int64_t sum = 0;
    for (int base = 0; base < emax; base += kEvTeam) {
        const int j = base + tl;
        const bool le = j < E && lu[j] <= ts;
        const uint32_t tb = (uint32_t)(__ballot(le) >> tbase) & kTeamMask;
        if (tb) pos = base + (31 - __clz(tb)) + 1;
        sum += j < E ? (int64_t)wt[j] : 0;
    }
    sum = team_sum_i64<TEAM>(sum);
    const int64_t total = sum + (int64_t)r.weight;  // weightedSize + weight, clhm :603

    // evict(): poll the head while weightedSize > capacity
    const bool need = valid && total > cap;
    int n_victims = need ? E + 1 : 0;
    int64_t after = need ? 0 : total;
    int64_t carry = 0;
    bool done = !need;
    for (int base = 0; base <= emax; base += kEvTeam) {
        if (__ballot(!done) == 0) break;
        const int j = base + tl;
        int64_t w = 0;
        if (need && j <= E) w = j < pos ? (int64_t)wt[j] : (j == pos ? (int64_t)r.weight : (int64_t)wt[j - 1]);
        int64_t incl = w;  // inclusive scan of the team's 64-bit weights
#pragma unroll
        for (int o2 = 1; o2 < kEvTeam; o2 <<= 1) {
            const int64_t t = (int64_t)shfl_u64((uint64_t)incl, tl >= o2 ? lane - o2 : lane);
            if (tl >= o2) incl += t;
        }
        const int64_t left = total - (carry + incl);
        const uint32_t tb = (uint32_t)(__ballot(!done && j <= E && left <= cap) >> tbase) & kTeamMask;
        const int l = tb ? __ffs(tb) - 1 : 0;
        const int64_t left_at = (int64_t)shfl_u64((uint64_t)left, tbase + l);
        const int64_t last_incl = (int64_t)shfl_u64((uint64_t)incl, tbase + kEvTeam - 1);
        if (tb && !done) {
            n_victims = base + l + 1;
            after = left_at;
            done = true;
        }
        carry += last_incl;
    }
    if (in_range && tl == 0) {
        mmp_evict_out o;
        o.insert_pos = 0; o.n_victims = 0; o.self_evicted = 0; o.pad = 0; o.weighted_size = 0; o.oldest_time = -1;
        if (valid) {
            o.insert_pos = pos;
            o.n_victims = n_victims;
            o.self_evicted = n_victims > pos ? 1 : 0;
            o.weighted_size = after;
            // oldestTime(): head of what is left, clhm :1125-1133
            const int h = n_victims;  // merged index of the new head
            if (h <= E) o.oldest_time = h < pos ? lu[h] : (h == pos ? ts : lu[h - 1]);
        }
        A.outs[i] = o;
    }
    announce_done(A.done);
}

}  // namespace mmp
