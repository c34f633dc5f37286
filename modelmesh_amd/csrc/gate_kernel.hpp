// gate_kernel.hpp — the request-level guards invokeModel evaluates around instance selection
// (SURVEY.md §8 rows a10, a11, a14, a20).  Pure scalar arithmetic: one lane per request, every
// branch transcribed from the cited Java lines with Java integer semantics.
#pragma once
#include "aux_kernels.hpp"

namespace mmp {

struct GateArgs {
    const mmp_gate_req *reqs;
    const mmp_model_row *models;
    const int32_t *ent_pod;
    const int64_t *ent_time;
    const mmp_pod_row *pods;
    const uint64_t *allowed;        // [T][W] over pod index, as loaded by mmp_types_load
    const uint8_t *has_allowed;     // [T]
    const StatsAcc *stats;          // ClusterStats of the committed snapshot
    const StatsAcc *tstats;         // [T_rows] typeSetStats(type) (MM.java:1432-1439): the subset a type may be placed on
    int32_t T_rows;
    const int32_t *excl_pod;
    const int64_t *excl_time;
    const int32_t *explicit_pool;
    mmp_gate_out *outs;
    int32_t n, n_models, P, W, T;
    int64_t now, in_use_expiry, min_space, min_churn;
    DoneFlag done;  // latency path (wave.hpp); {nullptr} otherwise
};

__device__ __forceinline__ int64_t jabs64(int64_t a) { return a < 0 ? (int64_t)(0ull - (uint64_t)a) : a; }
__device__ __forceinline__ int32_t jabs32(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }

// loadingChange / loadChange, MM.java:5536-5550
__device__ __forceinline__ bool loading_change(int32_t cur_in_prog, int32_t cur_threads, int32_t in_prog)
{
    if (in_prog == cur_in_prog) return false;
    if ((in_prog == 0) != (cur_in_prog == 0)) return true;
    if ((in_prog <= cur_threads) != (cur_in_prog <= cur_threads)) return true;
    return jabs32(in_prog - cur_in_prog) >= 3;
}
__device__ __forceinline__ bool load_change(int32_t cur_rpm, int32_t rpm)
{
    const int32_t diff = jabs32(cur_rpm - rpm);
    return diff >= 100 || (cur_rpm == 0 ? rpm != 0 : (100 * diff) / cur_rpm > 10);
}

// A launch lasts as long as its chain of dependent fetches, so everything a level of that chain needs is requested
// together before anything is evaluated: (1) the request; (2) the model row, the caller's own instance row, the first
// exclusions and the explicit pool; (3) the first copies and failure records, the type's stats row, its allowed bit;
// (4) the copies' instance flags.  The rules then run on registers; entries beyond kGatePre are fetched where needed.
constexpr int kGatePre = 4;

__device__ __forceinline__ mmp_gate_out gate_eval(const GateArgs &A, const mmp_gate_req &r)
{
    uint32_t bits = 0;
    int32_t initial = 0;
    // ---- level 2
    const bool have_model = r.model >= 0 && r.model < A.n_models;
    mmp_model_row m{};
    if (have_model) m = A.models[r.model];
    const bool self_in = r.self_pod >= 0 && r.self_pod < A.P;
    mmp_pod_row cur{};  // the caller's published record (publishInstanceRecord below)
    if (self_in) cur = A.pods[r.self_pod];
    int32_t x_pod[kGatePre], g_pool[kGatePre];
    int64_t x_time[kGatePre];
#pragma unroll
    for (int x = 0; x < kGatePre; x++) {
        const bool hx = have_model && x < r.n_excl, hg = have_model && x < r.n_explicit;
        x_pod[x] = hx ? A.excl_pod[r.excl_off + x] : -1;
        x_time[x] = hx ? A.excl_time[r.excl_off + x] : 0;
        g_pool[x] = hg ? A.explicit_pool[r.explicit_off + x] : -1;
    }
    // ---- level 3
    // typeSetStats(mr.getType()) at MM.java:5169 (loadLocal) and :2918 (onEviction); cluster-wide without a record
    const StatsAcc *st = have_model ? &A.tstats[(m.type < 0 || m.type >= A.T_rows) ? 0 : m.type] : A.stats;
    const int64_t total_cap = (int64_t)st->total_capacity, total_free = (int64_t)st->total_free;
    const int32_t copy_count = st->model_copy_count, inst_count = st->instance_count;
    int32_t l_iid[kGatePre], f_iid[kGatePre];
    int64_t l_t[kGatePre], f_t[kGatePre];
    bool l_in_table[kGatePre];
    bool typed = false;
    if (have_model) {
#pragma unroll
        for (int e = 0; e < kGatePre; e++) {
            l_iid[e] = e < m.n_loaded ? A.ent_pod[m.ent_off + e] : -1;
            l_t[e] = e < m.n_loaded ? A.ent_time[m.ent_off + e] : 0;
            f_iid[e] = e < m.n_failed ? A.ent_pod[m.ent_off + m.n_loaded + e] : -1;
            f_t[e] = e < m.n_failed ? A.ent_time[m.ent_off + m.n_loaded + e] : 0;
        }
        typed = A.T > 0 && m.type >= 0 && m.type < A.T && A.has_allowed[m.type];
        // ---- level 4
#pragma unroll
        for (int e = 0; e < kGatePre; e++)
            l_in_table[e] = l_iid[e] >= 0 && l_iid[e] < A.P && !(A.pods[l_iid[e]].flags & MMP_POD_TOMBSTONE);
    }
    const bool self_allowed =
        typed && self_in && ((A.allowed[(size_t)m.type * A.W + (r.self_pod >> 6)] >> (r.self_pod & 63)) & 1ull);

    if (have_model) {
        auto filtered_out = [&](int32_t iid, int64_t t) {  // MapFilteringSet.apply (MM.java:4279-4283)
            bool filtered = false;
#pragma unroll
            for (int x = 0; x < kGatePre; x++)
                if (x < r.n_excl && x_pod[x] == iid && (x_time[x] == MMP_ANY_TIME || x_time[x] == t)) filtered = true;
            for (int x = kGatePre; x < r.n_excl; x++) {
                const int32_t xp = A.excl_pod[r.excl_off + x];
                const int64_t xt = A.excl_time[r.excl_off + x];
                if (xp == iid && (xt == MMP_ANY_TIME || xt == t)) filtered = true;
            }
            return filtered;
        };
        auto in_explicit_pool = [&](int32_t iid) {
            bool hit = false;
#pragma unroll
            for (int x = 0; x < kGatePre; x++)
                if (x < r.n_explicit && g_pool[x] == iid) hit = true;
            for (int x = kGatePre; x < r.n_explicit; x++)
                if (A.explicit_pool[r.explicit_off + x] == iid) hit = true;
            return hit;
        };
        // ---- goLocal, MM.java:3598-3626 over filteredInstances = copies minus MapFilteringSet excludes
        int32_t n_f = 0;
        bool has_local = false;
        int64_t local_loaded = 0, oldest = INT64_MAX;
        auto go_local_visit = [&](int32_t iid, int64_t t) {
            if (filtered_out(iid, t)) return;
            n_f++;
            if (t < oldest) oldest = t;
            if (iid == r.self_pod && !has_local) { has_local = true; local_loaded = t; }
        };
#pragma unroll
        for (int e = 0; e < kGatePre; e++)
            if (e < m.n_loaded) go_local_visit(l_iid[e], l_t[e]);
        for (int e = kGatePre; e < m.n_loaded; e++) go_local_visit(A.ent_pod[m.ent_off + e], A.ent_time[m.ent_off + e]);
        bool go_local = false;
        if (n_f > 0 && has_local) {
            go_local = n_f == 1;
            if (!go_local && (r.flags & MMP_GATE_FAVOUR_SELF_FOR_HITS) && (r.flags & MMP_GATE_HAVE_CACHE_ENTRY)) {
                if (r.flags & MMP_GATE_ENTRY_DONE)
                    go_local = true;
                else if (oldest == local_loaded || age_of(oldest, A.now) < 1500)
                    go_local = true;
            }
        }
        if (go_local) bits |= MMP_GATE_GO_LOCAL;

        // ---- checkLoadFailureCount, MM.java:4607-4627 (MAX_LOAD_FAILURES = 3)
        {
            int count = 0;
            const int64_t cutoff = jsub64(A.now, A.in_use_expiry);
#pragma unroll
            for (int e = 0; e < kGatePre; e++)
                if (e < m.n_failed && f_t[e] > cutoff) count++;
            for (int e = kGatePre; e < m.n_failed && count < 3; e++)
                if (A.ent_time[m.ent_off + m.n_loaded + e] > cutoff) count++;
            if (count >= 3) bits |= MMP_GATE_FAILURES_BREACHED;
        }
        // ---- checkLoadLocationCount, MM.java:4590-4604 (MAX_LOAD_LOCATIONS = 5)
        {
            int count = 0;
#pragma unroll
            for (int e = 0; e < kGatePre; e++)
                if (e < m.n_loaded && l_in_table[e] && !in_explicit_pool(l_iid[e])) count++;
            for (int e = kGatePre; e < m.n_loaded && count < 5; e++) {
                const int32_t iid = A.ent_pod[m.ent_off + e];
                const bool in_table = iid >= 0 && iid < A.P && !(A.pods[iid].flags & MMP_POD_TOMBSTONE);
                if (in_table && !in_explicit_pool(iid)) count++;
            }
            if (count >= 5) bits |= MMP_GATE_LOCATIONS_BREACHED;
        }
        // ---- throwIfLocalLoadNotAllowed, MM.java:4003-4017
        {
            bool local_filtered = in_explicit_pool(r.self_pod);
#pragma unroll
            for (int e = 0; e < kGatePre; e++) {
                if (e < m.n_loaded && l_iid[e] == r.self_pod) local_filtered = true;
                if (e < m.n_failed && f_iid[e] == r.self_pod) local_filtered = true;
            }
            for (int e = kGatePre; e < m.n_loaded; e++)
                if (A.ent_pod[m.ent_off + e] == r.self_pod) local_filtered = true;
            for (int e = kGatePre; e < m.n_failed; e++)
                if (A.ent_pod[m.ent_off + m.n_loaded + e] == r.self_pod) local_filtered = true;
            const bool blocked = typed && !self_allowed;
            if (local_filtered || blocked) bits |= MMP_GATE_LOCAL_NOT_ALLOWED;
        }
    }

    // ---- churn guard, MM.java:3870-3884
    if (A.min_churn > 0) {
        const int64_t remaining = jsub64(r.cache_capacity, r.cache_weighted_size);
        if (remaining < A.min_space) {
            const int64_t lru = r.cache_oldest_time;
            if (lru >= 0 && lru != INT64_MAX && age_of(lru, A.now) < A.min_churn) bits |= MMP_GATE_CHURN_REJECT;
        }
    }

    // ---- loadLocal size prediction + early reject, MM.java:5158-5197
    if (r.flags & MMP_GATE_HAVE_SIZE_HINT) {
        initial = r.size_hint;
    } else if (r.loading_count > r.weight_predict_cutoff) {
        if (copy_count >= 10) {
            // (int)(totalCapacity - totalFree) / copyCount — the cast binds first (quirk B#7)
            const int32_t narrowed = (int32_t)(uint32_t)(uint64_t)jsub64(total_cap, total_free);
            const int32_t q = narrowed / copy_count;
            initial = (int32_t)(0u - (1u + (uint32_t)q));
        }
    }
    if (initial == 0) initial = r.loader_predicted;
    {
        const int32_t abs_size = jabs32(initial);
        if (r.flags & MMP_GATE_WE_CREATED_ENTRY) {
            if ((int64_t)abs_size > r.cache_capacity ||
                (r.last_used_time > 0 && (int64_t)abs_size > jsub64(r.cache_capacity, r.cache_weighted_size) &&
                 r.last_used_time < r.cache_oldest_time))
                bits |= MMP_GATE_EARLY_REJECT;
        }
    }

    // ---- onEviction reload rule, MM.java:2886-2920
    if (!(r.flags & MMP_GATE_ENTRY_FAILED) && r.loaded_time >= 0 &&
        jsub64(A.now, r.loaded_time) > 2 * r.load_timeout_ms) {
        if (total_cap > 0 && inst_count > 1 && (20 * total_free) / total_cap >= 1) bits |= MMP_GATE_RELOAD_ELSEWHERE;
    }

    // ---- publishInstanceRecord hysteresis, MM.java:5397-5468
    {
        const int64_t FREQ = 40000, MINP = 2000;
        const bool pre = r.flags & MMP_GATE_PRE_SHUTDOWN, force = r.flags & MMP_GATE_PUBLISH_FORCE;
        const int64_t last_done = jsub64(A.now, r.last_published);
        bool publish = true;
        if (!pre && (last_done < MINP || (!force && last_done < FREQ - 1000))) {
            publish = false;
        } else {
            const bool old = last_done > FREQ * 4;
            const bool have_cur = self_in && !(cur.flags & MMP_POD_TOMBSTONE);
            if (have_cur) {
                const bool cur_sd = cur.flags & MMP_POD_SHUTTING_DOWN, sd = r.flags & MMP_GATE_FRESH_SHUTTING_DOWN;
                // (a host may pass runtimeCache.oldestTime() as it is: -1 = the cache is empty, read as Long.MAX_VALUE, :5423-5425)
                const int64_t cap = r.fresh_capacity, used = r.fresh_used, oldest = r.fresh_lru == -1 ? INT64_MAX : r.fresh_lru;
                const int32_t count = r.fresh_count;
                if (!old) {
                    const int64_t d_lru = jabs64(jsub64(cur.lru_time, oldest));
                    const int64_t d_cnt = jabs32(cur.count - count);
                    int64_t fr = jsub64(cap, used);
                    fr = fr > 0 ? fr : 0;
                    if (cur_sd == sd && jabs64(jsub64(cur.capacity, cap)) < cap / 50 && d_lru < 20000 &&
                        (cur.lru_time == INT64_MAX || d_lru < jsub64(A.now, cur.lru_time) / 16) && d_cnt < 10 &&
                        (cur.count == 0 ? count == 0 : (d_cnt * 100) / cur.count < 15) &&
                        (cur.used == 0 ? used == 0 : (jabs64(jsub64(cur.used, used)) * 100) / cur.used < 20) &&
                        (remaining_of(cur.capacity, cur.used) < A.min_space) == (fr < A.min_space) &&
                        cur.loading_threads == r.fresh_loading_threads &&
                        !loading_change(cur.loading_in_progress, cur.loading_threads, r.fresh_in_progress) &&
                        !load_change(cur.rpm, r.fresh_rpm))
                        publish = false;
                } else if (cur_sd == sd && cur.capacity == cap && cur.count == count && cur.lru_time == oldest &&
                           cur.used == used && cur.loading_threads == r.fresh_loading_threads &&
                           cur.loading_in_progress == r.fresh_in_progress && cur.rpm == r.fresh_rpm) {
                    publish = false;
                }
            } else if (r.flags & MMP_GATE_FRESH_SHUTTING_DOWN) {
                publish = false;  // :5432-5436: no record of ours in the table and shutting down -> return without creating one
            }
        }
        if (publish) bits |= MMP_GATE_SHOULD_PUBLISH;
    }

    mmp_gate_out o;
    o.bits = bits;
    o.initial_size = initial;
    return o;
}

__global__ void gate_batch_kernel(GateArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) {
        const mmp_gate_req r = A.reqs[i];
        A.outs[i] = gate_eval(A, r);
    }
    announce_done(A.done);
}

// ---- the cache-hit route of invokeModel in ONE launch: for a request whose model is loaded somewhere, the guards that precede
// the routing (goLocal MM.java:3599-3626, the failure / location caps :4590-4627, throwIfLocalLoadNotAllowed :4003-4042, the
// churn guard :3870-3884 ...: gate_eval) AND the serve target among its copies (ForwardingLB.getNext :4315-4392: serve_eval) —
// what invokeModel asks per request, one after the other.  Both requests are fetched first, both evaluations are straight-line
// code on registers behind that, and both rows are stored at the end: the two dependent chains (request -> model row -> copies
// -> instance rows | counters) overlap instead of following each other in two launches (9.5 + 16 us at 100k requests).
__global__ void route_batch_kernel(GateArgs G, ServeArgs S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) {
        const mmp_gate_req gr = G.reqs[i];
        const mmp_serve_req sr = S.reqs[i];
        const mmp_serve_out so = serve_eval(S, sr);
        const mmp_gate_out go = gate_eval(G, gr);
        S.outs[i] = so;
        G.outs[i] = go;
    }
    announce_done(G.done);
}

// ONE route (what invokeModel asks per request): both requests and the request's own counter rows ride in the kernel arguments —
// nothing is fetched from the slot's pinned memory over the fabric before the decision can start (the exclusion pools, empty on a
// first try, stay where they are) — one wavefront, lane 0 evaluates, one fence, the flag.
constexpr int kRouteInlineCnt = 4;
struct RouteInline {
    mmp_gate_req g;
    mmp_serve_req s;
    mmp_serve_counter cnt[kRouteInlineCnt];
};
__global__ __launch_bounds__(128) void route_single_kernel(GateArgs G, ServeArgs S, RouteInline R)
{
    // the two evaluations are two dependent-load chains: side by side on two wavefronts (lane 0 of each)
    __shared__ mmp_serve_counter s_cnt[kRouteInlineCnt];
    if (threadIdx.x == 0) {
        G.outs[0] = gate_eval(G, R.g);
    } else if (threadIdx.x >= 64) {
        const int t = threadIdx.x - 64;
        if (t < kRouteInlineCnt) s_cnt[t] = R.cnt[t];
        wave_sync();
        if (t == 0) {
            mmp_serve_req sr = R.s;
            sr.cnt_off = 0;
            S.counters = s_cnt;
            S.outs[0] = serve_eval(S, sr);
        }
    }
    if (G.done.flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(G.done.flag, G.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace mmp
