// mmplace.hip — libmmplace: C-ABI (include/mmplace.h) over the gfx950 kernels.
//
// Host side only marshals: it keeps a staging copy of the instance table so
// single-row events (MM.java:1455 handleInstanceTableChange) can be applied,
// uploads, launches, and hands results back.  All ranking, filtering,
// shortlisting and victim selection runs in the kernels; there is no CPU
// implementation of the path in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is bound at run time (dlopen), not at link time
#include <rocprim/device/device_radix_sort.hpp>
#include <chrono>
#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mmplace.h"
#include "aux_kernels.hpp"
#include "cache_kernels.hpp"
#include "gate_kernel.hpp"
#include "ingest_kernels.hpp"
#include "place_kernel.hpp"
#include "rebalance_kernels.hpp"
#include "shard_kernels.hpp"
#include "multi_kernel.hpp"
#include "rank_sample.hpp"
#include "snapshot.hpp"
#include "types_kernel.hpp"
#include "upgrade_tracker.hpp"

using namespace mmp;

namespace {

// mmp_last_error() returns the calling thread's own last message: the library is entered concurrently (latency
// slots) and the JNI veneer hands this pointer to ThrowNew, so it must not be a string another thread can rewrite.
thread_local std::string g_thread_err;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(bytes, 256);
        want = (want + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// device storage of one committed snapshot
struct SnapBufs {
    DevBuf pods, lru, rem, cnt, rpm, orig, pos_of, elig, elig_nors, pref, has_pref, fullw, ge, pc, ph, nz, heads, bslots, bpm, bwin, bsurv, bpcs;
    DevBuf sel, rk;  // Snap::sel / ::rk
    DevBuf amul;     // Snap::amul
    int32_t amul_w = 0;  // words it holds
    DevBuf memo, memo_cand, memo_rk;  // Snap::memo / ::memo_cand / ::memo_rk (place_kernel.hpp: TypeMemo)
    DevBuf lmemo;                     // Snap::lmemo (place_kernel.hpp: LongMemo)
    DevBuf ctpos;  // Snap::ctpos
    uint64_t types_gen = 0;  // has_pref holds the type table of this generation
    int32_t n_bslots = 0;  // case (b) slots this snapshot has (place_kernel.hpp: BSlot), read back at commit
    void release()
    {
        for (DevBuf *b : {&pods, &lru, &rem, &cnt, &rpm, &orig, &pos_of, &elig, &elig_nors, &pref, &has_pref, &fullw, &ge, &pc, &ph, &nz, &heads, &bslots, &bpm, &bwin, &bsurv, &bpcs, &ctpos, &sel, &rk, &amul, &memo, &memo_cand, &memo_rk, &lmemo})
            b->release();
    }
};

// What decisions read besides the Snap columns, one set per committed snapshot (double-buffered with SnapBufs, so
// that a commit builds the next set while decisions keep reading the published one): the type table's allowed
// rows, the cluster / partition / type-set stats (device + host mirrors), the instance partitions, and the
// registry view resolved against THIS snapshot's rank positions.
struct SnapSide {
    DevBuf d_allowed, d_has_allowed, stats_acc, d_pts, d_prohib, pstats, tstats, rmodels;
    DevBuf mtw;  // PlaceArgs::mtw: rmodels[i].type, an int per model (valid whenever rmodels is)
    uint64_t types_gen = 0;  // d_allowed / d_has_allowed hold the type table of this generation
    bool rmodels_ok = false;
    std::vector<int32_t> pts_of;        // pod -> partition (-1: not in the table)
    std::vector<uint64_t> pts_prohib;   // [n_pts][tw] prohibited type rows
    int32_t n_pts = 0, pts_tw = 1;
    std::vector<StatsAcc> pstats_h, tstats_h;  // host mirrors (pstats_h has n_pts + 1 entries: the last is EMPTY_STATS)
    void release()
    {
        for (DevBuf *b : {&d_allowed, &d_has_allowed, &stats_acc, &d_pts, &d_prohib, &pstats, &tstats, &rmodels, &mtw}) b->release();
    }
};

}  // namespace

// Low-latency slot for small host-pointer batches (the single ensureLoaded / invokeModel request):
// pinned, device-mapped request / result buffers and a stream of its own, so that one decision never
// queues behind a 100k-decision batch or a commit (SURVEY.md §8b "Threading").
// Every pinned buffer the device and the host hand data through WHILE A KERNEL RUNS (completion flags, result rows of
// the latency slots, the resident kernel's request slots) is allocated fine-grained and device-mapped explicitly:
// hipHostMallocDefault would leave the choice to the process environment (HIP_HOST_COHERENT).
constexpr unsigned int kPinnedFlags = hipHostMallocCoherent | hipHostMallocMapped;
constexpr size_t kSelMaxBytes = (size_t)32 << 20;  // Snap::sel / ::rk are built up to this size each (per snapshot side)
constexpr int kTailStaticLds = 512;  // what place_tail_body keeps in static LDS beside place_block's own
constexpr int kTailBlocks = 16;    // workgroups of a split batch's tail launch (measured: see place_kernel.hpp)
constexpr int kMaxMissBufs = 64;   // streams with split batches in flight that get a buffer of their own (more: those batches go unsplit)
constexpr int kFastSlots = 4;
constexpr int kFastN = 4096;       // decisions per fast call
constexpr int kFastExtra = 16384;  // extra-exclusion pool entries per fast call
struct FastSlot {
    std::mutex mu;
    hipStream_t stream = nullptr;
    mmp_place_req *reqs = nullptr;  // hipHostMalloc'ed: same pointer is valid on the device
    int32_t *extra = nullptr;
    mmp_place_out *outs = nullptr;
    uint32_t *done = nullptr;  // pinned: the kernel stores the call's sequence number here when its results are visible
    uint32_t *blocks = nullptr;  // device: finished-workgroup counter of the call in flight
    // sequence number of the slot's last launch: bumped by the owner (under f.mu and the shared state lock), read by
    // quiesce_decisions without either — hence atomic
    std::atomic<uint32_t> seq{0};
};

struct mmp_ctx {
    mmp_config cfg{};
    hipStream_t stream = nullptr;
    // lock order: batch_mu (owner of `stream`, the s_* / r_* scratch and the commit's INPUTS — the staged instance
    // table, type table, replica-set list, registry — for a whole call) -> mu (the published snapshot pointers +
    // host staging as readers see it; decision paths hold it only while they capture the pointers and enqueue,
    // loaders hold it for the whole call, a commit only for its final pointer swap)
    std::mutex batch_mu, err_mu, cs_mu;
    // the state lock: decision paths that only capture the published pointers and enqueue take it SHARED (so that the
    // submission threads of mmp_issue_threads launch concurrently), whoever changes what they capture takes it exclusive
    std::shared_mutex mu;
    // submission threads (mmp_issue_threads); null: launches are issued by the caller.  Read and swapped under pool_mu
    // (a leaf lock): a submitter keeps its own reference for the duration of its call, so mmp_issue_threads(ctx, 0 / n)
    // on another thread cannot free the rings underneath it
    std::shared_ptr<struct IssuePool> pool;
    std::mutex pool_mu;
    // the resident decision kernel for single requests (place_kernel.hpp: place_resident_kernel; mmp_resident)
    struct Resident {
        // written under launch_mu (running, generation) / the state lock (enabled) and READ without either by every single request on its
        // way in (the optimistic checks of mmp_place_batch / resident_place; resident_ensure looks again under the locks): atomics
        // (tools/asan_lib.sh, ThreadSanitizer: plain bools here were its first two reports)
        std::atomic<bool> enabled{false}, running{false};
        std::atomic<uint32_t> generation{0};  // of the kernel launched last (1, 2, ...)
        hipStream_t stream = nullptr;
        ResidentSlot *slots = nullptr;  // pinned, device-mapped: host -> device
        ResidentAnswer *answers = nullptr;  // pinned, device-mapped: device -> host
        ResidentCtl *ctl = nullptr;     // pinned
        std::mutex launch_mu;           // taken AFTER the state lock
        std::mutex slot_mu[kResidentSlots];
        uint32_t seq[kResidentSlots] = {};
        std::atomic<uint32_t> rr{0};
        long long idle_ticks = 5'000'000;  // 50 ms at the 100 MHz wall clock
        std::atomic<uint64_t> launches{0}, served{0}, punted{0};
        std::atomic<int> slow{0};  // answers in a row that took longer than 20 ms (self-check in resident_place)
        std::atomic<int> punt_streak{0}, skip{0};  // hand-backs in a row; single requests still to be sent to the launch path directly
    } res;
    // caller-owned streams that *_dev calls were enqueued on (leaf lock cs_mu): whoever rewrites state a decision
    // kernel reads waits for them as well as for the library's own streams (quiesce_decisions)
    std::vector<hipStream_t> caller_streams;
    std::string err;
    FastSlot fast[kFastSlots];
    std::atomic<uint32_t> fast_rr{0};
    // mmp_profile(): HIP events around the kernels (not the staging copies) of each host-pointer call
    bool prof = false, prof_armed = false;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    double last_kernel_ms = -1.0;
    int32_t force_wave = 0;  // MMP_FORCE_WAVE=1: every decision takes the wave-per-decision kernel (tests)
    size_t lds_limit = 64 * 1024;    // LDS a workgroup may use on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    // a registry event's in-place rewrite is enqueued under the state lock and finishes after it: kernels that read the registry
    // from other streams are ordered behind reg_event while reg_pending (order_after_registry)
    DevBuf rt_sreqs, rt_souts, rt_cnt;  // mmp_route_batch: the serve half's staging (the gate half uses s_reqs / s_outs / s_a / s_c / s_d)
    hipEvent_t reg_event = nullptr;
    std::atomic<bool> reg_pending{false};
    std::atomic<size_t> lds_granted{48 * 1024};  // dynamic LDS the place kernels may be launched with so far
    int32_t no_caseb = 0;    // MMP_NO_CASEB=1: case (b) decisions never use the whole-window tables (tests: the wave path decides them)
    int32_t cfg_plan_sorted = 0;  // MMP_PLAN_SORTED=1: mmp_proactive_plan takes its sorted (fallback) path on every input; 2: never (tests)
    int32_t cfg_plan_fused = 1;   // MMP_PLAN_FUSED=0: the plan as its eight dependent launches (comparison; the one-launch form is the default)
    int32_t plan_grid_max = 0;    // workgroups the one-launch plan may have: all of them resident at once
    int32_t no_long_lds = 0; // MMP_NO_LONG_LDS=1: the long path reads its per-type tables from global memory (tests, comparison)
    int32_t long_split_from = -1;  // MMP_LONG_SPLIT_FROM=n: requests from which a full-cluster batch is split (default kLongSplitFrom)
    int32_t no_long_memo = 0;  // MMP_NO_LONG_MEMO=1: commit records no walks of the long shortlists (place_kernel.hpp: LongMemo)
    int32_t no_memo = 0;     // MMP_NO_MEMO=1: batches do not use the per-type shortlists (place_kernel.hpp: TypeMemo)
    int32_t memo_from = -1;  // MMP_MEMO_FROM=n: decisions from which a batch takes the kernel with the shortlists in front (default kMemoFrom)
    bool long_dense_env = false;  // MMP_LONG_DENSE_FROM was given
    int32_t long_dense_from = kLongDenseFrom;  // MMP_LONG_DENSE_FROM=n: decisions from which a full-cluster batch takes the 4-wavefront instantiation with its tables staged in LDS
    int32_t no_heads = 0;    // MMP_NO_HEADS=1: decisions do not use the per-type head windows (tests: lane_decide_r alone)
    // the split form of a large batch (place_kernel.hpp: place_memo_kernel + place_tail_kernel)
    int32_t no_split = 0;    // MMP_NO_SPLIT=1: never (the one-launch kernels with the check in front instead)
    int32_t split_from = -1; // MMP_SPLIT_FROM=n: decisions from which a batch is split (default kSplitFrom / kSplitFromC)
    int32_t tail_blocks = kTailBlocks;  // MMP_TAIL_BLOCKS: workgroups of the tail launch
    int32_t memo_lds_min = 0;  // MMP_MEMO_LDS_MIN=bytes: dynamic LDS the first launch of a split batch asks for at least (an occupancy cap: see kMemoLdsMin)
    int32_t split_notail = 0;  // MMP_SPLIT_NOTAIL=1: the tail launch is left out — the batch's results are INCOMPLETE (timing the first launch alone)
    // per stream that has issued a split batch: the words its first launches leave for its tails (launches of one stream are ordered,
    // so one buffer per stream will do) and a pinned pair the tail reports to: {undecided, of how many}
    struct MissBuf {
        hipStream_t st = nullptr;
        int32_t *words = nullptr;  // place_kernel.hpp: rest_buffer_ints — counters + kRestLists lists
        size_t cap = 0;            // ints
        int32_t *report = nullptr;
    };
    std::vector<MissBuf> miss_bufs;   // guarded by miss_mu (a leaf lock)
    std::vector<void *> miss_retired; // outgrown word buffers: a launch in flight may still read them; freed with the context
    int32_t *miss_reports = nullptr;  // pinned, kMaxMissBufs pairs
    std::mutex miss_mu;
    std::atomic<bool> split_off{false};  // a tail reported more than 1/32 of its batch: batches go unsplit until the next commit / registry event
    std::atomic<int64_t> n_split{0};     // split batches issued (mmp_split_batches)

    // host staging (inputs of the next commit)
    std::vector<mmp_pod_row> pods;
    int32_t n_types = 0, types_w = 0;
    std::vector<uint64_t> allowed, prefer;
    std::vector<uint8_t> has_allowed, has_prefer;
    std::vector<int32_t> replaced_rs;
    UpgradeTracker upgrades;

    // committed snapshot (double-buffered; `cur` is what decisions read): the rank-ordered columns and bitmaps,
    // and the instance partitions / subset stats / resolved registry view that belong to them (snapshot.hpp)
    SnapBufs sb[2];
    SnapSide side[2];
    int cur = 0;
    bool committed = false;
    Snap snap{};
    mmp_stats stats{};

    // pod-axis shard mode (mmp_shard_configure): this context owns the rank positions of
    // `ssnap` only; n_shards == 0 means the ordinary single-device snapshot
    int32_t shard = 0, n_shards = 0;
    ShardSnap ssnap{};
    DevBuf rk_rows, rk_idx, rk_tmp;  // ranking by sorting (snapshot.hpp)
    int32_t long_mode = -1;  // MMP_LONG_MODE: -1 auto (the long-shortlist kernel for snapshots whose instances are nearly all full), 0 never, 1 always (tests)
    bool snap_long = false;  // the committed snapshot takes place_batch_long_kernel
    bool snap_full = false;  // ... because (nearly) all of its instances are full (not only because a type is sparse)
    // the rows written since the last commit (a commit of a few changed rows re-ranks by insertion: delta_scatter_kernel)
    uint64_t types_gen = 1;       // bumped by every load of the type table; device copies remember the one they hold
    uint64_t d_prefer_gen = 0;
    std::vector<int32_t> sig_of;    // per instance: id of its ProhibitedTypeSet among the distinct ones (valid while the type table stands)
    std::vector<uint64_t> sig_rows;
    bool sig_valid = false;
    std::vector<int32_t> dirty;
    bool dirty_all = true;          // the table was replaced / grew: the next commit ranks from scratch
    bool order_total = false;       // the published order came from a total order (sort-legal rows): unchanged rows keep their order
    std::vector<int32_t> h_order, h_pos;  // host mirror of the published order (position -> row, row -> position), fetched lazily
    bool h_order_valid = false;
    int32_t single_block = 0;       // MMP_SINGLE_BLOCK=1: a single decision runs the batch kernel's workgroup (place_single_kernel)
    int32_t no_delta = 0;           // MMP_NO_DELTA=1: every commit ranks from scratch (tests)
    int64_t n_delta_commits = 0;
    int32_t rank_mode = 0;  // MMP_RANK_MODE: 0 auto (rank by sampling from kRankSortMinPods pods), 1 all-pairs, 2 by sampling whenever legal, 3 merge sort whenever legal (tests)
    Snap sview{};  // the lane path's view of this shard's slice (place_shard_fast_kernel)
    DevBuf f_offs, f_idx, f_reqs, f_outs, f_scan_tmp;  // speculative form: the compacted rest of a batch
    // ... and what ONE batch in flight owns until its rest count has been read, twice (slot = batch parity): an asynchronous
    // batch is completed after the NEXT one has been enqueued, so the two must not share flags, counter, word or exchange words
    DevBuf f_flags[2];             // per decision: undecided by the single exchange
    DevBuf f_cnt[2];               // finished workgroups << 32 | flagged decisions of the finish launch in flight (zero between launches)
    uint64_t *f_done = nullptr;    // pinned, words [0] and [8]: seq << 32 | flagged decisions, stored by the finish kernel's last workgroup
    uint32_t f_slot = 0;
    uint32_t f_seq = 0;
    // mmp_shard_place_batch_async_dev: the batch whose finish kernel is enqueued and whose rest count has not been read yet
    struct ShardPending {
        bool active = false;
        const void *d_reqs = nullptr, *d_extra = nullptr;
        void *d_outs = nullptr;
        int32_t n = 0;
        int64_t now = 0;
        uint32_t seq = 0, slot = 0;
    } pend;
    int32_t last_n_rest = 0;
    bool rank_pending = false;

    // RCCL group of the pod-axis shards (mmp_shard_group_init): collectives run on c->stream, inside the boundary
    std::mutex group_mu;  // one group-level call at a time (taken before batch_mu / mu by the group entry points)
    ncclComm_t comm = nullptr;
    mmp_exchange_fn xfn = nullptr;  // a host-supplied transport instead of RCCL (mmp_shard_group_set_exchange)
    void *xuser = nullptr;
    bool group = false;
    int32_t g_rank = 0, g_world = 1;
    DevBuf g_rankbuf, g_xf[2], g_x[6];

    // commit scratch
    DevBuf rank, occupancy, flag, rs_list, rs_bad, d_prefer;

    // model registry view
    DevBuf models, ent_pod, ent_time;
    int32_t n_models = 0, n_entries = 0;  // n_entries = used part of the entry arena (mmp_models_upsert appends)
    std::vector<int32_t> m_cnt;           // host shadow: entries per model (for the arena's garbage accounting)
    int64_t ent_live = 0;                 // entries still referenced by a row
    DevBuf u_idx, u_rows, u_cnt, u_offs, u_tmp;
    std::vector<uint64_t> u_stamp;        // per model: (call generation, row index) of the last row naming it
    uint32_t u_gen = 0;
    // (the registry view resolved against a snapshot, place_kernel.hpp: ResolvedModel, lives in SnapSide)

    // eviction caches
    DevBuf c_seg, c_lu, c_wt, c_cap;
    int32_t n_caches = 0;
    int64_t cache_entries = 0;  // entries of all deques loaded by mmp_caches_load (picks the eviction kernel's team size)

    // stateful keyed caches (mmp_caches_load_keyed / mmp_cache_replay): two layouts, ping-pong
    struct KeyedStore {
        DevBuf off, lu, wt, key, n;
    } ks[2];
    int ks_cur = 0;
    int32_t k_caches = 0;
    std::vector<int32_t> k_n;  // host mirror of the live entry counts
    DevBuf r_part;  // per-workgroup partials of the proactive plan's first pass
    DevBuf rs_split, rs_int;  // rank_sample.hpp: the sorted samples; range_of[P] | idx[P] | hist | cur | off
    DevBuf k_cap, k_wsize, k_oldest, k_ubm, k_ops, k_order, k_opoff, k_outs, k_ev, k_evoff, k_ids;  // k_ids: cache ids grouped by replay team width

    // wire-format ingestion: per-pod id attributes and the hash tables the parsers probe
    std::vector<uint32_t> id_order_v;
    std::vector<int32_t> replica_set_v;
    DevBuf idtab_hash, idtab_val, tytab_hash, tytab_val, j_buf, j_off, j_rows, j_aux, j_status, j_cnt, j_offs, j_tmp_pod, j_tmp_time,
        j_scan_tmp;
    uint32_t idtab_mask = 0, tytab_mask = 0;
    bool have_ids = false, have_types = false;
    int32_t unknown_type = 0, default_type = 0;

    // per-call scratch for the host-pointer entry points
    DevBuf s_reqs, s_outs, s_extra, s_a, s_b, s_c, s_d;
    // rebalancer scratch
    DevBuf r_ps, r_counts, r_keys, r_vals, r_keys2, r_vals2, r_tmp, r_out_model, r_out_lu;
};

namespace {

int fail(mmp_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_thread_err = buf;
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        c->err = buf;  // kept for debuggers; readers go through the thread-local copy
    }
    return code;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), MMP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// the side state of the published snapshot (read with c->mu held, or by the owner of c->batch_mu: only a commit,
// which holds batch_mu, flips c->cur)
inline SnapSide &cur_side(mmp_ctx *c) { return c->side[c->cur]; }

// Blocking copy on the context's own (non-blocking) stream.  The library never touches the legacy null
// stream: one synchronous hipMemset there at context creation was measured to serialise, for the rest of the
// process, kernels that the host issues on separate streams (8-stream step time 4.0 -> 9.8 us).
// tests/test_abi.py::test_library_never_uses_the_null_stream keeps it that way.
hipError_t copy_sync(mmp_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
constexpr int kRankSortMinPods = 8192;
constexpr int kSampleMinPods = 1024;  // (the sample path needs P >= its 512 samples; MMP_RANK_MODE=3: the merge sort instead)

// Kernel-time bracket of a host-pointer entry point (owner of c->batch_mu): KT_BEGIN after the H2D
// copies are enqueued, KT_END after the last kernel, kt_collect() once the stream has been synchronised.
#define KT_BEGIN(c, st)                                             \
    do {                                                            \
        (c)->prof_armed = false;                                    \
        if ((c)->prof && hipEventRecord((c)->pe0, (st)) == hipSuccess) (c)->prof_armed = true; \
    } while (0)
#define KT_END(c, st)                                               \
    do {                                                            \
        if ((c)->prof_armed && hipEventRecord((c)->pe1, (st)) != hipSuccess) (c)->prof_armed = false; \
    } while (0)
inline void kt_collect(mmp_ctx *c)
{
    if (!c->prof) return;
    float ms = -1.f;
    if (c->prof_armed && hipEventElapsedTime(&ms, c->pe0, c->pe1) != hipSuccess) ms = -1.f;
    c->last_kernel_ms = ms;
    c->prof_armed = false;
}

// Called with c->mu held, before device state that decisions read is overwritten: every decision
// kernel was enqueued under c->mu, so once the streams are idle nothing reads the old state.
hipError_t copy_sync(mmp_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}

// The resident kernel holds the snapshot it was launched with: whoever publishes another one or rewrites what it reads
// stops it (called with the state lock held, or from mmp_destroy); the next single request launches a new one.
void resident_stop(mmp_ctx *c)
{
    auto &R = c->res;
    if (!R.slots) return;
    std::lock_guard<std::mutex> g(R.launch_mu);
    if (!R.running) return;
    __atomic_store_n(&R.ctl->stop, R.generation.load(std::memory_order_relaxed), __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(R.stream);
    R.running = false;
}

hipError_t slot_wait(FastSlot *f, uint32_t seq);
hipError_t quiesce_decisions(mmp_ctx *c)
{
    resident_stop(c);
    // A latency slot has a kernel in flight only while its completion word lags its sequence number (the kernel stores the word
    // last; the owner bumps the number under the state lock, which the caller of this function holds or which no launch can pass
    // any more): an idle slot costs a load here, not a hipStreamSynchronize (the slots' synchronisations were 50-100 us under the
    // exclusive lock).  The number is read ONCE per slot: the quiescer waits for the launches that existed when it came, not for
    // the ones a busy owner keeps adding behind them (sequence numbers wrap: compare by difference).
    for (FastSlot &f : c->fast) {
        if (!f.done) continue;
        const uint32_t snap = f.seq.load(std::memory_order_acquire);
        if ((int32_t)(__atomic_load_n(f.done, __ATOMIC_ACQUIRE) - snap) >= 0) continue;
        hipError_t e = slot_wait(&f, snap);
        if (e != hipSuccess) return e;
    }
    std::vector<hipStream_t> cs;
    {
        std::lock_guard<std::mutex> g(c->cs_mu);
        cs = c->caller_streams;
    }
    for (hipStream_t st : cs) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(c->stream);
}

// a *_dev call was enqueued on `st` (called with c->mu held)
void note_caller_stream(mmp_ctx *c, hipStream_t st)
{
    if (st == c->stream) return;
    std::lock_guard<std::mutex> g(c->cs_mu);
    for (hipStream_t k : c->caller_streams)
        if (k == st) return;
    c->caller_streams.push_back(st);
}

// Re-resolve every model's entry list against a snapshot's rank positions, into that snapshot's side state.
// For the PUBLISHED snapshot (after the model table changed): call with c->mu held and the decision streams idle.
// For the snapshot a commit is building: the side is not visible to decisions yet; the caller owns c->batch_mu,
// which keeps the model table still.
int rebuild_resolved(mmp_ctx *c, SnapSide &sd, const Snap &snap, bool committed, bool sync = true)
{
    sd.rmodels_ok = false;
    if (!committed || c->n_models <= 0) return MMP_OK;  // (shard contexts: `snap` carries the whole table's pos_of — resolved positions are GLOBAL)
    HIP_TRY(c, sd.rmodels.ensure((size_t)c->n_models * sizeof(ResolvedModel)));
    HIP_TRY(c, sd.mtw.ensure((size_t)c->n_models * 4));
    hipLaunchKernelGGL(resolve_models_kernel, dim3(div_up(c->n_models, 256)), dim3(256), 0, c->stream, snap,
                       c->models.as<mmp_model_row>(), c->ent_pod.as<int32_t>(), c->n_models,
                       sd.rmodels.as<ResolvedModel>(), sd.mtw.as<int32_t>());
    HIP_TRY(c, hipGetLastError());
    if (sync) HIP_TRY(c, hipStreamSynchronize(c->stream));
    sd.rmodels_ok = true;
    return MMP_OK;
}
int rebuild_resolved(mmp_ctx *c) { return rebuild_resolved(c, cur_side(c), c->snap, c->committed); }

// ---- latency slots (FastSlot): pinned, device-mapped buffers + a completion flag ---------------------
// Small host-pointer calls do not stage through hipMemcpy and do not hold the context's batch stream: the
// kernel reads its requests from and writes its results to pinned host memory on the slot's own
// high-priority stream and announces completion through the slot's pinned flag (wave.hpp: announce_done).
FastSlot *slot_acquire(mmp_ctx *c, std::unique_lock<std::mutex> &lock)
{
    const uint32_t first = c->fast_rr.fetch_add(1, std::memory_order_relaxed);
    for (int k = 0; k < kFastSlots; k++) {
        FastSlot &cand = c->fast[(first + k) % kFastSlots];
        std::unique_lock<std::mutex> t(cand.mu, std::try_to_lock);
        if (t.owns_lock()) {
            lock = std::move(t);
            return &cand;
        }
    }
    FastSlot *f = &c->fast[first % kFastSlots];
    lock = std::unique_lock<std::mutex>(f->mu);
    return f;
}

// Spin on the slot's flag until launch number `seq` has completed (the owner passes its own launch's number, a quiescer the
// number it read when it came); the ordinary stream synchronisation is the fallback when it is late.
hipError_t slot_wait(FastSlot *f, uint32_t seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        if ((int32_t)(__atomic_load_n(f->done, __ATOMIC_ACQUIRE) - seq) >= 0) return hipSuccess;
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
        __builtin_ia32_pause();
    }
    return hipStreamSynchronize(f->stream);
}

hipError_t order_after_registry(mmp_ctx *c, hipStream_t st);
#ifdef MMP_XP_EMPTYTAIL  // (experiment builds only, tools/r6: what the runtime charges for a second launch per call, whatever it does)
__global__ void xp_noop_kernel(int32_t *p) { if (p == nullptr) __builtin_trap(); }
#endif
// The buffer of a split batch on `st` (`ints`: rest_buffer_ints of its first launch) and the stream's report pair; false: none to be
// had (too many streams, no memory) — the batch goes unsplit.  Also reads what the stream's last tail reported.
bool miss_buffer(mmp_ctx *c, hipStream_t st, size_t ints, int32_t **words, int32_t **report)
{
    std::lock_guard<std::mutex> g(c->miss_mu);
    mmp_ctx::MissBuf *mb = nullptr;
    for (auto &b : c->miss_bufs)
        if (b.st == st) mb = &b;
    if (!mb) {
        if ((int)c->miss_bufs.size() >= kMaxMissBufs) return false;
        if (!c->miss_reports) {
            if (hipHostMalloc(reinterpret_cast<void **>(&c->miss_reports), (size_t)kMaxMissBufs * 2 * sizeof(int32_t), kPinnedFlags) != hipSuccess) {
                c->miss_reports = nullptr;
                return false;
            }
            memset(c->miss_reports, 0, (size_t)kMaxMissBufs * 2 * sizeof(int32_t));
        }
        mmp_ctx::MissBuf nb;
        nb.st = st;
        nb.report = c->miss_reports + 2 * c->miss_bufs.size();
        c->miss_bufs.push_back(nb);
        mb = &c->miss_bufs.back();
    }
    if (mb->cap < ints) {
        size_t want = std::max<size_t>(ints, 65536);
        want = std::max(want, mb->cap + mb->cap / 2);
        void *p = nullptr;
        if (hipMalloc(&p, want * sizeof(int32_t)) != hipSuccess) return false;
        // the counters start at zero (every tail leaves them so); ordered before the stream's first launch on this buffer
        if (hipMemsetAsync(p, 0, (size_t)kRestLists * kRestCntStride * sizeof(int32_t), st) != hipSuccess) {
            (void)hipFree(p);
            return false;
        }
        if (mb->words) c->miss_retired.push_back(mb->words);  // (a launch in flight may still read it)
        mb->words = static_cast<int32_t *>(p);
        mb->cap = want;
    }
    // the last tail of this stream that has finished: more than 1/32 of its batch undecided -> the records do not fit these batches
    const int32_t undecided = __atomic_load_n(&mb->report[0], __ATOMIC_RELAXED), of = __atomic_load_n(&mb->report[1], __ATOMIC_RELAXED);
    if (of > 0 && (int64_t)undecided * 32 > (int64_t)of) c->split_off.store(true, std::memory_order_relaxed);
    *words = mb->words;
    *report = mb->report;
    return true;
}
// a new snapshot / registry: split batches get another chance (the reports of the old one's tails are void)
void split_reset(mmp_ctx *c)
{
    std::lock_guard<std::mutex> g(c->miss_mu);
    if (c->miss_reports) memset(c->miss_reports, 0, (size_t)kMaxMissBufs * 2 * sizeof(int32_t));
    c->split_off.store(false, std::memory_order_relaxed);
}
// what only some callers of place_launch bring: the caller's side of the single-caller form (d_reqs are mmp_place_req_c rows then),
// the declared length of the exclusion pool (bounded calls; extra_bound = 1 + entries, 0 = not declared)
struct PlaceOpts {
    const mmp_place_caller *caller = nullptr;
    int32_t extra_bound = 0;
};
int place_launch(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs,
                 hipStream_t st, uint32_t *done_flag = nullptr, uint32_t done_seq = 0, const mmp_place_req *inline_req = nullptr,
                 uint32_t *done_blocks = nullptr, const PlaceSegs *segs = nullptr, int32_t seg_blocks = 0, const GateArgs *fused_gate = nullptr,
                 const mmp_gate_req *fused_greq = nullptr, const PlaceOpts *opts = nullptr)
{
    if (n == 0) return MMP_OK;
    PlaceArgs A{};
    A.reqs = static_cast<const mmp_place_req *>(d_reqs);
    A.models = c->models.as<mmp_model_row>();
    A.rmodels = cur_side(c).rmodels_ok ? cur_side(c).rmodels.as<ResolvedModel>() : nullptr;
    A.mtw = cur_side(c).rmodels_ok ? cur_side(c).mtw.as<int32_t>() : nullptr;
    A.wins = (c->no_heads || c->n_shards > 0) ? nullptr : c->sb[c->cur].heads.as<TypeWin>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.extra = static_cast<const int32_t *>(d_extra);
    A.outs = static_cast<mmp_place_out *>(d_outs);
    A.n = n;
    A.n_models = c->n_models;
    A.now = now;
    A.n_dev = nullptr;
    A.force_wave = c->force_wave;
    A.n_pods_all = c->snap.P;
    A.done_flag = done_flag;
    A.done_seq = done_seq;
    A.extra_bound = opts ? opts->extra_bound : 0;
    const mmp_place_caller *caller = opts ? opts->caller : nullptr;
    if (caller && (segs || inline_req || done_flag)) return fail(c, MMP_EINVAL, "the single-caller form takes the batch kernels only");
    const int wpad = (c->snap.W + 1) & ~1;
    // one dynamic region: the lane phase's windows + scratch, re-used by the wave path's tiles (place_block)
    size_t lds = std::max<size_t>((size_t)kPlaceWaves * 2 * wpad * sizeof(uint64_t), (size_t)place_lane_lds(c->snap.T));
    // the long kernel on a full cluster: the case (b) tables of the snapshot's preferring types (place_kernel.hpp: BSlot)
    A.long_first = (c->snap_long && c->snap_full && A.rmodels) ? 1 : 0;
    // the long path's per-type tables in LDS when they are small (C3: 20 KB) and the launch fills the chip: measured on the full
    // cluster, 800k decisions per launch 70.8 -> 67.1 us; at 100k (1.5 wavefronts per SIMD) 20.4 -> 21.0 us, hence the size condition
    // With the recorded long walks (place_kernel.hpp: LongMemo) hardly a request searches those tables, and the barrier-free
    // instantiation is the faster one at every size (C3 full cluster, 200k / 400k / 800k requests per launch: 16.0 / 25.6 / 40.3 us against
    // 18.5 / 28.9 / 43.5 us; profiles/r6/long_records.txt)
    const int32_t dense_from = (A.long_first && c->snap.lmemo && !c->long_dense_env) ? INT32_MAX : c->long_dense_from;
    if (A.long_first && !inline_req && !done_flag && !c->no_long_lds && n >= dense_from) {
        const size_t tb = long_tables_bytes(c->snap.T, c->snap.W);
        // (only where they fit beside the wave tile and the static part: a device with 64 KB of LDS per workgroup keeps reading
        // them from global memory instead of failing the launch)
        if (tb <= 24 * 1024 && ((lds + 15) & ~(size_t)15) + tb + kPlaceStaticLds <= c->lds_limit) {
            const size_t off = (lds + 15) & ~(size_t)15;
            A.long_first = 1 + (int32_t)off;
            lds = off + tb;
        }
    }
    if (c->snap_long && !inline_req && !done_flag && A.wins && !c->no_caseb) {
        const SnapBufs &B = c->sb[c->cur];
        if (B.n_bslots > 0) {
            A.bslots = B.bslots.as<BSlot>();
            A.n_bslots = std::min(B.n_bslots, kBSlots);
            A.bwin = B.bwin.as<BLaunch>();
            A.bsurv = B.bsurv.as<uint64_t>();
            A.bpcs = B.bpcs.as<int32_t>();
        }
    }
    // the wave path's tile (two bitmaps of the whole table per wavefront) next to the static LDS of place_block;
    // gfx950 gives a workgroup up to 160 KB (c->lds_limit is the device's answer), which admits ~120k instances
    if (lds + kPlaceStaticLds > c->lds_limit)
        return fail(c, MMP_EINVAL, "instance table too large for the LDS staging tile (%d pods: %zu + %d bytes of LDS, the device "
                    "gives a workgroup %zu)", c->snap.P, lds, kPlaceStaticLds, c->lds_limit);
    if (lds > c->lds_granted.load(std::memory_order_acquire)) {  // beyond the default dynamic-LDS grant: ask once, for every variant
        static std::mutex grant_mu;  // (decision paths hold the state lock shared)
        std::lock_guard<std::mutex> gg(grant_mu);
        const int want = (int)(c->lds_limit - kPlaceStaticLds);
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_long4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_flag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_single_lean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(miss_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_multi_m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_multi_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_multi_long4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_c_m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_long_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_batch_long4_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want - kTailStaticLds));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_tail_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want - kTailStaticLds));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_long_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want - kTailStaticLds));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(place_long_tail_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want - kTailStaticLds));
        c->lds_granted.store((size_t)want, std::memory_order_release);
    }
    // batches with the per-type shortlists in front (place_batch_m_kernel / place_batch_c_m_kernel): head windows and the resolved registry
    // view in place, not the full-cluster regime (its shortlists span the table), none of the diagnostic routes, a launch that fills the chip
    const bool use_memo = !inline_req && !done_flag && !c->snap_long && !c->no_memo && !c->force_wave && A.wins && A.rmodels && c->snap.memo &&
                          n >= (c->memo_from >= 0 ? c->memo_from : (caller ? kMemoFromC : kMemoFrom));
    // NOBAR kernels: a region per wavefront (place_wave_lds) instead of the shared one
    const size_t lds_nobar = (size_t)kPlaceWaves * place_wave_lds(wpad);
    HIP_TRY(c, order_after_registry(c, st));
    // the split form: the check alone in a launch of its own, the rest in a dense tail behind it (place_kernel.hpp: place_memo_kernel)
    if (use_memo && !segs && !c->no_split && !c->split_off.load(std::memory_order_relaxed) &&
        n >= (c->split_from >= 0 ? c->split_from : (caller ? kSplitFromC : kSplitFrom))) {
        const int grid = div_up(n, kPlaceBlock);
        const int n_words = grid * kPlaceWaves;  // wavefronts of the first launch
        const int cap = rest_list_cap(n_words);
        int32_t *words = nullptr;
        int32_t *report = nullptr;
        if (lds + kPlaceStaticLds + kTailStaticLds <= c->lds_limit && miss_buffer(c, st, rest_buffer_ints(n_words), &words, &report) &&
            !c->split_off.load(std::memory_order_relaxed)) {
            size_t lds_memo = (size_t)kPlaceWaves * memo_stage_bytes(c->snap.T);  // a copy of the types' records per wavefront
            lds_memo = std::max(lds_memo, (size_t)c->memo_lds_min);
            if (c->split_notail) words = nullptr;  // (diagnostics: the undecided requests are not even recorded)
            if (caller) {
                hipLaunchKernelGGL(place_memo_c_kernel, dim3(grid), dim3(kPlaceBlock), lds_memo, st, c->snap, A, words, cap, *caller);
                if (!c->split_notail)
                    hipLaunchKernelGGL(place_tail_c_kernel, dim3(c->tail_blocks), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, words, cap, report, *caller);
            } else {
                hipLaunchKernelGGL(place_memo_kernel, dim3(grid), dim3(kPlaceBlock), lds_memo, st, c->snap, A, words, cap);
#ifdef MMP_XP_EMPTYTAIL
                if (getenv("MMP_XP_EMPTYTAIL")) {
                    hipLaunchKernelGGL(xp_noop_kernel, dim3(atoi(getenv("MMP_XP_EMPTYTAIL"))), dim3(kPlaceBlock), 0, st, words);
                    return MMP_OK;
                }
#endif
                if (!c->split_notail)
                    hipLaunchKernelGGL(place_tail_kernel, dim3(c->tail_blocks), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, words, cap, report);
            }
            HIP_TRY(c, hipGetLastError());
            c->n_split.fetch_add(1, std::memory_order_relaxed);
            return MMP_OK;
        }
    }
    // ... and on a full cluster: the recorded long walks alone in the first launch (place_kernel.hpp: place_long_memo_kernel)
    if (A.long_first && c->snap.lmemo && !segs && !inline_req && !done_flag && !c->force_wave && !c->no_split &&
        !c->split_off.load(std::memory_order_relaxed) && n >= (c->long_split_from >= 0 ? c->long_split_from : kLongSplitFrom)) {
        const int grid = div_up(n, kPlaceBlock);
        const int n_words = grid * kPlaceWaves;
        const int cap = rest_list_cap(n_words);
        int32_t *words = nullptr;
        int32_t *report = nullptr;
        // (the tail reads the tables from global memory: its requests are a handful)
        const size_t lds_tail = std::max<size_t>((size_t)kPlaceWaves * 2 * wpad * sizeof(uint64_t), (size_t)place_lane_lds(c->snap.T));
        if (lds_tail + kPlaceStaticLds + kTailStaticLds <= c->lds_limit && miss_buffer(c, st, rest_buffer_ints(n_words), &words, &report) &&
            !c->split_off.load(std::memory_order_relaxed)) {
            PlaceArgs At = A;
            At.long_first = 1;
            if (c->split_notail) words = nullptr;
            if (caller) {
                hipLaunchKernelGGL(place_long_memo_c_kernel, dim3(grid), dim3(kPlaceBlock), 0, st, c->snap, A, words, cap, *caller);
                if (!c->split_notail)
                    hipLaunchKernelGGL(place_long_tail_c_kernel, dim3(c->tail_blocks), dim3(kPlaceBlock), lds_tail, st, c->snap, At, wpad, words, cap, report, *caller);
            } else {
                hipLaunchKernelGGL(place_long_memo_kernel, dim3(grid), dim3(kPlaceBlock), 0, st, c->snap, A, words, cap);
                if (!c->split_notail)
                    hipLaunchKernelGGL(place_long_tail_kernel, dim3(c->tail_blocks), dim3(kPlaceBlock), lds_tail, st, c->snap, At, wpad, words, cap, report);
            }
            HIP_TRY(c, hipGetLastError());
            c->n_split.fetch_add(1, std::memory_order_relaxed);
            return MMP_OK;
        }
    }
    if (segs) {  // several request arrays, one launch (multi_kernel.hpp); n = the decisions of all of them
        if (c->snap_long && n >= c->long_dense_from)
            hipLaunchKernelGGL(place_multi_long4_kernel, dim3(seg_blocks), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *segs);
        else if (c->snap_long)
            hipLaunchKernelGGL(place_multi_long_kernel, dim3(seg_blocks), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *segs);
        else if (use_memo)
            hipLaunchKernelGGL(place_multi_m_kernel, dim3(seg_blocks), dim3(kPlaceBlock), lds_nobar, st, c->snap, A, wpad, *segs);
        else
            hipLaunchKernelGGL(place_multi_kernel, dim3(seg_blocks), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *segs);
    } else if (inline_req && fused_gate)  // one cache-miss route: the guards beside the load target (multi_kernel.hpp)
        hipLaunchKernelGGL(miss_single_kernel, dim3(1), dim3(128), lds, st, c->snap, A, wpad, *inline_req, *fused_gate, *fused_greq);
    else if (inline_req && !c->single_block)
        hipLaunchKernelGGL(place_single_lean_kernel, dim3(1), dim3(64), lds, st, c->snap, A, wpad, *inline_req);
    else if (inline_req)
        hipLaunchKernelGGL(place_single_kernel, dim3(1), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *inline_req);
    else if (done_flag && n > kPlaceBlock)
        hipLaunchKernelGGL(place_batch_flag_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds, st, c->snap, A, wpad,
                           done_blocks);
    else if (caller && c->snap_long && n >= dense_from)
        hipLaunchKernelGGL(place_batch_long4_c_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *caller);
    else if (caller && c->snap_long)
        hipLaunchKernelGGL(place_batch_long_c_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds_nobar, st, c->snap, A, wpad, *caller);
    else if (use_memo && !caller)
        hipLaunchKernelGGL(place_batch_m_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds_nobar, st, c->snap, A, wpad);
    else if (use_memo)
        hipLaunchKernelGGL(place_batch_c_m_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds_nobar, st, c->snap, A, wpad, *caller);
    else if (caller)
        hipLaunchKernelGGL(place_batch_c_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds, st, c->snap, A, wpad, *caller);
    else if (c->snap_long && n >= dense_from)
        hipLaunchKernelGGL(place_batch_long4_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds, st, c->snap, A, wpad);
    else if (c->snap_long)
        hipLaunchKernelGGL(place_batch_long_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds_nobar, st, c->snap, A, wpad);
    else
        hipLaunchKernelGGL(place_batch_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), lds, st, c->snap, A, wpad);
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
}

}  // namespace

extern "C" {

int mmp_abi_version(void) { return MMP_ABI_VERSION; }

int64_t mmp_min_space_units(int32_t dflt, int32_t threads, int64_t cap_units, int have_unload)
{
    // MM.java:765-771, Java int arithmetic
    const int32_t mn = (int32_t)((uint32_t)dflt * ((have_unload || threads <= 1) ? 1u : 2u));
    const int32_t a = (int32_t)((uint32_t)dflt * (uint32_t)threads);
    const int32_t b = (int32_t)(cap_units / 20);
    const int32_t target = a < b ? a : b;
    return mn > target ? mn : target;
}

int mmp_create(const mmp_config *cfg, mmp_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, MMP_EINVAL, "mmp_create: null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, MMP_ENODEVICE, "no HIP device available (%s); libmmplace has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MMP_EINVAL, "device %d out of range (have %d)", cfg->device, ndev);
    if ((e = hipSetDevice(cfg->device)) != hipSuccess)
        return fail(nullptr, MMP_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    mmp_ctx *c = new (std::nothrow) mmp_ctx();
    if (!c) return fail(nullptr, MMP_ENOMEM, "out of memory");
    c->cfg = *cfg;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) == hipSuccess && v > 0) c->lds_limit = (size_t)v;
    }
    if (const char *fw = getenv("MMP_FORCE_WAVE")) c->force_wave = fw[0] == '1';
    if (const char *nh = getenv("MMP_NO_HEADS")) c->no_heads = nh[0] == '1';
    if (const char *nm = getenv("MMP_NO_MEMO")) c->no_memo = nm[0] == '1';
    if (const char *nm = getenv("MMP_NO_LONG_MEMO")) c->no_long_memo = nm[0] == '1';
    if (const char *lf = getenv("MMP_LONG_SPLIT_FROM")) c->long_split_from = atoi(lf);
    if (const char *mf = getenv("MMP_MEMO_FROM")) c->memo_from = atoi(mf);
    if (const char *ns = getenv("MMP_NO_SPLIT")) c->no_split = ns[0] == '1';
    if (const char *sf = getenv("MMP_SPLIT_FROM")) c->split_from = atoi(sf);
    if (const char *tb = getenv("MMP_TAIL_BLOCKS")) c->tail_blocks = std::max(1, std::min(atoi(tb), kRestLists));
    if (const char *nt = getenv("MMP_SPLIT_NOTAIL")) c->split_notail = nt[0] == '1';
    if (const char *ml = getenv("MMP_MEMO_LDS_MIN")) c->memo_lds_min = std::max(0, std::min(atoi(ml), 60 * 1024));
    if (const char *nl = getenv("MMP_NO_LONG_LDS")) c->no_long_lds = nl[0] == '1';
    if (const char *nb = getenv("MMP_NO_CASEB")) c->no_caseb = nb[0] == '1';
    const bool want_resident = getenv("MMP_RESIDENT") && getenv("MMP_RESIDENT")[0] == '1';
    if (const char *rm = getenv("MMP_RANK_MODE")) c->rank_mode = atoi(rm);
    if (const char *lm = getenv("MMP_LONG_MODE")) c->long_mode = atoi(lm);
    if (const char *ld = getenv("MMP_LONG_DENSE_FROM")) {
        c->long_dense_from = atoi(ld);
        c->long_dense_env = true;
    }
    if (const char *nd = getenv("MMP_NO_DELTA")) c->no_delta = nd[0] == '1';
    if (const char *sb = getenv("MMP_SINGLE_BLOCK")) c->single_block = sb[0] == '1';
    if (const char *sp = getenv("MMP_PLAN_SORTED")) c->cfg_plan_sorted = atoi(sp);
    if (const char *pf = getenv("MMP_PLAN_FUSED")) c->cfg_plan_fused = atoi(pf);
    {
        // the one-launch plan waits at grid-wide barriers: every workgroup has to be on the chip (one per compute unit at most);
        // it keeps the bucket offsets in 64 KB of dynamic LDS beside its 23 KB of static
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(proactive_plan_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kPlanFusedLds) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, proactive_plan_fused_kernel, kPlanFusedBlock, kPlanFusedLds) == hipSuccess &&
            per_cu > 0)
            c->plan_grid_max = std::min(cus, kPlanFusedGrid);
        (void)hipGetLastError();
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return fail(nullptr, MMP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if ((e = hipEventCreateWithFlags(&c->reg_event, hipEventDisableTiming)) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, MMP_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
    }
    for (FastSlot &f : c->fast) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi is the numerically lowest = highest priority
        hipError_t e1 = hipStreamCreateWithPriority(&f.stream, hipStreamNonBlocking, hi);
        hipError_t e2 = hipHostMalloc(reinterpret_cast<void **>(&f.reqs), kFastN * sizeof(mmp_place_req), kPinnedFlags);
        hipError_t e3 = hipHostMalloc(reinterpret_cast<void **>(&f.extra), kFastExtra * sizeof(int32_t), kPinnedFlags);
        hipError_t e4 = hipHostMalloc(reinterpret_cast<void **>(&f.outs), kFastN * sizeof(mmp_place_out), kPinnedFlags);
        if (e4 == hipSuccess) e4 = hipHostMalloc(reinterpret_cast<void **>(&f.done), 64, kPinnedFlags);
        if (e4 == hipSuccess) *f.done = 0;
        if (e4 == hipSuccess) e4 = hipMalloc(reinterpret_cast<void **>(&f.blocks), 64);
        if (e4 == hipSuccess) e4 = hipMemsetAsync(f.blocks, 0, 64, c->stream);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
            mmp_destroy(c);
            return fail(nullptr, MMP_EHIP, "fast-slot allocation failed");
        }
    }
    (void)hipStreamSynchronize(c->stream);
    *out = c;
    if (want_resident && mmp_resident(c, 1) != MMP_OK) {
        *out = nullptr;
        mmp_destroy(c);
        return fail(nullptr, MMP_EHIP, "the resident decision kernel could not be set up");
    }
    return MMP_OK;
}

namespace {
void group_comm_destroy(ncclComm_t comm);  // defined with the RCCL binding below
}
extern "C" int mmp_issue_threads(mmp_ctx *c, int32_t n);
extern "C" int mmp_resident(mmp_ctx *c, int enable);
namespace {
void resident_stop(mmp_ctx *c);
}

void mmp_destroy(mmp_ctx *c)
{
    if (!c) return;
    (void)mmp_issue_threads(c, 0);  // launches still in the rings are issued, the helpers join
    (void)hipSetDevice(c->cfg.device);
    resident_stop(c);
    if (c->res.stream) (void)hipStreamDestroy(c->res.stream);
    if (c->res.slots) (void)hipHostFree(c->res.slots);
    if (c->res.ctl) (void)hipHostFree(c->res.ctl);
    if (c->res.answers) (void)hipHostFree(c->res.answers);
    c->res.slots = nullptr;
    for (FastSlot &f : c->fast) {
        if (f.stream) {
            (void)hipStreamSynchronize(f.stream);
            (void)hipStreamDestroy(f.stream);
        }
        if (f.reqs) (void)hipHostFree(f.reqs);
        if (f.done) (void)hipHostFree(f.done);
        if (f.blocks) (void)hipFree(f.blocks);
        if (f.extra) (void)hipHostFree(f.extra);
        if (f.outs) (void)hipHostFree(f.outs);
    }

    if (c->stream) (void)hipStreamSynchronize(c->stream);  // (an asynchronous pod-axis batch may still be writing its count)
    if (c->f_done) (void)hipHostFree(c->f_done);
    if (c->comm) {
        group_comm_destroy(c->comm);
        c->comm = nullptr;
    }
    for (DevBuf *b : {&c->g_rankbuf, &c->g_xf[0], &c->g_xf[1], &c->g_x[0], &c->g_x[1], &c->g_x[2], &c->g_x[3], &c->g_x[4], &c->g_x[5]}) b->release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->reg_event) (void)hipEventDestroy(c->reg_event);
    if (c->pe0) (void)hipEventDestroy(c->pe0);
    if (c->pe1) (void)hipEventDestroy(c->pe1);
    c->sb[0].release();
    c->sb[1].release();
    c->side[0].release();
    c->side[1].release();
    // (caller-owned streams a split batch ran on are the caller's to drain before it destroys the context, as for every *_dev call)
    for (auto &mb : c->miss_bufs)
        if (mb.words) (void)hipFree(mb.words);
    for (void *p : c->miss_retired) (void)hipFree(p);
    if (c->miss_reports) (void)hipHostFree(c->miss_reports);
    for (DevBuf *b : {&c->rank, &c->occupancy, &c->flag, &c->rs_list, &c->rs_bad, &c->d_prefer,
                      &c->models, &c->ent_pod, &c->ent_time, &c->c_seg,
                      &c->c_lu, &c->c_wt, &c->c_cap, &c->s_reqs, &c->s_outs, &c->s_extra, &c->s_a, &c->s_b,
                      &c->s_c, &c->s_d, &c->r_ps, &c->r_counts, &c->r_keys, &c->r_vals, &c->r_keys2, &c->r_vals2,
                      &c->r_tmp, &c->r_part, &c->rs_split, &c->rs_int, &c->r_out_model, &c->r_out_lu, &c->rt_sreqs, &c->rt_souts, &c->rt_cnt, &c->k_ids, &c->k_cap, &c->k_wsize, &c->k_oldest, &c->k_ubm, &c->k_ops, &c->k_order,
                      &c->k_opoff, &c->k_outs, &c->k_ev, &c->k_evoff, &c->idtab_hash, &c->idtab_val, &c->tytab_hash,
                      &c->tytab_val, &c->j_buf, &c->j_off, &c->j_rows, &c->j_aux, &c->j_status, &c->j_cnt, &c->j_offs, &c->j_tmp_pod,
                      &c->j_tmp_time, &c->j_scan_tmp, &c->rk_rows, &c->rk_idx, &c->rk_tmp, &c->u_idx, &c->u_rows, &c->u_cnt, &c->u_offs, &c->u_tmp, &c->f_flags[0], &c->f_flags[1], &c->f_offs, &c->f_idx, &c->f_reqs, &c->f_outs, &c->f_scan_tmp, &c->f_cnt[0], &c->f_cnt[1],
                      &c->ks[0].off, &c->ks[0].lu, &c->ks[0].wt,
                      &c->ks[0].key, &c->ks[0].n, &c->ks[1].off, &c->ks[1].lu, &c->ks[1].wt, &c->ks[1].key, &c->ks[1].n})
        b->release();
    delete c;
}

const char *mmp_last_error(mmp_ctx *) { return g_thread_err.c_str(); }

int mmp_backend(mmp_ctx *c) { return c ? 1 : 0; }

int mmp_profile(mmp_ctx *c, int enable)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gb(c->batch_mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (enable && !c->pe0) {
        HIP_TRY(c, hipEventCreate(&c->pe0));
        HIP_TRY(c, hipEventCreate(&c->pe1));
    }
    c->prof = enable != 0;
    c->last_kernel_ms = -1.0;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_backend");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_backend", e.what());
}

double mmp_last_kernel_ms(mmp_ctx *c) { return c ? c->last_kernel_ms : -1.0; }

int mmp_sync(mmp_ctx *c)
try {
    if (!c) return MMP_EINVAL;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_sync");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_sync", e.what());
}

/* ---- staging of the instance table -------------------------------------- */

namespace {
// a row written since the last commit; a long list is as good as "everything" (a context that never commits must not grow it)
inline void note_dirty(mmp_ctx *c, int32_t k)
{
    if (c->dirty_all) return;
    if (c->dirty.size() >= (size_t)8 * kDeltaRows) {
        c->dirty_all = true;
        c->dirty.clear();
        return;
    }
    c->dirty.push_back(k);
}
}  // namespace

int mmp_pods_load(mmp_ctx *c, const mmp_pod_row *rows, int32_t n)
try {
    if (!c || n < 0 || (n > 0 && !rows)) return fail(c, MMP_EINVAL, "mmp_pods_load: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->pods.assign(rows, rows + n);
    c->dirty_all = true;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pods_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pods_load", e.what());
}

int mmp_pods_upsert(mmp_ctx *c, const int32_t *idx, const mmp_pod_row *rows, int32_t n)
try {
    if (!c || n < 0 || (n > 0 && (!rows || !idx))) return fail(c, MMP_EINVAL, "mmp_pods_upsert: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    {  // all of the call or none of it: a bad index must not leave the rows before it applied
        int32_t count = (int32_t)c->pods.size();
        for (int32_t i = 0; i < n; i++) {
            if (idx[i] < 0 || idx[i] > count) return fail(c, MMP_EINVAL, "mmp_pods_upsert: index %d out of range", idx[i]);
            if (idx[i] == count) count++;
        }
    }
    for (int32_t i = 0; i < n; i++) {
        const int32_t k = idx[i];
        if (k == (int32_t)c->pods.size()) {
            c->pods.push_back(rows[i]);
            c->dirty_all = true;
        } else {
            c->pods[k] = rows[i];
            note_dirty(c, k);
        }
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pods_upsert");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pods_upsert", e.what());
}

int mmp_pods_remove(mmp_ctx *c, const int32_t *idx, int32_t n)
try {
    if (!c || n < 0 || (n > 0 && !idx)) return fail(c, MMP_EINVAL, "mmp_pods_remove: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    for (int32_t i = 0; i < n; i++)  // (all of the call or none of it)
        if (idx[i] < 0 || idx[i] >= (int32_t)c->pods.size()) return fail(c, MMP_EINVAL, "mmp_pods_remove: index %d out of range", idx[i]);
    for (int32_t i = 0; i < n; i++) {
        const int32_t k = idx[i];
        c->pods[k].flags |= MMP_POD_TOMBSTONE;
        c->pods[k].flags &= ~MMP_POD_LIVE;
        note_dirty(c, k);
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pods_remove");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pods_remove", e.what());
}

int mmp_types_load(mmp_ctx *c, int32_t n_types, const uint64_t *allowed, const uint64_t *prefer,
                   const uint8_t *has_allowed, const uint8_t *has_prefer)
try {
    if (!c || n_types < 0) return fail(c, MMP_EINVAL, "mmp_types_load: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    const int32_t W = div_up((int)c->pods.size(), 64);
    c->n_types = n_types;
    c->types_w = W;
    c->sig_valid = false;
    c->types_gen++;
    const size_t words = (size_t)n_types * W;
    c->allowed.assign(words, 0);
    c->prefer.assign(words, 0);
    c->has_allowed.assign(std::max(n_types, 1), 0);
    c->has_prefer.assign(std::max(n_types, 1), 0);
    for (int32_t t = 0; t < n_types; t++) {
        if (has_allowed && has_allowed[t]) {
            if (!allowed) return fail(c, MMP_EINVAL, "mmp_types_load: allowed bitmap missing");
            c->has_allowed[t] = 1;
            std::copy(allowed + (size_t)t * W, allowed + (size_t)(t + 1) * W, c->allowed.begin() + (size_t)t * W);
        }
        if (has_prefer && has_prefer[t]) {
            if (!prefer) return fail(c, MMP_EINVAL, "mmp_types_load: prefer bitmap missing");
            c->has_prefer[t] = 1;
            std::copy(prefer + (size_t)t * W, prefer + (size_t)(t + 1) * W, c->prefer.begin() + (size_t)t * W);
        }
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_types_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_types_load", e.what());
}

int mmp_types_from_labels(mmp_ctx *c, int32_t n_types, const uint64_t *required, const uint64_t *preferred,
                          const uint64_t *pod_labels, uint64_t *allowed_out, uint64_t *prefer_out,
                          uint8_t *has_allowed_out, uint8_t *has_prefer_out)
try {
    if (!c || n_types < 0 || (n_types > 0 && (!required || !preferred)))
        return fail(c, MMP_EINVAL, "mmp_types_from_labels: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    const int32_t P = (int32_t)c->pods.size();
    if (P > 0 && !pod_labels) return fail(c, MMP_EINVAL, "mmp_types_from_labels: pod_labels is null");
    const int32_t W = div_up(P, 64), T = n_types, R = T + 1;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const size_t words = (size_t)std::max(R * W, 1);
    // scratch: pods, labels, required, preferred, allowed, cpref, prefer, score, has_*
    HIP_TRY(c, c->s_reqs.ensure((size_t)std::max(P, 1) * sizeof(mmp_pod_row)));
    HIP_TRY(c, c->s_a.ensure((size_t)std::max(P, 1) * 8));
    HIP_TRY(c, c->s_b.ensure((size_t)std::max(T, 1) * 16));
    HIP_TRY(c, c->s_c.ensure(words * 8 * 3));
    HIP_TRY(c, c->s_d.ensure((size_t)std::max(P, 1) * 4 + (size_t)R * 2));
    uint64_t *d_req = c->s_b.as<uint64_t>(), *d_prf = d_req + std::max(T, 1);
    uint64_t *d_al = c->s_c.as<uint64_t>(), *d_cp = d_al + words, *d_pf = d_cp + words;
    int32_t *d_score = c->s_d.as<int32_t>();
    uint8_t *d_ha = reinterpret_cast<uint8_t *>(d_score + std::max(P, 1)), *d_hp = d_ha + R;
    if (P) {
        HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, c->pods.data(), (size_t)P * sizeof(mmp_pod_row), hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->s_a.p, pod_labels, (size_t)P * 8, hipMemcpyHostToDevice, st));
    }
    if (T) {
        HIP_TRY(c, hipMemcpyAsync(d_req, required, (size_t)T * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(d_prf, preferred, (size_t)T * 8, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(c, hipMemsetAsync(d_al, 0, words * 8 * 3, st));
    if (W > 0) {
        hipLaunchKernelGGL(type_sets_kernel, dim3(div_up(W, 4)), dim3(256), 0, st, c->s_reqs.as<mmp_pod_row>(),
                           c->s_a.as<uint64_t>(), P, W, T, d_req, d_prf, d_al, d_cp, d_score);
    }
    hipLaunchKernelGGL(type_prefer_kernel, dim3(R), dim3(256), 0, st, P, W, T, d_req, d_al, d_cp, d_score, d_pf, d_ha, d_hp);
    HIP_TRY(c, hipGetLastError());
    // install as the type table of the next commit (same staging the host-bitmap entry point fills)
    c->n_types = R;
    c->types_w = W;
    c->sig_valid = false;
    c->types_gen++;
    c->allowed.assign((size_t)R * W, 0);
    c->prefer.assign((size_t)R * W, 0);
    c->has_allowed.assign(R, 0);
    c->has_prefer.assign(R, 0);
    if (W > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->allowed.data(), d_al, (size_t)R * W * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(c->prefer.data(), d_pf, (size_t)R * W * 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(c, hipMemcpyAsync(c->has_allowed.data(), d_ha, R, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(c->has_prefer.data(), d_hp, R, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (allowed_out && W > 0) memcpy(allowed_out, c->allowed.data(), (size_t)R * W * 8);
    if (prefer_out && W > 0) memcpy(prefer_out, c->prefer.data(), (size_t)R * W * 8);
    if (has_allowed_out) memcpy(has_allowed_out, c->has_allowed.data(), R);
    if (has_prefer_out) memcpy(has_prefer_out, c->has_prefer.data(), R);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_types_from_labels");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_types_from_labels", e.what());
}

int mmp_replaced_rs_load(mmp_ctx *c, const int32_t *rs, int32_t n)
try {
    if (!c || n < 0 || (n > 0 && !rs)) return fail(c, MMP_EINVAL, "mmp_replaced_rs_load: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->replaced_rs.assign(rs, rs + n);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_replaced_rs_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_replaced_rs_load", e.what());
}

static void publish_upgrades(mmp_ctx *c)
{
    c->replaced_rs.clear();
    for (auto &e : c->upgrades.replaced) c->replaced_rs.push_back(e.first);
}

int mmp_upgrade_instance_added(mmp_ctx *c, int64_t labels_key, int32_t rs, int64_t start_time, int64_t now)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gb(c->batch_mu);  // replaced_rs is an input of the commit
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->upgrades.instance_added(labels_key, rs, start_time, now);
    publish_upgrades(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_upgrade_instance_added");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_upgrade_instance_added", e.what());
}

int mmp_upgrade_instance_removed(mmp_ctx *c, int64_t labels_key, int32_t rs, int64_t now)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gb(c->batch_mu);  // replaced_rs is an input of the commit
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->upgrades.instance_removed(labels_key, rs, now);
    publish_upgrades(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_upgrade_instance_removed");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_upgrade_instance_removed", e.what());
}

int mmp_upgrade_housekeeping(mmp_ctx *c, int64_t now)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gb(c->batch_mu);  // replaced_rs is an input of the commit
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->upgrades.housekeeping(now);
    publish_upgrades(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_upgrade_housekeeping");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_upgrade_housekeeping", e.what());
}

int mmp_upgrade_replaced(mmp_ctx *c, int32_t *rs_out, int64_t *expiry_out, int32_t max, int32_t *n_out)
try {
    if (!c || !n_out || max < 0 || (max > 0 && (!rs_out || !expiry_out))) return fail(c, MMP_EINVAL, "mmp_upgrade_replaced: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    int32_t i = 0;
    for (auto &e : c->upgrades.replaced) {
        if (i < max) {
            rs_out[i] = e.first;
            expiry_out[i] = e.second;
        }
        i++;
    }
    *n_out = i;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_upgrade_replaced");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_upgrade_replaced", e.what());
}

int mmp_models_load(mmp_ctx *c, const mmp_model_row *rows, int32_t n_models, const int32_t *ent_pod,
                    const int64_t *ent_time, int32_t n_entries)
try {
    if (!c || n_models < 0 || n_entries < 0 || (n_models > 0 && !rows) || (n_entries > 0 && (!ent_pod || !ent_time)))
        return fail(c, MMP_EINVAL, "mmp_models_load: bad argument");
    for (int32_t i = 0; i < n_models; i++) {
        const mmp_model_row &m = rows[i];
        if (m.n_loaded < 0 || m.n_failed < 0 || m.ent_off < 0 ||
            (int64_t)m.ent_off + m.n_loaded + m.n_failed > (int64_t)n_entries)
            return fail(c, MMP_EINVAL, "mmp_models_load: model %d entry range out of bounds", i);
    }
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, quiesce_decisions(c));  // the table is overwritten in place
    c->side[0].rmodels_ok = c->side[1].rmodels_ok = false;
    HIP_TRY(c, c->models.ensure((size_t)std::max(n_models, 1) * sizeof(mmp_model_row)));
    HIP_TRY(c, c->ent_pod.ensure((size_t)std::max(n_entries, 1) * sizeof(int32_t)));
    HIP_TRY(c, c->ent_time.ensure((size_t)std::max(n_entries, 1) * sizeof(int64_t)));
    if (n_models) HIP_TRY(c, copy_sync(c, c->models.p, rows, (size_t)n_models * sizeof(mmp_model_row), hipMemcpyHostToDevice));
    if (n_entries) {
        HIP_TRY(c, copy_sync(c, c->ent_pod.p, ent_pod, (size_t)n_entries * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->ent_time.p, ent_time, (size_t)n_entries * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    c->n_models = n_models;
    c->n_entries = n_entries;
    c->m_cnt.resize(n_models);
    c->ent_live = 0;
    for (int32_t i = 0; i < n_models; i++) {
        c->m_cnt[i] = rows[i].n_loaded + rows[i].n_failed;
        c->ent_live += c->m_cnt[i];
    }
    split_reset(c);
    return rebuild_resolved(c);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_models_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_models_load", e.what());
}

namespace {
// grow a device buffer, keeping its first `used` bytes
// Called with c->mu held (shared) right before a kernel that reads the registry (models / resolved rows / entry arena) is
// enqueued on `st`: a registry event whose rewrite kernel is still running on the context's stream is waited for ON THE DEVICE.
hipError_t order_after_registry(mmp_ctx *c, hipStream_t st)
{
    if (!c->reg_pending.load(std::memory_order_acquire) || st == c->stream) return hipSuccess;
    return hipStreamWaitEvent(st, c->reg_event, 0);
}

// Growing a table decisions read, without holding them off for the copy: the new allocation is filled beside the old one
// (batch_mu keeps other writers away), the pointer is swapped under the state lock, and the old allocation is freed once the
// kernels that captured it have drained.  `old_out` receives the old allocation (released by the caller after quiesce).
int grow_cow(mmp_ctx *c, DevBuf &b, size_t used, size_t want, DevBuf &fresh)
{
    if (want <= b.cap) return MMP_OK;
    HIP_TRY(c, fresh.ensure(std::max(want, b.cap * 2)));
    if (used && b.p) HIP_TRY(c, hipMemcpyAsync(fresh.p, b.p, used, hipMemcpyDeviceToDevice, c->stream));
    return MMP_OK;
}

int grow_keep(mmp_ctx *c, DevBuf &b, size_t used, size_t want)
{
    if (want <= b.cap) return MMP_OK;
    DevBuf nb;
    HIP_TRY(c, nb.ensure(std::max(want, b.cap * 2)));
    if (used && b.p) {
        const hipError_t e = copy_sync(c, nb.p, b.p, used, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            nb.release();
            HIP_TRY(c, e);
        }
    }
    b.release();
    b = nb;
    return MMP_OK;
}

// squeeze the garbage out of the entry arena (rows keep their order, entries their order inside a row).  Copy-on-write: the rows
// with their new offsets and the squeezed arena are built BESIDE the published ones (called with batch_mu, without the state
// lock: decisions keep reading the old tables), swapped in under the state lock, and the old ones freed when the kernels that
// captured them have drained.  (In place under the lock this held every decision off for the whole rebuild: the 0.4-0.6 ms
// maximum of a single decision under churn in round 3.)
int compact_registry(mmp_ctx *c)
{
    const int32_t n = c->n_models;
    hipStream_t st = c->stream;
    HIP_TRY(c, c->u_cnt.ensure((size_t)(n + 1) * 4));
    HIP_TRY(c, c->u_offs.ensure((size_t)(n + 1) * 4));
    size_t tmp = 0;
    HIP_TRY(c, rocprim::exclusive_scan(nullptr, tmp, c->u_cnt.as<int32_t>(), c->u_offs.as<int32_t>(), (int32_t)0, (size_t)n + 1,
                                       rocprim::plus<int32_t>(), st));
    HIP_TRY(c, c->u_tmp.ensure(std::max<size_t>(tmp, 16)));
    DevBuf nm, np, nt;
    const size_t cap = (size_t)std::max<int64_t>(c->ent_live + c->ent_live / 2, 1024);
    auto drop = [&] { nm.release(); np.release(); nt.release(); };
    if (nm.ensure(std::max<size_t>(c->models.cap, 16)) != hipSuccess || np.ensure(cap * 4) != hipSuccess || nt.ensure(cap * 8) != hipSuccess) {
        drop();
        return fail(c, MMP_ENOMEM, "compact_registry: out of device memory");
    }
    hipError_t e = hipMemcpyAsync(nm.p, c->models.p, (size_t)n * sizeof(mmp_model_row), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(model_counts_kernel, dim3(div_up(n + 1, 256)), dim3(256), 0, st, c->models.as<mmp_model_row>(), n,
                       c->u_cnt.as<int32_t>());
    (void)rocprim::exclusive_scan(c->u_tmp.p, tmp, c->u_cnt.as<int32_t>(), c->u_offs.as<int32_t>(), (int32_t)0, (size_t)n + 1,
                                  rocprim::plus<int32_t>(), st);
    hipLaunchKernelGGL(move_entries_kernel, dim3(div_up(std::max(n, 1), 256)), dim3(256), 0, st, nm.as<mmp_model_row>(), n,
                       c->u_offs.as<int32_t>(), c->ent_pod.as<int32_t>(), c->ent_time.as<int64_t>(), np.as<int32_t>(),
                       nt.as<int64_t>());
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        drop();
        HIP_TRY(c, e);
    }
    DevBuf om, op, ot;
    {
        std::lock_guard<std::shared_mutex> g(c->mu);  // the swap: three pointers
        om = c->models;
        op = c->ent_pod;
        ot = c->ent_time;
        c->models = nm;
        c->ent_pod = np;
        c->ent_time = nt;
        c->n_entries = (int32_t)c->ent_live;
    }
    const hipError_t q = quiesce_decisions(c);  // kernels that captured the old tables
    om.release();
    op.release();
    ot.release();
    HIP_TRY(c, q);
    return MMP_OK;
}
}  // namespace

int mmp_models_upsert(mmp_ctx *c, const int32_t *idx, const mmp_model_row *rows, int32_t n, const int32_t *ent_pod,
                      const int64_t *ent_time, int32_t n_entries)
try {
    if (!c || n < 0 || n_entries < 0 || (n > 0 && (!idx || !rows)) || (n_entries > 0 && (!ent_pod || !ent_time)))
        return fail(c, MMP_EINVAL, "mmp_models_upsert: bad argument");
    if (n == 0) return MMP_OK;
    // batch_mu keeps the registry's host bookkeeping and device tables to this call (every other writer of them
    // takes it); decisions are held off (c->mu + idle decision streams) only while the rows are rewritten in place —
    // the entries of the changed records are appended to the arena beyond anything a published row refers to, and
    // the new rows wait in scratch, before that.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    int32_t count = c->n_models;
    for (int32_t i = 0; i < n; i++) {
        const mmp_model_row &m = rows[i];
        if (idx[i] < 0 || idx[i] > count) return fail(c, MMP_EINVAL, "mmp_models_upsert: row %d names model %d of %d", i, idx[i], count);
        if (idx[i] == count) count++;
        if (m.n_loaded < 0 || m.n_failed < 0 || m.ent_off < 0 || (int64_t)m.ent_off + m.n_loaded + m.n_failed > (int64_t)n_entries)
            return fail(c, MMP_EINVAL, "mmp_models_upsert: row %d entry range out of bounds", i);
    }
    if ((int64_t)c->n_entries + n_entries > INT32_MAX) return fail(c, MMP_EINVAL, "mmp_models_upsert: entry arena overflow; reload the registry");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    // the last row wins when a model appears twice in one call (events are applied in order)
    std::vector<int32_t> last(n);
    {
        const uint64_t gen = (uint64_t)(++c->u_gen) << 32;
        if (c->u_stamp.size() < (size_t)count) c->u_stamp.resize(count, 0);
        for (int32_t i = 0; i < n; i++) c->u_stamp[idx[i]] = gen | (uint32_t)i;
        int32_t k = 0;
        for (int32_t i = 0; i < n; i++)
            if (c->u_stamp[idx[i]] == (gen | (uint32_t)i)) last[k++] = i;
        last.resize(k);
    }
    const int32_t k = (int32_t)last.size();
    const int32_t base = c->n_entries;
    std::vector<int32_t> h_idx(k);
    std::vector<mmp_model_row> h_rows(k);
    c->m_cnt.resize(count, 0);
    for (int32_t j = 0; j < k; j++) {
        const int32_t i = last[j];
        h_idx[j] = idx[i];
        h_rows[j] = rows[i];
        h_rows[j].ent_off += base;
        c->ent_live += (int64_t)(rows[i].n_loaded + rows[i].n_failed) - c->m_cnt[idx[i]];
        c->m_cnt[idx[i]] = rows[i].n_loaded + rows[i].n_failed;
    }
    // A table that has to grow moves (the old allocation is freed): only then is the state lock needed this early.
    const bool grows = (size_t)count * sizeof(mmp_model_row) > c->models.cap || (size_t)(base + n_entries) * 4 > c->ent_pod.cap ||
                       (size_t)(base + n_entries) * 8 > c->ent_time.cap ||
                       (cur_side(c).rmodels_ok && ((size_t)count * sizeof(ResolvedModel) > cur_side(c).rmodels.cap || (size_t)count * 4 > cur_side(c).mtw.cap));
    if (grows) {  // copy-on-write (grow_cow): decisions are held off for the pointer swap only
        DevBuf fm, fp, ft, fr, fw;
        auto drop = [&] { fm.release(); fp.release(); ft.release(); fr.release(); fw.release(); };
        int rc = grow_cow(c, c->models, (size_t)c->n_models * sizeof(mmp_model_row), (size_t)count * sizeof(mmp_model_row), fm);
        if (rc == MMP_OK) rc = grow_cow(c, c->ent_pod, (size_t)base * 4, (size_t)(base + n_entries) * 4, fp);
        if (rc == MMP_OK) rc = grow_cow(c, c->ent_time, (size_t)base * 8, (size_t)(base + n_entries) * 8, ft);
        // (the unpublished side's view is rebuilt from the model table by the next commit)
        if (rc == MMP_OK && cur_side(c).rmodels_ok)
            rc = grow_cow(c, cur_side(c).rmodels, (size_t)c->n_models * sizeof(ResolvedModel), (size_t)count * sizeof(ResolvedModel), fr);
        if (rc == MMP_OK && cur_side(c).rmodels_ok) rc = grow_cow(c, cur_side(c).mtw, (size_t)c->n_models * 4, (size_t)count * 4, fw);
        if (rc == MMP_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(c, MMP_EHIP, "mmp_models_upsert: growing the registry tables failed");
        if (rc != MMP_OK) {
            drop();
            return rc;
        }
        DevBuf olds[5];
        {
            std::lock_guard<std::shared_mutex> g(c->mu);
            if (fm.p) { olds[0] = c->models; c->models = fm; }
            if (fp.p) { olds[1] = c->ent_pod; c->ent_pod = fp; }
            if (ft.p) { olds[2] = c->ent_time; c->ent_time = ft; }
            if (fr.p) { olds[3] = cur_side(c).rmodels; cur_side(c).rmodels = fr; }
            if (fw.p) { olds[4] = cur_side(c).mtw; cur_side(c).mtw = fw; }
        }
        const hipError_t q = quiesce_decisions(c);  // kernels that captured the old allocations
        for (DevBuf &o : olds) o.release();
        HIP_TRY(c, q);
    }
    HIP_TRY(c, c->u_idx.ensure((size_t)k * 4));
    HIP_TRY(c, c->u_rows.ensure((size_t)k * sizeof(mmp_model_row)));
    if (n_entries) {
        HIP_TRY(c, hipMemcpyAsync(c->ent_pod.as<int32_t>() + base, ent_pod, (size_t)n_entries * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->ent_time.as<int64_t>() + base, ent_time, (size_t)n_entries * 8, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(c, hipMemcpyAsync(c->u_idx.p, h_idx.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->u_rows.p, h_rows.data(), (size_t)k * sizeof(mmp_model_row), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipStreamSynchronize(st));  // the pageable sources above are this call's stack / the caller's arrays
    {
        // Rows (and their resolved positions) are rewritten in place.  Under the state lock: the decisions in flight drain, the
        // rewrite kernel is ENQUEUED and an event recorded behind it; decisions launched from now on order themselves behind that
        // event on the device (order_after_registry).  The lock is not held while the kernel runs.
        std::lock_guard<std::shared_mutex> g(c->mu);
        HIP_TRY(c, quiesce_decisions(c));
        const bool resolved = cur_side(c).rmodels_ok && c->committed;
        KT_BEGIN(c, st);
        hipLaunchKernelGGL(upsert_models_kernel, dim3(div_up(k, 256)), dim3(256), 0, st, c->snap, c->u_idx.as<int32_t>(),
                           c->u_rows.as<mmp_model_row>(), k, c->ent_pod.as<int32_t>(), c->models.as<mmp_model_row>(),
                           resolved ? cur_side(c).rmodels.as<ResolvedModel>() : nullptr, resolved ? cur_side(c).mtw.as<int32_t>() : nullptr);
        KT_END(c, st);
        // From here on the rewrite may be running.  Nothing may leave this block with the lock released, the rewrite enqueued
        // and no way for later decisions to order themselves behind it: a launch or event error drains the stream first, and
        // the new counts are published only once the event stands.
        hipError_t le = hipGetLastError();
        if (le == hipSuccess) le = hipEventRecord(c->reg_event, st);
        if (le != hipSuccess) {
            (void)hipStreamSynchronize(st);
            HIP_TRY(c, le);
        }
        c->reg_pending.store(true, std::memory_order_release);
        c->n_models = count;
        c->n_entries = base + n_entries;
    }
    const hipError_t se = hipStreamSynchronize(st);
    if (se == hipSuccess) c->reg_pending.store(false, std::memory_order_release);  // (a failed wait leaves later launches ordered behind the event)
    HIP_TRY(c, se);
    kt_collect(c);
    // more garbage than live entries (and enough to matter): squeeze the arena
    if ((int64_t)c->n_entries - c->ent_live > std::max<int64_t>(c->ent_live, 1 << 16)) return compact_registry(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_models_upsert");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_models_upsert", e.what());
}

/* ---- commit: rank + permute + bitmaps + stats, all on the device --------- */

namespace {
// Partition the present instances by their ProhibitedTypeSet and build the per-partition / per-type
// ClusterStats of the snapshot being committed (snapshot.hpp "instance partitions").  Enqueues on st; the
// host mirrors are valid once the caller has synchronised st.  Call after cluster_stats_kernel.  N = the side
// state of the snapshot being built (its d_has_allowed / stats_acc are already filled).
// Rank the table from scratch by sampling (rank_sample.hpp): ranks of the rows [p_lo, p_hi) into `rank` (the others untouched).
// Only for a table on which the comparator is a strict total order (the caller's O(P) check).
int rank_by_sampling(mmp_ctx *c, const mmp_pod_row *d_pods, int32_t P, int64_t min_space, int64_t churn2, int32_t p_lo, int32_t p_hi,
                     int32_t *d_rank, hipStream_t st)
{
    const int32_t S = P >= 32768 ? 512 : 256;
    HIP_TRY(c, c->rs_split.ensure((size_t)2 * S * sizeof(RankRow)));
    HIP_TRY(c, c->rs_int.ensure(((size_t)2 * P + (2 * kCtrStride + 1) * (size_t)(S + 2)) * 4));
    RankRow *split = c->rs_split.as<RankRow>();
    int32_t *range_of = c->rs_int.as<int32_t>(), *idx = range_of + P, *hist = idx + P, *cur = hist + (S + 2) * kCtrStride,
            *off = cur + (S + 2) * kCtrStride;
    RankRow *srows = split + S;
    hipLaunchKernelGGL(sample_gather_kernel, dim3(div_up(S + 1, 256)), dim3(256), 0, st, d_pods, P, S, min_space, srows, hist, cur);
    hipLaunchKernelGGL(sample_sort_kernel, dim3(S), dim3(256), 0, st, srows, S, churn2, split);
    hipLaunchKernelGGL(sample_range_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, d_pods, P, S, min_space, churn2, split, range_of, hist);
    hipLaunchKernelGGL(sample_scatter_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, P, S, range_of, hist, cur, off, idx);
    hipLaunchKernelGGL(sample_rank_kernel, dim3(div_up(P, 16)), dim3(1024), 0, st, d_pods, P, min_space, churn2, range_of, off, idx, p_lo, p_hi,
                       d_rank);
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
}

struct ZeroBatch {  // collects buffers, clears them with one zero_regions_kernel launch
    ZeroList Z{};
    size_t most = 0;
    hipStream_t stream = nullptr;
    explicit ZeroBatch(hipStream_t st) : stream(st) {}
    void add(void *p, size_t bytes)
    {
        if (!p || bytes == 0) return;
        if (Z.n == kZeroRegions) launch(stream);  // (a full table goes out; the rest follows in another launch)
        Z.p[Z.n] = p;
        Z.bytes[Z.n] = (bytes + 3) & ~(size_t)3;  // (DevBuf allocations are rounded up: the tail bytes exist)
        most = std::max(most, bytes);
        Z.n++;
    }
    void launch(hipStream_t st)
    {
        if (Z.n == 0) return;
        const int blocks = (int)std::min<size_t>(std::max<size_t>(most / (256 * 4 * 4), 1), 256);
        hipLaunchKernelGGL(zero_regions_kernel, dim3(blocks), dim3(256), 0, st, Z);
        Z.n = 0;
        most = 0;
    }
};
// part 0: the host's interning of the partitions + the uploads; part 1: the kernels + the results back; 2: the results only; -1: all
int build_subset_stats(mmp_ctx *c, SnapSide &N, const mmp_pod_row *d_pods, int32_t P, int64_t min_space, hipStream_t st, int part = -1)
{
    const int32_t T = std::max(c->n_types, 1), Tw = div_up(T, 64);
    if (part <= 0) {
        N.pts_of.assign(P, -1);
        N.pts_prohib.clear();
        N.n_pts = 0;
        N.pts_tw = Tw;
        if (c->n_types > 0) {
            // every instance's ProhibitedTypeSet as an id of the distinct sets: a function of the type table alone, kept until
            // that table is reloaded (interning P signatures of T bits through a map was most of a small commit's host time)
            if (!c->sig_valid || (int32_t)c->sig_of.size() != P) {
                std::map<std::vector<uint64_t>, int32_t> intern;
                std::vector<uint64_t> sig(Tw);
                const int32_t Wf = c->types_w;
                c->sig_of.assign(P, 0);
                c->sig_rows.clear();
                for (int32_t p = 0; p < P; p++) {
                    std::fill(sig.begin(), sig.end(), 0);
                    for (int32_t t = 0; t < c->n_types; t++)
                        if (c->has_allowed[t] && !((c->allowed[(size_t)t * Wf + (p >> 6)] >> (p & 63)) & 1ull))
                            sig[t >> 6] |= 1ull << (t & 63);
                    auto it = intern.find(sig);
                    if (it == intern.end()) {
                        it = intern.emplace(sig, (int32_t)intern.size()).first;
                        c->sig_rows.insert(c->sig_rows.end(), sig.begin(), sig.end());
                    }
                    c->sig_of[p] = it->second;
                }
                c->sig_valid = true;
            }
            // partitions are numbered by first appearance among the instances PRESENT in this table
            std::vector<int32_t> number(c->sig_rows.size() / std::max(Tw, 1), -1);
            for (int32_t p = 0; p < P; p++) {
                if (c->pods[p].flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) continue;
                const int32_t sid = c->sig_of[p];
                if (number[sid] < 0) {
                    number[sid] = N.n_pts++;
                    N.pts_prohib.insert(N.pts_prohib.end(), c->sig_rows.begin() + (size_t)sid * Tw, c->sig_rows.begin() + (size_t)(sid + 1) * Tw);
                }
                N.pts_of[p] = number[sid];
            }
        }
        const int32_t NP = N.n_pts;
        N.pstats_h.assign(NP + 1, StatsAcc{});
        N.pstats_h[NP].global_lru = INT64_MAX;  // InstanceSetStatsTracker.EMPTY_STATS
        N.tstats_h.assign(T, StatsAcc{});
        HIP_TRY(c, N.d_pts.ensure(std::max<size_t>(P, 1) * 4));
        HIP_TRY(c, N.d_prohib.ensure(std::max<size_t>((size_t)NP * Tw, 1) * 8));
        HIP_TRY(c, N.pstats.ensure((size_t)(NP + 1) * sizeof(StatsAcc)));
        HIP_TRY(c, N.tstats.ensure((size_t)T * sizeof(StatsAcc)));
        if (P) HIP_TRY(c, hipMemcpyAsync(N.d_pts.p, N.pts_of.data(), (size_t)P * 4, hipMemcpyHostToDevice, st));
        if (NP) HIP_TRY(c, hipMemcpyAsync(N.d_prohib.p, N.pts_prohib.data(), (size_t)NP * Tw * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(N.pstats.p, N.pstats_h.data(), (size_t)(NP + 1) * sizeof(StatsAcc), hipMemcpyHostToDevice, st));
    }
    if (part != 0) {
        const int32_t NP = N.n_pts;
        if (part == 2) {
            // (the kernels ran inside the commit's level launches)
        } else if (NP > 0 && P > 0)
            hipLaunchKernelGGL(partition_stats_kernel, dim3(std::min(div_up(P, 256), 64)), dim3(256), 0, st, d_pods, P, min_space,
                               N.d_pts.as<int32_t>(), NP, N.pstats.as<StatsAcc>());
        if (part != 2)
            hipLaunchKernelGGL(subset_stats_finish_kernel, dim3(div_up(std::max(NP, T), 256)), dim3(256), 0, st,
                               N.stats_acc.as<StatsAcc>(), N.pstats.as<StatsAcc>(), NP, N.d_prohib.as<uint64_t>(), Tw, T,
                               c->n_types > 0 ? N.d_has_allowed.as<uint8_t>() : nullptr, N.tstats.as<StatsAcc>());
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(N.pstats_h.data(), N.pstats.p, (size_t)(NP + 1) * sizeof(StatsAcc), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(N.tstats_h.data(), N.tstats.p, (size_t)T * sizeof(StatsAcc), hipMemcpyDeviceToHost, st));
    }
    return MMP_OK;
}
}  // namespace

int mmp_snapshot_commit(mmp_ctx *c)
try {
    if (!c) return MMP_EINVAL;
    // Wait-free for decisions (SURVEY.md §8b "Threading"): the whole build runs with batch_mu only.  batch_mu keeps
    // the inputs still — every loader of the instance table, the type table, the registry and the replica-set list
    // takes it, and so does any other commit — and it owns c->stream and the commit scratch.  Everything a decision
    // reads is double-buffered (SnapBufs + SnapSide): the build fills the set that is NOT published, decisions keep
    // capturing the published one under c->mu, and c->mu is taken only for the pointer swap at the end.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (c->n_shards > 0)
        return fail(c, MMP_ESTATE, "context is a pod-axis shard: commit with mmp_shard_rank_dev + mmp_shard_commit_dev");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // The set about to be rewritten was published until the previous commit's swap; a kernel that still reads it
    // was enqueued before that swap.  Waiting for the decision streams (without c->mu: new decisions only ever
    // capture the published set) retires those.
    HIP_TRY(c, quiesce_decisions(c));
    const int32_t P = (int32_t)c->pods.size();
    const int32_t W = std::max(div_up(P, 64), 1);
    const int32_t T = std::max(c->n_types, 1);
    if (c->n_types > 0 && c->types_w != div_up(P, 64))
        return fail(c, MMP_ESTATE, "type bitmaps were loaded for a different pod count; reload them before commit");
    const size_t padded = (size_t)W * 64;
    SnapBufs &B = c->sb[1 - c->cur];
    SnapSide &N = c->side[1 - c->cur];
    hipStream_t st = c->stream;

    HIP_TRY(c, B.pods.ensure(std::max<size_t>(P, 1) * sizeof(mmp_pod_row)));
    HIP_TRY(c, B.lru.ensure(padded * 8));
    HIP_TRY(c, B.rem.ensure(padded * 8));
    HIP_TRY(c, B.cnt.ensure(padded * 4));
    HIP_TRY(c, B.rpm.ensure(padded * 4));
    HIP_TRY(c, B.orig.ensure(padded * 4));
    HIP_TRY(c, B.pos_of.ensure(padded * 4));
    HIP_TRY(c, B.elig.ensure((size_t)T * W * 8));
    HIP_TRY(c, B.elig_nors.ensure((size_t)T * W * 8));
    HIP_TRY(c, B.pref.ensure((size_t)T * W * 8));
    HIP_TRY(c, B.has_pref.ensure(T));
    HIP_TRY(c, B.fullw.ensure((size_t)W * 8));
    HIP_TRY(c, B.ge.ensure((size_t)kGeRows * W * 8));
    HIP_TRY(c, B.ctpos.ensure((size_t)(kGeRows + 1) * 4));
    HIP_TRY(c, B.pc.ensure((size_t)2 * T * (W + 1) * 4));
    HIP_TRY(c, B.nz.ensure((size_t)2 * T * (W + 1) * 4));
    HIP_TRY(c, B.ph.ensure((size_t)2 * T * (W + 1) * 8));
    // the long path's inverse tables (Snap::sel / ::rk): 512 bytes per type row and 64-position word, each, rebuilt at every commit —
    // 320 KB on C3, but a table of 100 type rows x 100 000 instances would be 80 MB a piece: beyond kSelMaxBytes they are not built
    // and a shortlist that spans the table takes the wave path
    const size_t sel_bytes = (size_t)2 * T * W * 64 * 4;
    const bool have_sel = sel_bytes <= kSelMaxBytes;
    if (have_sel) {
        HIP_TRY(c, B.sel.ensure(sel_bytes));
        HIP_TRY(c, B.rk.ensure(sel_bytes));
    }
    HIP_TRY(c, B.memo.ensure((size_t)kWinLds * sizeof(TypeMemo)));
    HIP_TRY(c, B.memo_cand.ensure((size_t)kWinLds * kMemoCand * 4));
    HIP_TRY(c, B.memo_rk.ensure((size_t)kWinLds * kMemoCand * 2));
    // rounded up to the 1 KB chunks place_block stages (rows beyond T are never read as windows)
    const size_t wins_bytes = (((size_t)std::max(T, kWinLds) * sizeof(TypeWin) + 1023) / 1024) * 1024;
    HIP_TRY(c, B.heads.ensure(wins_bytes));
    HIP_TRY(c, B.bslots.ensure(kBSlots * sizeof(BSlot) + 16));
    HIP_TRY(c, B.bpm.ensure((size_t)kBSlots * W * 64 * 4));
    HIP_TRY(c, B.bwin.ensure(kBSlots * sizeof(BLaunch)));
    HIP_TRY(c, B.bsurv.ensure((size_t)kBSlots * kBClasses * W * 8));
    HIP_TRY(c, B.bpcs.ensure((size_t)kBSlots * kBClasses * (W + 1) * 4));
    HIP_TRY(c, c->rank.ensure(padded * 4));
    HIP_TRY(c, c->occupancy.ensure(padded * 4));
    HIP_TRY(c, c->flag.ensure(sizeof(int32_t)));
    HIP_TRY(c, N.stats_acc.ensure(sizeof(StatsAcc)));
    HIP_TRY(c, c->rs_bad.ensure(padded));
    HIP_TRY(c, c->rs_list.ensure(std::max<size_t>(c->replaced_rs.size(), 1) * 4));
    HIP_TRY(c, N.d_allowed.ensure(std::max<size_t>((size_t)T * W, 1) * 8));
    HIP_TRY(c, c->d_prefer.ensure(std::max<size_t>((size_t)T * W, 1) * 8));
    HIP_TRY(c, N.d_has_allowed.ensure(T));

    const int64_t min_space = c->cfg.min_space_units;
    const int64_t churn2 = (int64_t)((uint64_t)c->cfg.min_churn_age_ms * 2u);
    // is the literal comparator a strict total order on these rows?  (see snapshot.hpp "ranking by sorting")
    bool versions_differ = false, full_low_lru = false, wide_count = false;
    int32_t n_present = 0, n_nonfull = 0;
    for (int32_t p = 0; p < P; p++) {
        const mmp_pod_row &r = c->pods[p];
        if (r.version != c->pods[0].version) versions_differ = true;
        if (r.count > (1 << 30) || r.count < -(1 << 30)) wide_count = true;  // :4676 is an int subtraction
        const uint64_t d = (uint64_t)r.capacity - (uint64_t)r.used;
        const int64_t rem = (int64_t)d > 0 ? (int64_t)d : 0;
        if (rem < min_space && r.lru_time <= churn2) full_low_lru = true;
        if (!(r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE))) {
            n_present++;
            if (!(rem < min_space)) n_nonfull++;
        }
    }
    const bool sort_legal = !(versions_differ && full_low_lru) && !wide_count;
    // A few rows changed since the published snapshot, and both orders are total: re-rank by insertion (snapshot.hpp
    // "a commit whose table differs ... in a few rows").  The unpublished set is rebuilt from the PUBLISHED one.
    DeltaRows dl{};
    std::vector<int32_t> new_order;
    bool delta = false;
    if (c->committed && !c->dirty_all && !c->no_delta && sort_legal && c->order_total && P >= 2 && P == c->snap.P) {
        std::vector<int32_t> chg(c->dirty);
        std::sort(chg.begin(), chg.end());
        chg.erase(std::unique(chg.begin(), chg.end()), chg.end());
        if ((int32_t)chg.size() <= kDeltaRows) {
            const SnapBufs &A = c->sb[c->cur];
            if (!c->h_order_valid) {  // the host's mirror of the published order, fetched when a delta first needs it
                c->h_order.resize(P);
                c->h_pos.resize(P);
                HIP_TRY(c, copy_sync(c, c->h_order.data(), A.orig.p, (size_t)P * 4, hipMemcpyDeviceToHost));
                for (int32_t q = 0; q < P; q++) c->h_pos[c->h_order[q]] = q;
                c->h_order_valid = true;
            }
            const int32_t K = (int32_t)chg.size();
            dl.K = K;
            std::vector<int32_t> rem_sorted(K);
            for (int32_t k = 0; k < K; k++) rem_sorted[k] = c->h_pos[chg[k]];
            std::sort(rem_sorted.begin(), rem_sorted.end());
            auto unchanged_at = [&](int32_t j) {  // the j-th row of the old order with the changed rows taken out
                int32_t pos = j;
                for (int32_t k = 0; k < K; k++)
                    if (rem_sorted[k] <= pos) pos++;
                return c->h_order[pos];
            };
            RankRow rr[kDeltaRows];
            for (int32_t k = 0; k < K; k++) rr[k] = make_rank_row(c->pods[chg[k]], min_space);
            // The order this path publishes must be STRICT, as the one from scratch is (scatter_pods_kernel reports two rows
            // of one rank as MMP_EORDER): a changed row that now ties with its neighbour or with another changed row — a
            // duplicate or unset id_order, every other field equal — sends the commit down the full path, which reports it.
            bool strict = true;
            for (int32_t k = 0; k < K; k++) {
                int32_t lo = 0, hi = P - K;
                while (lo < hi) {
                    const int32_t mid = (lo + hi) >> 1;
                    if (placement_less(make_rank_row(c->pods[unchanged_at(mid)], min_space), rr[k], churn2))
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                // (every unchanged row before lo is less than the row by the search; the one at lo must be greater, not equal)
                if (lo < P - K && !placement_less(rr[k], make_rank_row(c->pods[unchanged_at(lo)], min_space), churn2)) strict = false;
                int32_t before = 0;
                for (int32_t k2 = 0; k2 < K; k2++) {
                    if (k2 == k) continue;
                    const bool lt = placement_less(rr[k2], rr[k], churn2);
                    if (lt) before++;
                    else if (!placement_less(rr[k], rr[k2], churn2)) strict = false;  // neither before the other: a tie
                }
                dl.pod[k] = chg[k];
                dl.removed[k] = rem_sorted[k];
                dl.ins[k] = lo;
                dl.newrank[k] = lo + before;
                dl.rows[k] = c->pods[chg[k]];
            }
            // the host mirror of the NEW order (published with the snapshot below)
            if (strict) {
                new_order.assign(P, -1);
                std::vector<uint8_t> is_chg(P, 0);
                for (int32_t k = 0; k < K; k++) {
                    new_order[dl.newrank[k]] = chg[k];
                    is_chg[chg[k]] = 1;
                }
                int32_t w = 0;
                for (int32_t q = 0; q < P; q++) {
                    const int32_t pod = c->h_order[q];
                    if (is_chg[pod]) continue;
                    while (new_order[w] >= 0) w++;
                    new_order[w++] = pod;
                }
            }
            delta = strict;
        }
    }
    if (P && !delta) HIP_TRY(c, hipMemcpyAsync(B.pods.p, c->pods.data(), (size_t)P * sizeof(mmp_pod_row), hipMemcpyHostToDevice, st));
    {
        ZeroBatch zb(st);
        zb.add(c->rank.p, padded * 4);
        zb.add(c->occupancy.p, padded * 4);
        zb.add(c->flag.p, sizeof(int32_t));
        zb.add(B.bslots.p, kBSlots * sizeof(BSlot) + 16);
        zb.launch(st);
    }
    hipLaunchKernelGGL(zero_tails_kernel, dim3(1), dim3(64), 0, st, B.lru.as<int64_t>(), B.rem.as<int64_t>(), B.cnt.as<int32_t>(),
                       B.rpm.as<int32_t>(), B.orig.as<int32_t>(), B.pos_of.as<int32_t>(), P, (int32_t)padded);

    // the type table goes up only when it was reloaded since this buffer set / side / the staging copy last took it
    std::vector<uint8_t> hp(T, 0), ha(T, 0);
    for (int32_t t = 0; t < c->n_types; t++) {
        hp[t] = c->has_prefer[t];
        ha[t] = c->has_allowed[t];
    }
    if (B.types_gen != c->types_gen) HIP_TRY(c, hipMemcpyAsync(B.has_pref.p, hp.data(), T, hipMemcpyHostToDevice, st));
    if (N.types_gen != c->types_gen) HIP_TRY(c, hipMemcpyAsync(N.d_has_allowed.p, ha.data(), T, hipMemcpyHostToDevice, st));
    if (c->n_types > 0 && !c->allowed.empty()) {
        if (N.types_gen != c->types_gen)
            HIP_TRY(c, hipMemcpyAsync(N.d_allowed.p, c->allowed.data(), c->allowed.size() * 8, hipMemcpyHostToDevice, st));
        if (c->d_prefer_gen != c->types_gen)
            HIP_TRY(c, hipMemcpyAsync(c->d_prefer.p, c->prefer.data(), c->prefer.size() * 8, hipMemcpyHostToDevice, st));
    }
    const uint64_t types_gen_uploaded = c->types_gen;  // (marked current once the copies have landed: behind the synchronisation)
    const int32_t n_rs = (int32_t)c->replaced_rs.size();
    if (n_rs) HIP_TRY(c, hipMemcpyAsync(c->rs_list.p, c->replaced_rs.data(), (size_t)n_rs * 4, hipMemcpyHostToDevice, st));

    StatsAcc init{};
    init.global_lru = INT64_MAX;
    HIP_TRY(c, hipMemcpyAsync(N.stats_acc.p, &init, sizeof init, hipMemcpyHostToDevice, st));

    Snap S{};
    S.P = P;
    S.W = W;
    S.T = T;
    S.any_rs = c->replaced_rs.size() > 0;
    S.min_space = min_space;
    S.lru = B.lru.as<int64_t>();
    S.rem = B.rem.as<int64_t>();
    S.cnt = B.cnt.as<int32_t>();
    S.rpm = B.rpm.as<int32_t>();
    S.orig = B.orig.as<int32_t>();
    S.pos_of = B.pos_of.as<int32_t>();
    S.elig = B.elig.as<uint64_t>();
    S.elig_nors = B.elig_nors.as<uint64_t>();
    S.pref = B.pref.as<uint64_t>();
    S.has_pref = B.has_pref.as<uint8_t>();
    S.fullw = B.fullw.as<uint64_t>();
    S.ge = B.ge.as<uint64_t>();
    S.pc = B.pc.as<int32_t>();
    S.nz = B.nz.as<int32_t>();
    S.ph = B.ph.as<uint64_t>();
    S.ctpos = P > 0 ? B.ctpos.as<int32_t>() : nullptr;
    if (B.amul_w != W) {  // the audit hash's word multipliers: a function of the word index alone
        HIP_TRY(c, B.amul.ensure((size_t)W * 8));
        std::vector<uint64_t> am((size_t)W);
        for (int32_t w = 0; w < W; w++) am[(size_t)w] = audit_mul((uint64_t)w);
        HIP_TRY(c, hipMemcpyAsync(B.amul.p, am.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));  // (`am` leaves scope; once per table size)
        B.amul_w = W;
    }
    S.amul = B.amul.as<uint64_t>();
    S.sel = P > 0 && have_sel ? B.sel.as<int32_t>() : nullptr;
    S.rk = P > 0 && have_sel ? B.rk.as<int32_t>() : nullptr;
    S.memo = B.memo.as<TypeMemo>();
    S.memo_cand = B.memo_cand.as<int32_t>();
    S.memo_rk = B.memo_rk.as<int16_t>();
    S.lmemo = nullptr;
    if (P > 0 && have_sel && !c->no_long_memo) {  // the recorded walks of the long shortlists (place_kernel.hpp: LongMemo)
        HIP_TRY(c, B.lmemo.ensure((size_t)T * kLongLevels * sizeof(LongMemo)));
        S.lmemo = B.lmemo.as<LongMemo>();
    }

    bool next_long = c->long_mode == 1, next_full = false;
    {  // the partitions of the type constraints (host) and their uploads: inputs, like the table itself
        const int rc = build_subset_stats(c, N, B.pods.as<mmp_pod_row>(), P, min_space, st, 0);
        if (rc != MMP_OK) return rc;
    }
    KT_BEGIN(c, st);
    if (P > 0) {
        // (nearly) every instance full: getNext is in its LRU-window mode and whole-table shortlists are common
        next_long = c->long_mode == 1 || (c->long_mode != 0 && n_present > 0 && n_nonfull * 16 <= n_present);
        next_full = n_present > 0 && n_nonfull * 16 <= n_present;
        // all-pairs is embarrassingly parallel and wins below ~8k pods (measured: 10k pods 172 us all-pairs vs 120 us
        // sort; 50k pods 4.3 ms vs 0.25 ms); a merge sort of a few thousand 64-byte keys is latency bound
        const bool want_sort = c->rank_mode == 2 || c->rank_mode == 3 || (c->rank_mode == 0 && P >= kRankSortMinPods);
        if (delta) {
            const SnapBufs &A = c->sb[c->cur];
            hipLaunchKernelGGL(delta_scatter_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, A.pods.as<mmp_pod_row>(), A.pos_of.as<int32_t>(),
                               P, dl, B.pods.as<mmp_pod_row>(), c->rank.as<int32_t>(), B.lru.as<int64_t>(), B.rem.as<int64_t>(),
                               B.cnt.as<int32_t>(), B.rpm.as<int32_t>(), B.orig.as<int32_t>(), B.pos_of.as<int32_t>());
        } else if (want_sort && P >= kSampleMinPods && sort_legal && c->rank_mode != 3) {
            const int rc = rank_by_sampling(c, B.pods.as<mmp_pod_row>(), P, min_space, churn2, 0, P, c->rank.as<int32_t>(), st);
            if (rc != MMP_OK) return rc;
        } else if (want_sort && P >= 2 && sort_legal) {
            HIP_TRY(c, c->rk_rows.ensure((size_t)P * sizeof(RankRow)));
            HIP_TRY(c, c->rk_idx.ensure((size_t)P * sizeof(RankRow)));
            const PlacementRowLess less{churn2};
            size_t tmp_bytes = 0;
            HIP_TRY(c, rocprim::merge_sort(nullptr, tmp_bytes, c->rk_rows.as<RankRow>(), c->rk_idx.as<RankRow>(), (size_t)P, less, st));
            HIP_TRY(c, c->rk_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
            hipLaunchKernelGGL(rank_rows_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(), P, min_space,
                               c->rk_rows.as<RankRow>());
            HIP_TRY(c, rocprim::merge_sort(c->rk_tmp.p, tmp_bytes, c->rk_rows.as<RankRow>(), c->rk_idx.as<RankRow>(), (size_t)P, less, st));
            hipLaunchKernelGGL(rank_from_order_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, c->rk_idx.as<RankRow>(), P,
                               c->rank.as<int32_t>());
        } else {
            const int pb = div_up(P, kRankBlock);
            const int slices = std::max(1, std::min(64, 2048 / pb));
            hipLaunchKernelGGL(rank_pods_kernel, dim3(pb, slices), dim3(kRankBlock), 0, st, B.pods.as<mmp_pod_row>(), P,
                               min_space, churn2, slices, 0, P, c->rank.as<int32_t>());
        }
        if (!delta)
            hipLaunchKernelGGL(scatter_pods_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(), P,
                               min_space, c->rank.as<int32_t>(), c->occupancy.as<int32_t>(), B.lru.as<int64_t>(),
                               B.rem.as<int64_t>(), B.cnt.as<int32_t>(), B.rpm.as<int32_t>(), B.orig.as<int32_t>(),
                               B.pos_of.as<int32_t>(), c->flag.as<int32_t>());
        if (n_rs)
            hipLaunchKernelGGL(mark_replaced_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(),
                               P, c->rs_list.as<int32_t>(), n_rs, c->rs_bad.as<uint8_t>());
        // level 1: everything that needs only the table and the rank-ordered columns, in ONE launch (place_kernel.hpp CommitL1)
        CommitL1 L1{};
        L1.pods = B.pods.as<mmp_pod_row>();
        L1.P = P;
        L1.W = W;
        L1.T = T;
        L1.min_space = min_space;
        L1.orig = B.orig.as<int32_t>();
        L1.cnt = B.cnt.as<int32_t>();
        L1.allowed = N.d_allowed.as<uint64_t>();
        L1.prefer = c->d_prefer.as<uint64_t>();
        L1.has_allowed = N.d_has_allowed.as<uint8_t>();
        L1.has_prefer = B.has_pref.as<uint8_t>();
        L1.rs_bad = n_rs ? c->rs_bad.as<uint8_t>() : nullptr;
        L1.elig = B.elig.as<uint64_t>();
        L1.elig_nors = B.elig_nors.as<uint64_t>();
        L1.pref = B.pref.as<uint64_t>();
        L1.fullw = B.fullw.as<uint64_t>();
        L1.ge = B.ge.as<uint64_t>();
        L1.ctpos = B.ctpos.as<int32_t>();
        L1.acc = N.stats_acc.as<StatsAcc>();
        L1.pod_pts = N.d_pts.as<int32_t>();
        L1.NP = N.n_pts;
        L1.pstats = N.pstats.as<StatsAcc>();
        L1.nb_masks = div_up(T * W, 4);
        L1.nb_ge = div_up(kGeRows * W, 4);
        L1.nb_stats = std::min(div_up(P, 256), 512);
        L1.nb_pstats = N.n_pts > 0 ? std::min(div_up(P, 256), 64) : 0;
        hipLaunchKernelGGL(commit_level1_kernel, dim3(L1.nb_stats + L1.nb_pstats + 1 + L1.nb_masks + L1.nb_ge), dim3(256), 0, st, L1);
        // level 2: what needs the bitmaps / the stats: prefix tables, head windows (place_kernel.hpp: TypeWin), case (b) slots
        // (where case (b) starts per preferring type), partition / type subset stats
        int32_t *n_bslots_dev = reinterpret_cast<int32_t *>(static_cast<char *>(B.bslots.p) + kBSlots * sizeof(BSlot));
        CommitL2 L2{};
        L2.S = S;
        L2.pods = B.pods.as<mmp_pod_row>();
        L2.wins = B.heads.as<TypeWin>();
        L2.slots = B.bslots.as<BSlot>();
        L2.n_slots = n_bslots_dev;
        L2.acc = N.stats_acc.as<StatsAcc>();
        L2.pstats = N.pstats.as<StatsAcc>();
        L2.NP = N.n_pts;
        L2.Tw = N.pts_tw;
        L2.prohib = N.d_prohib.as<uint64_t>();
        L2.has_allowed = c->n_types > 0 ? N.d_has_allowed.as<uint8_t>() : nullptr;
        L2.tstats = N.tstats.as<StatsAcc>();
        L2.nb_finish = div_up(std::max(N.n_pts, T), 64);
        hipLaunchKernelGGL(commit_level2_kernel, dim3(4 * T + L2.nb_finish), dim3(64), 0, st, L2);
        // sel / rk from the prefix tables, and the per-type shortlists (TypeMemo) from the finished head windows (every row a decision
        // can name is written, valid or not): one launch; the registry view below is resolved against the shortlists' ranges
        hipLaunchKernelGGL(build_sel_memo_kernel, dim3((have_sel ? 2 * T * W : 0) + std::min(T, kWinLds) + (S.lmemo ? T : 0)), dim3(64), 0, st, S,
                           have_sel ? B.sel.as<int32_t>() : nullptr, have_sel ? B.rk.as<int32_t>() : nullptr,
                           B.heads.as<TypeWin>(), B.memo.as<TypeMemo>(), B.memo_cand.as<int32_t>(), B.memo_rk.as<int16_t>(),
                           B.lmemo.as<LongMemo>());
        // the running minimum of the case (b) candidates' rpm, then the survivor bitmaps of the rpm rule's four limits
        hipLaunchKernelGGL(prefix_min_rpm_kernel, dim3(kBSlots), dim3(64), 0, st, S, B.bslots.as<BSlot>(), n_bslots_dev, B.bpm.as<int32_t>(),
                           (int32_t)(W * 64));
        hipLaunchKernelGGL(build_bsurv_kernel, dim3(kBSlots), dim3(256), 0, st, S, B.bslots.as<BSlot>(), n_bslots_dev, B.bpm.as<int32_t>(),
                           (int32_t)(W * 64), B.bwin.as<BLaunch>(), B.bsurv.as<uint64_t>(), B.bpcs.as<int32_t>());
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, hipMemsetAsync(B.heads.p, 0, wins_bytes, st));
        HIP_TRY(c, hipMemsetAsync(B.memo.p, 0, (size_t)kWinLds * sizeof(TypeMemo), st));
        HIP_TRY(c, hipMemsetAsync(B.pc.p, 0, (size_t)2 * T * (W + 1) * 4, st));
        HIP_TRY(c, hipMemsetAsync(B.nz.p, 0, (size_t)2 * T * (W + 1) * 4, st));
        HIP_TRY(c, hipMemsetAsync(B.ph.p, 0, (size_t)2 * T * (W + 1) * 8, st));
        HIP_TRY(c, hipMemsetAsync(B.elig.p, 0, (size_t)T * W * 8, st));
        HIP_TRY(c, hipMemsetAsync(B.elig_nors.p, 0, (size_t)T * W * 8, st));
        HIP_TRY(c, hipMemsetAsync(B.pref.p, 0, (size_t)T * W * 8, st));
        HIP_TRY(c, hipMemsetAsync(B.fullw.p, 0, (size_t)W * 8, st));
        HIP_TRY(c, hipMemsetAsync(B.ge.p, 0, (size_t)kGeRows * W * 8, st));
    }
    {
        const int rc = build_subset_stats(c, N, B.pods.as<mmp_pod_row>(), P, min_space, st, P > 0 ? 2 : 1);
        if (rc != MMP_OK) return rc;
    }
    KT_END(c, st);
    // the registry view resolved against the new order (still invisible to decisions), queued before the one synchronisation
    {
        const int rc = rebuild_resolved(c, N, S, true, false);
        if (rc != MMP_OK) return rc;
    }
    int32_t bad = 0;
    StatsAcc acc{};
    HIP_TRY(c, hipMemcpyAsync(&bad, c->flag.p, sizeof bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&acc, N.stats_acc.p, sizeof acc, hipMemcpyDeviceToHost, st));
    int32_t n_bslots = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_bslots, static_cast<char *>(B.bslots.p) + kBSlots * sizeof(BSlot), sizeof n_bslots, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    B.types_gen = N.types_gen = c->d_prefer_gen = types_gen_uploaded;
    B.n_bslots = std::min(n_bslots, kBSlots);
    kt_collect(c);
    if (bad)
        return fail(c, MMP_EORDER,
                    "PLACEMENT_ORDER is not a total order on these rows (a full instance with lruTime <= "
                    "2*minChurnAgeMs next to differing instanceVersions); snapshot not published");

    // a type only a few instances may host: its first candidate is usually beyond a lane scan's reach, and only the
    // long variant carries the prefix-table jump that finds it without the wave path
    if (c->long_mode < 0 && acc.sparse_types) next_long = true;
    // publish: the only part of a commit a decision can ever wait for
    std::lock_guard<std::shared_mutex> g(c->mu);
    resident_stop(c);  // it answers for the snapshot it was launched with; the next single request starts one on the new
    c->order_total = sort_legal;
    if (delta) {
        c->h_order.swap(new_order);
        for (int32_t q = 0; q < P; q++) c->h_pos[c->h_order[q]] = q;
        c->n_delta_commits++;
    } else
        c->h_order_valid = false;
    c->dirty.clear();
    c->dirty_all = false;
    c->snap_long = next_long;
    c->snap_full = next_full;
    c->snap = S;
    c->cur = 1 - c->cur;
    c->committed = true;
    split_reset(c);
    c->stats.total_capacity = (int64_t)acc.total_capacity;
    c->stats.total_free = (int64_t)acc.total_free;
    c->stats.global_lru = (int64_t)acc.global_lru;
    c->stats.instance_count = acc.instance_count;
    c->stats.model_copy_count = acc.model_copy_count;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_snapshot_commit");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_snapshot_commit", e.what());
}

int mmp_delta_commits(mmp_ctx *c, int64_t *n_out)
{
    if (!c || !n_out) return fail(c, MMP_EINVAL, "mmp_delta_commits: null argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    *n_out = c->n_delta_commits;
    return MMP_OK;
}

int mmp_get_order(mmp_ctx *c, int32_t *order_out, int32_t *n_out)
try {
    if (!c || !order_out || !n_out) return fail(c, MMP_EINVAL, "mmp_get_order: null argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return fail(c, MMP_ESTATE, "mmp_get_order: a pod-axis shard holds only its own slice of the order");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int32_t n = c->stats.instance_count;  // absent rows sort last
    if (n) HIP_TRY(c, copy_sync(c, order_out, c->snap.orig, (size_t)n * 4, hipMemcpyDeviceToHost));
    *n_out = n;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_get_order");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_get_order", e.what());
}

int mmp_shortlists(mmp_ctx *c, mmp_shortlist_row *rows, int32_t cap_rows, int32_t *n_rows_out)
try {
    if (!c || !n_rows_out || (!rows && cap_rows > 0) || cap_rows < 0) return fail(c, MMP_EINVAL, "mmp_shortlists: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int32_t nt = c->snap.memo ? std::min(c->snap.T, kWinLds) : 0;
    std::vector<TypeMemo> h((size_t)std::max(nt, 1));  // (whole rows: the records' heads are read below)
    if (nt) HIP_TRY(c, copy_sync(c, h.data(), c->snap.memo, (size_t)nt * sizeof(TypeMemo), hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < 2 * nt && i < cap_rows; i++) {
        const MemoVar &v = h[i >> 1].v[i & 1];
        rows[i] = mmp_shortlist_row{v.valid, v.lo, v.hi, v.ccount};
    }
    *n_rows_out = 2 * nt;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shortlists");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shortlists", e.what());
}

int mmp_long_shortlists(mmp_ctx *c, mmp_shortlist_row *rows, int32_t cap_rows, int32_t *n_rows_out)
try {
    if (!c || !n_rows_out || (!rows && cap_rows > 0) || cap_rows < 0) return fail(c, MMP_EINVAL, "mmp_long_shortlists: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int32_t nt = c->snap.lmemo ? c->snap.T : 0;
    std::vector<LongMemo> h((size_t)std::max(nt, 1) * kLongLevels);
    if (nt) HIP_TRY(c, copy_sync(c, h.data(), c->snap.lmemo, (size_t)nt * kLongLevels * sizeof(LongMemo), hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < 2 * nt && i < cap_rows; i++) {
        const LongMemo &m = h[(size_t)(i >> 1) * kLongLevels];  // (the record of a request that excludes none of the first instances)
        const LongVar &v = m.v[i & 1];
        rows[i] = mmp_shortlist_row{v.valid, m.best0, v.end, v.ccount};
    }
    *n_rows_out = 2 * nt;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_long_shortlists");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_long_shortlists", e.what());
}

int mmp_split_batches(mmp_ctx *c, int64_t *n_split_out, int32_t *off_out)
{
    if (!c) return fail(c, MMP_EINVAL, "mmp_split_batches: null context");
    if (n_split_out) *n_split_out = c->n_split.load(std::memory_order_relaxed);
    if (off_out) *off_out = c->split_off.load(std::memory_order_relaxed) ? 1 : 0;
    return MMP_OK;
}

int mmp_cluster_stats(mmp_ctx *c, mmp_stats *out)
try {
    if (!c || !out) return fail(c, MMP_EINVAL, "mmp_cluster_stats: null argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    *out = c->stats;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_cluster_stats");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_cluster_stats", e.what());
}

namespace {
mmp_stats stats_of(const StatsAcc &a)
{
    mmp_stats s{};
    s.total_capacity = (int64_t)a.total_capacity;
    s.total_free = (int64_t)a.total_free;
    s.global_lru = (int64_t)a.global_lru;
    s.instance_count = a.instance_count;
    s.model_copy_count = a.model_copy_count;
    return s;
}
}  // namespace

int mmp_type_stats(mmp_ctx *c, int32_t type, mmp_stats *out)
try {
    if (!c || !out) return fail(c, MMP_EINVAL, "mmp_type_stats: null argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    const int32_t T = (int32_t)cur_side(c).tstats_h.size();
    *out = stats_of(cur_side(c).tstats_h[(type < 0 || type >= T) ? 0 : type]);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_type_stats");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_type_stats", e.what());
}

int mmp_partition_count(mmp_ctx *c, int32_t *n_out)
try {
    if (!c || !n_out) return fail(c, MMP_EINVAL, "mmp_partition_count: null argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    *n_out = cur_side(c).n_pts;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_partition_count");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_partition_count", e.what());
}

int mmp_partition_stats(mmp_ctx *c, int32_t partition, mmp_stats *out, uint64_t *prohibited_out, int32_t max_words)
try {
    if (!c || !out || max_words < 0 || (max_words > 0 && !prohibited_out))
        return fail(c, MMP_EINVAL, "mmp_partition_stats: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (partition < 0 || partition >= cur_side(c).n_pts) return fail(c, MMP_EINVAL, "mmp_partition_stats: no partition %d", partition);
    *out = stats_of(cur_side(c).pstats_h[partition]);
    for (int32_t w = 0; w < max_words; w++)
        prohibited_out[w] = w < cur_side(c).pts_tw ? cur_side(c).pts_prohib[(size_t)partition * cur_side(c).pts_tw + w] : 0;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_partition_stats");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_partition_stats", e.what());
}

int mmp_pod_partitions(mmp_ctx *c, int32_t *partition_out, int32_t max_pods, int32_t *n_out)
try {
    if (!c || !n_out || max_pods < 0 || (max_pods > 0 && !partition_out))
        return fail(c, MMP_EINVAL, "mmp_pod_partitions: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    *n_out = (int32_t)cur_side(c).pts_of.size();
    const int32_t m = std::min(*n_out, max_pods);
    if (m > 0) memcpy(partition_out, cur_side(c).pts_of.data(), (size_t)m * 4);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pod_partitions");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pod_partitions", e.what());
}

/* ---- wire-format ingestion (§8f-1) ------------------------------------------ */

namespace {
// Records per wavefront of the wave-path parsers: a group shares the lanes in the per-field phase (more
// lanes busy), but fewer wavefronts are in flight; keep >= ~8 wavefronts per SIMD before growing the group.
int ingest_group(int32_t n)
{
    static const int forced = [] {
        const char *e = getenv("MMP_JGROUP");
        return e ? atoi(e) : 0;
    }();
    if (forced >= 1 && forced <= kJGroup) return forced;
    int g = 1;
    while (g < kJGroup && n / (g * 2) >= 8192) g *= 2;
    return g;
}

// host side of the open-addressing table the device probes (ingest_kernels.hpp: tab_find)
int build_hash_table(mmp_ctx *c, const char *strs, const int32_t *off, int32_t n, DevBuf &d_hash, DevBuf &d_val,
                     uint32_t &mask_out, const char *what)
{
    uint32_t cap = 16;
    while (cap < (uint32_t)n * 2u) cap <<= 1;
    std::vector<uint64_t> hs(cap, 0);
    std::vector<int32_t> vs(cap, INT32_MIN);
    for (int32_t i = 0; i < n; i++) {
        if (off[i + 1] < off[i]) return fail(c, MMP_EINVAL, "%s: offsets not monotone at %d", what, i);
        const uint64_t h = fnv1a(strs + off[i], off[i + 1] - off[i]);
        uint32_t s = (uint32_t)(h ^ (h >> 32)) & (cap - 1);
        while (vs[s] != INT32_MIN) {
            if (hs[s] == h) return fail(c, MMP_EINVAL, "%s: entries %d and %d are equal or collide under FNV-1a", what, vs[s], i);
            s = (s + 1) & (cap - 1);
        }
        hs[s] = h;
        vs[s] = i;
    }
    HIP_TRY(c, d_hash.ensure((size_t)cap * 8));
    HIP_TRY(c, d_val.ensure((size_t)cap * 4));
    HIP_TRY(c, copy_sync(c, d_hash.p, hs.data(), (size_t)cap * 8, hipMemcpyHostToDevice));
    HIP_TRY(c, copy_sync(c, d_val.p, vs.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    mask_out = cap - 1;
    return MMP_OK;
}
}  // namespace

int mmp_pod_ids_load(mmp_ctx *c, const char *ids, const int32_t *id_off, int32_t n_pods, uint32_t *id_order_out,
                     int32_t *replica_set_out)
try {
    if (!c || n_pods < 0 || !id_off || (n_pods > 0 && !ids)) return fail(c, MMP_EINVAL, "mmp_pod_ids_load: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, quiesce_decisions(c));
    int rc = build_hash_table(c, ids, id_off, n_pods, c->idtab_hash, c->idtab_val, c->idtab_mask, "mmp_pod_ids_load");
    if (rc != MMP_OK) return rc;
    // String.compareTo on ASCII ids == bytewise comparison, shorter prefix first
    std::vector<int32_t> perm(n_pods);
    for (int32_t i = 0; i < n_pods; i++) perm[i] = i;
    auto less = [&](int32_t a, int32_t b) {
        const int la = id_off[a + 1] - id_off[a], lb = id_off[b + 1] - id_off[b];
        const int m = memcmp(ids + id_off[a], ids + id_off[b], (size_t)std::min(la, lb));
        return m != 0 ? m < 0 : la < lb;
    };
    std::sort(perm.begin(), perm.end(), less);
    c->id_order_v.assign(n_pods, 0);
    for (int32_t r = 0; r < n_pods; r++) c->id_order_v[perm[r]] = (uint32_t)r;
    std::unordered_map<std::string, int32_t> rs_intern;
    c->replica_set_v.assign(n_pods, -1);
    for (int32_t i = 0; i < n_pods; i++) {
        const int len = id_off[i + 1] - id_off[i];
        if (len < 7) continue;  // MM.java:4769: iid.length() > 6
        auto it = rs_intern.emplace(std::string(ids + id_off[i], 6), (int32_t)rs_intern.size());
        c->replica_set_v[i] = it.first->second;
    }
    const size_t old = c->pods.size();
    c->pods.resize(n_pods);
    c->dirty_all = true;
    for (size_t i = 0; i < (size_t)n_pods; i++) {
        if (i >= old) {
            c->pods[i] = mmp_pod_row{};
            c->pods[i].flags = MMP_POD_TOMBSTONE;
        }
        c->pods[i].id_order = c->id_order_v[i];
        c->pods[i].replica_set = c->replica_set_v[i];
    }
    c->have_ids = true;
    if (id_order_out && n_pods) memcpy(id_order_out, c->id_order_v.data(), (size_t)n_pods * 4);
    if (replica_set_out && n_pods) memcpy(replica_set_out, c->replica_set_v.data(), (size_t)n_pods * 4);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pod_ids_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pod_ids_load", e.what());
}

int mmp_pods_ingest_json(mmp_ctx *c, const char *buf, const int64_t *off, int32_t n, const int32_t *pod_idx,
                         const uint8_t *live, int64_t *start_time_out, int32_t *status_out)
try {
    if (!c || n < 0 || (n > 0 && (!buf || !off || !pod_idx || !status_out)))
        return fail(c, MMP_EINVAL, "mmp_pods_ingest_json: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->have_ids) return fail(c, MMP_ESTATE, "mmp_pods_ingest_json: load the instance ids first (mmp_pod_ids_load)");
    if (n == 0) return MMP_OK;
    const int32_t P = (int32_t)c->pods.size();
    std::vector<mmp_pod_row> rows(n);
    for (int32_t i = 0; i < n; i++) {
        const int32_t k = pod_idx[i];
        if (k < 0 || k >= P) return fail(c, MMP_EINVAL, "mmp_pods_ingest_json: record %d names pod %d", i, k);
        if (off[i + 1] < off[i]) return fail(c, MMP_EINVAL, "mmp_pods_ingest_json: offsets not monotone at %d", i);
        mmp_pod_row r{};
        r.id_order = c->id_order_v[k];
        r.replica_set = c->replica_set_v[k];
        r.flags = (!live || live[i]) ? MMP_POD_LIVE : 0u;
        rows[i] = r;
    }
    const int64_t bytes = off[n] - off[0];
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->j_buf.ensure((size_t)bytes + 16));  // the wave path stages whole dwords
    HIP_TRY(c, c->j_off.ensure((size_t)(n + 1) * 8));
    HIP_TRY(c, c->j_rows.ensure((size_t)n * sizeof(mmp_pod_row)));
    HIP_TRY(c, c->j_aux.ensure((size_t)n * 8));
    HIP_TRY(c, c->j_status.ensure((size_t)n * 4));
    std::vector<int64_t> rel(n + 1);
    for (int32_t i = 0; i <= n; i++) rel[i] = off[i] - off[0];
    HIP_TRY(c, hipMemcpyAsync(c->j_buf.p, buf + off[0], (size_t)bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->j_off.p, rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->j_rows.p, rows.data(), (size_t)n * sizeof(mmp_pod_row), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->j_aux.p, 0, (size_t)n * 8, st));
    KT_BEGIN(c, st);
    const int grp = ingest_group(n);
    hipLaunchKernelGGL(ingest_pods_kernel, dim3(div_up(n, kJWaves * grp)), dim3(kJBlock), 0, st, c->j_buf.as<char>(), c->j_off.as<int64_t>(), n,
                       grp, c->j_rows.as<mmp_pod_row>(), c->j_aux.as<int64_t>(), c->j_status.as<int32_t>());
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    std::vector<int64_t> stt(n);
    HIP_TRY(c, hipMemcpyAsync(rows.data(), c->j_rows.p, (size_t)n * sizeof(mmp_pod_row), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(stt.data(), c->j_aux.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(status_out, c->j_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    for (int32_t i = 0; i < n; i++) {
        if (status_out[i] == 0) {
            c->pods[pod_idx[i]] = rows[i];
            note_dirty(c, pod_idx[i]);
        }
        if (start_time_out) start_time_out[i] = stt[i];
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pods_ingest_json");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pods_ingest_json", e.what());
}

int mmp_type_names_load(mmp_ctx *c, const char *names, const int32_t *name_off, int32_t n_types, int32_t unknown_type)
try {
    if (!c || n_types < 0 || !name_off || (n_types > 0 && !names)) return fail(c, MMP_EINVAL, "mmp_type_names_load: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = build_hash_table(c, names, name_off, n_types, c->tytab_hash, c->tytab_val, c->tytab_mask, "mmp_type_names_load");
    if (rc != MMP_OK) return rc;
    c->unknown_type = unknown_type;
    c->default_type = unknown_type;
    static const char kDefault[] = "NLCLASSIFIER";
    for (int32_t i = 0; i < n_types; i++)
        if (name_off[i + 1] - name_off[i] == (int)sizeof(kDefault) - 1 && memcmp(names + name_off[i], kDefault, sizeof(kDefault) - 1) == 0)
            c->default_type = i;
    c->have_types = true;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_type_names_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_type_names_load", e.what());
}

int mmp_models_ingest_json(mmp_ctx *c, const char *buf, const int64_t *off, int32_t n_models, int64_t *last_unload_out,
                           int32_t *status_out)
try {
    if (!c || n_models < 0 || (n_models > 0 && (!buf || !off || !status_out)))
        return fail(c, MMP_EINVAL, "mmp_models_ingest_json: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (!c->have_ids) return fail(c, MMP_ESTATE, "mmp_models_ingest_json: load the instance ids first (mmp_pod_ids_load)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, quiesce_decisions(c));  // the registry view is replaced in place
    hipStream_t st = c->stream;
    const int32_t n = n_models;
    c->side[0].rmodels_ok = c->side[1].rmodels_ok = false;
    if (n == 0) {
        c->n_models = 0;
        c->n_entries = 0;
        c->ent_live = 0;
        c->m_cnt.clear();
        return MMP_OK;
    }
    for (int32_t i = 0; i < n; i++)
        if (off[i + 1] < off[i]) return fail(c, MMP_EINVAL, "mmp_models_ingest_json: offsets not monotone at %d", i);
    const int64_t bytes = off[n] - off[0];
    HIP_TRY(c, c->j_buf.ensure((size_t)bytes + 16));  // the wave path stages whole dwords
    HIP_TRY(c, c->j_off.ensure((size_t)(n + 1) * 8));
    HIP_TRY(c, c->j_aux.ensure((size_t)n * 8 + 8));
    HIP_TRY(c, c->j_status.ensure((size_t)n * 4));
    HIP_TRY(c, c->models.ensure((size_t)n * sizeof(mmp_model_row)));
    std::vector<int64_t> rel(n + 1);
    for (int32_t i = 0; i <= n; i++) rel[i] = off[i] - off[0];
    HIP_TRY(c, hipMemcpyAsync(c->j_buf.p, buf + off[0], (size_t)bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->j_off.p, rel.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->models.p, 0, (size_t)n * sizeof(mmp_model_row), st));
    IngestModelsArgs A{};
    A.buf = c->j_buf.as<char>();
    A.off = c->j_off.as<int64_t>();
    A.n = n;
    A.ids = HashTab{c->idtab_hash.as<uint64_t>(), c->idtab_val.as<int32_t>(), c->idtab_mask};
    A.types = c->have_types ? HashTab{c->tytab_hash.as<uint64_t>(), c->tytab_val.as<int32_t>(), c->tytab_mask}
                            : HashTab{nullptr, nullptr, 0};
    A.unknown_type = c->have_types ? c->unknown_type : 0;
    A.default_type = c->have_types ? c->default_type : 0;
    A.rows = c->models.as<mmp_model_row>();
    A.last_unload = c->j_aux.as<int64_t>();
    A.status = c->j_status.as<int32_t>();
    // one pass: every record's entries are parked at slot off / 6 (an entry takes >= 6 bytes of JSON), the
    // counts are scanned, and the entries move to their CSR position — no host round trip in between
    const size_t ent_cap = (size_t)(bytes / 6 + 2);
    HIP_TRY(c, c->j_tmp_pod.ensure(ent_cap * 4));
    HIP_TRY(c, c->j_tmp_time.ensure(ent_cap * 8));
    HIP_TRY(c, c->ent_pod.ensure(ent_cap * 4));
    HIP_TRY(c, c->ent_time.ensure(ent_cap * 8));
    HIP_TRY(c, c->j_cnt.ensure((size_t)(n + 1) * 4));
    HIP_TRY(c, c->j_offs.ensure((size_t)(n + 1) * 4));
    size_t scan_bytes = 0;
    HIP_TRY(c, rocprim::exclusive_scan(nullptr, scan_bytes, c->j_cnt.as<int32_t>(), c->j_offs.as<int32_t>(), (int32_t)0,
                                       (size_t)n + 1, rocprim::plus<int32_t>(), st));
    HIP_TRY(c, c->j_scan_tmp.ensure(std::max<size_t>(scan_bytes, 16)));
    HIP_TRY(c, hipMemsetAsync(c->j_cnt.as<int32_t>() + n, 0, 4, st));
    A.cnt = c->j_cnt.as<int32_t>();
    A.ent_pod = c->j_tmp_pod.as<int32_t>();
    A.ent_time = c->j_tmp_time.as<int64_t>();
    A.grp = ingest_group(n);
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(ingest_models_kernel, dim3(div_up(n, kJWaves * A.grp)), dim3(kJBlock), 0, st, A);
    HIP_TRY(c, rocprim::exclusive_scan(c->j_scan_tmp.p, scan_bytes, c->j_cnt.as<int32_t>(), c->j_offs.as<int32_t>(), (int32_t)0,
                                       (size_t)n + 1, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(compact_entries_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, A.off, n, A.cnt, c->j_offs.as<int32_t>(),
                       A.ent_pod, A.ent_time, A.rows, c->ent_pod.as<int32_t>(), c->ent_time.as<int64_t>());
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    int32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->j_offs.as<int32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(status_out, c->j_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (last_unload_out) HIP_TRY(c, hipMemcpyAsync(last_unload_out, c->j_aux.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    c->m_cnt.resize(n);
    HIP_TRY(c, hipMemcpyAsync(c->m_cnt.data(), c->j_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    c->n_models = n;
    c->n_entries = total;
    c->ent_live = total;
    return rebuild_resolved(c);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_models_ingest_json");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_models_ingest_json", e.what());
}

int mmp_pods_get(mmp_ctx *c, mmp_pod_row *rows_out, int32_t max_rows, int32_t *n_out)
try {
    if (!c || !n_out || max_rows < 0 || (max_rows > 0 && !rows_out)) return fail(c, MMP_EINVAL, "mmp_pods_get: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    *n_out = (int32_t)c->pods.size();
    const int32_t m = std::min(*n_out, max_rows);
    if (m > 0) memcpy(rows_out, c->pods.data(), (size_t)m * sizeof(mmp_pod_row));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_pods_get");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_pods_get", e.what());
}

int mmp_models_get(mmp_ctx *c, mmp_model_row *rows_out, int32_t max_models, int32_t *ent_pod_out, int64_t *ent_time_out,
                   int32_t max_entries, int32_t *n_models_out, int32_t *n_entries_out)
try {
    if (!c || !n_models_out || !n_entries_out || max_models < 0 || max_entries < 0)
        return fail(c, MMP_EINVAL, "mmp_models_get: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_models_out = c->n_models;
    *n_entries_out = c->n_entries;
    const int32_t m = std::min(c->n_models, max_models), e = std::min(c->n_entries, max_entries);
    if (m > 0 && rows_out) HIP_TRY(c, copy_sync(c, rows_out, c->models.p, (size_t)m * sizeof(mmp_model_row), hipMemcpyDeviceToHost));
    if (e > 0 && ent_pod_out) HIP_TRY(c, copy_sync(c, ent_pod_out, c->ent_pod.p, (size_t)e * 4, hipMemcpyDeviceToHost));
    if (e > 0 && ent_time_out) HIP_TRY(c, copy_sync(c, ent_time_out, c->ent_time.p, (size_t)e * 8, hipMemcpyDeviceToHost));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_models_get");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_models_get", e.what());
}

/* ---- pod-axis sharding --------------------------------- */

int mmp_shard_configure(mmp_ctx *c, int32_t shard, int32_t n_shards)
try {
    if (!c || n_shards < 1 || n_shards > kMaxShards || shard < 0 || shard >= n_shards)
        return fail(c, MMP_EINVAL, "mmp_shard_configure: need 0 <= shard < n_shards <= %d", kMaxShards);
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    c->shard = shard;
    c->n_shards = n_shards;
    c->committed = false;
    c->rank_pending = false;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_configure");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_configure", e.what());
}

int32_t mmp_shard_xchg_slots(int32_t phase, int32_t n_shards)
{
    switch (phase) {
    case 1: return kX1;
    case 2: return kX2;
    case 3: return kX3;
    case 4: return kX4;
    case 5: return x5_slots(n_shards);
    case 6: return kX6;
    default: return 0;
    }
}

int32_t mmp_shard_xchg_is_sum(int32_t phase) { return phase == 5 ? 1 : 0; }

int mmp_shard_rank_dev(mmp_ctx *c, void *d_rank)
try {
    if (!c || !d_rank) return fail(c, MMP_EINVAL, "mmp_shard_rank_dev: null argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (c->n_shards < 1) return fail(c, MMP_ESTATE, "mmp_shard_configure has not been called");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int32_t P = (int32_t)c->pods.size();
    if (P >= kShardMaxPods)
        return fail(c, MMP_EINVAL, "pod-axis mode: %d instances exceed the %d the exchange words can carry", P, kShardMaxPods - 1);
    SnapBufs &B = c->sb[1 - c->cur];
    hipStream_t st = c->stream;
    HIP_TRY(c, B.pods.ensure(std::max<size_t>(P, 1) * sizeof(mmp_pod_row)));
    if (P) {
        HIP_TRY(c, hipMemcpyAsync(B.pods.p, c->pods.data(), (size_t)P * sizeof(mmp_pod_row), hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemsetAsync(d_rank, 0, (size_t)P * 4, st));
        const int per = div_up(P, c->n_shards);
        const int p_lo = std::min(P, c->shard * per), p_hi = std::min(P, p_lo + per);
        if (p_hi > p_lo) {
            const int64_t churn2 = (int64_t)((uint64_t)c->cfg.min_churn_age_ms * 2u);
            const int64_t min_space = c->cfg.min_space_units;
            // the slice against all rows is (P / shards) x P comparator calls; from where an unsharded commit sorts (8192^2 pairs)
            // sorting the whole table is cheaper — when the comparator is a total order on it (snapshot.hpp "ranking by sorting")
            bool versions_differ = false, full_low_lru = false, wide_count = false;
            for (int32_t p = 0; p < P; p++) {
                const mmp_pod_row &r = c->pods[p];
                if (r.version != c->pods[0].version) versions_differ = true;
                if (r.count > (1 << 30) || r.count < -(1 << 30)) wide_count = true;
                const uint64_t d = (uint64_t)r.capacity - (uint64_t)r.used;
                const int64_t rem = (int64_t)d > 0 ? (int64_t)d : 0;
                if (rem < min_space && r.lru_time <= churn2) full_low_lru = true;
            }
            const bool sort_legal = !(versions_differ && full_low_lru) && !wide_count;
            const bool want_sort = c->rank_mode == 2 || c->rank_mode == 3 || (c->rank_mode == 0 && (int64_t)(p_hi - p_lo) * P >= (int64_t)kRankSortMinPods * kRankSortMinPods);
            if (want_sort && sort_legal && P >= kSampleMinPods && c->rank_mode != 3) {
                const int rc = rank_by_sampling(c, B.pods.as<mmp_pod_row>(), P, min_space, churn2, p_lo, p_hi, static_cast<int32_t *>(d_rank), st);
                if (rc != MMP_OK) return rc;
            } else if (want_sort && sort_legal && P >= 2) {
                HIP_TRY(c, c->rk_rows.ensure((size_t)P * sizeof(RankRow)));
                HIP_TRY(c, c->rk_idx.ensure((size_t)P * sizeof(RankRow)));
                const PlacementRowLess less{churn2};
                size_t tmp_bytes = 0;
                HIP_TRY(c, rocprim::merge_sort(nullptr, tmp_bytes, c->rk_rows.as<RankRow>(), c->rk_idx.as<RankRow>(), (size_t)P, less, st));
                HIP_TRY(c, c->rk_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
                hipLaunchKernelGGL(rank_rows_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(), P, min_space,
                                   c->rk_rows.as<RankRow>());
                HIP_TRY(c, rocprim::merge_sort(c->rk_tmp.p, tmp_bytes, c->rk_rows.as<RankRow>(), c->rk_idx.as<RankRow>(), (size_t)P, less, st));
                hipLaunchKernelGGL(rank_from_order_range_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, c->rk_idx.as<RankRow>(), P, p_lo, p_hi,
                                   static_cast<int32_t *>(d_rank));
            } else {
                const int pb = div_up(p_hi - p_lo, kRankBlock);
                const int slices = std::max(1, std::min(64, 2048 / pb));
                hipLaunchKernelGGL(rank_pods_kernel, dim3(pb, slices), dim3(kRankBlock), 0, st, B.pods.as<mmp_pod_row>(), P,
                                   min_space, churn2, slices, p_lo, p_hi, static_cast<int32_t *>(d_rank));
            }
            HIP_TRY(c, hipGetLastError());
        }
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    c->rank_pending = true;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_rank_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_rank_dev", e.what());
}

int mmp_shard_commit_dev(mmp_ctx *c, const void *d_rank)
try {
    if (!c || !d_rank) return fail(c, MMP_EINVAL, "mmp_shard_commit_dev: null argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (c->n_shards < 1 || !c->rank_pending) return fail(c, MMP_ESTATE, "mmp_shard_commit_dev: call mmp_shard_rank_dev first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, quiesce_decisions(c));
    const int32_t P = (int32_t)c->pods.size();
    const int32_t W = std::max(div_up(P, 64), 1);
    const int32_t Wfull = div_up(P, 64);
    const int32_t T = std::max(c->n_types, 1);
    if (c->n_types > 0 && c->types_w != Wfull)
        return fail(c, MMP_ESTATE, "type bitmaps were loaded for a different pod count; reload them before commit");
    const int32_t Wl = div_up(W, c->n_shards);
    const int32_t w_lo = std::min(W, c->shard * Wl);
    const int32_t Wn = std::min(W, w_lo + Wl) - w_lo;
    const int32_t Wn1 = std::max(Wn, 1);
    const size_t padded = (size_t)Wn1 * 64, padded_full = (size_t)W * 64;
    SnapBufs &B = c->sb[1 - c->cur];  // B.pods was filled by mmp_shard_rank_dev
    SnapSide &N = c->side[1 - c->cur];
    hipStream_t st = c->stream;

    HIP_TRY(c, B.lru.ensure(padded * 8));
    HIP_TRY(c, B.rem.ensure(padded * 8));
    HIP_TRY(c, B.cnt.ensure(padded * 4));
    HIP_TRY(c, B.rpm.ensure(padded * 4));
    HIP_TRY(c, B.orig.ensure(padded * 4));
    HIP_TRY(c, B.pos_of.ensure(padded_full * 4));
    HIP_TRY(c, B.elig.ensure((size_t)T * Wn1 * 8));
    HIP_TRY(c, B.elig_nors.ensure((size_t)T * Wn1 * 8));
    HIP_TRY(c, B.pref.ensure((size_t)T * Wn1 * 8));
    HIP_TRY(c, B.has_pref.ensure(T));
    HIP_TRY(c, B.fullw.ensure((size_t)Wn1 * 8));
    HIP_TRY(c, B.ge.ensure((size_t)kGeRows * Wn1 * 8));
    HIP_TRY(c, c->occupancy.ensure(padded_full * 4));
    HIP_TRY(c, c->flag.ensure(sizeof(int32_t)));
    HIP_TRY(c, N.stats_acc.ensure(sizeof(StatsAcc)));
    HIP_TRY(c, c->rs_bad.ensure(padded_full));
    HIP_TRY(c, c->rs_list.ensure(std::max<size_t>(c->replaced_rs.size(), 1) * 4));
    HIP_TRY(c, N.d_allowed.ensure(std::max<size_t>((size_t)T * W, 1) * 8));
    HIP_TRY(c, c->d_prefer.ensure(std::max<size_t>((size_t)T * W, 1) * 8));
    HIP_TRY(c, N.d_has_allowed.ensure(T));

    {
        ZeroBatch zb(st);
        zb.add(c->occupancy.p, padded_full * 4);
        zb.add(c->flag.p, sizeof(int32_t));
        zb.add(B.lru.p, padded * 8);
        zb.add(B.rem.p, padded * 8);
        zb.add(B.cnt.p, padded * 4);
        zb.add(B.rpm.p, padded * 4);
        zb.add(B.orig.p, padded * 4);
        zb.add(B.pos_of.p, padded_full * 4);
        zb.add(B.elig.p, (size_t)T * Wn1 * 8);
        zb.add(B.elig_nors.p, (size_t)T * Wn1 * 8);
        zb.add(B.pref.p, (size_t)T * Wn1 * 8);
        zb.add(B.fullw.p, (size_t)Wn1 * 8);
        zb.add(B.ge.p, (size_t)kGeRows * Wn1 * 8);
        zb.launch(st);
        HIP_TRY(c, hipGetLastError());
    }

    std::vector<uint8_t> hp(T, 0), ha(T, 0);
    for (int32_t t = 0; t < c->n_types; t++) {
        hp[t] = c->has_prefer[t];
        ha[t] = c->has_allowed[t];
    }
    HIP_TRY(c, hipMemcpyAsync(B.has_pref.p, hp.data(), T, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(N.d_has_allowed.p, ha.data(), T, hipMemcpyHostToDevice, st));
    if (c->n_types > 0 && !c->allowed.empty()) {
        HIP_TRY(c, hipMemcpyAsync(N.d_allowed.p, c->allowed.data(), c->allowed.size() * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->d_prefer.p, c->prefer.data(), c->prefer.size() * 8, hipMemcpyHostToDevice, st));
    }
    B.types_gen = N.types_gen = c->d_prefer_gen = 0;  // (uploaded unconditionally here; "unknown" makes an ordinary commit upload again)
    const int32_t n_rs = (int32_t)c->replaced_rs.size();
    if (n_rs) HIP_TRY(c, hipMemcpyAsync(c->rs_list.p, c->replaced_rs.data(), (size_t)n_rs * 4, hipMemcpyHostToDevice, st));
    const int64_t min_space = c->cfg.min_space_units;
    StatsAcc init{};
    init.global_lru = INT64_MAX;
    HIP_TRY(c, hipMemcpyAsync(N.stats_acc.p, &init, sizeof init, hipMemcpyHostToDevice, st));

    if (P > 0) {
        hipLaunchKernelGGL(scatter_shard_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(), P,
                           static_cast<const int32_t *>(d_rank), c->occupancy.as<int32_t>(), w_lo * 64, (w_lo + Wn) * 64,
                           B.lru.as<int64_t>(), B.rem.as<int64_t>(), B.cnt.as<int32_t>(), B.rpm.as<int32_t>(),
                           B.orig.as<int32_t>(), B.pos_of.as<int32_t>(), c->flag.as<int32_t>());
        if (n_rs)
            hipLaunchKernelGGL(mark_replaced_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(),
                               P, c->rs_list.as<int32_t>(), n_rs, c->rs_bad.as<uint8_t>());
        if (Wn > 0)
            hipLaunchKernelGGL(build_masks_shard_kernel, dim3(div_up(T * Wn, 4)), dim3(256), 0, st, B.pods.as<mmp_pod_row>(),
                               P, Wfull, w_lo, Wn, T, min_space, B.orig.as<int32_t>(), N.d_allowed.as<uint64_t>(),
                               N.d_has_allowed.as<uint8_t>(), c->d_prefer.as<uint64_t>(), B.has_pref.as<uint8_t>(),
                               n_rs ? c->rs_bad.as<uint8_t>() : nullptr, B.elig.as<uint64_t>(),
                               B.elig_nors.as<uint64_t>(), B.pref.as<uint64_t>(), B.fullw.as<uint64_t>());
        const int32_t P_local = std::max(0, std::min(P - w_lo * 64, Wn * 64));
        if (Wn > 0)  // the count-threshold bitmaps of the slice (the lane path's count break)
            hipLaunchKernelGGL(build_ge_kernel, dim3(div_up(kGeRows * Wn, 4)), dim3(256), 0, st, B.cnt.as<int32_t>(), P_local, Wn,
                               B.ge.as<uint64_t>());
        hipLaunchKernelGGL(cluster_stats_kernel, dim3(std::min(div_up(P, 256), 512)), dim3(256), 0, st,
                           B.pods.as<mmp_pod_row>(), P, min_space, N.stats_acc.as<StatsAcc>());
        HIP_TRY(c, hipGetLastError());
    }
    {
        const int rc = build_subset_stats(c, N, B.pods.as<mmp_pod_row>(), P, min_space, st);
        if (rc != MMP_OK) return rc;
    }
    int32_t bad = 0;
    StatsAcc acc{};
    HIP_TRY(c, hipMemcpyAsync(&bad, c->flag.p, sizeof bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&acc, N.stats_acc.p, sizeof acc, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    c->rank_pending = false;
    if (bad)
        return fail(c, MMP_EORDER, "PLACEMENT_ORDER is not a total order on these rows; snapshot not published");

    ShardSnap S{};
    S.P = P;
    S.W = W;
    S.T = T;
    S.any_rs = n_rs > 0;
    S.min_space = min_space;
    S.shard = c->shard;
    S.n_shards = c->n_shards;
    S.Wl = Wl;
    S.w_lo = w_lo;
    S.Wn = Wn;
    S.lru = B.lru.as<int64_t>();
    S.rem = B.rem.as<int64_t>();
    S.cnt = B.cnt.as<int32_t>();
    S.rpm = B.rpm.as<int32_t>();
    S.orig = B.orig.as<int32_t>();
    S.pos_of = B.pos_of.as<int32_t>();
    S.elig = B.elig.as<uint64_t>();
    S.elig_nors = B.elig_nors.as<uint64_t>();
    S.pref = B.pref.as<uint64_t>();
    S.has_pref = B.has_pref.as<uint8_t>();
    S.fullw = B.fullw.as<uint64_t>();
    c->ssnap = S;
    Snap V{};  // the slice as the lane-per-decision path sees it: local positions, global pos_of
    V.P = std::max(0, std::min(P - w_lo * 64, Wn * 64));
    V.W = Wn;
    V.T = T;
    V.any_rs = S.any_rs;
    V.min_space = min_space;
    V.lru = S.lru;
    V.rem = S.rem;
    V.cnt = S.cnt;
    V.rpm = S.rpm;
    V.orig = S.orig;
    V.pos_of = S.pos_of;
    V.elig = S.elig;
    V.elig_nors = S.elig_nors;
    V.pref = S.pref;
    V.has_pref = S.has_pref;
    V.fullw = S.fullw;
    V.ge = B.ge.as<uint64_t>();
    V.pos_base = w_lo * 64;
    V.w_base = w_lo;
    V.more_after = (w_lo + Wn) * 64 < P ? 1 : 0;
    Snap M{};  // what the non-placement entry points read in shard mode
    M.P = P;
    M.W = W;
    M.T = T;
    M.any_rs = S.any_rs;
    M.min_space = min_space;
    M.pos_of = S.pos_of;
    {
        // the slice's per-type head windows (place_kernel.hpp: TypeWin, built over the VIEW) and the registry's resolved rows
        // (global rank positions; resolve_req<true> translates them): what place_shard_fast_kernel decides from
        const size_t wins_bytes = (size_t)std::max(T, kWinLds) * sizeof(TypeWin);
        HIP_TRY(c, B.heads.ensure(wins_bytes));
        HIP_TRY(c, hipMemsetAsync(B.heads.p, 0, wins_bytes, st));
        if (V.P > 0 && V.W > 0) hipLaunchKernelGGL(build_wins_kernel, dim3(T), dim3(64), 0, st, V, B.heads.as<TypeWin>());
        HIP_TRY(c, hipGetLastError());
        const int rc = rebuild_resolved(c, N, M, true);  // (synchronizes the stream)
        if (rc != MMP_OK) return rc;
    }
    c->sview = V;
    c->snap = M;
    c->cur = 1 - c->cur;
    c->committed = true;
    c->stats.total_capacity = (int64_t)acc.total_capacity;
    c->stats.total_free = (int64_t)acc.total_free;
    c->stats.global_lru = (int64_t)acc.global_lru;
    c->stats.instance_count = acc.instance_count;
    c->stats.model_copy_count = acc.model_copy_count;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_commit_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_commit_dev", e.what());
}

namespace {
// one phase of the general protocol on `st` (called with c->mu held); n_dev: see PlaceArgs::n_dev
int shard_phase_launch(mmp_ctx *c, int32_t phase, const void *d_reqs, int32_t n, const int32_t *n_dev, const void *d_extra,
                       int64_t now, void *const *d_xchg, void *d_outs, hipStream_t st)
{
    HIP_TRY(c, order_after_registry(c, st));
    PlaceArgs A{};
    A.reqs = static_cast<const mmp_place_req *>(d_reqs);
    A.models = c->models.as<mmp_model_row>();
    A.rmodels = nullptr;
    A.wins = nullptr;
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.extra = static_cast<const int32_t *>(d_extra);
    A.outs = static_cast<mmp_place_out *>(d_outs);
    A.n = n;
    A.n_models = c->n_models;
    A.now = now;
    A.n_dev = n_dev;
    A.force_wave = 0;
    A.n_pods_all = c->ssnap.P;
    A.done_flag = nullptr;
    A.done_seq = 0;
    XchgPtrs X{static_cast<int64_t *>(d_xchg[0]), static_cast<int64_t *>(d_xchg[1]), static_cast<int64_t *>(d_xchg[2]),
               static_cast<int64_t *>(d_xchg[3]), static_cast<int64_t *>(d_xchg[4]), static_cast<int64_t *>(d_xchg[5])};
    const ShardSnap &S = c->ssnap;
    const int wpad = (std::max(S.Wn, 1) + 1) & ~1;
    const size_t lds = (size_t)kPlaceWaves * 2 * wpad * sizeof(uint64_t);
    if (lds > 64 * 1024) return fail(c, MMP_EINVAL, "shard slice too large for the LDS staging tile (%d words)", S.Wn);
    const int blocks = std::min(div_up(n, kPlaceWaves), 256 * 8);
    const dim3 grid(blocks), block(kPlaceWaves * 64);
    switch (phase) {
    case 1: hipLaunchKernelGGL(place_shard_kernel<1>, grid, block, lds, st, S, A, X, wpad); break;
    case 2: hipLaunchKernelGGL(place_shard_kernel<2>, grid, block, lds, st, S, A, X, wpad); break;
    case 3: hipLaunchKernelGGL(place_shard_kernel<3>, grid, block, lds, st, S, A, X, wpad); break;
    case 4: hipLaunchKernelGGL(place_shard_kernel<4>, grid, block, lds, st, S, A, X, wpad); break;
    case 5: hipLaunchKernelGGL(place_shard_kernel<5>, grid, block, lds, st, S, A, X, wpad); break;
    case 6: hipLaunchKernelGGL(place_shard_kernel<6>, grid, block, lds, st, S, A, X, wpad); break;
    default: hipLaunchKernelGGL(place_shard_finish_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, S, A, X); break;
    }
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
}
}  // namespace

int mmp_shard_place_phase_dev(mmp_ctx *c, int32_t phase, const void *d_reqs, int32_t n, const void *d_extra, int64_t now,
                              void *const *d_xchg, void *d_outs, void *stream)
try {
    if (!c || phase < 1 || phase > 7 || n < 0 || !d_xchg || (n > 0 && (!d_reqs || (phase == 7 && !d_outs))))
        return fail(c, MMP_EINVAL, "mmp_shard_place_phase_dev: bad argument");
    for (int i = 0; i < 6; i++)
        if (n > 0 && !d_xchg[i]) return fail(c, MMP_EINVAL, "mmp_shard_place_phase_dev: exchange buffer %d is null", i + 1);
    std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published shard snapshot + enqueue; no wait
    if (c->n_shards < 1 || !c->committed) return fail(c, MMP_ESTATE, "no committed shard snapshot");
    if (n == 0) return MMP_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    note_caller_stream(c, st);
    return shard_phase_launch(c, phase, d_reqs, n, nullptr, d_extra, now, d_xchg, d_outs, st);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_phase_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_phase_dev", e.what());
}

int32_t mmp_shard_fast_slots(void) { return kXF; }

namespace {
PlaceArgs shard_args(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs)
{
    PlaceArgs A{};
    A.reqs = static_cast<const mmp_place_req *>(d_reqs);
    A.models = c->models.as<mmp_model_row>();
    A.rmodels = cur_side(c).rmodels_ok ? cur_side(c).rmodels.as<ResolvedModel>() : nullptr;
    A.wins = (c->no_heads || c->sview.P <= 0) ? nullptr : c->sb[c->cur].heads.as<TypeWin>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.extra = static_cast<const int32_t *>(d_extra);
    A.outs = static_cast<mmp_place_out *>(d_outs);
    A.n = n;
    A.n_models = c->n_models;
    A.now = now;
    A.n_dev = nullptr;
    A.force_wave = 0;
    A.n_pods_all = c->ssnap.P;
    A.done_flag = nullptr;
    A.done_seq = 0;
    return A;
}
}  // namespace

int mmp_shard_place_fast_dev(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_xf,
                             void *stream)
try {
    if (!c || n < 0 || (n > 0 && (!d_reqs || !d_xf))) return fail(c, MMP_EINVAL, "mmp_shard_place_fast_dev: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (c->n_shards < 1 || !c->committed) return fail(c, MMP_ESTATE, "no committed shard snapshot");
    if (n == 0) return MMP_OK;
    const PlaceArgs A = shard_args(c, d_reqs, n, d_extra, now, nullptr);
    note_caller_stream(c, static_cast<hipStream_t>(stream));
    HIP_TRY(c, order_after_registry(c, static_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(place_shard_fast_kernel, dim3(div_up(n, kPlaceBlock)), dim3(kPlaceBlock), (size_t)place_lane_lds(c->sview.T),
                       static_cast<hipStream_t>(stream), c->sview, A, c->shard, static_cast<int64_t *>(d_xf));
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_fast_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_fast_dev", e.what());
}

namespace {
// After the all-reduce: the decided rows are written, the rest flagged and COUNTED; the count reaches the host through a
// pinned word (place_shard_fast_finish_kernel).  Three steps so that a caller may leave between the first and the second
// (mmp_shard_place_batch_async_dev): enqueue the kernel; synchronise and read the count; only when it is not zero, flags ->
// exclusive scan -> gather of the undecided requests (decision order, identical on every shard).  Called with c->mu held.
int shard_finish_prepare(mmp_ctx *c, uint32_t slot, int32_t n, hipStream_t st, uint32_t *seq_out)
{
    HIP_TRY(c, c->f_flags[slot].ensure((size_t)(n + 1) * 4));
    if (!c->f_cnt[slot].p) {
        HIP_TRY(c, c->f_cnt[slot].ensure(16));
        HIP_TRY(c, hipMemsetAsync(c->f_cnt[slot].p, 0, 16, st));
    }
    if (!c->f_done) {
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->f_done), 128, kPinnedFlags));
        c->f_done[0] = c->f_done[8] = 0;
    }
    *seq_out = ++c->f_seq;
    return MMP_OK;
}
int shard_finish_enqueue(mmp_ctx *c, uint32_t slot, int32_t n, const void *d_xf, void *d_outs, hipStream_t st, uint32_t *seq_out)
{
    uint32_t seq = 0;
    const int prc = shard_finish_prepare(c, slot, n, st, &seq);
    if (prc != MMP_OK) return prc;
    hipLaunchKernelGGL(place_shard_fast_finish_kernel, dim3(div_up(n + 1, 256)), dim3(256), 0, st, static_cast<const int64_t *>(d_xf),
                       n, c->ssnap.any_rs, static_cast<mmp_place_out *>(d_outs), c->f_flags[slot].as<int32_t>(),
                       c->f_cnt[slot].as<unsigned long long>(), c->f_done + 8 * slot, seq);
    HIP_TRY(c, hipGetLastError());
    *seq_out = seq;
    return MMP_OK;
}
// The count arrives in pinned memory with the finish kernel's last workgroup: the host polls the word (no copy, and no wait for
// whatever a later batch has enqueued behind that kernel); a word that stays away for 2 ms is waited for with the stream.
// Everything the host does with the count afterwards is enqueued on the same stream, i.e. ordered behind the kernel.
int shard_finish_collect(mmp_ctx *c, uint32_t slot, hipStream_t st, uint32_t seq, int32_t *n_rest_out)
{
    const uint64_t *w = c->f_done + 8 * slot;
    uint64_t word = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        word = __atomic_load_n(w, __ATOMIC_ACQUIRE);
        if ((uint32_t)(word >> 32) == seq) break;
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            HIP_TRY(c, hipStreamSynchronize(st));
            word = __atomic_load_n(w, __ATOMIC_ACQUIRE);
            if ((uint32_t)(word >> 32) != seq) return fail(c, MMP_EHIP, "shard finish kernel did not report its count");
            break;
        }
        __builtin_ia32_pause();
    }
    *n_rest_out = (int32_t)(uint32_t)word;
    return MMP_OK;
}
int shard_rest_gather(mmp_ctx *c, uint32_t slot, const void *d_reqs, int32_t n, int32_t n_rest, hipStream_t st)
{
    HIP_TRY(c, c->f_offs.ensure((size_t)(n + 1) * 4));
    HIP_TRY(c, c->f_idx.ensure((size_t)n_rest * 4));
    HIP_TRY(c, c->f_reqs.ensure((size_t)n_rest * sizeof(mmp_place_req)));
    HIP_TRY(c, c->f_outs.ensure((size_t)n_rest * sizeof(mmp_place_out)));
    const int32_t *flags = c->f_flags[slot].as<int32_t>();
    size_t scan_bytes = 0;
    HIP_TRY(c, rocprim::exclusive_scan(nullptr, scan_bytes, flags, c->f_offs.as<int32_t>(), (int32_t)0, (size_t)n + 1, rocprim::plus<int32_t>(), st));
    HIP_TRY(c, c->f_scan_tmp.ensure(std::max<size_t>(scan_bytes, 16)));
    HIP_TRY(c, rocprim::exclusive_scan(c->f_scan_tmp.p, scan_bytes, flags, c->f_offs.as<int32_t>(), (int32_t)0, (size_t)n + 1,
                                       rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(place_shard_gather_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, static_cast<const mmp_place_req *>(d_reqs),
                       n, flags, c->f_offs.as<int32_t>(), c->f_reqs.as<mmp_place_req>(), c->f_idx.as<int32_t>());
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
}
int shard_finish_launch(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_xf, void *d_outs, hipStream_t st, int32_t *n_rest_out)
{
    uint32_t seq = 0;  // (the step-wise calls: one batch at a time, slot 0; the caller reads d_outs next: synchronise)
    int rc = shard_finish_enqueue(c, 0, n, d_xf, d_outs, st, &seq);
    if (rc == MMP_OK) rc = shard_finish_collect(c, 0, st, seq, n_rest_out);
    if (rc == MMP_OK && *n_rest_out > 0) rc = shard_rest_gather(c, 0, d_reqs, n, *n_rest_out, st);
    if (rc == MMP_OK) HIP_TRY(c, hipStreamSynchronize(st));
    return rc;
}
}  // namespace

int mmp_shard_place_fast_finish_dev(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_xf, void *d_outs, void *stream,
                                    int32_t *n_rest_out, void **d_rest_reqs_out, void **d_rest_outs_out)
try {
    if (!c || n < 0 || !n_rest_out || !d_rest_reqs_out || !d_rest_outs_out || (n > 0 && (!d_reqs || !d_xf || !d_outs)))
        return fail(c, MMP_EINVAL, "mmp_shard_place_fast_finish_dev: bad argument");
    *n_rest_out = 0;
    *d_rest_reqs_out = *d_rest_outs_out = nullptr;
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (c->n_shards < 1 || !c->committed) return fail(c, MMP_ESTATE, "no committed shard snapshot");
    // the step-wise calls use count slot 0 and the context's rest buffers, which an asynchronous batch still in flight may own
    if (c->pend.active) return fail(c, MMP_ESTATE, "mmp_shard_place_fast_finish_dev: an asynchronous batch is pending (mmp_shard_wait first)");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    note_caller_stream(c, st);
    int32_t n_rest = 0;
    {
        const int rc = shard_finish_launch(c, d_reqs, n, d_xf, d_outs, st, &n_rest);
        if (rc != MMP_OK) return rc;
    }
    *n_rest_out = n_rest;
    if (n_rest) {
        *d_rest_reqs_out = c->f_reqs.p;
        *d_rest_outs_out = c->f_outs.p;
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_fast_finish_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_fast_finish_dev", e.what());
}

int mmp_shard_place_fast_scatter_dev(mmp_ctx *c, int32_t n_rest, void *d_outs, void *stream)
try {
    if (!c || n_rest < 0 || (n_rest > 0 && !d_outs)) return fail(c, MMP_EINVAL, "mmp_shard_place_fast_scatter_dev: bad argument");
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (c->n_shards < 1 || !c->committed) return fail(c, MMP_ESTATE, "no committed shard snapshot");
    if (n_rest == 0) return MMP_OK;
    hipLaunchKernelGGL(place_shard_scatter_kernel, dim3(div_up(n_rest, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       c->f_outs.as<mmp_place_out>(), c->f_idx.as<int32_t>(), n_rest, static_cast<mmp_place_out *>(d_outs));
    HIP_TRY(c, hipGetLastError());
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_fast_scatter_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_fast_scatter_dev", e.what());
}

/* ---- RCCL inside the boundary: the pod-axis group ------------------------- */
namespace {
// librccl is bound at run time: a host without RCCL (or a CPU-only test box) still loads libmmplace, and a process
// that already carries a copy (PyTorch bundles one) reuses it instead of mapping a second one.
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
// why_out: receives the reason when the call returns null (copied out under the binding's lock).
// MMP_RCCL_PATH=<file> names the one library to bind (a host with its own RCCL build; also how the tests reach the
// "no RCCL on this host" answer on a box that has one).
RcclApi *rccl_api(std::string *why_out = nullptr)
{
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> g(mu);
    if (api.h) return &api;
    struct Why {  // the reason leaves with every null return, whichever one it is
        RcclApi &a;
        std::string *out;
        ~Why() { if (out && !a.h) *out = a.why; }
    } why_guard{api, why_out};
    const char *only = getenv("MMP_RCCL_PATH");
    if (only && *only) {
        api.h = dlopen(only, RTLD_NOW | RTLD_LOCAL);
    } else {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (int pass = 0; pass < 2 && !api.h; pass++)
            for (const char *nm : names) {
                api.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (api.h) break;
            }
        if (!api.h) api.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!api.h) {
        const char *e = dlerror();  // ONE call: dlerror() clears the message it returns
        api.why = e ? e : "librccl.so not found";
        return nullptr;
    }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.h, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.h, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.h, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        api.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
        dlclose(api.h);
        api.h = nullptr;
        return nullptr;
    }
    return &api;
}

void group_comm_destroy(ncclComm_t comm)
{
    if (RcclApi *R = rccl_api()) (void)R->CommDestroy(comm);
}

#define RCCL_TRY(c, expr)                                                                                  \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return fail((c), MMP_EHIP, "%s failed: %s (%s:%d)", #expr,                                     \
                        rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r_) : "rccl error", __FILE__, __LINE__); \
    } while (0)

// all-reduce in place on the context's stream; a group of one without a communicator has nothing to exchange
int group_allreduce(mmp_ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op)
{
    if (count == 0) return MMP_OK;
    if (c->xfn) {
        const int rc = c->xfn(c->xuser, buf, (int64_t)count, dt == ncclInt64 ? 1 : 0, op == ncclMin ? 1 : 0, c->stream);
        return rc == 0 ? MMP_OK : fail(c, MMP_EHIP, "the host's exchange callback failed (%d)", rc);
    }
    if (!c->comm) return MMP_OK;
    RCCL_TRY(c, rccl_api()->AllReduce(buf, buf, count, dt, op, c->comm, c->stream));
    return MMP_OK;
}

// the six exchange phases + the result rows for `n` request rows (n_dev: the real row count on the device, or null)
int group_general(mmp_ctx *c, const void *d_reqs, int32_t n, const int32_t *n_dev, const void *d_extra, int64_t now, void *d_outs)
{
    void *xs[6];
    for (int ph = 1; ph <= 6; ph++) {
        HIP_TRY(c, c->g_x[ph - 1].ensure((size_t)std::max(n, 1) * mmp_shard_xchg_slots(ph, c->n_shards) * 8));
        xs[ph - 1] = c->g_x[ph - 1].p;
    }
    for (int ph = 1; ph <= 7; ph++) {
        {
            std::lock_guard<std::shared_mutex> g(c->mu);
            const int rc = shard_phase_launch(c, ph, d_reqs, n, n_dev, d_extra, now, xs, d_outs, c->stream);
            if (rc != MMP_OK) return rc;
        }
        if (ph <= 6) {
            const int rc = group_allreduce(c, xs[ph - 1], (size_t)n * mmp_shard_xchg_slots(ph, c->n_shards), ncclInt64,
                                           mmp_shard_xchg_is_sum(ph) ? ncclSum : ncclMin);
            if (rc != MMP_OK) return rc;
        }
    }
    return MMP_OK;
}
}  // namespace

int mmp_shard_unique_id(void *id_out)
{
    if (!id_out) return fail(nullptr, MMP_EINVAL, "mmp_shard_unique_id: null argument");
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (!R) return fail(nullptr, MMP_ENODEVICE, "RCCL is not available: %s", why.c_str());
    ncclUniqueId id;
    RCCL_TRY(nullptr, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return MMP_OK;
}

int mmp_shard_group_init(mmp_ctx *c, const void *unique_id, int32_t rank, int32_t world)
try {
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id && !c->xfn))
        return fail(c, MMP_EINVAL, "mmp_shard_group_init: bad argument (a group of more than one shard needs a unique id or an exchange callback)");
    std::lock_guard<std::mutex> gg(c->group_mu);
    if (c->group) return fail(c, MMP_ESTATE, "mmp_shard_group_init: the context already belongs to a group");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (unique_id) {
        std::string why;
        RcclApi *R = rccl_api(&why);
        if (!R) return fail(c, MMP_ENODEVICE, "RCCL is not available on this host: %s", why.c_str());
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        RCCL_TRY(c, R->CommInitRank(&c->comm, world, id, rank));
    }
    const int rc = mmp_shard_configure(c, rank, world);
    if (rc != MMP_OK) {
        if (c->comm) (void)rccl_api()->CommDestroy(c->comm);
        c->comm = nullptr;
        return rc;
    }
    c->g_rank = rank;
    c->g_world = world;
    c->group = true;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_group_init");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_group_init", e.what());
}

int mmp_shard_group_set_exchange(mmp_ctx *c, mmp_exchange_fn fn, void *user)
{
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gg(c->group_mu);
    if (c->group) return fail(c, MMP_ESTATE, "mmp_shard_group_set_exchange: set the callback before mmp_shard_group_init");
    c->xfn = fn;
    c->xuser = user;
    return MMP_OK;
}

namespace {
int shard_resolve_pending(mmp_ctx *c);
}
int mmp_shard_group_destroy(mmp_ctx *c)
{
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gg(c->group_mu);
    if (!c->group) return MMP_OK;
    (void)hipSetDevice(c->cfg.device);
    (void)shard_resolve_pending(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl_api()->CommDestroy(c->comm);
    c->comm = nullptr;
    c->group = false;
    return MMP_OK;
}

/* Sharded commit, collective over the group: every shard ranks its slice of the instances against all of them,
 * ncclAllReduce(SUM) of int32 rank[P] on the library's stream, every shard scatters the positions it owns. */
int mmp_shard_commit(mmp_ctx *c)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gg(c->group_mu);
    if (!c->group) return fail(c, MMP_ESTATE, "mmp_shard_commit: call mmp_shard_group_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    {
        const int rc0 = shard_resolve_pending(c);  // an open asynchronous batch is decided against the snapshot it was issued on
        if (rc0 != MMP_OK) return rc0;
    }
    size_t P;
    {
        std::lock_guard<std::shared_mutex> g(c->mu);
        P = c->pods.size();
    }
    HIP_TRY(c, c->g_rankbuf.ensure(std::max<size_t>(P, 1) * 4));
    int rc = mmp_shard_rank_dev(c, c->g_rankbuf.p);  // synchronises c->stream
    if (rc != MMP_OK) return rc;
    rc = group_allreduce(c, c->g_rankbuf.p, P, ncclInt32, ncclSum);
    if (rc != MMP_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return mmp_shard_commit_dev(c, c->g_rankbuf.p);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_commit");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_commit", e.what());
}

/* One batch on the pod-axis group, device pointers, collective: returns when d_outs is complete on every shard.
 * fast kernel -> ncclAllReduce(MIN, 2 int64 per decision) -> decided rows + compaction of the rest -> the six
 * exchange phases over a sub-batch of fixed CAPACITY (max(1024, n / 16) rows; the real count stays on the device, so
 * no host round trip sits between the kernels and every collective has a size the host knows) -> scatter.  Only
 * when more decisions than that capacity need the six phases — the host learns it with the results — are they run
 * again at their exact size. */
namespace {
int shard_place_batch_locked(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs, int32_t *n_rest_out,
                             bool wait);
int shard_resolve_pending(mmp_ctx *c);
}
int mmp_shard_place_batch_dev(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs,
                              int32_t *n_rest_out)
try {
    if (!c || n < 0 || (n > 0 && (!d_reqs || !d_outs))) return fail(c, MMP_EINVAL, "mmp_shard_place_batch_dev: bad argument");
    std::lock_guard<std::mutex> gg(c->group_mu);
    return shard_place_batch_locked(c, d_reqs, n, d_extra, now, d_outs, n_rest_out, true);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_batch_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_batch_dev", e.what());
}

namespace {
// the batch of mmp_shard_place_batch_async_dev that is still open: wait for its finish kernel, read the rest count, run the six
// phases over the rest if there is one.  The caller holds c->group_mu.  n_rest identical on every shard (the same reduced
// words): the group stays in step.
int shard_resolve_pending(mmp_ctx *c)
{
    if (!c->pend.active) return MMP_OK;
    const mmp_ctx::ShardPending P = c->pend;
    c->pend.active = false;
    hipStream_t st = c->stream;
    int32_t n_rest = 0;
    // the count is polled WITHOUT the state lock (every decision path holds it shared: a spin under the exclusive lock stalled them
    // for the whole wait); only the gather's buffers need it
    int rc = shard_finish_collect(c, P.slot, st, P.seq, &n_rest);
    if (rc == MMP_OK && n_rest > 0) {
        std::lock_guard<std::shared_mutex> g(c->mu);
        rc = shard_rest_gather(c, P.slot, P.d_reqs, P.n, n_rest, st);
    }
    if (rc != MMP_OK) return rc;
    c->last_n_rest = n_rest;
    if (n_rest > 0) {
        rc = group_general(c, c->f_reqs.p, n_rest, nullptr, P.d_extra, P.now, c->f_outs.p);
        if (rc != MMP_OK) return rc;
        hipLaunchKernelGGL(place_shard_scatter_kernel, dim3(div_up(n_rest, 256)), dim3(256), 0, st, c->f_outs.as<mmp_place_out>(),
                           c->f_idx.as<int32_t>(), n_rest, static_cast<mmp_place_out *>(P.d_outs), nullptr);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(st));
    }
    return MMP_OK;
}

// the body of the group's batch calls; the caller holds c->group_mu (and nothing that is ordered behind it).  wait = false:
// returns with the fast kernel, the all-reduce and the finish kernel enqueued (mmp_shard_place_batch_async_dev)
int shard_place_batch_locked(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs, int32_t *n_rest_out,
                             bool wait = true)
{
    if (!c->group) return fail(c, MMP_ESTATE, "mmp_shard_place_batch_dev: call mmp_shard_group_init first");
    if (n_rest_out) *n_rest_out = 0;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if (n == 0) {
        rc = shard_resolve_pending(c);
        c->last_n_rest = rc == MMP_OK ? 0 : c->last_n_rest;
        return rc;
    }
    // This batch is enqueued FIRST, on the buffers of the other slot; the batch before it — whose finish kernel has had the time
    // of these three launches to end — is completed afterwards, and whatever its rest needs is enqueued behind this batch.
    hipStream_t st = c->stream;
    const uint32_t slot = c->f_slot ^= 1u;
    uint32_t seq = 0;
    if (c->n_shards == 1) {
        // a group of one shard: the slice's answer IS the answer — one launch writes rows, rest flags and count (no exchange words,
        // no all-reduce, no finish kernel)
        std::lock_guard<std::shared_mutex> g(c->mu);
        if (!c->committed) return fail(c, MMP_ESTATE, "no committed shard snapshot");
        rc = shard_finish_prepare(c, slot, n, st, &seq);
        if (rc != MMP_OK) return rc;
        const PlaceArgs A = shard_args(c, d_reqs, n, d_extra, now, d_outs);
        hipLaunchKernelGGL(place_shard_fast_direct_kernel, dim3(div_up(n + 1, kPlaceBlock)), dim3(kPlaceBlock), (size_t)place_lane_lds(c->sview.T),
                           st, c->sview, A, c->shard, c->ssnap.any_rs, c->f_flags[slot].as<int32_t>(), c->f_cnt[slot].as<unsigned long long>(),
                           c->f_done + 8 * slot, seq);
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, c->g_xf[slot].ensure((size_t)n * kXF * 8));
        rc = mmp_shard_place_fast_dev(c, d_reqs, n, d_extra, now, c->g_xf[slot].p, st);
        if (rc != MMP_OK) return rc;
        rc = group_allreduce(c, c->g_xf[slot].p, (size_t)n * kXF, ncclInt64, ncclMin);
        if (rc != MMP_OK) return rc;
        std::lock_guard<std::shared_mutex> g(c->mu);
        rc = shard_finish_enqueue(c, slot, n, c->g_xf[slot].p, d_outs, st, &seq);
        if (rc != MMP_OK) return rc;
    }
    rc = shard_resolve_pending(c);
    if (rc != MMP_OK) return rc;
    c->pend.slot = slot;
    c->pend.active = true;
    c->pend.d_reqs = d_reqs;
    c->pend.d_extra = d_extra;
    c->pend.d_outs = d_outs;
    c->pend.n = n;
    c->pend.now = now;
    c->pend.seq = seq;
    if (!wait) return MMP_OK;
    // Round 2 kept the rest count on the device and ALWAYS ran the six-exchange protocol over a fixed-capacity sub-batch (seven
    // launches and six collectives that mostly found no rows): 99 us per 100k decisions at one shard against 50 us for the form
    // that asks.
    rc = shard_resolve_pending(c);
    if (rc != MMP_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(st));  // the caller reads d_outs next
    if (n_rest_out) *n_rest_out = c->last_n_rest;
    return MMP_OK;
}
}  // namespace

int mmp_shard_place_batch_async_dev(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs)
try {
    if (!c || n < 0 || (n > 0 && (!d_reqs || !d_outs))) return fail(c, MMP_EINVAL, "mmp_shard_place_batch_async_dev: bad argument");
    std::lock_guard<std::mutex> gg(c->group_mu);
    return shard_place_batch_locked(c, d_reqs, n, d_extra, now, d_outs, nullptr, false);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_batch_async_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_batch_async_dev", e.what());
}

int mmp_shard_wait(mmp_ctx *c, int32_t *n_rest_out)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gg(c->group_mu);
    if (!c->group) return fail(c, MMP_ESTATE, "mmp_shard_wait: call mmp_shard_group_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int rc = shard_resolve_pending(c);
    if (rc != MMP_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_rest_out) *n_rest_out = c->last_n_rest;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_wait");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_wait", e.what());
}

/* The same with host pointers (what a JVM holds): staged through the context's scratch. */
int mmp_shard_place_batch(mmp_ctx *c, const mmp_place_req *reqs, int32_t n, const int32_t *extra_pool, int32_t n_extra,
                          int64_t now, mmp_place_out *outs, int32_t *n_rest_out)
try {
    if (!c || n < 0 || n_extra < 0 || (n > 0 && (!reqs || !outs)) || (n_extra > 0 && !extra_pool))
        return fail(c, MMP_EINVAL, "mmp_shard_place_batch: bad argument");
    for (int32_t i = 0; i < n; i++)
        if (reqs[i].n_extra < 0 || reqs[i].extra_off < 0 || (int64_t)reqs[i].extra_off + reqs[i].n_extra > n_extra)
            return fail(c, MMP_EINVAL, "mmp_shard_place_batch: request %d extra range out of bounds", i);
    if (n_rest_out) *n_rest_out = 0;
    if (n == 0) return MMP_OK;
    // lock order of the group entry points: group_mu, then batch_mu (mmp_shard_commit and mmp_shard_group_init take
    // batch_mu through the calls they make while holding group_mu)
    std::lock_guard<std::mutex> gg(c->group_mu);
    std::lock_guard<std::mutex> gb(c->batch_mu);  // the staging scratch
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_place_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_place_out)));
    HIP_TRY(c, c->s_extra.ensure((size_t)std::max(n_extra, 1) * 4));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_place_req), hipMemcpyHostToDevice, c->stream));
    if (n_extra) HIP_TRY(c, hipMemcpyAsync(c->s_extra.p, extra_pool, (size_t)n_extra * 4, hipMemcpyHostToDevice, c->stream));
    const int rc = shard_place_batch_locked(c, c->s_reqs.p, n, c->s_extra.p, now, c->s_outs.p, n_rest_out, true);
    if (rc != MMP_OK) return rc;
    HIP_TRY(c, copy_sync(c, outs, c->s_outs.p, (size_t)n * sizeof(mmp_place_out), hipMemcpyDeviceToHost));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_shard_place_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_shard_place_batch", e.what());
}

/* ---- submission threads ---------------------------------------------------
 * One host thread needs ~3 us per kernel launch (HIP's launch path), which is more than a 100k-decision batch takes the
 * GPU when several are in flight.  mmp_issue_threads(ctx, n) starts n helper threads that spin on per-helper rings;
 * while they run, mmp_place_batch_dev only validates, appends a descriptor (tens of ns) and returns, and the helpers
 * launch in parallel — every caller stream is pinned to one helper, so launches on one stream keep their order.
 * mmp_issue_flush waits until everything submitted so far has been launched (and reports the first launch error);
 * mmp_stream_retire, mmp_issue_threads and mmp_destroy flush first.  A commit or a registry event does NOT (it holds
 * the state lock the helpers launch under, and it need not): a helper captures the published snapshot when it launches,
 * so a descriptor that is still in a ring when another snapshot is published is decided against that one, and the
 * stream it is bound for was noted at submission, so whoever rewrites state afterwards waits for it. */
struct IssueItem {
    const void *d_reqs, *d_extra;
    void *d_outs;
    hipStream_t st;
    int64_t now;
    int32_t n;
};
struct IssueRing {
    static constexpr uint32_t kCap = 4096;
    IssueItem items[kCap];
    alignas(64) std::atomic<uint32_t> tail{0};  // producers (under push_mu)
    alignas(64) std::atomic<uint32_t> head{0};  // the helper
    std::mutex push_mu;
};
struct IssuePool {
    std::vector<std::thread> th;
    std::vector<IssueRing *> rings;
    std::atomic<bool> stop{false};
    std::atomic<bool> closed{false};  // set (then every push_mu taken once) before the helpers are told to stop: a submitter
                                      // that still holds a reference launches by itself instead of appending
    ~IssuePool()
    {
        for (IssueRing *R : rings) delete R;
    }
    std::atomic<int> first_rc{MMP_OK};
    std::mutex map_mu;
    std::vector<std::pair<hipStream_t, int>> stream_ring;  // caller stream -> helper, assigned round-robin at first sight
    int next_ring = 0;
};

namespace {
void issue_helper(mmp_ctx *c, IssuePool *P, IssueRing *R)
{
    (void)hipSetDevice(c->cfg.device);
    uint32_t idle = 0;
    for (;;) {
        const uint32_t h = R->head.load(std::memory_order_relaxed);
        if (h == R->tail.load(std::memory_order_acquire)) {
            if (P->stop.load(std::memory_order_acquire)) return;
            if (++idle > 64) __builtin_ia32_pause();
            continue;
        }
        idle = 0;
        const IssueItem it = R->items[h % IssueRing::kCap];
        int rc;
        {
            std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue
            rc = !c->committed ? fail(c, MMP_ESTATE, "no committed snapshot")
                               : place_launch(c, it.d_reqs, it.n, it.d_extra, it.now, it.d_outs, it.st);
        }
        if (rc != MMP_OK) {
            int expect = MMP_OK;
            (void)P->first_rc.compare_exchange_strong(expect, rc);
        }
        R->head.store(h + 1, std::memory_order_release);
    }
}

int issue_ring_of(IssuePool *P, hipStream_t st)
{
    std::lock_guard<std::mutex> g(P->map_mu);
    for (auto &kv : P->stream_ring)
        if (kv.first == st) return kv.second;
    const int r = P->next_ring++ % (int)P->rings.size();
    P->stream_ring.emplace_back(st, r);
    return r;
}

void pool_get(mmp_ctx *c, std::shared_ptr<IssuePool> &out)
{
    std::lock_guard<std::mutex> g(c->pool_mu);
    out = c->pool;
}

// everything submitted so far has been handed to HIP (called without the state lock)
int issue_flush_pool(IssuePool *P)
{
    if (!P) return MMP_OK;
    for (IssueRing *R : P->rings)
        while (R->head.load(std::memory_order_acquire) != R->tail.load(std::memory_order_acquire)) __builtin_ia32_pause();
    return P->first_rc.exchange(MMP_OK);
}
int issue_flush(mmp_ctx *c)
{
    std::shared_ptr<IssuePool> P;
    pool_get(c, P);
    return issue_flush_pool(P.get());
}
}  // namespace

int mmp_issue_threads(mmp_ctx *c, int32_t n)
try {
    if (!c || n < 0 || n > 64) return fail(c, MMP_EINVAL, "mmp_issue_threads: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::shared_ptr<IssuePool> old;
    {
        std::lock_guard<std::mutex> g(c->pool_mu);
        old.swap(c->pool);  // new submitters launch by themselves from here on
    }
    if (old) {
        old->closed.store(true, std::memory_order_release);
        for (IssueRing *R : old->rings) std::lock_guard<std::mutex> barrier(R->push_mu);  // appends in flight have landed
        const int rc = issue_flush_pool(old.get());
        old->stop.store(true, std::memory_order_release);
        for (std::thread &t : old->th) t.join();
        old.reset();  // the rings go with the last reference (a submitter may still hold one; it sees `closed`)
        if (rc != MMP_OK) return rc;
    }
    if (n == 0) return MMP_OK;
    auto P = std::make_shared<IssuePool>();
    for (int i = 0; i < n; i++) P->rings.push_back(new IssueRing());
    for (int i = 0; i < n; i++) P->th.emplace_back(issue_helper, c, P.get(), P->rings[i]);
    {
        std::lock_guard<std::mutex> g(c->pool_mu);
        c->pool = P;
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_issue_threads");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_issue_threads", e.what());
}

int mmp_issue_flush(mmp_ctx *c)
{
    if (!c) return MMP_EINVAL;
    const int rc = issue_flush(c);
    return rc == MMP_OK ? MMP_OK : fail(c, rc, "a launch submitted through the issue threads failed (%d)", rc);
}

/* ---- the resident decision kernel: single requests without a launch --------- */
namespace {
constexpr int kResidentFallback = 1;  // "take the launch path" (never an MMP_* code)

// make sure a resident kernel is running on the published snapshot (lock order: state lock, then launch_mu)
int resident_ensure(mmp_ctx *c)
{
    auto &R = c->res;
    std::shared_lock<std::shared_mutex> gs(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return kResidentFallback;
    std::lock_guard<std::mutex> g(R.launch_mu);
    if (R.running && __atomic_load_n(&R.ctl->exited, __ATOMIC_ACQUIRE) != R.generation) return MMP_OK;
    static const bool dbg = getenv("MMP_RESIDENT_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[resident] ensure: running %d exited %u -> (re)launch #%llu\n", (int)R.running.load(), R.ctl->exited,
                     (unsigned long long)R.launches.load() + 1);
    if (R.running) {
        HIP_TRY(c, hipStreamSynchronize(R.stream));  // it left by itself (idle): returns at once
        R.running = false;
        if (dbg) fprintf(stderr, "[resident] previous kernel retired\n");
    }
    {
        uint32_t g2 = R.generation.load(std::memory_order_relaxed) + 1;
        if (g2 == 0) g2 = 1;
        R.generation.store(g2, std::memory_order_release);
    }
    PlaceArgs A{};
    A.models = c->models.as<mmp_model_row>();
    A.rmodels = cur_side(c).rmodels_ok ? cur_side(c).rmodels.as<ResolvedModel>() : nullptr;
    A.wins = c->no_heads ? nullptr : c->sb[c->cur].heads.as<TypeWin>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.n = 1;
    A.n_models = c->n_models;
    A.n_pods_all = c->snap.P;
    HIP_TRY(c, order_after_registry(c, R.stream));
    hipLaunchKernelGGL(place_resident_kernel, dim3(1), dim3(64), (size_t)kPlaceLaneLds, R.stream, c->snap, A, R.slots, R.answers, R.ctl,
                       R.idle_ticks, R.generation.load(std::memory_order_relaxed));
    HIP_TRY(c, hipGetLastError());
    R.running = true;
    R.launches.fetch_add(1, std::memory_order_relaxed);
    if (dbg) fprintf(stderr, "[resident] launched\n");
    return MMP_OK;
}

int resident_place(mmp_ctx *c, const mmp_place_req &rq, int64_t now, mmp_place_out *out)
{
    auto &R = c->res;
    if (R.skip.load(std::memory_order_relaxed) > 0) {
        R.skip.fetch_sub(1, std::memory_order_relaxed);
        return kResidentFallback;
    }
    // a slot of our own for the duration of the call
    const uint32_t first = R.rr.fetch_add(1, std::memory_order_relaxed);
    int si = -1;
    std::unique_lock<std::mutex> sl;
    for (int k = 0; k < kResidentSlots && si < 0; k++) {
        const int cand = (int)((first + k) % kResidentSlots);
        std::unique_lock<std::mutex> t(R.slot_mu[cand], std::try_to_lock);
        if (t.owns_lock()) {
            sl = std::move(t);
            si = cand;
        }
    }
    if (si < 0) {
        si = (int)(first % kResidentSlots);
        sl = std::unique_lock<std::mutex>(R.slot_mu[si]);
    }
    ResidentSlot *S = &R.slots[si];
    ResidentAnswer *Ans = &R.answers[si];
    uint32_t seq = (R.seq[si] + 1) & 0xfffffu;  // 20 bits ride in the bell next to now_ms
    if (seq == 0) seq = 1;
    R.seq[si] = seq;
    S->req = rq;
    __atomic_store_n(&S->bell, ((uint64_t)seq << kResidentNowBits) | ((uint64_t)now & ((1ull << kResidentNowBits) - 1ull)),
                     __ATOMIC_RELEASE);  // the bell: written last
    bool restarted = false;  // this call (re)started the kernel, or waited for a commit to do so: not a sample for the self-check
    if (!R.running || __atomic_load_n(&R.ctl->exited, __ATOMIC_ACQUIRE) == R.generation) {
        const int rc = resident_ensure(c);
        if (rc != MMP_OK) return rc;
        restarted = true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const uint32_t d = __atomic_load_n(&Ans->done, __ATOMIC_ACQUIRE);
        if ((d & ~kResidentPunt) == seq) {
            if (d & kResidentPunt) {
                R.punted.fetch_add(1, std::memory_order_relaxed);
                // a table on which most decisions need the wave path (every instance a candidate ...) makes the resident
                // attempt pure overhead: after 8 hand-backs in a row the next 4096 single requests go to the launch path directly
                if (R.punt_streak.fetch_add(1, std::memory_order_relaxed) + 1 >= 8) {
                    R.punt_streak.store(0, std::memory_order_relaxed);
                    R.skip.store(4096, std::memory_order_relaxed);
                }
                return kResidentFallback;  // a shape the resident wavefront leaves to the launch path
            }
            R.punt_streak.store(0, std::memory_order_relaxed);
            *out = Ans->out;
            R.served.fetch_add(1, std::memory_order_relaxed);
            // Self-check: an answer takes ~10 us.  Three answers IN A ROW slower than 20 ms — from a kernel that was running
            // when the request was posted and was not restarted while it waited (a commit stops it) — switch the resident path
            // off for this context and the launch path takes over (what that looked like before the resident stream got a
            // priority level of its own: 50 ms per call, the time the resident kernel needs to idle out of the way of a launch
            // queued behind it).  A request thread that lost its core for a few milliseconds is not such a sample.
            const bool slow_answer = !restarted && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20);
            if (!slow_answer && !restarted) R.slow.store(0, std::memory_order_relaxed);
            if (slow_answer && R.slow.fetch_add(1) + 1 >= 3) {
                R.enabled = false;
                (void)fail(c, MMP_OK, "resident decision kernel disabled: three answers in a row took longer than 20 ms on this host");
            }
            return MMP_OK;
        }
        if ((spins & 255) == 255) {
            if (__atomic_load_n(&R.ctl->exited, __ATOMIC_ACQUIRE) == R.generation) {  // it left (idle, or stopped for a commit) before it saw the bell
                if (getenv("MMP_RESIDENT_DEBUG"))
                    fprintf(stderr, "[resident] slot %d waits for seq %u: done %#x bell seq %u, spins %u\n", si, seq, Ans->done,
                            (unsigned)(S->bell >> kResidentNowBits), spins);
                const int rc = resident_ensure(c);
                if (rc != MMP_OK) return rc;
                restarted = true;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
                return fail(c, MMP_EHIP, "the resident decision kernel did not answer within 200 ms");
        }
        __builtin_ia32_pause();
    }
}
}  // namespace

int mmp_resident(mmp_ctx *c, int enable)
try {
    if (!c) return MMP_EINVAL;
    std::lock_guard<std::mutex> gb(c->batch_mu);
    auto &R = c->res;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (enable && !R.slots) {
        // A stream of the LOWEST priority: HIP multiplexes the streams of one priority level onto a small pool of hardware
        // queues, and a kernel that never ends blocks whatever else lands on its queue.  On the latency slots' level (highest)
        // it held their launches back until it idled out — 50 ms per request that it had handed back to the launch path
        // (tools/micro/single_calls.cc, SINGLE_CALLS_FLAT=1).  Nothing else in the library uses the lowest level.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_TRY(c, hipStreamCreateWithPriority(&R.stream, hipStreamNonBlocking, lo));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&R.slots), sizeof(ResidentSlot) * kResidentSlots, kPinnedFlags));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&R.ctl), sizeof(ResidentCtl), kPinnedFlags));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&R.answers), sizeof(ResidentAnswer) * kResidentSlots, kPinnedFlags));
        memset(R.answers, 0, sizeof(ResidentAnswer) * kResidentSlots);
        memset(R.slots, 0, sizeof(ResidentSlot) * kResidentSlots);
        memset(R.ctl, 0, sizeof(ResidentCtl));
        int khz = 100000;
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->cfg.device);
        long long ms = 50;
        if (const char *e = getenv("MMP_RESIDENT_IDLE_MS")) ms = std::max(1, atoi(e));
        R.idle_ticks = (long long)khz * ms;
    }
    if (!enable) {
        std::lock_guard<std::shared_mutex> g(c->mu);
        R.enabled = false;
        resident_stop(c);
        return MMP_OK;
    }
    R.enabled = true;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_resident");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_resident", e.what());
}

int mmp_resident_stats(mmp_ctx *c, uint64_t *launches, uint64_t *served, uint64_t *punted)
{
    if (!c) return MMP_EINVAL;
    if (launches) *launches = c->res.launches.load();
    if (served) *served = c->res.served.load();
    if (punted) *punted = c->res.punted.load();
    return MMP_OK;
}

/* ---- decisions ---------------------------------------------------------- */

int mmp_place_multi_dev(mmp_ctx *c, int32_t k, const void *const *d_reqs, const int32_t *n, const void *const *d_extra, int64_t now,
                        void *const *d_outs, void *stream)
try {
    if (!c || k < 0 || (k > 0 && (!d_reqs || !n || !d_outs))) return fail(c, MMP_EINVAL, "mmp_place_multi_dev: bad argument");
    for (int32_t i = 0; i < k; i++)
        if (n[i] < 0 || (n[i] > 0 && (!d_reqs[i] || !d_outs[i]))) return fail(c, MMP_EINVAL, "mmp_place_multi_dev: bad argument (array %d)", i);
    {  // launches handed to submission threads earlier are issued first: this call launches from the calling thread, in stream order
        std::shared_ptr<IssuePool> P;
        pool_get(c, P);
        if (P) {
            const int rc = issue_flush(c);
            if (rc != MMP_OK) return fail(c, rc, "a launch submitted through the issue threads failed (%d)", rc);
        }
    }
    std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue; no wait
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
    hipStream_t st = static_cast<hipStream_t>(stream);
    note_caller_stream(c, st);
    for (int32_t i0 = 0; i0 < k;) {  // kMaxSegs arrays per launch
        PlaceSegs G{};
        int32_t blocks = 0;
        int64_t total = 0;
        int32_t i = i0;
        for (; i < k && G.n_segs < kMaxSegs; i++) {
            if (n[i] == 0) continue;
            PlaceSeg &sg = G.seg[G.n_segs++];
            sg.reqs = static_cast<const mmp_place_req *>(d_reqs[i]);
            sg.outs = static_cast<mmp_place_out *>(d_outs[i]);
            sg.extra = d_extra ? static_cast<const int32_t *>(d_extra[i]) : nullptr;
            sg.first_block = blocks;
            sg.n = n[i];
            blocks += div_up(n[i], kPlaceBlock);
            total += n[i];
        }
        i0 = i;
        if (G.n_segs == 0) continue;
        if (total > INT32_MAX - kPlaceBlock * (int64_t)kMaxSegs) return fail(c, MMP_EINVAL, "mmp_place_multi_dev: more than 2^31 decisions in one launch");
        const int rc = place_launch(c, G.seg[0].reqs, (int32_t)total, G.seg[0].extra, now, G.seg[0].outs, st, nullptr, 0, nullptr, nullptr, &G, blocks);
        if (rc != MMP_OK) return rc;
    }
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_multi_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_multi_dev", e.what());
}

int mmp_place_batch_dev(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int64_t now, void *d_outs,
                        void *stream)
try {
    if (!c || n < 0 || (n > 0 && (!d_reqs || !d_outs))) return fail(c, MMP_EINVAL, "mmp_place_batch_dev: bad argument");
    std::shared_ptr<IssuePool> P;
    pool_get(c, P);
    if (P) {  // submission threads: append and return
        if (n == 0) return MMP_OK;
        hipStream_t st = static_cast<hipStream_t>(stream);
        IssueRing *R = P->rings[issue_ring_of(P.get(), st)];
        {
            std::shared_lock<std::shared_mutex> g(c->mu);
            if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
            if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
            note_caller_stream(c, st);
        }
        std::lock_guard<std::mutex> gp(R->push_mu);
        if (!P->closed.load(std::memory_order_acquire)) {
            const uint32_t t = R->tail.load(std::memory_order_relaxed);
            while (t - R->head.load(std::memory_order_acquire) >= IssueRing::kCap) __builtin_ia32_pause();
            R->items[t % IssueRing::kCap] = IssueItem{d_reqs, d_extra, d_outs, st, now, n};
            R->tail.store(t + 1, std::memory_order_release);
            return MMP_OK;
        }
        // the pool was retired while this call held its reference: launch here, like a context without helpers
    }
    std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue; no wait
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
    note_caller_stream(c, static_cast<hipStream_t>(stream));
    return place_launch(c, d_reqs, n, d_extra, now, d_outs, static_cast<hipStream_t>(stream));
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_batch_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_batch_dev", e.what());
}

int mmp_stream_retire(mmp_ctx *c, void *stream)
try {
    if (!c) return MMP_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)issue_flush(c);  // launches submitted for this stream have reached it
    {
        std::lock_guard<std::mutex> g(c->cs_mu);
        auto it = std::find(c->caller_streams.begin(), c->caller_streams.end(), st);
        if (it == c->caller_streams.end()) return MMP_OK;
        c->caller_streams.erase(it);
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(st));  // what was enqueued on it has finished reading the library's tables
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_stream_retire");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_stream_retire", e.what());
}

int mmp_place_batch_dev2(mmp_ctx *c, const void *d_reqs, int32_t n, const void *d_extra, int32_t n_extra_pool, int64_t now,
                         void *d_outs, void *stream)
try {
    if (!c || n < 0 || n_extra_pool < 0 || (n > 0 && (!d_reqs || !d_outs)) || (n_extra_pool > 0 && !d_extra))
        return fail(c, MMP_EINVAL, "mmp_place_batch_dev2: bad argument");
    std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue; no wait
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
    note_caller_stream(c, static_cast<hipStream_t>(stream));
    PlaceOpts o;
    o.extra_bound = n_extra_pool + 1;
    return place_launch(c, d_reqs, n, d_extra, now, d_outs, static_cast<hipStream_t>(stream), nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr,
                        nullptr, &o);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_batch_dev2");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_batch_dev2", e.what());
}

int mmp_place_batch_c_dev(mmp_ctx *c, const mmp_place_caller *caller, const void *d_reqs, int32_t n, const void *d_extra,
                          int32_t n_extra_pool, int64_t now, void *d_outs, void *stream)
try {
    if (!c || !caller || n < 0 || n_extra_pool < 0 || (n > 0 && (!d_reqs || !d_outs)) || (n_extra_pool > 0 && !d_extra))
        return fail(c, MMP_EINVAL, "mmp_place_batch_c_dev: bad argument");
    std::shared_lock<std::shared_mutex> g(c->mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
    note_caller_stream(c, static_cast<hipStream_t>(stream));
    PlaceOpts o;
    o.caller = caller;
    o.extra_bound = n_extra_pool + 1;
    return place_launch(c, d_reqs, n, d_extra, now, d_outs, static_cast<hipStream_t>(stream), nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr,
                        nullptr, &o);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_batch_c_dev");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_batch_c_dev", e.what());
}

int mmp_place_batch(mmp_ctx *c, const mmp_place_req *reqs, int32_t n, const int32_t *extra_pool, int32_t n_extra,
                    int64_t now, mmp_place_out *outs);
int mmp_place_batch_c(mmp_ctx *c, const mmp_place_caller *caller, const mmp_place_req_c *reqs, int32_t n, const int32_t *extra_pool,
                      int32_t n_extra, int64_t now, mmp_place_out *outs)
try {
    if (!c || !caller || n < 0 || n_extra < 0 || (n > 0 && (!reqs || !outs)) || (n_extra > 0 && !extra_pool))
        return fail(c, MMP_EINVAL, "mmp_place_batch_c: bad argument");
    for (int32_t i = 0; i < n; i++)
        if (reqs[i].n_extra < 0 || reqs[i].extra_off < 0 || (int64_t)reqs[i].extra_off + reqs[i].n_extra > n_extra)
            return fail(c, MMP_EINVAL, "mmp_place_batch_c: request %d extra range out of bounds", i);
    if (n <= kFastN && n_extra <= kFastExtra) {
        // a handful of requests: the latency path of mmp_place_batch (slots, the single-decision kernels, the resident kernel) on
        // the same decisions written out as mmp_place_req rows
        std::vector<mmp_place_req> full((size_t)n);
        for (int32_t i = 0; i < n; i++) {
            mmp_place_req &r = full[(size_t)i];
            r.model = reqs[i].model;
            r.self_pod = caller->self_pod;
            r.flags = caller->flags;
            r.pick = reqs[i].pick;
            r.last_used = reqs[i].last_used;
            r.extra_off = reqs[i].extra_off;
            r.n_extra = reqs[i].n_extra;
            r.fresh_lru = caller->fresh_lru;
            r.fresh_capacity = caller->fresh_capacity;
            r.fresh_used = caller->fresh_used;
            r.fresh_count = caller->fresh_count;
            r.fresh_rpm = caller->fresh_rpm;
        }
        return mmp_place_batch(c, full.data(), n, extra_pool, n_extra, now, outs);
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::lock_guard<std::mutex> gb(c->batch_mu);
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_place_req_c)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_place_out)));
    HIP_TRY(c, c->s_extra.ensure((size_t)std::max(n_extra, 1) * 4));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_place_req_c), hipMemcpyHostToDevice, st));
    if (n_extra) HIP_TRY(c, hipMemcpyAsync(c->s_extra.p, extra_pool, (size_t)n_extra * 4, hipMemcpyHostToDevice, st));
    {
        std::lock_guard<std::shared_mutex> g(c->mu);
        if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
        if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
        PlaceOpts o;
        o.caller = caller;
        KT_BEGIN(c, st);
        const int rc = place_launch(c, c->s_reqs.p, n, c->s_extra.p, now, c->s_outs.p, st, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr,
                                    nullptr, &o);
        if (rc != MMP_OK) return rc;
        KT_END(c, st);
    }
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_place_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_batch_c");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_batch_c", e.what());
}

int mmp_place_batch(mmp_ctx *c, const mmp_place_req *reqs, int32_t n, const int32_t *extra_pool, int32_t n_extra,
                    int64_t now, mmp_place_out *outs)
try {
    if (!c || n < 0 || n_extra < 0 || (n > 0 && (!reqs || !outs)) || (n_extra > 0 && !extra_pool))
        return fail(c, MMP_EINVAL, "mmp_place_batch: bad argument");
    for (int32_t i = 0; i < n; i++)
        if (reqs[i].n_extra < 0 || reqs[i].extra_off < 0 || (int64_t)reqs[i].extra_off + reqs[i].n_extra > n_extra)
            return fail(c, MMP_EINVAL, "mmp_place_batch: request %d extra range out of bounds", i);
    if (n == 0) {
        std::lock_guard<std::shared_mutex> g(c->mu);
        if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
        if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
        return MMP_OK;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));

    if (n == 1 && reqs[0].n_extra == 0 && c->res.enabled && now >= 0 && (now >> kResidentNowBits) == 0) {
        // a single request without exclusions of its own: the resident kernel, no launch at all
        const int rc = resident_place(c, reqs[0], now, outs);
        if (rc != kResidentFallback) return rc;
    }
    if (n <= kFastN && n_extra <= kFastExtra) {
        // latency path: the kernel reads the requests from, and writes the results to, pinned host
        // memory over the fabric — no staging copies, no contention with batches on c->stream
        static const bool trace = getenv("MMP_TRACE_SLOT") != nullptr;  // where a single decision's wall time goes (stderr)
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> fl;
        FastSlot *f = slot_acquire(c, fl);
        memcpy(f->reqs, reqs, (size_t)n * sizeof(mmp_place_req));
        if (n_extra) memcpy(f->extra, extra_pool, (size_t)n_extra * sizeof(int32_t));
        const auto t1 = std::chrono::steady_clock::now();
        {
            std::shared_lock<std::shared_mutex> g(c->mu);
            if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
            if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
            const int rc = place_launch(c, f->reqs, n, f->extra, now, f->outs, f->stream, f->done, ++f->seq,
                                        (n == 1 && reqs[0].n_extra == 0) ? &reqs[0] : nullptr, f->blocks);
            if (rc != MMP_OK) return rc;
        }
        const auto t2 = std::chrono::steady_clock::now();
        HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
        memcpy(outs, f->outs, (size_t)n * sizeof(mmp_place_out));
        if (trace) {
            static thread_local double acc[3] = {0, 0, 0};
            static thread_local int cnt = 0;
            const auto t3 = std::chrono::steady_clock::now();
            acc[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
            acc[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
            acc[2] += std::chrono::duration<double, std::micro>(t3 - t2).count();
            if (++cnt == 2000) {
                fprintf(stderr, "[slot path] per call: slot + copy %.2f us, lock + launch %.2f us, wait for the flag %.2f us\n", acc[0] / cnt,
                        acc[1] / cnt, acc[2] / cnt);
                acc[0] = acc[1] = acc[2] = 0;
                cnt = 0;
            }
        }
        return MMP_OK;
    }

    std::lock_guard<std::mutex> gb(c->batch_mu);
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_place_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_place_out)));
    HIP_TRY(c, c->s_extra.ensure((size_t)std::max(n_extra, 1) * 4));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_place_req), hipMemcpyHostToDevice, st));
    if (n_extra) HIP_TRY(c, hipMemcpyAsync(c->s_extra.p, extra_pool, (size_t)n_extra * 4, hipMemcpyHostToDevice, st));
    {
        std::lock_guard<std::shared_mutex> g(c->mu);
        if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
        if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard: use mmp_shard_place_phase_dev");
        KT_BEGIN(c, st);
        const int rc = place_launch(c, c->s_reqs.p, n, c->s_extra.p, now, c->s_outs.p, st);
        if (rc != MMP_OK) return rc;
        KT_END(c, st);
    }
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_place_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_place_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_place_batch", e.what());
}

int mmp_serve_batch(mmp_ctx *c, const mmp_serve_req *reqs, int32_t n, const mmp_serve_counter *counters, int32_t n_counters,
                    const int32_t *excl_pod, const int64_t *excl_time, int32_t n_excl, int64_t now, mmp_serve_out *outs)
try {
    if (!c || n < 0 || n_excl < 0 || n_counters < 0 || (n > 0 && (!reqs || !outs)) || (n_counters > 0 && !counters) ||
        (n_excl > 0 && (!excl_pod || !excl_time)))
        return fail(c, MMP_EINVAL, "mmp_serve_batch: bad argument");
    for (int32_t i = 0; i < n; i++) {
        if (reqs[i].n_excl < 0 || reqs[i].excl_off < 0 || (int64_t)reqs[i].excl_off + reqs[i].n_excl > n_excl)
            return fail(c, MMP_EINVAL, "mmp_serve_batch: request %d exclude range out of bounds", i);
        if (reqs[i].n_cnt < 0 || reqs[i].cnt_off < 0 || (int64_t)reqs[i].cnt_off + reqs[i].n_cnt > n_counters)
            return fail(c, MMP_EINVAL, "mmp_serve_batch: request %d counter range out of bounds", i);
    }
    auto args = [&](int32_t cnt) {
        ServeArgs A{};
        A.models = c->models.as<mmp_model_row>();
        A.ent_pod = c->ent_pod.as<int32_t>();
        A.ent_time = c->ent_time.as<int64_t>();
        A.n = cnt;
        A.n_models = c->n_models;
        A.P = c->snap.P;
        A.now = now;
        A.done = DoneFlag{nullptr, nullptr, 0};
        return A;
    };
    // latency path (see slot_acquire): the slot's pinned buffers take the requests, the counters behind them, and the
    // exclusion pairs in the int pool; one launch + the completion flag, no staging copies — what the LB's one call per
    // request (MM.java:4315) needs.  Everything a call brings is O(copies): 48 B + 16 B per listed copy.
    constexpr int kSlotReqs = 1024, kSlotCounters = 8192, kSlotExcl = kFastExtra / 4;
    static_assert((size_t)kSlotReqs * sizeof(mmp_serve_req) + (size_t)kSlotCounters * sizeof(mmp_serve_counter) <= (size_t)kFastN * sizeof(mmp_place_req),
                  "requests + counters fit the slot's request buffer");
    static_assert((size_t)kSlotReqs * sizeof(mmp_serve_out) <= (size_t)kFastN * sizeof(mmp_place_out), "results fit the slot's result buffer");
    if (n > 0 && n <= kSlotReqs && n_counters <= kSlotCounters && n_excl <= kSlotExcl) {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        std::unique_lock<std::mutex> fl;
        FastSlot *f = slot_acquire(c, fl);
        char *rb = reinterpret_cast<char *>(f->reqs);
        memcpy(rb, reqs, (size_t)n * sizeof(mmp_serve_req));
        mmp_serve_counter *cb = reinterpret_cast<mmp_serve_counter *>(rb + (size_t)kSlotReqs * sizeof(mmp_serve_req));
        if (n_counters) memcpy(cb, counters, (size_t)n_counters * sizeof(mmp_serve_counter));
        int32_t *pool = f->extra;
        if (n_excl) {
            memcpy(pool, excl_pod, (size_t)n_excl * 4);
            memcpy(pool + 2 * kSlotExcl, excl_time, (size_t)n_excl * 8);
        }
        {
            std::shared_lock<std::shared_mutex> g(c->mu);  // capture the registry view + enqueue
            if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
            ServeArgs A = args(n);
            A.reqs = reinterpret_cast<const mmp_serve_req *>(rb);
            A.counters = cb;
            A.excl_pod = pool;
            A.excl_time = reinterpret_cast<const int64_t *>(pool + 2 * kSlotExcl);
            A.outs = reinterpret_cast<mmp_serve_out *>(f->outs);
            A.done = DoneFlag{f->done, f->blocks, ++f->seq};
            HIP_TRY(c, order_after_registry(c, f->stream));
            hipLaunchKernelGGL(serve_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, f->stream, A);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
        memcpy(outs, f->outs, (size_t)n * sizeof(mmp_serve_out));
        return MMP_OK;
    }
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_serve_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_serve_out)));
    HIP_TRY(c, c->s_a.ensure((size_t)std::max(n_counters, 1) * sizeof(mmp_serve_counter)));
    HIP_TRY(c, c->s_c.ensure((size_t)std::max(n_excl, 1) * 4));
    HIP_TRY(c, c->s_d.ensure((size_t)std::max(n_excl, 1) * 8));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_serve_req), hipMemcpyHostToDevice, st));
    if (n_counters) HIP_TRY(c, hipMemcpyAsync(c->s_a.p, counters, (size_t)n_counters * sizeof(mmp_serve_counter), hipMemcpyHostToDevice, st));
    if (n_excl) {
        HIP_TRY(c, hipMemcpyAsync(c->s_c.p, excl_pod, (size_t)n_excl * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->s_d.p, excl_time, (size_t)n_excl * 8, hipMemcpyHostToDevice, st));
    }
    ServeArgs A = args(n);
    A.reqs = c->s_reqs.as<mmp_serve_req>();
    A.counters = c->s_a.as<mmp_serve_counter>();
    A.excl_pod = c->s_c.as<int32_t>();
    A.excl_time = c->s_d.as<int64_t>();
    A.outs = c->s_outs.as<mmp_serve_out>();
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(serve_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, A);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_serve_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_serve_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_serve_batch", e.what());
}

namespace {
GateArgs gate_args(mmp_ctx *c, int32_t n, int64_t now, int64_t in_use_expiry)
{
    GateArgs A{};
    A.models = c->models.as<mmp_model_row>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.ent_time = c->ent_time.as<int64_t>();
    A.pods = c->sb[c->cur].pods.as<mmp_pod_row>();
    A.allowed = cur_side(c).d_allowed.as<uint64_t>();
    A.has_allowed = cur_side(c).d_has_allowed.as<uint8_t>();
    A.stats = cur_side(c).stats_acc.as<StatsAcc>();
    A.tstats = cur_side(c).tstats.as<StatsAcc>();
    A.T_rows = std::max(c->n_types, 1);
    A.n = n;
    A.n_models = c->n_models;
    A.P = c->snap.P;
    A.W = c->snap.W;
    A.T = c->n_types;
    A.now = now;
    A.in_use_expiry = in_use_expiry;
    A.min_space = c->cfg.min_space_units;
    A.min_churn = c->cfg.min_churn_age_ms;
    A.done = DoneFlag{nullptr, nullptr, 0};
    return A;
}
}  // namespace

int mmp_gate_batch(mmp_ctx *c, const mmp_gate_req *reqs, int32_t n, const int32_t *excl_pod, const int64_t *excl_time,
                   int32_t n_excl, const int32_t *explicit_pool, int32_t n_explicit, int64_t now, int64_t in_use_expiry,
                   mmp_gate_out *outs)
try {
    if (!c || n < 0 || n_excl < 0 || n_explicit < 0 || (n > 0 && (!reqs || !outs)) ||
        (n_excl > 0 && (!excl_pod || !excl_time)) || (n_explicit > 0 && !explicit_pool))
        return fail(c, MMP_EINVAL, "mmp_gate_batch: bad argument");
    for (int32_t i = 0; i < n; i++) {
        const mmp_gate_req &r = reqs[i];
        if (r.n_excl < 0 || r.excl_off < 0 || (int64_t)r.excl_off + r.n_excl > n_excl || r.n_explicit < 0 ||
            r.explicit_off < 0 || (int64_t)r.explicit_off + r.n_explicit > n_explicit)
            return fail(c, MMP_EINVAL, "mmp_gate_batch: request %d pool range out of bounds", i);
    }
    constexpr int kGatePool = kFastExtra / 4;  // the slot's int pool holds excl_pod | explicit | excl_time (as int64)
    if (n > 0 && (size_t)n * sizeof(mmp_gate_req) <= kFastN * sizeof(mmp_place_req) &&
        (size_t)n * sizeof(mmp_gate_out) <= kFastN * sizeof(mmp_place_out) && n_excl <= kGatePool && n_explicit <= kGatePool) {
        // latency path (see slot_acquire)
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        std::unique_lock<std::mutex> fl;
        FastSlot *f = slot_acquire(c, fl);
        memcpy(f->reqs, reqs, (size_t)n * sizeof(mmp_gate_req));
        int32_t *pool = f->extra;
        if (n_excl) {
            memcpy(pool, excl_pod, (size_t)n_excl * 4);
            memcpy(pool + 2 * kGatePool, excl_time, (size_t)n_excl * 8);
        }
        if (n_explicit) memcpy(pool + kGatePool, explicit_pool, (size_t)n_explicit * 4);
        {
            std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue
            if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
            GateArgs A = gate_args(c, n, now, in_use_expiry);
            A.reqs = reinterpret_cast<const mmp_gate_req *>(f->reqs);
            A.excl_pod = pool;
            A.explicit_pool = pool + kGatePool;
            A.excl_time = reinterpret_cast<const int64_t *>(pool + 2 * kGatePool);
            A.outs = reinterpret_cast<mmp_gate_out *>(f->outs);
            A.done = DoneFlag{f->done, f->blocks, ++f->seq};
            HIP_TRY(c, order_after_registry(c, f->stream));
            hipLaunchKernelGGL(gate_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, f->stream, A);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
        memcpy(outs, f->outs, (size_t)n * sizeof(mmp_gate_out));
        return MMP_OK;
    }
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_gate_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_gate_out)));
    HIP_TRY(c, c->s_a.ensure((size_t)std::max(n_explicit, 1) * 4));
    HIP_TRY(c, c->s_c.ensure((size_t)std::max(n_excl, 1) * 4));
    HIP_TRY(c, c->s_d.ensure((size_t)std::max(n_excl, 1) * 8));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_gate_req), hipMemcpyHostToDevice, st));
    if (n_explicit) HIP_TRY(c, hipMemcpyAsync(c->s_a.p, explicit_pool, (size_t)n_explicit * 4, hipMemcpyHostToDevice, st));
    if (n_excl) {
        HIP_TRY(c, hipMemcpyAsync(c->s_c.p, excl_pod, (size_t)n_excl * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->s_d.p, excl_time, (size_t)n_excl * 8, hipMemcpyHostToDevice, st));
    }
    GateArgs A = gate_args(c, n, now, in_use_expiry);
    A.reqs = c->s_reqs.as<mmp_gate_req>();
    A.excl_pod = c->s_c.as<int32_t>();
    A.excl_time = c->s_d.as<int64_t>();
    A.explicit_pool = c->s_a.as<int32_t>();
    A.outs = c->s_outs.as<mmp_gate_out>();
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(gate_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, A);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_gate_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_gate_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_gate_batch", e.what());
}

int mmp_miss_batch(mmp_ctx *c, const mmp_gate_req *greqs, const mmp_place_req *preqs, int32_t n, const int32_t *excl_pod,
                   const int64_t *excl_time, int32_t n_excl, const int32_t *explicit_pool, int32_t n_explicit, const int32_t *extra_pool,
                   int32_t n_extra, int64_t now, int64_t in_use_expiry, mmp_gate_out *gouts, mmp_place_out *pouts)
try {
    if (!c || n < 0 || n_excl < 0 || n_explicit < 0 || n_extra < 0 || (n > 0 && (!greqs || !preqs || !gouts || !pouts)) ||
        (n_excl > 0 && (!excl_pod || !excl_time)) || (n_explicit > 0 && !explicit_pool) || (n_extra > 0 && !extra_pool))
        return fail(c, MMP_EINVAL, "mmp_miss_batch: bad argument");
    for (int32_t i = 0; i < n; i++) {
        const mmp_gate_req &g = greqs[i];
        const mmp_place_req &r = preqs[i];
        if (g.n_excl < 0 || g.excl_off < 0 || (int64_t)g.excl_off + g.n_excl > n_excl || g.n_explicit < 0 || g.explicit_off < 0 ||
            (int64_t)g.explicit_off + g.n_explicit > n_explicit)
            return fail(c, MMP_EINVAL, "mmp_miss_batch: request %d pool range out of bounds", i);
        if (r.n_extra < 0 || r.extra_off < 0 || (int64_t)r.extra_off + r.n_extra > n_extra)
            return fail(c, MMP_EINVAL, "mmp_miss_batch: request %d extra range [%d, +%d) outside the pool of %d", i, r.extra_off, r.n_extra, n_extra);
        if (g.model != r.model) return fail(c, MMP_EINVAL, "mmp_miss_batch: request %d names two models", i);
    }
    if (n == 0) return MMP_OK;
    // One latency slot, TWO launches on its stream (the guards, then the load targets: in stream order, so the second kernel's
    // completion flag covers both), ONE wait: what two calls did in two launch-wait-return round trips.
    constexpr int kMissN = 256, kMissPool = kFastExtra / 8;  // place extras | gate excl_pod | gate explicit | gate excl_time (2 ints each)
    constexpr size_t kMissGreqOff = (size_t)kMissN * sizeof(mmp_place_req), kMissGoutOff = (size_t)kMissN * sizeof(mmp_place_out);
    static_assert(kMissGreqOff + (size_t)kMissN * sizeof(mmp_gate_req) <= (size_t)kFastN * sizeof(mmp_place_req), "miss slot layout");
    static_assert(kMissGoutOff + (size_t)kMissN * sizeof(mmp_gate_out) <= (size_t)kFastN * sizeof(mmp_place_out), "miss slot results");
    if (n > kMissN || n_extra > kMissPool || n_excl > kMissPool || n_explicit > kMissPool) {  // large batches: the two calls
        const int rc = mmp_gate_batch(c, greqs, n, excl_pod, excl_time, n_excl, explicit_pool, n_explicit, now, in_use_expiry, gouts);
        return rc != MMP_OK ? rc : mmp_place_batch(c, preqs, n, extra_pool, n_extra, now, pouts);
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::unique_lock<std::mutex> fl;
    FastSlot *f = slot_acquire(c, fl);
    char *base = reinterpret_cast<char *>(f->reqs), *obase = reinterpret_cast<char *>(f->outs);
    memcpy(base, preqs, (size_t)n * sizeof(mmp_place_req));
    memcpy(base + kMissGreqOff, greqs, (size_t)n * sizeof(mmp_gate_req));
    int32_t *pool = f->extra;
    if (n_extra) memcpy(pool, extra_pool, (size_t)n_extra * 4);
    if (n_excl) {
        memcpy(pool + kMissPool, excl_pod, (size_t)n_excl * 4);
        memcpy(pool + 3 * kMissPool, excl_time, (size_t)n_excl * 8);
    }
    if (n_explicit) memcpy(pool + 2 * kMissPool, explicit_pool, (size_t)n_explicit * 4);
    {
        std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue
        if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
        if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard");
        GateArgs G = gate_args(c, n, now, in_use_expiry);
        G.reqs = reinterpret_cast<const mmp_gate_req *>(base + kMissGreqOff);
        G.excl_pod = pool + kMissPool;
        G.explicit_pool = pool + 2 * kMissPool;
        G.excl_time = reinterpret_cast<const int64_t *>(pool + 3 * kMissPool);
        G.outs = reinterpret_cast<mmp_gate_out *>(obase + kMissGoutOff);
        G.done = DoneFlag{nullptr, nullptr, 0};
        const bool one = n == 1 && preqs[0].n_extra == 0;
        if (one) {  // ONE launch: both requests in the kernel arguments, guards and load target on two wavefronts
            const int rc = place_launch(c, f->reqs, n, pool, now, f->outs, f->stream, f->done, ++f->seq, &preqs[0], f->blocks, nullptr, 0, &G,
                                        &greqs[0]);
            if (rc != MMP_OK) return rc;
        } else {
            HIP_TRY(c, order_after_registry(c, f->stream));
            hipLaunchKernelGGL(gate_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, f->stream, G);
            HIP_TRY(c, hipGetLastError());
            const int rc = place_launch(c, f->reqs, n, pool, now, f->outs, f->stream, f->done, ++f->seq,
                                        (n == 1 && preqs[0].n_extra == 0) ? &preqs[0] : nullptr, f->blocks);
            if (rc != MMP_OK) return rc;
        }
    }
    HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
    memcpy(pouts, obase, (size_t)n * sizeof(mmp_place_out));
    memcpy(gouts, obase + kMissGoutOff, (size_t)n * sizeof(mmp_gate_out));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_miss_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_miss_batch", e.what());
}

int mmp_route_batch(mmp_ctx *c, const mmp_gate_req *greqs, const mmp_serve_req *sreqs, int32_t n, const mmp_serve_counter *counters,
                    int32_t n_counters, const int32_t *excl_pod, const int64_t *excl_time, int32_t n_excl,
                    const int32_t *explicit_pool, int32_t n_explicit, int64_t now, int64_t in_use_expiry, mmp_gate_out *gouts,
                    mmp_serve_out *souts)
try {
    if (!c || n < 0 || n_counters < 0 || n_excl < 0 || n_explicit < 0 || (n > 0 && (!greqs || !sreqs || !gouts || !souts)) ||
        (n_counters > 0 && !counters) || (n_excl > 0 && (!excl_pod || !excl_time)) || (n_explicit > 0 && !explicit_pool))
        return fail(c, MMP_EINVAL, "mmp_route_batch: bad argument");
    for (int32_t i = 0; i < n; i++) {
        const mmp_gate_req &g = greqs[i];
        const mmp_serve_req &r = sreqs[i];
        if (g.n_excl < 0 || g.excl_off < 0 || (int64_t)g.excl_off + g.n_excl > n_excl || g.n_explicit < 0 || g.explicit_off < 0 ||
            (int64_t)g.explicit_off + g.n_explicit > n_explicit || r.n_excl < 0 || r.excl_off < 0 ||
            (int64_t)r.excl_off + r.n_excl > n_excl || r.n_cnt < 0 || r.cnt_off < 0 || (int64_t)r.cnt_off + r.n_cnt > n_counters)
            return fail(c, MMP_EINVAL, "mmp_route_batch: request %d pool range out of bounds", i);
        if (g.model != r.model) return fail(c, MMP_EINVAL, "mmp_route_batch: request %d names two models", i);
    }
    // The per-request seam (invokeModel asks for ONE route at a time): a latency slot, as mmp_gate_batch / mmp_serve_batch take
    // for small calls — pinned, device-mapped buffers on the slot's own stream, one launch, completion through the pinned flag;
    // nothing is staged and the call never queues behind a commit or a large batch.
    constexpr int kRouteN = 256, kRouteCnt = 4096, kRoutePool = kFastExtra / 4;
    constexpr size_t kRouteSreqOff = (size_t)kRouteN * sizeof(mmp_gate_req), kRouteCntOff = kRouteSreqOff + (size_t)kRouteN * sizeof(mmp_serve_req),
                     kRouteSoutOff = (((size_t)kRouteN * sizeof(mmp_gate_out)) + 15) & ~(size_t)15;
    static_assert(kRouteCntOff + (size_t)kRouteCnt * sizeof(mmp_serve_counter) <= (size_t)kFastN * sizeof(mmp_place_req), "route slot layout");
    static_assert(kRouteSoutOff + (size_t)kRouteN * sizeof(mmp_serve_out) <= (size_t)kFastN * sizeof(mmp_place_out), "route slot results");
    static_assert(kRouteSreqOff % 16 == 0 && kRouteCntOff % 16 == 0, "route slot alignment");
    if (n > 0 && n <= kRouteN && n_counters <= kRouteCnt && n_excl <= kRoutePool && n_explicit <= kRoutePool) {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        std::unique_lock<std::mutex> fl;
        FastSlot *f = slot_acquire(c, fl);
        char *base = reinterpret_cast<char *>(f->reqs);
        memcpy(base, greqs, (size_t)n * sizeof(mmp_gate_req));
        memcpy(base + kRouteSreqOff, sreqs, (size_t)n * sizeof(mmp_serve_req));
        if (n_counters) memcpy(base + kRouteCntOff, counters, (size_t)n_counters * sizeof(mmp_serve_counter));
        int32_t *pool = f->extra;
        if (n_excl) {
            memcpy(pool, excl_pod, (size_t)n_excl * 4);
            memcpy(pool + 2 * kRoutePool, excl_time, (size_t)n_excl * 8);
        }
        if (n_explicit) memcpy(pool + kRoutePool, explicit_pool, (size_t)n_explicit * 4);
        char *obase = reinterpret_cast<char *>(f->outs);
        {
            std::shared_lock<std::shared_mutex> g(c->mu);  // capture the published snapshot + enqueue
            if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
            if (c->n_shards > 0) return fail(c, MMP_ESTATE, "context is a pod-axis shard");
            GateArgs G = gate_args(c, n, now, in_use_expiry);
            G.reqs = reinterpret_cast<const mmp_gate_req *>(base);
            G.excl_pod = pool;
            G.explicit_pool = pool + kRoutePool;
            G.excl_time = reinterpret_cast<const int64_t *>(pool + 2 * kRoutePool);
            G.outs = reinterpret_cast<mmp_gate_out *>(obase);
            G.done = DoneFlag{f->done, f->blocks, ++f->seq};
            ServeArgs S{};
            S.reqs = reinterpret_cast<const mmp_serve_req *>(base + kRouteSreqOff);
            S.models = c->models.as<mmp_model_row>();
            S.ent_pod = c->ent_pod.as<int32_t>();
            S.ent_time = c->ent_time.as<int64_t>();
            S.counters = reinterpret_cast<const mmp_serve_counter *>(base + kRouteCntOff);
            S.excl_pod = G.excl_pod;
            S.excl_time = G.excl_time;
            S.outs = reinterpret_cast<mmp_serve_out *>(obase + kRouteSoutOff);
            S.n = n;
            S.n_models = c->n_models;
            S.P = c->snap.P;
            S.now = now;
            S.done = DoneFlag{nullptr, nullptr, 0};
            HIP_TRY(c, order_after_registry(c, f->stream));
            if (n == 1 && n_counters <= kRouteInlineCnt && sreqs[0].n_cnt <= kRouteInlineCnt) {  // ONE route: the requests ride in the kernel arguments
                RouteInline R{};
                R.g = greqs[0];
                R.s = sreqs[0];
                for (int32_t j = 0; j < sreqs[0].n_cnt; j++) R.cnt[j] = counters[sreqs[0].cnt_off + j];
                hipLaunchKernelGGL(route_single_kernel, dim3(1), dim3(128), 0, f->stream, G, S, R);
            } else
                hipLaunchKernelGGL(route_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, f->stream, G, S);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
        memcpy(gouts, obase, (size_t)n * sizeof(mmp_gate_out));
        memcpy(souts, obase + kRouteSoutOff, (size_t)n * sizeof(mmp_serve_out));
        return MMP_OK;
    }
    std::lock_guard<std::mutex> gb(c->batch_mu);  // (as mmp_gate_batch's batch path: owns c->stream and the scratch)
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_gate_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_gate_out)));
    HIP_TRY(c, c->rt_sreqs.ensure((size_t)n * sizeof(mmp_serve_req)));
    HIP_TRY(c, c->rt_souts.ensure((size_t)n * sizeof(mmp_serve_out)));
    HIP_TRY(c, c->rt_cnt.ensure((size_t)std::max(n_counters, 1) * sizeof(mmp_serve_counter)));
    HIP_TRY(c, c->s_a.ensure((size_t)std::max(n_explicit, 1) * 4));
    HIP_TRY(c, c->s_c.ensure((size_t)std::max(n_excl, 1) * 4));
    HIP_TRY(c, c->s_d.ensure((size_t)std::max(n_excl, 1) * 8));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, greqs, (size_t)n * sizeof(mmp_gate_req), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rt_sreqs.p, sreqs, (size_t)n * sizeof(mmp_serve_req), hipMemcpyHostToDevice, st));
    if (n_counters) HIP_TRY(c, hipMemcpyAsync(c->rt_cnt.p, counters, (size_t)n_counters * sizeof(mmp_serve_counter), hipMemcpyHostToDevice, st));
    if (n_explicit) HIP_TRY(c, hipMemcpyAsync(c->s_a.p, explicit_pool, (size_t)n_explicit * 4, hipMemcpyHostToDevice, st));
    if (n_excl) {
        HIP_TRY(c, hipMemcpyAsync(c->s_c.p, excl_pod, (size_t)n_excl * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->s_d.p, excl_time, (size_t)n_excl * 8, hipMemcpyHostToDevice, st));
    }
    GateArgs G = gate_args(c, n, now, in_use_expiry);
    G.reqs = c->s_reqs.as<mmp_gate_req>();
    G.excl_pod = c->s_c.as<int32_t>();
    G.excl_time = c->s_d.as<int64_t>();
    G.explicit_pool = c->s_a.as<int32_t>();
    G.outs = c->s_outs.as<mmp_gate_out>();
    ServeArgs S{};
    S.reqs = c->rt_sreqs.as<mmp_serve_req>();
    S.models = c->models.as<mmp_model_row>();
    S.ent_pod = c->ent_pod.as<int32_t>();
    S.ent_time = c->ent_time.as<int64_t>();
    S.counters = c->rt_cnt.as<mmp_serve_counter>();
    S.excl_pod = c->s_c.as<int32_t>();
    S.excl_time = c->s_d.as<int64_t>();
    S.outs = c->rt_souts.as<mmp_serve_out>();
    S.n = n;
    S.n_models = c->n_models;
    S.P = c->snap.P;
    S.now = now;
    S.done = DoneFlag{nullptr, nullptr, 0};
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(route_batch_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, G, S);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(gouts, c->s_outs.p, (size_t)n * sizeof(mmp_gate_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(souts, c->rt_souts.p, (size_t)n * sizeof(mmp_serve_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_route_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_route_batch", e.what());
}

int mmp_proactive_plan(mmp_ctx *c, int32_t default_units, int64_t now, int32_t max_out, int32_t *out_model,
                       int64_t *out_last_used, mmp_proactive_info *info)
try {
    return mmp_proactive_plan_subset(c, -1, nullptr, 0, default_units, now, max_out, out_model, out_last_used, info);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_proactive_plan");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_proactive_plan", e.what());
}

int mmp_proactive_plan_subset(mmp_ctx *c, int32_t partition, const int32_t *skip_models, int32_t n_skip, int32_t default_units,
                              int64_t now, int32_t max_out, int32_t *out_model, int64_t *out_last_used,
                              mmp_proactive_info *info)
try {
    if (!c || !info || max_out < 0 || n_skip < 0 || (n_skip > 0 && !skip_models) || (max_out > 0 && (!out_model || !out_last_used)))
        return fail(c, MMP_EINVAL, "mmp_proactive_plan: bad argument");
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (partition >= cur_side(c).n_pts || partition < -1) return fail(c, MMP_EINVAL, "mmp_proactive_plan: no partition %d", partition);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const int32_t M = c->n_models, P = c->snap.P;
    const int nb = std::max(div_up(std::max(M, 1), kCompactBlock), 1);
    HIP_TRY(c, c->r_ps.ensure(sizeof(PlanScalars)));
    HIP_TRY(c, c->r_counts.ensure((size_t)(nb + 1) * 4));
    HIP_TRY(c, c->r_keys.ensure((size_t)std::max(M, 1) * 8));
    HIP_TRY(c, c->r_keys2.ensure((size_t)std::max(M, 1) * 8));
    HIP_TRY(c, c->r_vals.ensure((size_t)std::max(M, 1) * 4));
    HIP_TRY(c, c->r_vals2.ensure((size_t)std::max(M, 1) * 4));
    HIP_TRY(c, c->r_out_model.ensure((size_t)std::max(max_out, 1) * 4));
    HIP_TRY(c, c->r_out_lu.ensure((size_t)std::max(max_out, 1) * 8));
    PlanScalars *ps = c->r_ps.as<PlanScalars>();
    const StatsAcc *stats = cur_side(c).stats_acc.as<StatsAcc>();
    const mmp_pod_row *pods = c->sb[c->cur].pods.as<mmp_pod_row>();
    const mmp_model_row *models = c->models.as<mmp_model_row>();
    int32_t *counts = nullptr;
    HIP_TRY(c, hipMemsetAsync(ps, 0, sizeof(PlanScalars), st));
    PlanSubset U{};
    U.global = stats;
    U.stats = partition >= 0 ? cur_side(c).pstats.as<StatsAcc>() + partition : stats;
    U.pod_pts = cur_side(c).d_pts.as<int32_t>();
    U.prohib = partition >= 0 ? cur_side(c).d_prohib.as<uint64_t>() + (size_t)partition * cur_side(c).pts_tw : nullptr;
    U.skip = nullptr;
    U.pts = partition;
    U.n_types = c->n_types;
    if (n_skip > 0) {  // models already triggered for an earlier partition of this run
        std::vector<uint8_t> mask((size_t)std::max(M, 1), 0);
        for (int32_t i = 0; i < n_skip; i++)
            if (skip_models[i] >= 0 && skip_models[i] < M) mask[skip_models[i]] = 1;
        HIP_TRY(c, c->s_d.ensure(mask.size()));
        HIP_TRY(c, hipMemcpyAsync(c->s_d.p, mask.data(), mask.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));  // `mask` leaves scope
        U.skip = c->s_d.as<uint8_t>();
    }
    // The bucketed plan: six dependent launches, every size read on the device, ONE read of the result at the end.
    const int32_t n_cnt_words = 3 * (kPlanBuckets + 1) + 3 * kPlanBuckets + 1;  // (even: the one-launch plan's chunk totals follow, 8 bytes each)
    static_assert((3 * (kPlanBuckets + 1) + 3 * kPlanBuckets + 1) % 2 == 0, "chunk totals are 8-byte aligned");
    HIP_TRY(c, c->r_counts.ensure((size_t)std::max(n_cnt_words + 2 * div_up(std::max(M, 1), kPlanFusedBlock), nb + 1) * 4));
    HIP_TRY(c, c->r_part.ensure((size_t)nb * std::max(sizeof(PlanPartial), (size_t)kPlanPartWords * 8)));
    counts = c->r_counts.as<int32_t>();
    int32_t *hist = counts, *dcnt = hist + kPlanBuckets, *dge = dcnt + kPlanBuckets, *off = dge + kPlanBuckets,
            *cur = off + kPlanBuckets + 1, *doff = cur + kPlanBuckets + 1;
    PlanPartial *part = c->r_part.as<PlanPartial>();
    KT_BEGIN(c, st);  // device span of the whole plan
    // The one-launch form prefetches registry / instance rows unconditionally and waits at grid-wide barriers: it needs a registry
    // and a table to read (an empty or never-loaded one takes the eight launches, which guard every access: ADVICE r5), and every
    // workgroup on the chip at once — plan_grid_max is what the occupancy query granted at mmp_create; a launch never asks for more
    const bool fused_ok = c->cfg_plan_fused && c->plan_grid_max > 0 && M > 0 && P > 0 && models != nullptr && pods != nullptr;
    if (fused_ok) {
        // ONE launch, the steps separated by grid-wide barriers (rebalance_kernels.hpp: proactive_plan_fused_kernel)
        const int G = std::max(std::min(div_up(std::max(M, 1), kPlanFusedBlock), (int)c->plan_grid_max), 1);
        hipLaunchKernelGGL(proactive_plan_fused_kernel, dim3(G), dim3(kPlanFusedBlock), kPlanFusedLds, st, pods, P, models, M, U, default_units, now,
                           ps, c->r_part.as<uint64_t>(), hist, c->r_keys2.as<int32_t>(), c->r_keys.as<int64_t>(), c->r_vals.as<int32_t>(),
                           c->r_vals2.as<int32_t>(), reinterpret_cast<uint64_t *>(counts + n_cnt_words), max_out,
                           c->r_out_model.as<int32_t>(), c->r_out_lu.as<int64_t>());
    } else {
        hipLaunchKernelGGL(proactive_space_scalars_kernel, dim3(std::min(std::max(div_up(P, 256), 1), 512)), dim3(256), 0, st, pods, P,
                           U, default_units, now, ps);
        hipLaunchKernelGGL(proactive_qualify_kernel, dim3(nb), dim3(kCompactBlock), 0, st, models, M, U, ps, part, hist);
        hipLaunchKernelGGL(proactive_hist_kernel, dim3(nb), dim3(kCompactBlock), 0, st, models, M, U, ps, part, nb, hist);
        hipLaunchKernelGGL(proactive_scan_kernel, dim3(1), dim3(kPlanScanBlock), 0, st, ps, hist, off, cur);
        hipLaunchKernelGGL(proactive_bin_kernel, dim3(nb), dim3(kCompactBlock), 0, st, models, M, U, ps, cur, c->r_keys.as<int64_t>(),
                           c->r_vals.as<int32_t>());
        hipLaunchKernelGGL(proactive_bucket_rank_kernel, dim3(kPlanBuckets / 4), dim3(256), 0, st, c->r_keys.as<int64_t>(),
                           c->r_vals.as<int32_t>(), off, ps, c->r_vals2.as<int32_t>(), dcnt, dge);
        hipLaunchKernelGGL(proactive_scan2_kernel, dim3(1), dim3(kPlanScanBlock), 0, st, ps, dcnt, dge, doff);
        hipLaunchKernelGGL(proactive_emit_kernel, dim3(nb), dim3(kCompactBlock), 0, st, c->r_keys.as<int64_t>(), c->r_vals.as<int32_t>(),
                           c->r_vals2.as<int32_t>(), doff, ps, max_out, c->r_out_model.as<int32_t>(), c->r_out_lu.as<int64_t>());
    }
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    PlanScalars h{};
    HIP_TRY(c, hipMemcpyAsync(&h, ps, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    // (overflow == 2: the one-launch plan gave up at a grid barrier — its workgroups were not all on the chip; the general path below)
    if (h.overflow && c->cfg_plan_sorted == 2) return fail(c, MMP_ESTATE, "mmp_proactive_plan: a key bucket overflowed and MMP_PLAN_SORTED=2 forbids the sorted path");
    if (!h.overflow && c->cfg_plan_sorted != 1) {
        const int32_t n_copy = std::min(h.n_selected, max_out);
        if (n_copy > 0) {
            HIP_TRY(c, copy_sync(c, out_model, c->r_out_model.p, (size_t)n_copy * 4, hipMemcpyDeviceToHost));
            HIP_TRY(c, copy_sync(c, out_last_used, c->r_out_lu.p, (size_t)n_copy * 8, hipMemcpyDeviceToHost));
        }
    } else {
        // lastUsed values piled on a few milliseconds (a bucket above kPlanBucketMax): the general path — order-preserving
        // compaction, a stable descending radix sort sized by the qualified count the host has just read, run starts
        HIP_TRY(c, hipMemsetAsync(ps, 0, sizeof(PlanScalars), st));
        KT_BEGIN(c, st);
        hipLaunchKernelGGL(proactive_space_kernel, dim3(std::min(std::max(div_up(P, 256), 1), 512)), dim3(256), 0, st, pods, P,
                           U, default_units, ps);
        hipLaunchKernelGGL(proactive_scalars_kernel, dim3(1), dim3(64), 0, st, U, default_units, now, ps);
        hipLaunchKernelGGL(proactive_count_kernel, dim3(nb), dim3(kCompactBlock), 0, st, models, M, U, ps, counts,
                           &ps->n_candidates);
        hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, st, counts, nb, &ps->n_qualified);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(&h, ps, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        const int32_t nq = h.n_qualified;
        if (nq <= 0) {
            KT_END(c, st);
            HIP_TRY(c, hipStreamSynchronize(st));
            kt_collect(c);
        } else {
            hipLaunchKernelGGL(proactive_scatter_kernel, dim3(nb), dim3(kCompactBlock), 0, st, models, M, U, ps, counts,
                               c->r_keys.as<int64_t>(), c->r_vals.as<int32_t>());
            // stable descending radix sort: equal lastUsed keep registry order, so the first one seen wins
            size_t tmp_bytes = 0;
            HIP_TRY(c, rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, c->r_keys.as<int64_t>(), c->r_keys2.as<int64_t>(),
                                                      c->r_vals.as<int32_t>(), c->r_vals2.as<int32_t>(), (size_t)nq, 0, 64, st));
            HIP_TRY(c, c->r_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
            HIP_TRY(c, rocprim::radix_sort_pairs_desc(c->r_tmp.p, tmp_bytes, c->r_keys.as<int64_t>(), c->r_keys2.as<int64_t>(),
                                                      c->r_vals.as<int32_t>(), c->r_vals2.as<int32_t>(), (size_t)nq, 0, 64, st));
            const int nb2 = div_up(nq, kCompactBlock);
            hipLaunchKernelGGL(distinct_count_kernel, dim3(nb2), dim3(kCompactBlock), 0, st, c->r_keys2.as<int64_t>(), nq, counts);
            hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, st, counts, nb2, &ps->n_distinct);
            hipLaunchKernelGGL(distinct_scatter_kernel, dim3(nb2), dim3(kCompactBlock), 0, st, c->r_keys2.as<int64_t>(),
                               c->r_vals2.as<int32_t>(), nq, counts, ps, max_out, c->r_out_model.as<int32_t>(),
                               c->r_out_lu.as<int64_t>());
            hipLaunchKernelGGL(proactive_final_kernel, dim3(1), dim3(64), 0, st, ps);
            KT_END(c, st);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&h, ps, sizeof h, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            kt_collect(c);
            const int32_t n_copy = std::min(h.n_selected, max_out);
            if (n_copy > 0) {
                HIP_TRY(c, copy_sync(c, out_model, c->r_out_model.p, (size_t)n_copy * 4, hipMemcpyDeviceToHost));
                HIP_TRY(c, copy_sync(c, out_last_used, c->r_out_lu.p, (size_t)n_copy * 8, hipMemcpyDeviceToHost));
            }
        }
    }
    info->size_estimate = h.size_estimate;
    info->free_count = h.free_count;
    info->total_count = h.total_count;
    info->n_candidates = h.n_candidates;
    info->n_selected = h.n_selected;
    info->error = h.error;
    info->space_to_fill = h.space_to_fill;
    info->cutoff = h.cutoff;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_proactive_plan_subset");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_proactive_plan_subset", e.what());
}

// rateTrackingTask: the one body of mmp_scaleup_plan (conc == null) and mmp_scaleup_plan_conc
static int scaleup_plan_impl(mmp_ctx *c, const char *fn, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                             const mmp_scaleup_params *p, const mmp_conc_params *cp, mmp_scaleup_out *outs, mmp_conc_out *conc_outs,
                             uint8_t *overloaded_out, int32_t *skipped, mmp_conc_result *result)
{
    const bool latency = cp != nullptr;
    if (!c || !p || !skipped || n < 0 || (n > 0 && (!entries || !outs)) || (latency && (!result || (n > 0 && (!conc || !conc_outs)))))
        return fail(c, MMP_EINVAL, "%s: bad argument", fn);
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    const int32_t P = c->snap.P;
    if (P > 0 && !overloaded_out) return fail(c, MMP_EINVAL, "%s: overloaded_out is null", fn);
    if (P > 0) memset(overloaded_out, 0, (size_t)P);
    for (int32_t i = 0; i < n; i++) {
        outs[i] = mmp_scaleup_out{};
        outs[i].new_i1 = entries[i].earlier_use_iteration;
        outs[i].new_i2 = entries[i].last_used_iteration;
        if (latency) {
            conc_outs[i] = mmp_conc_out{};
            conc_outs[i].new_prior_sum = conc[i].prior_sum;
            conc_outs[i].new_prior_count = conc[i].prior_count;
        }
    }
    if (latency) {  // (a run that returns early leaves the task's field as it was)
        *result = mmp_conc_result{};
        result->average_model_parallelism = cp->average_model_parallelism;
    }
    // the three early returns of the task, MM.java:5646-5648, :5658-5660, :5667-5669
    const int64_t time_delta = (int64_t)((uint64_t)p->now - (uint64_t)p->last_check_time);
    *skipped = (time_delta * 5 < p->rate_check_interval_ms * 3 || c->stats.instance_count < 2 || n == 0) ? 1 : 0;
    if (*skipped) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_cache_entry)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_scaleup_out)));
    HIP_TRY(c, c->s_a.ensure((size_t)std::max(P, 1)));
    HIP_TRY(c, c->s_b.ensure(2 * sizeof(int32_t)));  // [0] overloaded instances, [1] latency-based: getExcludeSet's maxRpm
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, entries, (size_t)n * sizeof(mmp_cache_entry), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->s_b.p, 0, 2 * sizeof(int32_t), st));
    if (latency) {
        HIP_TRY(c, c->s_c.ensure((size_t)n * sizeof(mmp_conc_entry)));
        HIP_TRY(c, c->s_d.ensure((size_t)n * sizeof(mmp_conc_out) + sizeof(mmp_conc_result)));
        HIP_TRY(c, hipMemcpyAsync(c->s_c.p, conc, (size_t)n * sizeof(mmp_conc_entry), hipMemcpyHostToDevice, st));
    }
    mmp_conc_result *d_res = latency ? reinterpret_cast<mmp_conc_result *>(static_cast<char *>(c->s_d.p) + (size_t)n * sizeof(mmp_conc_out)) : nullptr;
    const int32_t a = (int32_t)((uint32_t)p->scale_up_rpm_threshold * 4u);
    const int32_t b = (int32_t)((uint32_t)p->our_rpm - 2u * (uint32_t)p->scale_up_rpm_threshold);
    const mmp_pod_row *pods = c->sb[c->cur].pods.as<mmp_pod_row>();
    KT_BEGIN(c, st);
    if (latency)  // getExcludeSet's threshold comes from the task's averageModelParallelism (:5836)
        hipLaunchKernelGGL(conc_exclude_rpms_kernel, dim3(1), dim3(64), 0, st, cp->average_model_parallelism, p->our_rpm,
                           c->s_b.as<int32_t>() + 1, d_res);
    if (P > 0)
        hipLaunchKernelGGL(overloaded_pods_kernel, dim3(div_up(P, 256)), dim3(256), 0, st, pods, P, p->self_pod,
                           a > b ? a : b, latency ? c->s_b.as<int32_t>() + 1 : nullptr, c->s_a.as<uint8_t>(), c->s_b.as<int32_t>());
    ScaleupArgs A;
    A.entries = c->s_reqs.as<mmp_cache_entry>();
    A.models = c->models.as<mmp_model_row>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.ent_time = c->ent_time.as<int64_t>();
    A.stats = cur_side(c).stats_acc.as<StatsAcc>();
    A.tstats = cur_side(c).tstats.as<StatsAcc>();
    A.T_rows = std::max(c->n_types, 1);
    A.has_tc = c->n_types > 0 ? 1 : 0;
    A.overloaded = c->s_a.as<uint8_t>();
    A.excluded_count = c->s_b.as<int32_t>();
    A.outs = c->s_outs.as<mmp_scaleup_out>();
    A.p = *p;
    A.n = n;
    A.n_models = c->n_models;
    A.P = P;
    A.conc = latency ? c->s_c.as<mmp_conc_entry>() : nullptr;
    A.conc_outs = latency ? c->s_d.as<mmp_conc_out>() : nullptr;
    A.conc_res = d_res;
    A.dyn_const = latency ? cp->dynamic_rpm_scale_constant : 0;
    hipLaunchKernelGGL(scaleup_plan_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, A);
    if (latency) hipLaunchKernelGGL(conc_average_kernel, dim3(1), dim3(64), 0, st, n, d_res);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_scaleup_out), hipMemcpyDeviceToHost, st));
    if (P > 0) HIP_TRY(c, hipMemcpyAsync(overloaded_out, c->s_a.p, (size_t)P, hipMemcpyDeviceToHost, st));
    if (latency) {
        HIP_TRY(c, hipMemcpyAsync(conc_outs, c->s_d.p, (size_t)n * sizeof(mmp_conc_out), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(result, d_res, sizeof(mmp_conc_result), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
}

int mmp_scaleup_plan(mmp_ctx *c, const mmp_cache_entry *entries, int32_t n, const mmp_scaleup_params *p,
                     mmp_scaleup_out *outs, uint8_t *overloaded_out, int32_t *skipped)
try {
    return scaleup_plan_impl(c, "mmp_scaleup_plan", entries, nullptr, n, p, nullptr, outs, nullptr, overloaded_out, skipped, nullptr);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_scaleup_plan");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_scaleup_plan", e.what());
}

int mmp_scaleup_plan_conc(mmp_ctx *c, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                          const mmp_scaleup_params *p, const mmp_conc_params *cp, mmp_scaleup_out *outs, mmp_conc_out *conc_outs,
                          uint8_t *overloaded_out, int32_t *skipped, mmp_conc_result *result)
try {
    if (!cp) return fail(c, MMP_EINVAL, "mmp_scaleup_plan_conc: conc_params is null");
    return scaleup_plan_impl(c, "mmp_scaleup_plan_conc", entries, conc, n, p, cp, outs, conc_outs, overloaded_out, skipped, result);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_scaleup_plan_conc");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_scaleup_plan_conc", e.what());
}

// the janitor's scale-down: the one body of mmp_scaledown_plan (conc == null) and mmp_scaledown_plan_conc
static int scaledown_plan_impl(mmp_ctx *c, const char *fn, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                               const mmp_scaledown_params *p, int64_t dyn_const, uint8_t *removed_out)
{
    if (!c || !p || n < 0 || (n > 0 && (!entries || !removed_out)))
        return fail(c, MMP_EINVAL, "%s: bad argument", fn);
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_cache_entry)));
    HIP_TRY(c, c->s_a.ensure((size_t)n));
    HIP_TRY(c, c->s_b.ensure((size_t)n));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, entries, (size_t)n * sizeof(mmp_cache_entry), hipMemcpyHostToDevice, st));
    if (conc) {
        HIP_TRY(c, c->s_c.ensure((size_t)n * sizeof(mmp_conc_entry)));
        HIP_TRY(c, hipMemcpyAsync(c->s_c.p, conc, (size_t)n * sizeof(mmp_conc_entry), hipMemcpyHostToDevice, st));
    }
    ScaledownArgs A;
    A.entries = c->s_reqs.as<mmp_cache_entry>();
    A.models = c->models.as<mmp_model_row>();
    A.ent_pod = c->ent_pod.as<int32_t>();
    A.ent_time = c->ent_time.as<int64_t>();
    A.pods = c->sb[c->cur].pods.as<mmp_pod_row>();
    A.pos_of = c->snap.pos_of;
    // instanceSetStats(): with type constraints, the stats of the partition this instance belongs to
    // (EMPTY_STATS when it is not in the table), cluster-wide otherwise
    {
        const int32_t sp = p->self_pod;
        const int32_t k = (sp >= 0 && sp < (int32_t)cur_side(c).pts_of.size()) ? cur_side(c).pts_of[sp] : -1;
        A.stats = c->n_types > 0 ? cur_side(c).pstats.as<StatsAcc>() + (k >= 0 ? k : cur_side(c).n_pts) : cur_side(c).stats_acc.as<StatsAcc>();
    }
    A.decide = c->s_a.as<uint8_t>();
    A.removed = c->s_b.as<uint8_t>();
    A.p = *p;
    A.n = n;
    A.n_models = c->n_models;
    A.P = c->snap.P;
    A.conc = conc ? c->s_c.as<mmp_conc_entry>() : nullptr;
    A.dyn_const = dyn_const;
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(scaledown_decide_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, A);
    hipLaunchKernelGGL(scaledown_budget_kernel, dim3(1), dim3(64), 0, st, A);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(removed_out, c->s_b.p, (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
}

int mmp_scaledown_plan(mmp_ctx *c, const mmp_cache_entry *entries, int32_t n, const mmp_scaledown_params *p,
                       uint8_t *removed_out)
try {
    return scaledown_plan_impl(c, "mmp_scaledown_plan", entries, nullptr, n, p, 0, removed_out);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_scaledown_plan");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_scaledown_plan", e.what());
}

int mmp_scaledown_plan_conc(mmp_ctx *c, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                            const mmp_scaledown_params *p, int64_t dynamic_rpm_scale_constant, uint8_t *removed_out)
try {
    if (n > 0 && !conc) return fail(c, MMP_EINVAL, "mmp_scaledown_plan_conc: conc is null");
    return scaledown_plan_impl(c, "mmp_scaledown_plan_conc", entries, conc, n, p, dynamic_rpm_scale_constant, removed_out);
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_scaledown_plan_conc");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_scaledown_plan_conc", e.what());
}

int mmp_migration_plan(mmp_ctx *c, const mmp_cache_entry *entries, int32_t n, int32_t self_pod, int64_t now,
                       int64_t cutoff_age_ms, uint8_t *action_out, uint8_t *wait_out)
try {
    if (!c || n < 0 || (n > 0 && (!entries || !action_out || !wait_out)))
        return fail(c, MMP_EINVAL, "mmp_migration_plan: bad argument");
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (!c->committed) return fail(c, MMP_ESTATE, "no committed snapshot");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_cache_entry)));
    HIP_TRY(c, c->s_a.ensure((size_t)n));
    HIP_TRY(c, c->s_b.ensure((size_t)n));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, entries, (size_t)n * sizeof(mmp_cache_entry), hipMemcpyHostToDevice, st));
    KT_BEGIN(c, st);
    hipLaunchKernelGGL(migration_plan_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, c->s_reqs.as<mmp_cache_entry>(), n,
                       c->models.as<mmp_model_row>(), c->n_models, c->ent_pod.as<int32_t>(), self_pod,
                       (int64_t)((uint64_t)now - (uint64_t)cutoff_age_ms), c->s_a.as<uint8_t>(), c->s_b.as<uint8_t>());
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(action_out, c->s_a.p, (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(wait_out, c->s_b.p, (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_migration_plan");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_migration_plan", e.what());
}

/* ---- stateful keyed caches (a12 + a13) ------------------------------------- */

int mmp_caches_load_keyed(mmp_ctx *c, int32_t n_caches, const int32_t *seg_off, const int64_t *last_used,
                          const int32_t *weight, const int32_t *key, const int64_t *capacity, const mmp_ubm_state *ubm)
try {
    if (!c || n_caches < 0 || !seg_off || (n_caches > 0 && !capacity)) return fail(c, MMP_EINVAL, "mmp_caches_load_keyed: bad argument");
    if (seg_off[0] != 0) return fail(c, MMP_EINVAL, "mmp_caches_load_keyed: seg_off[0] must be 0");
    for (int32_t i = 0; i < n_caches; i++)
        if (seg_off[i + 1] < seg_off[i]) return fail(c, MMP_EINVAL, "mmp_caches_load_keyed: seg_off not monotone at %d", i);
    const int32_t E = seg_off[n_caches];
    if (E > 0 && (!last_used || !weight || !key)) return fail(c, MMP_EINVAL, "mmp_caches_load_keyed: null entry arrays");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    mmp_ctx::KeyedStore &K = c->ks[c->ks_cur];
    HIP_TRY(c, K.off.ensure((size_t)(n_caches + 1) * 4));
    HIP_TRY(c, K.lu.ensure((size_t)std::max(E, 1) * 8));
    HIP_TRY(c, K.wt.ensure((size_t)std::max(E, 1) * 4));
    HIP_TRY(c, K.key.ensure((size_t)std::max(E, 1) * 4));
    HIP_TRY(c, K.n.ensure((size_t)std::max(n_caches, 1) * 4));
    HIP_TRY(c, c->k_cap.ensure((size_t)std::max(n_caches, 1) * 8));
    HIP_TRY(c, c->k_wsize.ensure((size_t)std::max(n_caches, 1) * 8));
    HIP_TRY(c, c->k_oldest.ensure((size_t)std::max(n_caches, 1) * 8));
    HIP_TRY(c, c->k_ubm.ensure((size_t)std::max(n_caches, 1) * sizeof(mmp_ubm_state)));
    c->k_n.assign(n_caches, 0);
    std::vector<int64_t> ws(std::max(n_caches, 1), 0), old(std::max(n_caches, 1), -1);
    std::vector<mmp_ubm_state> us(std::max(n_caches, 1));
    for (int32_t i = 0; i < n_caches; i++) {
        c->k_n[i] = seg_off[i + 1] - seg_off[i];
        if (c->k_n[i] > 0) old[i] = last_used[seg_off[i]];  // oldestTime as the last write left it (clhm :1129-1133)
        for (int32_t e = seg_off[i]; e < seg_off[i + 1]; e++) ws[i] += weight[e] < 0 ? -(int64_t)weight[e] : weight[e];
        if (ubm)
            us[i] = ubm[i];
        else {
            us[i] = mmp_ubm_state{};
            us[i].reserved = -1;
        }
    }
    HIP_TRY(c, copy_sync(c, K.off.p, seg_off, (size_t)(n_caches + 1) * 4, hipMemcpyHostToDevice));
    if (E) {
        HIP_TRY(c, copy_sync(c, K.lu.p, last_used, (size_t)E * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, K.wt.p, weight, (size_t)E * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, K.key.p, key, (size_t)E * 4, hipMemcpyHostToDevice));
    }
    if (n_caches) {
        HIP_TRY(c, copy_sync(c, K.n.p, c->k_n.data(), (size_t)n_caches * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->k_cap.p, capacity, (size_t)n_caches * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->k_wsize.p, ws.data(), (size_t)n_caches * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->k_oldest.p, old.data(), (size_t)n_caches * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->k_ubm.p, us.data(), (size_t)n_caches * sizeof(mmp_ubm_state), hipMemcpyHostToDevice));
    }
    c->k_caches = n_caches;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_caches_load_keyed");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_caches_load_keyed", e.what());
}

int mmp_cache_replay(mmp_ctx *c, const mmp_cache_op *ops, int32_t n_ops, int64_t now, mmp_cache_op_out *outs,
                     int32_t *evicted_keys, int32_t max_evicted, int32_t *n_evicted_slots)
try {
    if (!c || n_ops < 0 || max_evicted < 0 || (n_ops > 0 && (!ops || !outs)) || (max_evicted > 0 && !evicted_keys) ||
        !n_evicted_slots)
        return fail(c, MMP_EINVAL, "mmp_cache_replay: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    const int32_t NC = c->k_caches;
    if (NC <= 0 && n_ops > 0) return fail(c, MMP_ESTATE, "no keyed caches loaded");
    *n_evicted_slots = 0;
    if (n_ops == 0) return MMP_OK;
    // group the operations by cache (stable) and size the new layout: every insert may add a slot
    std::vector<int32_t> cnt(NC + 1, 0), ins(NC, 0);
    for (int32_t i = 0; i < n_ops; i++) {
        const mmp_cache_op &o = ops[i];
        if (o.cache < 0 || o.cache >= NC) return fail(c, MMP_EINVAL, "mmp_cache_replay: op %d names cache %d", i, o.cache);
        if (o.op < 0 || o.op > MMP_COP_UBM_INSERT_FAILED_PLACEHOLDER) return fail(c, MMP_EINVAL, "mmp_cache_replay: op %d has code %d", i, o.op);
        cnt[o.cache + 1]++;
        if (o.op == MMP_COP_PUT_IF_ABSENT || o.op == MMP_COP_UBM_INSERT_NEW_ENTRY || o.op == MMP_COP_UBM_INSERT_FAILED_PLACEHOLDER)
            ins[o.cache]++;
    }
    std::vector<int32_t> op_off(NC + 1, 0), new_off(NC + 1, 0), ev_off(NC + 1, 0);
    // team width per cache by the deque slots its replay needs: [0] 8 lanes, [1] 16 lanes, [2] the whole wavefront
    int32_t tile = 1, tiles[3] = {1, 1, 1};
    std::vector<int32_t> ids[3];
    for (int32_t k = 0; k < NC; k++) {
        op_off[k + 1] = op_off[k] + cnt[k + 1];
        const int32_t slots = c->k_n[k] + ins[k];
        new_off[k + 1] = new_off[k] + slots;
        ev_off[k + 1] = ev_off[k] + (cnt[k + 1] ? slots : 0);  // a cache cannot evict more than it ever held
        const int w = slots + 1 <= kTeam8Slots ? 0 : slots + 1 <= kTeam16Slots ? 1 : 2;
        ids[w].push_back(k);
        if (cnt[k + 1]) {
            tile = std::max(tile, slots + 1);
            tiles[w] = std::max(tiles[w], slots + 1);
        }
    }
    if (tile > kCacheTile) return fail(c, MMP_EINVAL, "mmp_cache_replay: a cache needs %d deque slots, the tile holds %d", tile, kCacheTile);
    if (ev_off[NC] > max_evicted) return fail(c, MMP_EINVAL, "mmp_cache_replay: evicted_keys needs room for %d keys", ev_off[NC]);
    std::vector<int32_t> order(n_ops), fill(op_off.begin(), op_off.end() - 1);
    for (int32_t i = 0; i < n_ops; i++) order[fill[ops[i].cache]++] = i;
    std::vector<mmp_cache_op> grouped((size_t)n_ops);  // the kernel reads an operation with ONE load (no index in between)
    for (int32_t q = 0; q < n_ops; q++) grouped[q] = ops[order[q]];

    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    mmp_ctx::KeyedStore &S = c->ks[c->ks_cur], &D = c->ks[1 - c->ks_cur];
    const int32_t Enew = new_off[NC];
    HIP_TRY(c, D.off.ensure((size_t)(NC + 1) * 4));
    HIP_TRY(c, D.lu.ensure((size_t)std::max(Enew, 1) * 8));
    HIP_TRY(c, D.wt.ensure((size_t)std::max(Enew, 1) * 4));
    HIP_TRY(c, D.key.ensure((size_t)std::max(Enew, 1) * 4));
    HIP_TRY(c, D.n.ensure((size_t)NC * 4));
    HIP_TRY(c, c->k_ops.ensure((size_t)n_ops * sizeof(mmp_cache_op)));
    HIP_TRY(c, c->k_order.ensure((size_t)n_ops * 4));
    HIP_TRY(c, c->k_opoff.ensure((size_t)(NC + 1) * 4));
    HIP_TRY(c, c->k_outs.ensure((size_t)n_ops * sizeof(mmp_cache_op_out)));
    HIP_TRY(c, c->k_ev.ensure((size_t)std::max(ev_off[NC], 1) * 4));
    HIP_TRY(c, c->k_evoff.ensure((size_t)(NC + 1) * 4));
    HIP_TRY(c, hipMemcpyAsync(D.off.p, new_off.data(), (size_t)(NC + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->k_ops.p, grouped.data(), (size_t)n_ops * sizeof(mmp_cache_op), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->k_order.p, order.data(), (size_t)n_ops * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->k_opoff.p, op_off.data(), (size_t)(NC + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->k_evoff.p, ev_off.data(), (size_t)(NC + 1) * 4, hipMemcpyHostToDevice, st));
    ReplayArgs A;
    A.src = CacheStore{S.off.as<int32_t>(), S.lu.as<int64_t>(), S.wt.as<int32_t>(), S.key.as<int32_t>(), S.n.as<int32_t>()};
    A.dst = CacheStore{D.off.as<int32_t>(), D.lu.as<int64_t>(), D.wt.as<int32_t>(), D.key.as<int32_t>(), D.n.as<int32_t>()};
    A.capacity = c->k_cap.as<int64_t>();
    A.weighted_size = c->k_wsize.as<int64_t>();
    A.oldest = c->k_oldest.as<int64_t>();
    A.ubm = c->k_ubm.as<mmp_ubm_state>();
    A.ops = c->k_ops.as<mmp_cache_op>();
    A.op_order = c->k_order.as<int32_t>();
    A.op_off = c->k_opoff.as<int32_t>();
    A.outs = c->k_outs.as<mmp_cache_op_out>();
    A.evicted = c->k_ev.as<int32_t>();
    A.ev_off = c->k_evoff.as<int32_t>();
    A.n_caches = NC;
    A.tile = tile;
    A.now = now;
    HIP_TRY(c, c->k_ids.ensure((size_t)NC * 4));
    {
        std::vector<int32_t> all;
        all.reserve(NC);
        for (auto &v : ids) all.insert(all.end(), v.begin(), v.end());
        HIP_TRY(c, hipMemcpyAsync(c->k_ids.p, all.data(), (size_t)NC * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));  // `all` leaves scope (the other uploads above are from vectors that live to the end)
    }
    KT_BEGIN(c, st);
    const int32_t *d_ids = c->k_ids.as<int32_t>();
    for (int w = 0; w < 3; w++)
        if ((int32_t)ids[w].size() == NC) d_ids = nullptr;  // one width for every cache: the list is the identity
    if (!ids[0].empty()) {
        A.tile = tiles[0];
        hipLaunchKernelGGL(cache_replay_kernel<8>, dim3(div_up((int)ids[0].size(), 8)), dim3(64), (size_t)tiles[0] * 24 * 8, st, A, d_ids,
                           (int32_t)ids[0].size());
    }
    if (!ids[1].empty()) {
        A.tile = tiles[1];
        hipLaunchKernelGGL(cache_replay_kernel<16>, dim3(div_up((int)ids[1].size(), 4)), dim3(64), (size_t)tiles[1] * 24 * 4, st, A,
                           d_ids ? d_ids + ids[0].size() : nullptr, (int32_t)ids[1].size());
    }
    if (!ids[2].empty()) {
        A.tile = tiles[2];
        hipLaunchKernelGGL(cache_replay_kernel<64>, dim3((int)ids[2].size()), dim3(64), (size_t)tiles[2] * 24, st, A,
                           d_ids ? d_ids + ids[0].size() + ids[1].size() : nullptr, (int32_t)ids[2].size());
    }
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(outs, c->k_outs.p, (size_t)n_ops * sizeof(mmp_cache_op_out), hipMemcpyDeviceToHost, st));
    if (ev_off[NC]) HIP_TRY(c, hipMemcpyAsync(evicted_keys, c->k_ev.p, (size_t)ev_off[NC] * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(c->k_n.data(), D.n.p, (size_t)NC * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    c->ks_cur = 1 - c->ks_cur;
    *n_evicted_slots = ev_off[NC];
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_cache_replay");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_cache_replay", e.what());
}

int mmp_cache_read(mmp_ctx *c, int32_t cache, int32_t max_entries, int64_t *last_used, int32_t *weight, int32_t *key,
                   int32_t *n_out, int64_t *capacity, int64_t *weighted_size, mmp_ubm_state *ubm)
try {
    if (!c || !n_out || max_entries < 0) return fail(c, MMP_EINVAL, "mmp_cache_read: bad argument");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    if (cache < 0 || cache >= c->k_caches) return fail(c, MMP_EINVAL, "mmp_cache_read: cache %d out of range", cache);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    mmp_ctx::KeyedStore &K = c->ks[c->ks_cur];
    int32_t off = 0;
    HIP_TRY(c, copy_sync(c, &off, K.off.as<int32_t>() + cache, 4, hipMemcpyDeviceToHost));
    const int32_t n = c->k_n[cache];
    *n_out = n;
    const int32_t m = std::min(n, max_entries);
    if (m > 0) {
        if (last_used) HIP_TRY(c, copy_sync(c, last_used, K.lu.as<int64_t>() + off, (size_t)m * 8, hipMemcpyDeviceToHost));
        if (weight) HIP_TRY(c, copy_sync(c, weight, K.wt.as<int32_t>() + off, (size_t)m * 4, hipMemcpyDeviceToHost));
        if (key) HIP_TRY(c, copy_sync(c, key, K.key.as<int32_t>() + off, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    if (capacity) HIP_TRY(c, copy_sync(c, capacity, c->k_cap.as<int64_t>() + cache, 8, hipMemcpyDeviceToHost));
    if (weighted_size) HIP_TRY(c, copy_sync(c, weighted_size, c->k_wsize.as<int64_t>() + cache, 8, hipMemcpyDeviceToHost));
    if (ubm) HIP_TRY(c, copy_sync(c, ubm, c->k_ubm.as<mmp_ubm_state>() + cache, sizeof(mmp_ubm_state), hipMemcpyDeviceToHost));
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_cache_read");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_cache_read", e.what());
}

int mmp_caches_load(mmp_ctx *c, int32_t n_caches, const int32_t *seg_off, const int64_t *last_used,
                    const int32_t *weight, const int64_t *capacity)
try {
    if (!c || n_caches < 0 || !seg_off || (n_caches > 0 && !capacity)) return fail(c, MMP_EINVAL, "mmp_caches_load: bad argument");
    if (seg_off[0] != 0) return fail(c, MMP_EINVAL, "mmp_caches_load: seg_off[0] must be 0");
    for (int32_t i = 0; i < n_caches; i++)
        if (seg_off[i + 1] < seg_off[i]) return fail(c, MMP_EINVAL, "mmp_caches_load: seg_off not monotone at %d", i);
    const int32_t E = seg_off[n_caches];
    if (E > 0 && (!last_used || !weight)) return fail(c, MMP_EINVAL, "mmp_caches_load: null entry arrays");
    std::lock_guard<std::mutex> gb(c->batch_mu);
    std::lock_guard<std::shared_mutex> g(c->mu);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, quiesce_decisions(c));  // small eviction batches read these tables from the latency slots' streams
    HIP_TRY(c, c->c_seg.ensure((size_t)(n_caches + 1) * 4));
    HIP_TRY(c, c->c_lu.ensure((size_t)std::max(E, 1) * 8));
    HIP_TRY(c, c->c_wt.ensure((size_t)std::max(E, 1) * 4));
    HIP_TRY(c, c->c_cap.ensure((size_t)std::max(n_caches, 1) * 8));
    HIP_TRY(c, copy_sync(c, c->c_seg.p, seg_off, (size_t)(n_caches + 1) * 4, hipMemcpyHostToDevice));
    if (E) {
        HIP_TRY(c, copy_sync(c, c->c_lu.p, last_used, (size_t)E * 8, hipMemcpyHostToDevice));
        HIP_TRY(c, copy_sync(c, c->c_wt.p, weight, (size_t)E * 4, hipMemcpyHostToDevice));
    }
    if (n_caches) HIP_TRY(c, copy_sync(c, c->c_cap.p, capacity, (size_t)n_caches * 8, hipMemcpyHostToDevice));
    c->n_caches = n_caches;
    c->cache_entries = E;
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_caches_load");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_caches_load", e.what());
}

namespace {
// eight lanes per evaluation while the deques are short, sixteen otherwise (aux_kernels.hpp)
void evict_launch(mmp_ctx *c, const EvictArgs &A, int32_t n, hipStream_t st)
{
    if (c->cache_entries <= (int64_t)24 * std::max(c->n_caches, 1))
        hipLaunchKernelGGL(evict_batch_kernel<8>, dim3(div_up(n, kEvBlock / 8)), dim3(kEvBlock), 0, st, A);
    else
        hipLaunchKernelGGL(evict_batch_kernel<16>, dim3(div_up(n, kEvBlock / 16)), dim3(kEvBlock), 0, st, A);
}
}  // namespace

int mmp_evict_batch(mmp_ctx *c, const mmp_evict_req *reqs, int32_t n, int64_t now, mmp_evict_out *outs)
try {
    if (!c || n < 0 || (n > 0 && (!reqs || !outs))) return fail(c, MMP_EINVAL, "mmp_evict_batch: bad argument");
    if (n > 0 && (size_t)n * sizeof(mmp_evict_out) <= kFastN * sizeof(mmp_place_out)) {
        // latency path (see slot_acquire): one launch on a slot stream, no staging copies, no batch lock
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        std::unique_lock<std::mutex> fl;
        FastSlot *f = slot_acquire(c, fl);
        memcpy(f->reqs, reqs, (size_t)n * sizeof(mmp_evict_req));
        {
            std::shared_lock<std::shared_mutex> g(c->mu);  // capture the cache tables + enqueue
            if (c->n_caches <= 0) return fail(c, MMP_ESTATE, "no caches loaded");
            EvictArgs A;
            A.reqs = reinterpret_cast<const mmp_evict_req *>(f->reqs);
            A.seg_off = c->c_seg.as<int32_t>();
            A.last_used = c->c_lu.as<int64_t>();
            A.weight = c->c_wt.as<int32_t>();
            A.capacity = c->c_cap.as<int64_t>();
            A.outs = reinterpret_cast<mmp_evict_out *>(f->outs);
            A.n = n;
            A.n_caches = c->n_caches;
            A.now = now;
            A.done = DoneFlag{f->done, f->blocks, ++f->seq};
            evict_launch(c, A, n, f->stream);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, slot_wait(f, f->seq.load(std::memory_order_relaxed)));
        memcpy(outs, f->outs, (size_t)n * sizeof(mmp_evict_out));
        return MMP_OK;
    }
    // batch_mu owns c->stream and the scratch for the whole call, and every writer of the state this call reads
    // (commit, the loaders, registry events) takes it too: the published snapshot cannot change underneath.  The
    // state lock c->mu is NOT held: latency-path calls (mmp_place_batch / _gate / _evict on the slots) keep flowing.
    std::lock_guard<std::mutex> gb(c->batch_mu);
    if (c->n_caches <= 0 && n > 0) return fail(c, MMP_ESTATE, "no caches loaded");
    if (n == 0) return MMP_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->s_reqs.ensure((size_t)n * sizeof(mmp_evict_req)));
    HIP_TRY(c, c->s_outs.ensure((size_t)n * sizeof(mmp_evict_out)));
    HIP_TRY(c, hipMemcpyAsync(c->s_reqs.p, reqs, (size_t)n * sizeof(mmp_evict_req), hipMemcpyHostToDevice, st));
    EvictArgs A;
    A.reqs = c->s_reqs.as<mmp_evict_req>();
    A.seg_off = c->c_seg.as<int32_t>();
    A.last_used = c->c_lu.as<int64_t>();
    A.weight = c->c_wt.as<int32_t>();
    A.capacity = c->c_cap.as<int64_t>();
    A.outs = c->s_outs.as<mmp_evict_out>();
    A.n = n;
    A.n_caches = c->n_caches;
    A.now = now;
    A.done = DoneFlag{nullptr, nullptr, 0};
    KT_BEGIN(c, st);
    evict_launch(c, A, n, st);
    KT_END(c, st);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(outs, c->s_outs.p, (size_t)n * sizeof(mmp_evict_out), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    kt_collect(c);
    return MMP_OK;
} catch (const std::bad_alloc &) {
    return fail(c, MMP_ENOMEM, "%s: out of host memory", "mmp_evict_batch");
} catch (const std::exception &e) {
    return fail(c, MMP_EHIP, "%s: %s", "mmp_evict_batch", e.what());
}

}  // extern "C"
