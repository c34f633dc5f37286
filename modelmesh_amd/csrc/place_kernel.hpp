// place_kernel.hpp — load-target selection, one wavefront per decision.
//
// Device form of CacheMissForwardingLB.getNext + filter (MM.java:4760-5005).
// The Java walks clusterState (a skip list in PLACEMENT_ORDER) with sequential
// `break`s; here the pods are already stored in that order (snapshot.hpp), so:
//   filter()                    -> eligibility bitmap E (staged per wave in LDS,
//                                  per-request exclusions cleared with ds_and)
//   it.next() (best)            -> first set bit of E           (ballot + ctz)
//   "break at first violator"   -> min position over the break predicates
//   candidates list             -> bits of E in (best, firstBreak)
//   index-th non-null candidate -> wave prefix-popcount + select-in-word
// Every quirk of SURVEY.md Appendix B is kept (curInst substitution, fresh rpm,
// bestIsFull not recomputed, >>2 vs "half", ...).  `us` is evaluated in its
// closed form `pod == self`: with unique instance ids an eligible self is never
// excluded and the `!us &&` toggle never fires (tests/test_place_parity_gpu.py
// checks this against the literal CPU restatement).
#pragma once
#include "snapshot.hpp"

namespace mmp {

#ifndef MMP_PLACE_WAVES
#define MMP_PLACE_WAVES 4  // measured on C3, 100k decisions per launch: see DESIGN.md §4.1 (workgroup size)
#endif
constexpr int kPlaceWaves = MMP_PLACE_WAVES;  // waves (decisions in flight) per workgroup
constexpr int kPlaceBlock = kPlaceWaves * 64;

// Phase clock (tools/phase_clock.py builds a second library with -DMMP_PHASE_CLOCK; the product build carries
// none of it): wave-level s_memtime deltas between the markers of lane_decide / place_block, one row per wavefront.
#ifdef MMP_PHASE_CLOCK
__device__ unsigned int g_phase[4096][16];  // one row per wavefront of the launch (the last launch's values stay)
#define PHASE_ROW() (((blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)) & 4095)
#define PHASE_T0() unsigned long long ph_t_ = __builtin_amdgcn_s_memtime()
#define PHASE(k)                                                                                   \
    do {                                                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
        if (lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) g_phase[PHASE_ROW()][k] = (unsigned int)(t_ - ph_t_); \
        ph_t_ = t_;                                                                                \
    } while (0)
#define PHASE_COUNT(k, n)                                                                          \
    do {                                                                                           \
        if (lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) g_phase[PHASE_ROW()][k] += (unsigned int)(n); \
    } while (0)
#define PHASE_WHY(bit)  /* (long_memo_try) why a request was left to the walk: slot 7 counts such lanes, slot 9 ors the reasons */ \
    do {                                                                                           \
        atomicAdd(&g_phase[PHASE_ROW()][7], 1u);                                                   \
        atomicOr(&g_phase[PHASE_ROW()][9], (unsigned int)(bit));                                   \
        g_phase[PHASE_ROW()][13] = (unsigned int)r.model;                                          \
    } while (0)
#define PHASE_ABS(k)  /* the constant-rate clock all XCDs share (10 ns ticks, low 32 bits): when a wavefront started / ended, tools/r6/wave_timeline.py */ \
    do {                                                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                            \
        if (lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) g_phase[PHASE_ROW()][k] = (unsigned int)t_; \
    } while (0)
#else
#define PHASE_WHY(bit) do { } while (0)
#define PHASE_ABS(k) do { } while (0)
#define PHASE_T0() do { } while (0)
#define PHASE(k) do { } while (0)
#define PHASE_COUNT(k, n) do { } while (0)
#endif

// A request with its indirections followed (request -> model row -> instanceIds / failedIn -> rank
// positions), as the lane-per-decision path keeps it in registers.
constexpr int kInlineExcl = 8;
struct __attribute__((aligned(16))) ResolvedReq {
    int32_t type;     // bitmap row; -1: unknown model -> null
    int32_t selfpos;  // rank position of the caller, -1 if not in the table
    uint32_t flags, pick;
    int64_t last_used, fresh_lru, f_rem;
    int32_t fresh_count, fresh_rpm;
    int32_t n_excl;   // |loaded ∪ failed ∪ tried ∪ explicit|; > kInlineExcl: left to the wave path
    int32_t model;
    int32_t excl_pos[kInlineExcl];  // rank positions of the excluded pods, -1 = not in the table
    // Late-bound request exclusions (place_block only): the request's own exclusions (tried / explicit, <= kLateExtra)
    // are kept apart from the model's, so that the pod -> position gather they need — a third dependent load
    // level that 5 % of the requests, hence nearly every wavefront, has — is first USED after the workgroup barrier
    // and the window staging of lane_decide_win instead of being waited for inside resolve_one.  n_late < 0: not
    // late-bound, excl_pos already holds everything.  lane_decide_r's callers merge first (merge_late_extras).
    int32_t n_late, n_model;
    int32_t late_pos[4];
};
constexpr int kLateExtra = 4;
static_assert(sizeof(ResolvedReq) == 112, "ResolvedReq is 112 bytes");

// The registry view with its indirections followed once per (registry, snapshot) pair instead of once
// per decision: model row -> instanceIds / failedIn -> rank positions.  Rebuilt by
// resolve_models_kernel whenever the model table is reloaded or a snapshot is committed; it takes two
// levels (entry list, pos_of) out of the dependent-load chain of lane_decide.
constexpr int kResolvedInline = 6;
struct __attribute__((aligned(16))) ResolvedModel {
    int32_t type;    // bitmap row, already clamped to the snapshot's rows
    int32_t n_ents;  // |instanceIds| + |loadFailedInstanceIds|
    int32_t pos[kResolvedInline];  // rank positions of the first entries, -1 = pod not in the table
};
static_assert(sizeof(ResolvedModel) == 32, "ResolvedModel is 32 bytes");

// ---- per-type head windows ------------------------------------------------------------------------
// getNext reads, for one request, a few 64-pod words of the type's candidate bitmap right behind the type's first
// eligible position, the rows of one or two instances there, and one threshold word — and lane_decide_r fetches them
// through L2 one dependent load after the other (profiles/r1/phase_clock_place_batch_C3.txt).  All of it is the
// same few hundred bytes for every request of a type, so commit() packs it once per type (build_wins_kernel) into
// a TypeWin: the eligibility / preference / fullness words of a kWinWords-word window that starts at the type's
// first eligible word, the rows (lruTime, remaining, count, rpm, pod index) of the first kWinRows eligible and
// of the first kWinRows preferred-and-eligible positions in it, and — counts being non-decreasing along the
// window, which PLACEMENT_ORDER guarantees for non-full instances of one version (MM.java:4669-4677) and the build
// checks — for every count threshold the window position from which it holds.  place_block stages the windows
// of the first kWinLds types into LDS beside the request fetch; lane_decide_win is lane_decide_r's simple case
// (MM.java:4806-4991) on that window, with the request's exclusions and the caller's own entry applied, and
// reports kLaneHeadMiss — lane_decide_r then decides on the same resolved request — whenever an answer would need a
// bit, a row or a threshold from outside the window (case (b), the replay list, the replica-set retry, a shortlist
// that runs past the window ...).
constexpr int kWinWords = 6;   // 64-pod words per window
constexpr int kWinRows = 12;   // rows kept per list: with <= kInlineExcl exclusions the best position is among the first 9
#ifndef MMP_WIN_LDS
#define MMP_WIN_LDS 12
#endif
// type rows staged in LDS (types beyond take lane_decide_r).  12: windows (13 KB) + per-lane scratch (24 KB) + lists keep a
// workgroup under 40 KB, i.e. FOUR workgroups per CU for launches that overlap on several streams (measured, C3, 4
// streams: 16 rows 3.70 us per step, 8 rows 3.20)
constexpr int kWinLds = MMP_WIN_LDS;
constexpr int32_t kWinPosPreferred = 1 << 30;  // WinRow::pos bit: the instance is one of the type's preferred instances
constexpr int32_t kWinPosMask = kWinPosPreferred - 1;
struct __attribute__((aligned(16))) WinRow {
    int64_t lru, rem;
    int32_t cnt, rpm, orig;
    int32_t pos;  // rank position | kWinPosPreferred; -1: no such row
};
static_assert(sizeof(WinRow) == 32, "WinRow is 32 bytes");
struct __attribute__((aligned(16))) TypeWin {
    int32_t valid;   // 0: every decision of this type takes lane_decide_r
    int32_t w0;      // first word of the window (the word of the type's first eligible position)
    int32_t nw;      // words of the window that exist (it is cut at the end of the table)
    uint32_t flags;  // bit 1: the type has preferred instances (getPreferredInstances(type) != null); bit 2: the precomputed
                     // preference step does not apply (see build_wins_kernel)
    uint64_t E[kWinWords];   // elig words
    uint64_t Pm[kWinWords];  // pref words; all ones when the type has no preferred instances (D = E & Pm either way)
    uint64_t F[kWinWords];   // fullw words
    uint16_t ct[kGeRows];    // ct[r]: window positions with count < kGeBase + r == first window position with count >= kGeBase + r
    uint16_t pad_[2];
    WinRow rowsE[kWinRows];  // rows of the first eligible positions of the window
    WinRow rowsP[kWinRows];  // rows of the first preferred eligible positions (types with preferred instances)
};
static_assert(sizeof(TypeWin) % 16 == 0, "TypeWin is staged in 16-byte pieces");
static_assert(kGeRows % 2 == 0, "ct + pad_ keep the rows 16-byte aligned");

// ---- per-type shortlists (rounds 5-6) ---------------------------------------------------------------
// What lane_decide_win computes from a head window depends on the REQUEST in four ways only: its exclusions, the caller's own
// entry (position + fresh record), the rpm rule's inputs (lastUsedTime, the fresh rpm) and the random pick.  The first two reach
// the shortlist only when one of those positions lies inside it — and a shortlist is a few dozen positions at the head of an
// order of thousands.  For every other request of the type the walk — first eligible instance, preference step, break scans,
// count, audit hash — yields one of TWO shortlists, selected by the fresh-row test of MM.java:4913-4922 (which compares the
// CALLER's fresh record with the best instance: one bit per request).  commit() runs lane_decide_win once per (type, bit) with no
// exclusions (build_sel_memo_kernel: build_memo_body) and keeps: the positions the result depends on [lo, hi), the candidates' count and
// audit-hash sum, the best row's fields the rpm rule reads, the candidates' pod indices in shortlist order (Snap::memo_cand) and — round 6
// — for every position of the type's window what it IS to the list (Snap::memo_rk: the best instance, the type's first eligible
// instance, candidate number k, nothing).  memo_try decides a request from those records:
//   * no position of its own inside [lo, hi): the recorded list as it stands (round 5);
//   * round 6: own positions inside the list that leave its SHAPE alone — an exclusion that is candidate k of the list takes that
//     candidate out (count - 1, its own term off the linear audit hash, the pick skips rank k: at most two per request); the caller
//     as candidate k is one more candidate with the rpm class / favourSelf / ABORT_REQUEST rules of MM.java:4931, :4951-4991; the
//     caller as the best instance with favourSelf is ABORT_REQUEST at once (:4891-4895); a position that is no candidate (another
//     type's instance, the instance that ends the list) changes nothing;
//   * anything that would change the walk itself — the best / first eligible instance excluded, the caller as the best instance
//     without favourSelf, a third excluded candidate, the fresh-row break with the caller next in line — is left to the ordinary
//     lane phase (in the same wavefront: place_batch_m_kernel & co; or in the tail launch of the split form: place_memo_kernel +
//     place_tail_kernel below).
// The model's own exclusions are classified when its registry row is resolved (resolve_model_row -> PlaceArgs::mtw, one word per
// model: type row, "needs the ordinary path", the window offsets of up to two excluded candidates), so a covered request never reads
// the 32-byte registry row.  What else was tried with these records — a second phase per workgroup, a wavefront per 256 requests,
// launch-wide lists, consumer workgroups — is in profiles/r5/shortlist_experiments.
constexpr int kMemoCand = kWinWords * 64;  // positions of a window == the most candidates a recorded list can have
// PlaceArgs::mtw, one word per model
constexpr uint32_t kMtwTypeMask = 15u;     // bits 0-3: the model's type row if it has recorded shortlists (< kWinLds), else 15
constexpr uint32_t kMtwTail = 1u << 4;     // the model's exclusions change the walk for either list: the ordinary path decides
constexpr uint32_t kMtwTail1 = 1u << 5;    // ... for the list of a FULL caller (fresh-row break on): the instance that ends it is excluded
constexpr int kMtwOffShift = 8;            // bits 8-16, 17-25: window offset + 1 of the model's first / second exclusion that is a candidate
constexpr uint32_t kMtwOffMask = 511u;     //   of list 0 (0: none)
constexpr uint32_t kMtwNone = kMtwTypeMask | kMtwTail | kMtwTail1;
static_assert(kWinLds <= 15 && kMemoCand < 511, "mtw packs a type row in 4 bits and a window offset + 1 in 9");
constexpr int kMtwMaxEnts = 32;            // entries of a model the classification walks (more: the ordinary path)
// Snap::memo_rk values (int16 per window position): k >= 1: candidate k of list 0 (k = 0 is the best instance)
constexpr int kRkNone = -1;                // nothing to the list
constexpr int kRkFirst = -2;               // the type's first eligible instance where that is not the best one (a preference step was taken)
struct __attribute__((aligned(16))) MemoVar {
    int32_t valid;       // 0: lane_decide_win gave no answer from the window for this (type, bit)
    int32_t lo, hi;      // own positions of a request matter only inside [lo, hi)
    int32_t ccount;      // candidates (:4940)
    uint64_t hsum;       // audit-hash sum of the candidate words (linear in the candidate bits: wave.hpp)
    uint32_t hash;       // ... folded
    int32_t pad_;
};
static_assert(sizeof(MemoVar) == 32, "MemoVar is 32 bytes");
struct __attribute__((aligned(16))) TypeMemoHead {
    int64_t b_rem, b_lru;  // the best instance's remaining space and lruTime (the fresh-row test compares against them)
    int32_t best_is_full, b_rpm, best_idx;
    int32_t plain;         // 1: no preference step was taken (the best instance is the type's first eligible one, bestpos == best0)
    MemoVar v[2];          // [fresh-row break does not fire, fires]
    int32_t w0;            // first word of the type's window: memo_rk / mtw offsets count from position w0 * 64
    int32_t bestpos, best0;
    int32_t e_rpm;         // rpm of the first eligible instance's snapshot row: the class of the caller's own entry (quirk B#2)
    int32_t sbk;           // the caller's own entry would END the list (its break rule, :4909-4922 with curInst = bestEntry's row):
                           // only where a preference step was taken; such requests take the ordinary path
    int32_t end0;          // where list 0 ends (kNoPos: at the end of the table)
    int32_t pad_[2];
};
static_assert(sizeof(TypeMemoHead) == 128, "TypeMemoHead is 128 bytes");
// ... and, in the same 512-byte row, the head of list 0 — the pod indices of its first kMemoNear candidates and what the first
// kMemoNear positions from the type's first eligible instance on are to it (the first entries of memo_cand / memo_rk again, the latter
// counted from `best0` instead of the window's start): everything a request of the type normally touches in one contiguous piece that
// the lean launch copies into LDS, a copy per wavefront and no barrier (place_memo_body).
constexpr int kMemoNear = 64;
struct __attribute__((aligned(16))) TypeMemo : TypeMemoHead {
    int32_t cand64[kMemoNear];
    int16_t rk64[kMemoNear];
};
static_assert(sizeof(TypeMemo) == 512, "TypeMemo is 512 bytes");
// bytes of the table the lean launch stages per wavefront: whole 1 KB pieces (the table is allocated for kWinLds rows)
__host__ __device__ constexpr int memo_stage_bytes(int type_rows)
{
    return (((type_rows < kWinLds ? type_rows : kWinLds) * (int)sizeof(TypeMemo) + 1023) / 1024) * 1024;
}

// ---- recorded walks of the long shortlists (round 6) --------------------------------------------------------------------------------
// On a cluster whose instances are (nearly) all full a shortlist spans thousands of positions, so a request always has positions of
// its own inside it — and the prefix-table phase (lane_decide_r<..., LONG>) already treats those as corrections of a list it never
// builds: an excluded candidate takes one off the count, its term off the linear hash, one rank off the pick.  What it still walks
// per request — the first eligible instance, the preference step, the break scans, the prefix-table differences: a third of a
// wavefront's time — depends on the request only if an exclusion or the caller sits on one of the few positions that STEER the walk:
// [best0, bestpos] and the instance that ends the list.  commit runs the walk once per (type, fresh-row bit) without exclusions
// (build_long_memo_kernel) and long_memo_try answers from the record whenever none of those positions is the request's own.
struct __attribute__((aligned(16))) LongVar {
    int32_t valid;   // 0: no record for this (type, bit) — the walk left the common shape
    int32_t end;     // the list is best ∪ candidates of [bestpos + 1, end); the instance at `end` ended it (S.P: nothing did)
    int32_t ccount;  // 1 + the row's candidate bits in [bestpos + 1, end)
    int32_t g0;      // the number (pc's numbering) of the first candidate bit at or behind bestpos + 1
    uint64_t hsum;   // audit-hash sum of that list
    uint64_t pad_;
};
struct __attribute__((aligned(16))) LongMemo {
    int64_t b_rem, b_lru;  // the best instance's row: what the fresh-row test (:4913-4922) compares the caller's fresh record with
    int32_t best_is_full, b_rpm, best_idx, has_pm;
    int32_t best0, bestpos, e_rpm, sbk;  // e_rpm / sbk: the first eligible instance's rpm / the break rule of the caller's own entry (:4909-4922 on that row)
    LongVar v[2];          // [fresh-row break does not fire, fires]
    int32_t pad_[4];
};
static_assert(sizeof(LongVar) == 32 && sizeof(LongMemo) == 128, "LongMemo is one 128-byte line");
// Snap::lmemo[type * kLongLevels + l]: the walk of a request that excludes exactly the type's first l eligible instances — a model with
// a copy on the first instance of the order is one request in a few thousand, but a wavefront that holds one runs both the check and
// the walk, and at 100k requests per launch the slowest wavefront IS the launch (tools/r6/wave_timeline.py)
constexpr int kLongLevels = 3;
struct LongCap {  // what lane_decide_r<..., LONG> leaves for build_long_memo_kernel
    int32_t nsb;  // in: the fresh-row bit to assume
    int32_t ok;
    LongMemo m;   // head fields + v[nsb]
};

struct PlaceArgs {
    const mmp_place_req *reqs;
    const mmp_model_row *models;
    const ResolvedModel *rmodels;  // null: not built (pod-axis shard contexts)
    const int32_t *mtw;            // rmodels[i].type as an array of its own (the shortlist kernel's gather: 4 bytes per model instead of a 32-byte row)
    const TypeWin *wins;           // null: not built (pod-axis shard contexts, MMP_NO_HEADS=1)
    const struct BSlot *bslots;    // case (b) slots of the snapshot (long kernel only; see BSlot), n_bslots of them,
    const struct BLaunch *bwin;    // ... their whole-window tables (build_bsurv_kernel)
    const uint64_t *bsurv;
    const int32_t *bpcs;
    int32_t n_bslots;
    int32_t long_first;            // (0: no; 1 + o: yes, and with o != 0 the per-type tables the lanes search — elig, pref, pc, nz — are staged in
                                   // LDS at byte offset o of the dynamic region: long_tables_bytes(T, W))
                                   // long kernel on a snapshot whose instances are (nearly) all full: most shortlists span the table, so the
                                   // first lane phase runs the prefix-table instantiation at once instead of window -> lane -> long
    const int32_t *ent_pod;  // model entries: loaded ids then failed ids
    const int32_t *extra;    // per-request extra exclusions
    mmp_place_out *outs;
    int32_t n;
    int32_t n_models;
    int64_t now;
    const int32_t *n_dev;  // pod-axis rest sub-batch: the row count lives on the device (min(n, *n_dev) rows are real); null otherwise
    int32_t force_wave;  // diagnostics: hand every decision to the wave-per-decision kernel
    int32_t n_pods_all;  // pod slots of the whole table (bounds of pos_of; == Snap::P unless the Snap is a shard view)
    // Latency path only (wave.hpp: announce_done); nullptr otherwise.  The workgroup counter of launches with
    // more than one workgroup is an argument of place_batch_flag_kernel, not part of this block.
    uint32_t *done_flag;
    uint32_t done_seq;
    int32_t extra_bound;  // bounded calls: 1 + the entries of `extra` the caller declared; 0: not declared, requests are followed as they are
};

// The request of decision d.  FORM kReq64: an mmp_place_req row.  kReqC: the single-caller form — an mmp_place_req_c row (24 B) and
// the caller's side from the kernel's arguments (wave-uniform: the caller's position, remaining space and every test on them are
// scalar work then).
enum { kReq64 = 0, kReqC = 1 };
template <int FORM>
__device__ __forceinline__ mmp_place_req fetch_req(const PlaceArgs &A, const mmp_place_caller &C, int d)
{
    if (FORM == kReq64) return A.reqs[d];
    const mmp_place_req_c q = reinterpret_cast<const mmp_place_req_c *>(A.reqs)[d];
    mmp_place_req r;
    r.model = q.model;
    r.self_pod = C.self_pod;
    r.flags = C.flags;
    r.pick = q.pick;
    r.last_used = q.last_used;
    r.extra_off = q.extra_off;
    r.n_extra = q.n_extra;
    r.fresh_lru = C.fresh_lru;
    r.fresh_capacity = C.fresh_capacity;
    r.fresh_used = C.fresh_used;
    r.fresh_count = C.fresh_count;
    r.fresh_rpm = C.fresh_rpm;
    return r;
}
// a request whose exclusion range leaves the pool the caller declared (PlaceArgs::extra_bound != 0)
__device__ __forceinline__ bool bad_extra_range(const PlaceArgs &A, const mmp_place_req &rq)
{
    return rq.n_extra < 0 || (rq.n_extra > 0 && (rq.extra_off < 0 || (int64_t)rq.extra_off + rq.n_extra > (int64_t)A.extra_bound - 1));
}

// (int)(double) with Java narrowing semantics
__device__ __forceinline__ int32_t jd2i(double d)
{
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}

__device__ __forceinline__ bool test_bit(const uint64_t *m, int pos) { return (m[pos >> 6] >> (pos & 63)) & 1ull; }

// bits of word w that fall inside positions [lo, hi)
__device__ __forceinline__ uint64_t clip_word(uint64_t v, int w, int lo, int hi)
{
    const int wl = lo >> 6, wh = hi >> 6;
    if (w < wl || w > wh) return 0;
    if (w == wl) v &= (~0ull) << (lo & 63);
    if (w == wh) v &= (1ull << (hi & 63)) - 1ull;
    return v;
}

// first position >= start whose bit is set in (ew & andmask); kNoPos if none
__device__ __forceinline__ int first_set_from(const uint64_t *ew, const uint64_t *andmask, int start, int W)
{
    const int lane = lane_id();
    if (start >= W * 64) return kNoPos;
    const int w0 = start >> 6;
    for (int base = w0; base < W; base += 64) {
        const int w = base + lane;
        uint64_t v = 0;
        if (w < W) {
            v = ew[w];
            if (andmask) v &= andmask[w];
            if (w == w0) v &= (~0ull) << (start & 63);
        }
        const uint64_t b = __ballot(v != 0);
        if (b) {
            const int l = __ffsll((unsigned long long)b) - 1;
            const uint64_t vv = readlane_u64(v, l);
            return (base + l) * 64 + (__ffsll((unsigned long long)vv) - 1);
        }
    }
    return kNoPos;
}

// first position in [start,end) with bit set in (ew&andmask) and
// cnt >= 10 && cnt > thr   (MM.java:4925-4926).  Four 64-pod groups per trip: their count loads are
// independent, so a scan that has to walk several groups pays one memory latency per four groups.
__device__ __forceinline__ int first_count_break(const uint64_t *ew, const uint64_t *andmask, int start,
                                                 int end, const int32_t *cnt, int32_t thr)
{
    const int lane = lane_id();
    if (start >= end) return kNoPos;
    const int g1 = (end - 1) >> 6;
    for (int g = start >> 6; g <= g1; g += 4) {
        uint64_t word[4];
        int32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            word[j] = 0;
            if (g + j <= g1) {
                uint64_t w = ew[g + j];
                if (andmask) w &= andmask[g + j];
                word[j] = clip_word(w, g + j, start, end);
            }
        }
        if ((word[0] | word[1] | word[2] | word[3]) == 0) continue;  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = word[j] ? cnt[(g + j) * 64 + lane] : 0;  // columns are padded to W*64
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool hit = ((word[j] >> lane) & 1ull) && c[j] >= 10 && c[j] > thr;
            const uint64_t b = __ballot(hit);
            if (b) return (g + j) * 64 + (__ffsll((unsigned long long)b) - 1);
        }
    }
    return kNoPos;
}

// The count break (MM.java:4925-4926) of the wave path through the snapshot's threshold bitmaps (Snap::ge, as the lane path):
// the first position in [start, end) set in ew & andmask & G — 64 WORDS per step instead of 64 pods per step of the count
// column.  (On a churned C3 fleet the single-decision path met shortlists of ~2700 instances: eleven dependent trips through
// the count column, 40-60 us per decision; one step here.)
__device__ __forceinline__ int first_ge_break(const uint64_t *ew, const uint64_t *andmask, const uint64_t *G, int start, int end)
{
    const int lane = lane_id();
    if (start >= end) return kNoPos;
    const int w0 = start >> 6, w1 = (end - 1) >> 6;
    for (int base = w0; base <= w1; base += 64) {
        const int w = base + lane;
        uint64_t v = 0;
        if (w <= w1) {
            v = ew[w] & G[w];
            if (andmask) v &= andmask[w];
            v = clip_word(v, w, start, end);
        }
        const uint64_t b = __ballot(v != 0);
        if (b) {
            const int l = __ffsll((unsigned long long)b) - 1;
            const uint64_t vv = readlane_u64(v, l);
            return (base + l) * 64 + (__ffsll((unsigned long long)vv) - 1);
        }
    }
    return kNoPos;
}

// first position >= start with bit set in ew and
// (lru - oldest) > abs_ms && (lru - oldest) > rel   (MM.java:4862-4866)
__device__ __forceinline__ int first_lru_break(const uint64_t *ew, int start, int W, const int64_t *lru,
                                               int64_t oldest, int64_t abs_ms, int64_t rel)
{
    const int lane = lane_id();
    if (start >= W * 64) return kNoPos;
    for (int g = start >> 6; g < W; g++) {
        const uint64_t word = clip_word(ew[g], g, start, W * 64);
        if (word == 0) continue;
        const int64_t d = jsub64(lru[g * 64 + lane], oldest);
        const bool hit = ((word >> lane) & 1ull) && d > abs_ms && d > rel;
        const uint64_t b = __ballot(hit);
        if (b) return g * 64 + (__ffsll((unsigned long long)b) - 1);
    }
    return kNoPos;
}

// One wavefront per type row (see TypeWin).
__device__ __forceinline__ void build_wins_kernel_body(int bid, int nblk, Snap S, TypeWin *__restrict__ wins)
{
    const int t = bid, lane = lane_id();
    const int P = S.P, W = S.W;
    const uint64_t *E = S.elig + (size_t)t * W;
    const uint64_t *Pm = S.pref + (size_t)t * W;
    const bool has_pm = S.has_pref[t] != 0;
    TypeWin *out = &wins[t];
    const int best0 = first_set_from(E, nullptr, 0, W);
    const int w0 = best0 == kNoPos ? 0 : best0 >> 6;
    const int nw = best0 == kNoPos ? 0 : (W - w0 < kWinWords ? W - w0 : kWinWords);
    const int lo = w0 * 64, hi = (w0 + nw) * 64 < P ? (w0 + nw) * 64 : P;  // window positions [lo, hi)
    // counts must not decrease along the window: then "count >= T" holds exactly from position lo + ct[T - kGeBase] on
    bool mono = true;
    for (int base = lo; base < hi; base += 64) {
        const int p = base + lane;
        const bool bad = p + 1 < hi && S.cnt[p] > S.cnt[p + 1];
        if (__ballot(bad)) mono = false;
    }
    if (lane < kWinWords) {
        const bool in = lane < nw;
        out->E[lane] = in ? E[w0 + lane] : 0ull;
        out->Pm[lane] = in ? (has_pm ? Pm[w0 + lane] : ~0ull) : 0ull;
        out->F[lane] = in ? S.fullw[w0 + lane] : 0ull;
    }
    if (lane < kGeRows) {
        // first window position whose count reaches kGeBase + lane (binary search on the non-decreasing counts)
        const int32_t T = kGeBase + lane;
        int a = lo, b = hi;
        while (a < b) {
            const int m = (a + b) >> 1;
            if (mono && S.cnt[m] < T)
                a = m + 1;
            else
                b = m;
        }
        out->ct[lane] = (uint16_t)(a - lo);
    }
    if (lane < 2) out->pad_[lane] = 0;
    // rows of the first kWinRows set bits of E (lanes 0..kWinRows-1) and of E & Pm (lanes 32..32+kWinRows-1)
    {
        const bool second = lane >= 32;
        const int k = lane & 31;
        if (k < kWinRows) {
            int pos = -1, left = k;
            for (int j = 0; j < nw && pos < 0; j++) {
                uint64_t v = E[w0 + j];
                if (second) v = has_pm ? (v & Pm[w0 + j]) : 0ull;
                const int c = __popcll((unsigned long long)v);
                if (left < c)
                    pos = (w0 + j) * 64 + select_kth_bit(v, left);
                else
                    left -= c;
            }
            WinRow r;
            r.lru = r.rem = 0;
            r.cnt = r.rpm = 0;
            r.orig = -1;
            r.pos = -1;
            if (pos >= 0) {
                r.lru = S.lru[pos];
                r.rem = S.rem[pos];
                r.cnt = S.cnt[pos];
                r.rpm = S.rpm[pos];
                r.orig = S.orig[pos];
                r.pos = pos | ((has_pm && ((Pm[pos >> 6] >> (pos & 63)) & 1ull)) ? kWinPosPreferred : 0);
            }
            (second ? out->rowsP : out->rowsE)[k] = r;
        }
    }
    if (lane == 0) {
        out->valid = (best0 != kNoPos && mono && nw > 0) ? 1 : 0;
        out->w0 = w0;
        out->nw = nw;
        // The preference step of a decision whose best instance is not a preferred one (MM.java:4817-4888, case (a)) moves to
        // the first preferred eligible position q provided no FULL eligible instance lies before it.  Without exclusions q is
        // rowsP[0] for every decision of the type, and exclusions only remove instances: as long as rowsP[0] itself is not
        // excluded it stays the first, and a stretch without full instances stays one.  Bit 2 says the shortcut does NOT
        // hold for this type: no preferred eligible position in the window, or a full eligible instance before it.
        uint32_t fl = has_pm ? 2u : 0u;
        if (has_pm) {
            int q = kNoPos;
            for (int j = 0; j < nw && q == kNoPos; j++) {
                const uint64_t v = E[w0 + j] & Pm[w0 + j];
                if (v) q = (w0 + j) * 64 + (__ffsll((unsigned long long)v) - 1);
            }
            bool blocked = q == kNoPos || q >= hi;
            for (int w = w0; !blocked && w <= (q >> 6); w++) {
                uint64_t v = E[w] & S.fullw[w];
                if (w == (q >> 6)) v &= (1ull << (q & 63)) - 1ull;  // positions before q
                if (v) blocked = true;
            }
            if (blocked) fl |= 4u;
        }
        out->flags = fl;
    }
}
__global__ __launch_bounds__(64) void build_wins_kernel(Snap S, TypeWin *__restrict__ wins)
{
    build_wins_kernel_body((int)blockIdx.x, (int)gridDim.x, S, wins);
}

// ---- case (b) of getNext on a full cluster (round 3) --------------------------------------------------------------
// A type with preferred instances whose most desirable eligible instance is FULL and not preferred takes the non-simple
// case (b), MM.java:4853-4887: the shortlist is every preferred eligible instance whose lruTime lies within
// max(120 s, age(oldest) / 4) of the oldest, each with its OWN rpm in the rpm rule (:4875).  On a cluster where every
// instance is full and the caches are about equally old — the steady state of a mesh — that window is the whole table:
// thousands of candidates per decision, which the wave path walked with two dependent rpm loads per candidate (40 us
// per decision; 70 of the 90 us a 100k batch took on such a fleet).  Nearly all of it is the same for every request
// of the type: full instances of one version stand in lruTime order, so the window is a position range (best0, lim)
// that depends on the type and the launch's clock only; with it the smallest candidate rpm, hence minLoad and the four
// rpm limits the rule can apply (:4957-4972), hence — per limit — the bitmap of candidates the rule leaves in.
// commit() records per type where case (b) starts and checks the order (BSlot, build_bslots_kernel), builds the
// running minimum of the candidates' rpm along the order (prefix_min_rpm_kernel) and, for the window that reaches the
// last full instance, limits, the five survivor bitmaps and their running counts (build_bsurv_kernel); a decision
// is then the LONG phase's arithmetic — range counts and hash terms from prefix tables, a correction per exclusion
// that is a candidate, a binary search for the index-th survivor.  The tables are anchored at the type's first eligible
// position p0; a request whose own first eligible instance lies behind it (it excludes p0 ...: every eligible position
// before its first one is among its exclusions, so the candidates there fall out as excluded candidates do) uses them as
// long as its window ends where the type's does.  The caller itself as the first instance, an excluded candidate that
// may hold the minimum, a window that ends elsewhere: the wave path.  (Measured, C3 with every instance full and equally old:
// the type's p0 happens to be preferred there, so case (b) is the ~10 requests per 100k that exclude it — and one such
// decision on the wave path took 80 us, which was the whole launch: 90 us per 100k before, 20 us without them.)
constexpr int kBSlots = 4;     // preferring types with a case (b) slot per snapshot (more: the wave path)
constexpr int kBClasses = 5;   // which clauses of the rpm rule apply: none, < 1 day, < 12 min, < 5 s, < -1 s (:4964-4972)
struct __attribute__((aligned(16))) BSlot {
    int32_t type;       // bitmap row
    int32_t best0;      // p0, the type's first eligible position: full
    int32_t end;        // positions [best0, end) are present, full and in lruTime order
    int32_t best_orig;  // pod index at best0
    int64_t oldest;     // lruTime at best0
    int32_t slot;       // row of the running-minimum table
    int32_t pad;
};
static_assert(sizeof(BSlot) == 32, "BSlot is 32 bytes");

// One wavefront per type row: the slot of a preferring type (valid: type >= 0), in the order of the type rows.
// `slots` = kBSlots rows, `n_slots` = how many are valid (written by the last type's wavefront: one launch, T blocks, so
// the slot index is assigned by an atomic ticket and the table is sorted by nothing — a decision finds its slot by type).
__device__ __forceinline__ void build_bslots_kernel_body(int bid, int nblk, Snap S, const mmp_pod_row *__restrict__ pods, BSlot *__restrict__ slots,
                                                          int32_t *__restrict__ n_slots)
{
    const int t = bid, lane = lane_id();
    if (!S.has_pref[t]) return;
    const int P = S.P, W = S.W;
    const uint64_t *E = S.elig + (size_t)t * W, *Pm = S.pref + (size_t)t * W;
    const int best0 = first_set_from(E, nullptr, 0, W);
    if (best0 == kNoPos) return;
    if (!test_bit(S.fullw, best0)) return;  // case (b) needs a full first instance (whether a request's first one is preferred is the request's)
    // [best0, end): present rows; all full, lruTime non-decreasing (one version: PLACEMENT_ORDER :4669-4674)
    int end = P;
    bool ok = true;
    for (int base = best0; base < P; base += 64) {
        const int p = base + lane;
        bool absent = false, bad = false;
        if (p < P) {
            absent = (pods[S.orig[p]].flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)) != 0;
            if (!absent) {
                if (!test_bit(S.fullw, p)) bad = true;
                if (p > best0 && jsub64(S.lru[p], S.lru[p - 1]) < 0) bad = true;
            }
        }
        const uint64_t ab = __ballot(absent), bb = __ballot(bad);
        const int first_absent = ab ? base + (__ffsll((unsigned long long)ab) - 1) : kNoPos;
        if (bb && base + (__ffsll((unsigned long long)bb) - 1) < first_absent) ok = false;
        if (first_absent != kNoPos) {
            end = first_absent;
            break;
        }
    }
    if (!ok) return;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(n_slots, 1);
    slot = readlane_i32(slot, 0);
    if (slot >= kBSlots) return;  // (n_slots may exceed kBSlots: readers clamp)
    if (lane == 0) {
        BSlot b;
        b.type = t;
        b.best0 = best0;
        b.end = end;
        b.best_orig = S.orig[best0];
        b.oldest = S.lru[best0];
        b.slot = slot;
        b.pad = 0;
        slots[slot] = b;
    }
}
__global__ __launch_bounds__(64) void build_bslots_kernel(Snap S, const mmp_pod_row *__restrict__ pods, BSlot *__restrict__ slots,
                                                          int32_t *__restrict__ n_slots)
{
    build_bslots_kernel_body((int)blockIdx.x, (int)gridDim.x, S, pods, slots, n_slots);
}

// ---- a commit as LEVELS of its dependency graph: every table build that needs only the rank-ordered columns runs in one launch,
// every build that needs only those tables in the next (a commit is a dozen small builds, each a few microseconds of launch
// latency when they queue one behind the other: 94 of the 170 us of a commit after a few changed rows) ----------------------------
struct CommitL1 {
    const mmp_pod_row *pods;
    int32_t P, W, T;
    int64_t min_space;
    const int32_t *orig, *cnt;
    const uint64_t *allowed, *prefer;
    const uint8_t *has_allowed, *has_prefer, *rs_bad;
    uint64_t *elig, *elig_nors, *pref, *fullw, *ge;
    int32_t *ctpos;
    StatsAcc *acc;
    const int32_t *pod_pts;
    int32_t NP;
    StatsAcc *pstats;
    int32_t nb_masks, nb_ge, nb_stats, nb_pstats;  // workgroups per part; one more builds ctpos
};
// level 1 (needs the scattered columns + the table): type bitmaps, count-threshold bitmaps, ctpos, cluster stats, partition stats
__global__ __launch_bounds__(256) void commit_level1_kernel(CommitL1 A)
{
    int b = blockIdx.x;
    if (b < A.nb_stats) {  // (the longest part first)
        cluster_stats_kernel_body(b, A.nb_stats, A.pods, A.P, A.min_space, A.acc);
        return;
    }
    b -= A.nb_stats;
    if (b < A.nb_pstats) {
        partition_stats_kernel_body(b, A.nb_pstats, A.pods, A.P, A.min_space, A.pod_pts, A.NP, A.pstats);
        return;
    }
    b -= A.nb_pstats;
    if (b == 0) {
        build_ctpos_block(A.cnt, A.P, A.ctpos);
        return;
    }
    b -= 1;
    if (b < A.nb_masks) {
        build_masks_kernel_body(b, A.nb_masks, A.pods, A.P, A.W, A.T, A.min_space, A.orig, A.allowed, A.has_allowed, A.prefer,
                                A.has_prefer, A.rs_bad, A.elig, A.elig_nors, A.pref, A.fullw);
        return;
    }
    b -= A.nb_masks;
    build_ge_kernel_body(b, A.nb_ge, A.cnt, A.P, A.W, A.ge);
}

struct CommitL2 {
    Snap S;
    const mmp_pod_row *pods;
    TypeWin *wins;
    BSlot *slots;
    int32_t *n_slots;
    StatsAcc *acc;
    StatsAcc *pstats;
    int32_t NP, Tw;
    const uint64_t *prohib;
    const uint8_t *has_allowed;
    StatsAcc *tstats;
    int32_t nb_finish;
};
// level 2 (needs level 1): prefix tables (2T wavefronts), head windows (T), case-(b) slots (T), subset stats (partitions / types)
__global__ __launch_bounds__(64) void commit_level2_kernel(CommitL2 A)
{
    int b = blockIdx.x;
    const int T = A.S.T;
    if (b < 2 * T) {
        build_prefix_kernel_body(b, 2 * T, A.S.elig, A.S.pref, T, A.S.W, const_cast<int32_t *>(A.S.pc), const_cast<uint64_t *>(A.S.ph),
                                 const_cast<int32_t *>(A.S.nz), A.acc);
        return;
    }
    b -= 2 * T;
    if (b < T) {
        build_wins_kernel_body(b, T, A.S, A.wins);
        return;
    }
    b -= T;
    if (b < T) {
        build_bslots_kernel_body(b, T, A.S, A.pods, A.slots, A.n_slots);
        return;
    }
    b -= T;
    subset_stats_finish_kernel_body(b, A.nb_finish, A.acc, A.pstats, A.NP, A.prohib, A.Tw, T, A.has_allowed, A.tstats);
}

// pm[slot][p] = min rpm over the preferred eligible positions in (best0, p]; INT32_MAX before the first one.
// One wavefront per slot, 64 positions per step.
// sel / rk (Snap): one wavefront per 64-position word of a candidate bitmap row (variant 0: elig, 1: elig & pref), a lane per position
__device__ __forceinline__ void build_sel_body(int bid, const Snap &S, int32_t *__restrict__ sel, int32_t *__restrict__ rk)
{
    const int W = S.W, T = S.T;
    const int row = bid / W, w = bid - row * W;  // row = variant * T + type
    const int type = row >= T ? row - T : row;
    uint64_t v = S.elig[(size_t)type * W + w];
    if (row >= T) v &= S.pref[(size_t)type * W + w];
    const int base = S.pc[(size_t)row * (W + 1) + w];
    const int l = lane_id();
    const bool bit = (v >> l) & 1ull;
    const int k = base + __popcll((unsigned long long)(v & ((1ull << l) - 1ull)));
    const size_t r0 = (size_t)row * (size_t)W * 64;
    rk[r0 + (size_t)w * 64 + l] = bit ? k : -1;
    if (bit) sel[r0 + k] = w * 64 + l;
}

__global__ __launch_bounds__(64) void prefix_min_rpm_kernel(Snap S, const BSlot *__restrict__ slots, const int32_t *__restrict__ n_slots,
                                                            int32_t *__restrict__ pm, int32_t stride)
{
    const int s = blockIdx.x, lane = lane_id();
    const int n = *n_slots < kBSlots ? *n_slots : kBSlots;
    if (s >= n) return;
    const BSlot b = slots[s];
    const uint64_t *E = S.elig + (size_t)b.type * S.W, *Pm = S.pref + (size_t)b.type * S.W;
    int32_t *out = pm + (size_t)s * stride;
    int32_t carry = INT32_MAX;
    for (int base = (b.best0 >> 6) << 6; base < b.end; base += 64) {
        const int p = base + lane;
        int32_t v = INT32_MAX;
        if (p > b.best0 && p < b.end && (((E[p >> 6] & Pm[p >> 6]) >> (p & 63)) & 1ull)) v = S.rpm[p];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {  // inclusive min-scan
            const int32_t tv = __shfl_up(v, o, 64);
            if (lane >= o && tv < v) v = tv;
        }
        v = v < carry ? v : carry;
        if (p < S.W * 64) out[p] = v;
        carry = readlane_i32(v, 63);
    }
}

// Per slot, built at commit (build_bsurv_kernel): the whole-window tables.
struct BLaunch {
    int32_t lim;        // window = positions (best0, lim)
    int32_t min_load;   // max(100, min rpm of the candidates); 0: no candidate in the window (the replay list, :4879-4884)
    int32_t min_rpm;    // that minimum itself
    int32_t limit[kBClasses];  // rpm limit of class c (c = 0: none applies -> INT32_MAX)
    int32_t wlo, whi;   // words of the window
};
struct BLds {
    const BSlot *slots;     // global
    int32_t n_slots;
    int32_t W;
    const BLaunch *launch;  // [n_slots]
    const uint64_t *surv;   // [n_slots][kBClasses][W]: candidates the rule of class c leaves in (class 0: all candidates)
    const int32_t *pcs;     // [n_slots][kBClasses][W + 1]: running counts of surv
};

struct RpmRule {
    bool active;  // lastUsedAgo < FIVE_DAYS_MS
    int64_t ago;
    int32_t min_load, m11, m15, m3, m4;
    __device__ __forceinline__ void init(int64_t ago_, int32_t min_rpm)
    {
        ago = ago_;
        active = ago < 5LL * 24 * 3600 * 1000;
        min_load = min_rpm > 100 ? min_rpm : 100;       // MM.java:4957
        // :4958  (int)(1.1 * minLoad), (int)(1.5 * minLoad) in integer arithmetic.  For every 0 <= x < 2^31:
        // (int)(1.1 * x) == min(x + x / 10, INT_MAX) — double(1.1) lies 8.9e-17 above 1.1, so the rounded product
        // never falls below an integer it should reach and is never more than 2.4e-7 off — and 1.5 * x is exact:
        // (int)(1.5 * x) == min(x + x / 2, INT_MAX)  (Java's narrowing saturates).  Checked exhaustively over all
        // 2^31 values (the CPU test suite samples it: test_rpm_thresholds_in_integer_arithmetic); the two
        // conversions and two double products were a tenth of the decision's instruction time.
        const uint32_t x = (uint32_t)min_load;
        const uint32_t t11 = x + x / 10u, t15 = x + (x >> 1);
        m11 = t11 > 0x7fffffffu ? INT32_MAX : (int32_t)t11;
        m15 = t15 > 0x7fffffffu ? INT32_MAX : (int32_t)t15;
        m3 = (int32_t)((uint32_t)min_load * 3u);
        m4 = (int32_t)((uint32_t)min_load * 4u);
    }
    // MM.java:4961-4972
    __device__ __forceinline__ bool nulls(int32_t rpm) const
    {
        return active && rpm >= 100 &&
               ((ago < -1000LL && rpm > m11) || (ago < 5000LL && rpm > m15) ||
                (ago < 12LL * 60 * 1000 && rpm > m3) || (ago < 24LL * 3600 * 1000 && rpm > m4));
    }
    // The same rule as ONE threshold: the age clauses are nested (-1 s < 5 s < 12 min < 1 day), so the clauses that
    // apply to this request are a suffix of the list and "some applicable clause has rpm > m_i" is rpm > the
    // smallest applicable m_i (taken explicitly: 3x / 4x wrap like the Java's int products, so the m_i need not
    // be ordered).  rpm nulls iff rpm >= 100 && rpm > limit(); limit() == INT32_MAX when no clause applies.
    __device__ __forceinline__ int32_t limit() const
    {
        int32_t t = INT32_MAX;
        if (active && ago < 24LL * 3600 * 1000) {
            t = m4;
            if (ago < 12LL * 60 * 1000) {
                t = m3 < t ? m3 : t;
                if (ago < 5000LL) {
                    t = m15 < t ? m15 : t;
                    if (ago < -1000LL) t = m11 < t ? m11 : t;
                }
            }
        }
        return t;
    }
};

// index-th set bit (in position order) over fw words [wlo, whi].  A shortlist is normally a few
// words, so the non-empty words are visited one by one with scalar code (ballot + readlane); only a
// trip with many non-empty words takes the wave-wide prefix scan.
__device__ __forceinline__ int select_in_range(const uint64_t *fw, int wlo, int whi, int index)
{
    const int lane = lane_id();
    int running = 0;
    for (int base = wlo; base <= whi; base += 64) {
        const int w = base + lane;
        const uint64_t v = (w <= whi) ? fw[w] : 0ull;
        uint64_t nz = __ballot(v != 0);
        if (__popcll((unsigned long long)nz) <= 8) {
            while (nz) {
                const int l = __ffsll((unsigned long long)nz) - 1;
                nz &= nz - 1;
                const uint64_t vv = readlane_u64(v, l);
                const int c = __popcll((unsigned long long)vv);
                if (index < running + c) return (base + l) * 64 + select_kth_bit(vv, index - running);
                running += c;
            }
            continue;
        }
        const int c = __popcll((unsigned long long)v);
        const int incl = wave_incl_scan_i32(c);
        const int total = readlane_i32(incl, 63);
        if (index < running + total) {
            const uint64_t b = __ballot(running + incl > index);
            const int l = __ffsll((unsigned long long)b) - 1;
            const uint64_t vv = readlane_u64(v, l);
            const int before = running + readlane_i32(incl, l) - readlane_i32(c, l);
            return (base + l) * 64 + select_kth_bit(vv, index - before);
        }
        running += total;
    }
    return kNoPos;
}

__device__ __forceinline__ void write_out(mmp_place_out *o, int32_t chosen, int32_t best, int32_t n,
                                          uint32_t hash)
{
    if (lane_id() == 0) {
        o->chosen = chosen;
        o->best = best;
        o->n_candidates = n;
        o->hash = hash;
    }
}

// Stage the eligibility bitmap of this decision into LDS and clear the
// CacheMissExcludeSet members (MM.java:4740-4743).
__device__ __forceinline__ void stage_eligible(const Snap &S, const uint64_t *src, uint64_t *ew,
                                               const int32_t *ents, int32_t n_ents, const int32_t *extra,
                                               int32_t n_extra)
{
    const int lane = lane_id();
    for (int w = lane; w < S.W; w += 64) ew[w] = src[w];
    wave_sync();
    const int nex = n_ents + n_extra;
    for (int i = lane; i < nex; i += 64) {
        const int32_t pod = i < n_ents ? ents[i] : extra[i - n_ents];
        if (pod >= 0 && pod < S.P) {
            const int pos = S.pos_of[pod];
            atomicAnd((unsigned long long *)&ew[pos >> 6], ~(1ull << (pos & 63)));
        }
    }
    wave_sync();
}

// global rank position -> position inside the view (-1: not in it); the identity on a whole snapshot
__device__ __forceinline__ int32_t view_pos(const Snap &S, int32_t gpos)
{
    const int32_t l = gpos - S.pos_base;
    return (l < 0 || l >= S.P) ? -1 : l;
}

// pod index -> position inside the view; P_all = pod slots of the whole table (== S.P on a whole snapshot)
template <bool VIEW>
__device__ __forceinline__ int32_t pod_view_pos(const Snap &S, int32_t pod, int32_t P_all)
{
    if (!VIEW) return (pod >= 0 && pod < P_all) ? S.pos_of[pod] : -1;
    return (pod >= 0 && pod < P_all) ? view_pos(S, S.pos_of[pod]) : -1;
}

// Follow request -> model row -> exclusion lists -> rank positions (one lane).  VIEW: S is a shard's view
// of its slice (positions are translated into it, pos_of is bounded by the whole table's pod count).
template <bool VIEW, bool LATE = false>
__device__ __forceinline__ ResolvedReq resolve_req(const Snap &S, const PlaceArgs &A, const mmp_place_req &rq);

template <bool VIEW, bool LATE = false, int FORM = kReq64>
__device__ __forceinline__ ResolvedReq resolve_one(const Snap &S, const PlaceArgs &A, int d, const mmp_place_caller &C = mmp_place_caller{})
{
    const mmp_place_req rq = fetch_req<FORM>(A, C, d);
    return resolve_req<VIEW, LATE>(S, A, rq);
}

template <bool VIEW, bool LATE>
__device__ __forceinline__ ResolvedReq resolve_req(const Snap &S, const PlaceArgs &A, const mmp_place_req &rq)
{
    const int32_t P_all = VIEW ? A.n_pods_all : S.P;
    ResolvedReq r;
    r.flags = rq.flags;
    r.pick = rq.pick;
    r.last_used = rq.last_used;
    r.fresh_lru = rq.fresh_lru;
    r.f_rem = remaining_of(rq.fresh_capacity, rq.fresh_used);
    r.fresh_count = rq.fresh_count;
    r.fresh_rpm = rq.fresh_rpm;
    r.model = rq.model;
    r.selfpos = pod_view_pos<VIEW>(S, rq.self_pod, P_all);
    r.n_late = -1;
    r.n_model = 0;
#pragma unroll
    for (int j = 0; j < kLateExtra; j++) r.late_pos[j] = -1;
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) r.excl_pos[i] = -1;
    // The request's own exclusions (tried-this-request / explicit) depend on the request alone: their pool
    // entries and rank positions are fetched NOW, beside the model row, instead of behind it — with 5 % of the
    // requests carrying some, nearly every wavefront would otherwise walk request -> model row -> pool ->
    // pos_of as four dependent loads (tools/phase_clock.py: this step was the longest of the decision).
    constexpr int kEarlyExtra = 4;
    int32_t xpos[kEarlyExtra];
    const bool early_extra = rq.n_extra > 0 && rq.n_extra <= kEarlyExtra;
#pragma unroll
    for (int j = 0; j < kEarlyExtra; j++) {
        xpos[j] = -1;
        if (early_extra && j < rq.n_extra) xpos[j] = pod_view_pos<VIEW>(S, A.extra[rq.extra_off + j], P_all);
    }
    if (rq.model < 0 || rq.model >= A.n_models) {
        r.type = -1;
        r.n_excl = 0;
    } else if (A.rmodels && A.rmodels[rq.model].n_ents <= kResolvedInline) {
        const ResolvedModel m = A.rmodels[rq.model];
        r.type = m.type;
        r.n_excl = m.n_ents + rq.n_extra;
        if (r.n_excl <= kInlineExcl) {
#pragma unroll
            for (int i = 0; i < kResolvedInline; i++)
                if (i < m.n_ents) r.excl_pos[i] = VIEW ? (m.pos[i] < 0 ? -1 : view_pos(S, m.pos[i])) : m.pos[i];  // resolved = GLOBAL rank positions
            if (LATE && (early_extra || rq.n_extra == 0)) {
                r.n_model = m.n_ents;
                r.n_late = rq.n_extra;
#pragma unroll
                for (int j = 0; j < kLateExtra; j++) r.late_pos[j] = xpos[j];  // first use of the gathered positions: by the caller
            } else if (early_extra) {
#pragma unroll
                for (int i = 0; i < kInlineExcl; i++) {  // slot n_ents + j takes extra j
                    const int j = i - m.n_ents;
                    int32_t v = r.excl_pos[i];
#pragma unroll
                    for (int q = 0; q < kEarlyExtra; q++)
                        if (j == q && q < rq.n_extra) v = xpos[q];
                    r.excl_pos[i] = v;
                }
            } else if (rq.n_extra > 0) {
#pragma unroll
                for (int i = 0; i < kInlineExcl; i++) {
                    if (i >= m.n_ents && i < r.n_excl) {
                        r.excl_pos[i] = pod_view_pos<VIEW>(S, A.extra[rq.extra_off + i - m.n_ents], P_all);
                    }
                }
            }
        }
    } else {
        const mmp_model_row m = A.models[rq.model];
        r.type = (m.type < 0 || m.type >= S.T) ? 0 : m.type;
        const int32_t n_ents = m.n_loaded + m.n_failed;
        r.n_excl = n_ents + rq.n_extra;
        if (r.n_excl <= kInlineExcl) {
#pragma unroll
            for (int i = 0; i < kInlineExcl; i++) {
                if (i < r.n_excl) {
                    const int32_t pod = i < n_ents ? A.ent_pod[m.ent_off + i] : A.extra[rq.extra_off + i - n_ents];
                    r.excl_pos[i] = pod_view_pos<VIEW>(S, pod, P_all);
                }
            }
        }
    }
    return r;
}

// Follow one model's entry list through pos_of once (see ResolvedModel), and classify its exclusions against the type's recorded
// shortlists (`mtw`: see kMtw*).
__device__ __forceinline__ ResolvedModel resolve_model_row(const Snap &S, const mmp_model_row &m, const int32_t *__restrict__ ent_pod, uint32_t &mtw)
{
    ResolvedModel r;
    r.type = (m.type < 0 || m.type >= S.T) ? 0 : m.type;
    r.n_ents = m.n_loaded + m.n_failed;
#pragma unroll
    for (int k = 0; k < kResolvedInline; k++) {
        int32_t pos = -1;
        if (k < r.n_ents) {
            const int32_t pod = ent_pod[m.ent_off + k];
            if (pod >= 0 && pod < S.P) pos = S.pos_of[pod];
        }
        r.pos[k] = pos;
    }
    mtw = kMtwNone;
    if (S.memo && r.type < kWinLds && r.n_ents <= kMtwMaxEnts) {
        const TypeMemo &M = S.memo[r.type];
        const MemoVar v0 = M.v[0], v1 = M.v[1];
        const int wl = M.w0 * 64;
        const int16_t *RK = S.memo_rk + (size_t)r.type * kMemoCand;
        uint32_t w = (uint32_t)r.type, o1 = 0, o2 = 0;
        for (int k = 0; k < r.n_ents; k++) {
            const int32_t pod = ent_pod[m.ent_off + k];
            if (pod < 0 || pod >= S.P) continue;
            const int32_t pos = S.pos_of[pod];
            if (v0.valid && pos >= v0.lo && pos < v0.hi) {
                const int rk = RK[pos - wl];
                if (rk == 0 || rk == kRkFirst)
                    w |= kMtwTail;  // the best / first eligible instance is excluded: another walk
                else if (rk > 0) {
                    const uint32_t off = (uint32_t)(pos - wl) + 1u;
                    if (off == o1 || off == o2) continue;
                    if (!o1)
                        o1 = off;
                    else if (!o2)
                        o2 = off;
                    else
                        w |= kMtwTail;  // a third excluded candidate
                }
            }
            // list 1 (the caller is full: the list ends at the first candidate behind the best instance, MM.java:4909-4922) depends on
            // the best / first eligible instance and on the instance that ends it
            if (v1.valid && pos >= v1.lo && pos < v1.hi && (pos == M.best0 || pos == M.bestpos || pos == v1.hi - 1)) w |= kMtwTail1;
        }
        mtw = w | (o1 << kMtwOffShift) | (o2 << (kMtwOffShift + 9));
    }
    return r;
}

// One lane per model.
__global__ void resolve_models_kernel(Snap S, const mmp_model_row *__restrict__ models,
                                      const int32_t *__restrict__ ent_pod, int32_t n_models,
                                      ResolvedModel *__restrict__ out, int32_t *__restrict__ mtw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_models) return;
    uint32_t w;
    const ResolvedModel r = resolve_model_row(S, models[i], ent_pod, w);
    out[i] = r;
    mtw[i] = (int32_t)w;
}

// Registry events (mmp_models_upsert): rows[i] replaces models[idx[i]]; its entries were appended to the
// entry arena and rows[i].ent_off already points there.  `resolved` may be null (no committed snapshot /
// shard context).  idx holds no duplicates (the host keeps the last row per model).
__global__ void upsert_models_kernel(Snap S, const int32_t *__restrict__ idx, const mmp_model_row *__restrict__ rows, int32_t n,
                                     const int32_t *__restrict__ ent_pod, mmp_model_row *__restrict__ models,
                                     ResolvedModel *__restrict__ resolved, int32_t *__restrict__ mtw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mmp_model_row m = rows[i];
    models[idx[i]] = m;
    if (resolved) {
        uint32_t w;
        const ResolvedModel r = resolve_model_row(S, m, ent_pod, w);
        resolved[idx[i]] = r;
        mtw[idx[i]] = (int32_t)w;
    }
}

// Arena compaction: offs = exclusive scan of the per-model entry counts; entries move to the fresh arrays.
__global__ void model_counts_kernel(const mmp_model_row *__restrict__ models, int32_t n, int32_t *__restrict__ cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = models[i].n_loaded + models[i].n_failed;
    if (i == n) cnt[n] = 0;
}

__global__ void move_entries_kernel(mmp_model_row *__restrict__ models, int32_t n, const int32_t *__restrict__ offs,
                                    const int32_t *__restrict__ old_pod, const int64_t *__restrict__ old_time,
                                    int32_t *__restrict__ new_pod, int64_t *__restrict__ new_time)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mmp_model_row m = models[i];
    const int32_t o = offs[i], k = m.n_loaded + m.n_failed;
    for (int32_t e = 0; e < k; e++) {
        new_pod[o + e] = old_pod[m.ent_off + e];
        new_time[o + e] = old_time[m.ent_off + e];
    }
    models[i].ent_off = o;
}

// ---- one LANE per decision -----------------------------------------------------------------------
// SQ counters of the wave-per-decision kernel (profiles/r1): ~815 instructions per decision, almost
// all of them wave-uniform control flow — the kernel is bound by instruction issue, not by memory.
// A shortlist is a handful of 64-pod words, and every step of getNext on it is plain 64-bit word
// arithmetic, so a single lane can carry a whole decision: 64 decisions per wavefront.  The one step
// that needed a per-pod column, the count break (MM.java:4925-4926), reads a threshold bitmap built at
// commit (Snap::ge) instead.  Decisions that leave the common shape — more than kInlineExcl exclusions,
// the replaced-replica-set retry, a preferred pod behind a full one, preference with a full best
// (case (b), per-candidate rpm), a count threshold beyond the bitmap rows — are handed, with their
// resolved request, to the wave-per-decision kernel, which remains the general implementation.
struct LaneWords {
    const uint64_t *E;  // eligibility row of the type
    int32_t ex[kInlineExcl];
    uint64_t touched;   // bit (w & 63) set if some exclusion falls into a word congruent to w: the
                        // shortlist words are rarely among them, and then the compares are skipped
    __device__ __forceinline__ void set_touched()
    {
        touched = 0;
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++)
            if (ex[i] >= 0) touched |= 1ull << ((ex[i] >> 6) & 63);
    }
    __device__ __forceinline__ uint64_t at(int w) const
    {
        uint64_t v = E[w];
        if ((touched >> (w & 63)) & 1ull) {
#pragma unroll
            for (int i = 0; i < kInlineExcl; i++)
                if ((ex[i] >> 6) == w) v &= ~(1ull << (ex[i] & 63));  // ex[i] == -1 never matches
        }
        return v;
    }
};

// A lane walks 64-pod words one after the other; a decision whose scans or shortlist span more than this many
// words is handed to the wave path, which takes 64 words per step (measured on a cluster where EVERY instance
// is full — the LRU-window mode of MM.java:4911-4917, where the caller's own lruTime can keep the loop from
// ever breaking and the shortlist is the whole table: 69 us per 100k decisions with all of it on one lane).
constexpr int kLaneSpan = 8;
constexpr int kFarWords = 24;  // non-empty words a two-condition scan visits through the prefix table before giving up

// first set bit of f(w) at a position in [start, stop); kNoPos if none.  `far` is set when the scan was
// given up after kLaneSpan words with words still to go (the answer is then unknown).
// `nz_all` (may be null) + `row`: the next-non-empty-word row (Snap::nz) of the bitmap that f() is a subset of (f =
// that bitmap minus the request's exclusions / the self bit).  With it the scan visits only the bitmap's non-empty
// words (a type that only a handful of instances may host has its first eligible pod anywhere in the order):
// kLaneSpan of them plus `tries` — an exclusion can empty a visited word at most kInlineExcl + 1 times; a scan for
// bits that ALSO pass a second test (count threshold, fullness) is given kFarWords.
template <class F>
__device__ __forceinline__ int lane_first(F f, int start, int stop, bool &far, const int32_t *nz_all = nullptr,
                                          size_t row = 0, int tries = kInlineExcl + 2)
{
    if (start >= stop) return kNoPos;
    const int w0 = start >> 6, wl = (stop - 1) >> 6;
    if (nz_all) {
        // the word itself and the index of the next non-empty one behind it are fetched together: one load
        // latency per visited word, and the empty stretches cost nothing
        const int32_t *nz = nz_all + row;
        int w = w0;
        for (int budget = kLaneSpan + tries; budget > 0 && w <= wl; budget--) {
            uint64_t v = f(w);
            const int nxt = nz[w + 1];
            if (w == w0) v &= (~0ull) << (start & 63);
            if (w == wl && (stop & 63)) v &= (1ull << (stop & 63)) - 1ull;
            if (v) return w * 64 + (__ffsll((unsigned long long)v) - 1);
            w = nxt;
        }
        if (w > wl) return kNoPos;
        far = true;
        return kNoPos;
    }
    const int wstop = wl - w0 >= kLaneSpan ? w0 + kLaneSpan - 1 : wl;
    for (int w = w0; w <= wstop; w++) {
        uint64_t v = f(w);
        if (w == w0) v &= (~0ull) << (start & 63);
        if (w == wl && (stop & 63)) v &= (1ull << (stop & 63)) - 1ull;
        if (v) return w * 64 + (__ffsll((unsigned long long)v) - 1);
    }
    if (wstop == wl) return kNoPos;
    far = true;
    return kNoPos;
}

// Outcome of lane_decide.  kLaneWave: the decision left the common shape and needs the general path.
// The last two only on a shard view (VIEW = true): the view holds no eligible pod for this decision /
// the view cannot decide it alone (a scan ran off the end of the view, or kLaneWave).
// kLaneLong: the shortlist spans more than kLaneSpan words — decided by the LONG instantiation of this same
// function (a later phase of the kernel), which counts / hashes / selects through the prefix tables of Snap
// instead of walking the words.
enum { kLaneDone = 0, kLaneWave = 1, kLaneNoneHere = 2, kLaneIncomplete = 3, kLaneLong = 4, kLaneHeadMiss = 5, kLaneCaseB = 6 };
// kLaneCaseB: case (b) on a snapshot that has slots for it (BSlot): the LONG phase decides it from their tables
// (a scan given up after kLaneSpan words also reports kLaneLong when the snapshot has prefix tables: the LONG
// instantiation's scans jump through them)

// excl_pos := the model's exclusions followed by the late-bound ones (what lane_decide_r expects)
__device__ __forceinline__ void merge_late_extras(ResolvedReq &r)
{
    if (r.n_late <= 0) {
        r.n_late = -1;
        return;
    }
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int j = i - r.n_model;
        int32_t v = r.excl_pos[i];
#pragma unroll
        for (int q = 0; q < kLateExtra; q++)
            if (j == q && q < r.n_late) v = r.late_pos[q];
        r.excl_pos[i] = v;
    }
    r.n_late = -1;
}

// The request's <= kInlineExcl excluded rank positions in ascending order (none = INT32_MAX, behind every position): a 19-comparator
// network.  The long path's corrections (which exclusions are candidates, the words whose hash term changes, how many candidates
// lie before the index-th survivor) are then ONE pass each over neighbours instead of all-pairs loops.
__device__ __forceinline__ void sort_excl(const int32_t (&ex)[kInlineExcl], int32_t (&sx)[kInlineExcl])
{
    static_assert(kInlineExcl == 8, "the network sorts 8 values");
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) sx[i] = ex[i] < 0 ? INT32_MAX : ex[i];
#define MMP_CE(a, b) do { const int32_t lo_ = sx[a] < sx[b] ? sx[a] : sx[b], hi_ = sx[a] < sx[b] ? sx[b] : sx[a]; sx[a] = lo_; sx[b] = hi_; } while (0)
    MMP_CE(0, 1); MMP_CE(2, 3); MMP_CE(4, 5); MMP_CE(6, 7);
    MMP_CE(0, 2); MMP_CE(1, 3); MMP_CE(4, 6); MMP_CE(5, 7);
    MMP_CE(1, 2); MMP_CE(5, 6); MMP_CE(0, 4); MMP_CE(3, 7);
    MMP_CE(1, 5); MMP_CE(2, 6);
    MMP_CE(1, 4); MMP_CE(3, 6);
    MMP_CE(2, 4); MMP_CE(3, 5);
    MMP_CE(3, 4);
#undef MMP_CE
}
__device__ __forceinline__ uint64_t bits_below(int pos) { return (1ull << (pos & 63)) - 1ull; }

// The first word w in [lo, hi) with pc[w + 1] > T, else hi (pc non-decreasing: running counts).  (A 16-way search — 15
// independent probes per round, two rounds instead of eight dependent loads — was measured on the 100k launch, round 4: 15.2 us
// against 14.3: the table rows are L1-resident and the extra instructions cost more than the round trips.)
__device__ __forceinline__ int first_word_over(const int32_t *pc, int lo, int hi, int T)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pc[mid + 1] > T)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// commit(): per slot the whole-window tables — window = (p0, end), i.e. every full present instance behind p0 (what the
// window of MM.java:4862-4866 is on a cluster whose caches are about equally old; a decision checks that ITS window does
// reach `end`, lane_case_b) — the smallest candidate rpm, the rule's limits, the five survivor bitmaps and their running
// counts.  One workgroup of 256 threads per slot.
__global__ __launch_bounds__(256) void build_bsurv_kernel(Snap S, const BSlot *__restrict__ slots, const int32_t *__restrict__ n_slots,
                                                          const int32_t *__restrict__ pm, int32_t pm_stride, BLaunch *__restrict__ launch,
                                                          uint64_t *__restrict__ surv, int32_t *__restrict__ pcs)
{
    const int s = blockIdx.x;
    const int n = *n_slots < kBSlots ? *n_slots : kBSlots;
    if (s >= n) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const int W = S.W, nwaves = (int)(blockDim.x >> 6);
    const BSlot b = slots[s];
    __shared__ BLaunch Ls;
    if (threadIdx.x == 0) {
        const int lim = b.end;
        const int32_t mn = lim - 1 > b.best0 ? pm[(size_t)b.slot * pm_stride + (lim - 1)] : INT32_MAX;
        BLaunch L;
        L.lim = lim;
        L.wlo = (b.best0 + 1) >> 6;
        L.whi = lim - 1 > b.best0 ? (lim - 1) >> 6 : L.wlo;
        L.min_load = 0;
        L.min_rpm = mn;
        L.limit[0] = INT32_MAX;
        if (mn != INT32_MAX) {
            RpmRule rule;
            rule.init(0, mn);
            L.min_load = rule.min_load;
            const int32_t t1 = rule.m4, t2 = rule.m3 < t1 ? rule.m3 : t1, t3 = rule.m15 < t2 ? rule.m15 : t2,
                          t4 = rule.m11 < t3 ? rule.m11 : t3;  // RpmRule::limit(): the applicable clauses are a suffix
            L.limit[1] = t1;
            L.limit[2] = t2;
            L.limit[3] = t3;
            L.limit[4] = t4;
        } else {
            L.limit[1] = L.limit[2] = L.limit[3] = L.limit[4] = INT32_MAX;
        }
        Ls = L;
        launch[s] = L;
    }
    __syncthreads();
    const int lim = Ls.lim;
    int32_t limc[kBClasses];
#pragma unroll
    for (int c = 0; c < kBClasses; c++) limc[c] = Ls.limit[c];
    const uint64_t *E = S.elig + (size_t)b.type * S.W, *Pm = S.pref + (size_t)b.type * S.W;
    uint64_t *sv = surv + (size_t)s * kBClasses * W;
    for (int w = wave; w < W; w += nwaves) {
        const int p = w * 64 + lane;
        const bool d = p > b.best0 && p < lim && (((E[w] & Pm[w]) >> lane) & 1ull);
        const int32_t rp = S.rpm[p];  // (the column is padded to whole words)
#pragma unroll
        for (int c = 0; c < kBClasses; c++) {
            const uint64_t keep = __ballot(d && !(rp >= 100 && rp > limc[c]));  // :4963 (class 0: no clause applies)
            if (lane == 0) sv[(size_t)c * W + w] = keep;
        }
    }
    __threadfence_block();
    __syncthreads();
    int32_t *pc = pcs + (size_t)s * kBClasses * (W + 1);
    for (int c = wave; c < kBClasses; c += nwaves) {
        int32_t carry = 0;
        if (lane == 0) pc[(size_t)c * (W + 1)] = 0;
        for (int base = 0; base < W; base += 64) {
            const int w = base + lane;
            const int32_t nb = w < W ? __popcll((unsigned long long)sv[(size_t)c * W + w]) : 0;
            const int32_t incl = wave_incl_scan_i32(nb);
            if (w < W) pc[(size_t)c * (W + 1) + w + 1] = carry + incl;
            carry += readlane_i32(incl, 63);
        }
    }
}

// Case (b) for one decision (one lane) from the snapshot's whole-window tables.  ex[] = the request's excluded rank positions (-1: none).
// kLaneDone, or kLaneWave when the request does not fit the slot (see BSlot).
__device__ __forceinline__ int lane_case_b(const Snap &S, const ResolvedReq &r, const int32_t (&ex)[kInlineExcl], int type, int best0,
                                           int64_t now, const BLds &B, mmp_place_out &o)
{
    int s = -1;
    for (int i = 0; i < B.n_slots; i++)
        if (B.slots[i].type == type) s = i;
    if (s < 0) return kLaneWave;
    const BSlot b = B.slots[s];
    const BLaunch &L = B.launch[s];
    const int selfpos = r.selfpos;
    if (best0 < b.best0 || best0 >= L.lim || selfpos == best0 || L.min_load == 0 || L.min_load > 500000000) return kLaneWave;
    const int W = B.W, lim = L.lim, wlo = L.wlo, whi = L.whi;
    {
        // the tables hold the window that reaches the last full present instance: this request's own window, from its own
        // first instance's lruTime on this launch's clock, must reach it too (:4862-4866; the instances stand in lruTime order)
        const int64_t oldest = best0 == b.best0 ? b.oldest : S.lru[best0];
        if (oldest == 0) return kLaneWave;
        const int64_t rel = age_of(oldest, now) / 4;
        const int64_t win = rel > 120000LL ? rel : 120000LL;
        if (lim - 1 > best0 && jsub64(S.lru[lim - 1], oldest) > win) return kLaneWave;
    }
    const uint64_t *sv = B.surv + (size_t)s * kBClasses * W;
    const int32_t *pc = B.pcs + (size_t)s * kBClasses * (W + 1);
    const bool favour = (r.flags & MMP_REQ_FAVOUR_SELF) != 0;
    // class 0 = every candidate of the type's window (p0, lim); those before this request's own first instance are among its
    // exclusions (they are eligible and lie before it) and fall out below like any excluded candidate
    auto cand_bit = [&](int p) { return p > b.best0 && p < lim && ((sv[p >> 6] >> (p & 63)) & 1ull); };
    // the request's exclusions that are candidates: each takes one off the counts behind it and changes its word's hash term.
    // In ascending position order (sort_excl), with everything they read fetched together: the candidate word and the rpm of each
    int32_t sx[kInlineExcl];
    sort_excl(ex, sx);
    uint64_t cw[kInlineExcl];   // the class-0 (all candidates) word of the slot's position, 0 if it lies outside the window
    int32_t erpm[kInlineExcl];
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int e = sx[i];
        const bool in = e > b.best0 && e < lim;  // (INT32_MAX, the padding, is not)
        cw[i] = in ? sv[e >> 6] : 0ull;
        erpm[i] = in ? S.rpm[e] : INT32_MAX;
    }
    uint32_t xmask = 0;  // slots of sx[] that are distinct removed candidates
    bool holds_min = false, self_removed = false;
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const bool dup = i > 0 && sx[i - 1] == sx[i];
        const uint32_t bit = dup ? 0u : (uint32_t)((cw[i] >> (sx[i] & 63)) & 1ull);
        xmask |= bit << i;
        holds_min |= bit && erpm[i] <= L.min_rpm;  // it holds the minimum the limits were derived from (:4957)
        self_removed |= bit && sx[i] == selfpos;
    }
    if (holds_min) return kLaneWave;
    const int n_removed = __popc(xmask);
    const bool self_in_c = selfpos >= 0 && cand_bit(selfpos);  // (an excluded caller is never `us`: a removed candidate above)
    o.best = best0 == b.best0 ? b.best_orig : S.orig[best0];
    o.n_candidates = 0;
    o.hash = 0;
    o.chosen = MMP_NONE;
    if (self_in_c && !self_removed && favour) return kLaneDone;  // :4871-4873 return null: the caller itself is to load it
    const int32_t *pc0 = pc;
    const int ccount = pc0[whi + 1] - pc0[wlo] - n_removed;
    if (ccount <= 0) return kLaneWave;  // no preferred instance left in the window: the replay list (:4879-4884)
    auto term = [&](uint64_t v, int w) { return audit_term(v, (uint64_t)w); };
    // audit hash of the candidates: the words strictly inside the window are whole words of the type's candidate bitmap
    // (prefix table ph, variant 1 = eligible & preferred), the two end words are the clipped ones of class 0
    const uint64_t *PH = S.ph + ((size_t)S.T + type) * (size_t)(S.W + 1);
    uint64_t hsum = term(sv[wlo], wlo);
    if (whi > wlo) hsum += term(sv[whi], whi) + (PH[whi] - PH[wlo + 1]);
    // the rpm rule (:4951-4980): which clauses apply is the request's (lastUsedTime); the limits are the window's
    const int64_t ago = age_of(r.last_used, now);
    int c = 0;
    if (ccount >= 2 && ago < 5LL * 24 * 3600 * 1000 && ago < 24LL * 3600 * 1000) {
        c = 1;
        if (ago < 12LL * 60 * 1000) {
            c = 2;
            if (ago < 5000LL) {
                c = 3;
                if (ago < -1000LL) c = 4;
            }
        }
    }
    const uint64_t *svc = sv + (size_t)c * W;
    const int32_t *pcc = pc + (size_t)c * (W + 1);
    // per removed candidate: its word among the class-c survivors and the running count in front of that word (fetched together)
    uint64_t sw_[kInlineExcl];
    int32_t pw_[kInlineExcl];
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const bool on = (xmask >> i) & 1u;
        sw_[i] = on ? svc[sx[i] >> 6] : 0ull;
        pw_[i] = on ? pcc[sx[i] >> 6] : 0;
    }
    int removed_surv = 0;
    const int p0c = pcc[wlo];
    // every removed candidate takes its own term out of the audit hash (linear in the candidate bits, wave.hpp) and, if the rule
    // of this class left it in, one off the survivors
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int e = sx[i];
        const uint64_t bit = (uint64_t)((xmask >> i) & 1u);
        if (bit) hsum -= S.amul[e >> 6] << (e & 63);
        removed_surv += (int)(bit & (sw_[i] >> (e & 63)));
    }
    const int remaining = pcc[whi + 1] - p0c - removed_surv;
    if (remaining <= 0) return kLaneWave;  // (cannot happen: the instance with the smallest rpm is never filtered)
    const int index = remaining <= 1 ? 0 : (int)(((uint64_t)r.pick * (uint64_t)(uint32_t)remaining) >> 32);
    int t = index;
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int e = sx[i];
        if (((xmask >> i) & 1u) && ((sw_[i] >> (e & 63)) & 1ull)) {
            const int rho = (pw_[i] - p0c) + __popcll((unsigned long long)(sw_[i] & bits_below(e)));
            if (rho <= t) t++;
        }
    }
    const int T = t + p0c;  // the word that holds it: the first w in [wlo, whi] with pcc[w + 1] > T
    const int lo = first_word_over(pcc, wlo, whi, T);
    const int cpos = lo * 64 + select_kth_bit(svc[lo], T - pcc[lo]);
    o.n_candidates = ccount;
    o.hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
    o.chosen = S.orig[cpos];
    if (!favour && cpos == selfpos) o.chosen = MMP_SELF;  // :4989-4991
    return kLaneDone;
}

template <bool VIEW, bool LONG = false>
__device__ __forceinline__ int lane_decide_r(const Snap &S, const PlaceArgs &A, const ResolvedReq &r, mmp_place_out &o, const BLds Bt = BLds{},
                                             LongCap *lc = nullptr);

template <bool VIEW, bool LONG = false, int FORM = kReq64>
__device__ __forceinline__ int lane_decide(const Snap &S, const PlaceArgs &A, int d, mmp_place_out &o, const BLds Bt = BLds{},
                                           const mmp_place_caller &C = mmp_place_caller{})
{
    const ResolvedReq r = resolve_one<VIEW, false, FORM>(S, A, d, C);
    return lane_decide_r<VIEW, LONG>(S, A, r, o, Bt);
}

// Bt: the staged case (b) tables of the long kernel (null: case (b) is the wave path's).  The first lane phase (LONG = false)
// answers kLaneLong for a case (b) decision when they exist, the LONG phase decides it from them.
// lc (LONG only; commit's build_long_memo_kernel): the walk for a request without exclusions or a caller's entry, the fresh-row bit
// forced to lc->nsb — returns once the list is known and leaves it there (see LongMemo).
template <bool VIEW, bool LONG>
__device__ __forceinline__ int lane_decide_r(const Snap &S, const PlaceArgs &A, const ResolvedReq &r, mmp_place_out &o, const BLds Bt, LongCap *lc)
{
    PHASE_T0();
    o.chosen = MMP_NONE;
    o.best = -1;
    o.n_candidates = 0;
    o.hash = 0;
    bool fb = false, far = false;
    const int P = S.P, W = S.W;
    do {
        if (r.type < 0) break;  // unknown model -> null
        if (r.n_excl > kInlineExcl || A.force_wave) {
            fb = true;
            break;
        }
        const int type = r.type;
        LaneWords L;
        L.E = S.elig + (size_t)type * W;
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++) L.ex[i] = r.excl_pos[i];
        L.set_touched();
        auto ew = [&](int w) { return L.at(w); };
        const int selfpos = r.selfpos;
        const bool favour = (r.flags & MMP_REQ_FAVOUR_SELF) != 0;

        // this type's rows in the [2][T][W+1] tables of its two candidate bitmaps (null on shard views)
        const size_t pc_e = (size_t)type * (size_t)(W + 1), pc_ep = ((size_t)S.T + type) * (size_t)(W + 1);
        const int best0 = lane_first(ew, 0, P, far, LONG ? S.nz : nullptr, pc_e);
        if (far) {
            fb = true;
            break;
        }
        if (best0 == kNoPos) {
            if (VIEW) return kLaneNoneHere;
            if (S.any_rs) fb = true;  // retry ignoring excludeReplicaSets, MM.java:4797-4804
            break;
        }
        PHASE(1);  // first eligible pod
        const int64_t f_lru = r.fresh_lru, f_rem = r.f_rem;
        const int32_t f_rpm = r.fresh_rpm;
        const int64_t e_lru = S.lru[best0], e_rem = S.rem[best0];
        const int32_t e_cnt = S.cnt[best0], e_rpm = S.rpm[best0];
        bool us = best0 == selfpos;
        int64_t b_lru = us ? f_lru : e_lru, b_rem = us ? f_rem : e_rem;
        int32_t b_cnt = us ? r.fresh_count : e_cnt, b_rpm = us ? f_rpm : e_rpm;
        const bool best_is_full = b_rem < S.min_space;  // :4811, never recomputed
        const bool has_pm = S.has_pref[type] != 0;
        const uint64_t *Pm = S.pref + (size_t)type * W;
        int bestpos = best0;
        if (has_pm && !((Pm[best0 >> 6] >> (best0 & 63)) & 1ull)) {
            if (best_is_full) {  // case (b)
                if (!VIEW && Bt.n_slots > 0) {
                    if (!LONG) return kLaneCaseB;  // the LONG phase decides it from the staged tables
                    const int code = lane_case_b(S, r, L.ex, type, best0, A.now, Bt, o);
                    if (code == kLaneDone) return kLaneDone;
                }
                fb = true;
                break;
            }
            // case (a): the first preferred pod, provided no full pod comes before it
            auto ewp = [&](int w) { return L.at(w) & Pm[w]; };
            const int q1 = lane_first(ewp, best0 + 1, P, far, LONG ? S.nz : nullptr, pc_ep);
            if (q1 == kNoPos) {  // (or given up: `far`)  // none preferred: the replay list ends at the first full pod
                fb = true;
                break;
            }
            auto ewf = [&](int w) { return L.at(w) & S.fullw[w]; };
            if (lane_first(ewf, best0 + 1, q1, far, LONG ? S.nz : nullptr, pc_e, kFarWords) != kNoPos || far) {
                fb = true;
                break;
            }
            bestpos = q1;
            b_lru = S.lru[q1];
            b_rem = S.rem[q1];
            b_cnt = S.cnt[q1];
            b_rpm = S.rpm[q1];
            us = q1 == selfpos;
        }
        auto dw = [&](int w) {  // eligible ∧ (preference treated as required, :4905)
            uint64_t v = L.at(w);
            if (has_pm) v &= Pm[w];
            return v;
        };
        PHASE(2);  // best row + preference step
        const int32_t best_idx = S.orig[bestpos];
        if (us && favour) {  // :4891-4895
            o.chosen = MMP_SELF;
            o.best = best_idx;
            break;
        }
        const int64_t oldest = b_lru;
        bool ns_break, self_break;
        if (best_is_full) {
            const int64_t rel = age_of(oldest, A.now) / 10;
            const int64_t d1 = jsub64(f_lru, oldest), d2 = jsub64(e_lru, oldest);
            ns_break = d1 > 45000LL && d1 > rel;  // :4913-4917
            self_break = d2 > 45000LL && d2 > rel;
        } else {
            const int64_t q = b_rem >> 2;  // :4922
            ns_break = f_rem < S.min_space || f_rem < q;
            self_break = e_rem < S.min_space || e_rem < q;
        }
        if (LONG && lc) ns_break = lc->nsb != 0;
        const int start = bestpos + 1;
        const bool self_in_d = selfpos >= start && selfpos < P && ((dw(selfpos >> 6) >> (selfpos & 63)) & 1ull);
        int end = P;
        if (ns_break) {
            auto dns = [&](int w) {
                uint64_t v = dw(w);
                if ((selfpos >> 6) == w) v &= ~(1ull << (selfpos & 63));  // the self pod breaks on its own rule
                return v;
            };
            const int p1 = lane_first(dns, start, P, far, LONG ? S.nz : nullptr, has_pm ? pc_ep : pc_e);
            end = p1 < end ? p1 : end;
        }
        if (self_in_d && self_break) end = selfpos < end ? selfpos : end;
        if (!best_is_full) {
            const int32_t thr = (int32_t)((uint32_t)b_cnt + (uint32_t)(b_cnt >> 2));  // :4926
            const int64_t T = thr < kGeBase - 1 ? (int64_t)kGeBase : (int64_t)thr + 1;  // count >= 10 && count > thr
            if (T >= kGeBase + kGeRows) {
                fb = true;
                break;
            }
            const uint64_t *G = S.ge + (size_t)(T - kGeBase) * W;
            auto dc = [&](int w) { return dw(w) & G[w]; };
            // Inside the non-decreasing head of the count column (Snap::ctpos) the threshold holds from one position on: the
            // break is the first CANDIDATE at or behind it — no walk over the words in front of it (a shortlist of thousands of
            // instances with counts below the threshold made that walk give up: the wave path, 40 us for a single decision)
            int pc = kNoPos, from2 = start;
            if (!VIEW && S.ctpos) {
                const int mono_end = S.ctpos[kGeRows], first_ge = S.ctpos[T - kGeBase];
                if (start < mono_end) {
                    const int lo = first_ge > start ? first_ge : start, hi = mono_end < end ? mono_end : end;
                    pc = lane_first(dw, lo, hi, far, LONG ? S.nz : nullptr, has_pm ? pc_ep : pc_e);
                    from2 = mono_end;
                }
            }
            if (pc == kNoPos && !far && from2 < end)
                pc = lane_first(dc, from2, end, far, LONG ? S.nz : nullptr, has_pm ? pc_ep : pc_e, kFarWords);
            end = pc < end ? pc : end;
        }
        PHASE(3);  // break scans
        if (far) {  // a scan was given up: the wave path takes 64 words per step
            fb = true;
            break;
        }
        if (VIEW && S.more_after && end >= P) return kLaneIncomplete;  // the shortlist runs into the next shard
        const bool is_long = ((end > start ? end - 1 : bestpos) >> 6) - (bestpos >> 6) >= kLaneSpan;
        if (is_long && (!LONG || !S.pc || !S.sel)) {
            if (VIEW || !S.pc || !S.sel) {  // (no prefix / inverse tables: shard views; snapshots whose tables would be too large, kSelMaxBytes)
                fb = true;
                break;
            }
            return kLaneLong;
        }
        const bool self_in_c = self_in_d && selfpos < end;
        if (self_in_c && favour) {  // :4931-4933
            o.chosen = MMP_SELF;
            o.best = best_idx;
            break;
        }
        // candidates = {best} ∪ D∩[start,end): count + hash
        const int wlo = bestpos >> 6, whi = end > start ? (end - 1) >> 6 : wlo;
        auto cand = [&](int w) {
            uint64_t v = clip_word(dw(w), w, start, end);
            if (w == wlo) v |= 1ull << (bestpos & 63);
            return v;
        };
        int ccount = 0;
        uint64_t hsum = 0;
        auto term = [&](uint64_t v, int w) { return audit_term(v, (uint64_t)((VIEW ? S.w_base : 0) + w)); };
        // LONG: the shortlist is the best instance and the RAW candidate bits (eligible, and preferred if the type prefers) of
        // [start, end) minus the request's exclusions that are among them.  Counts and audit-hash sums of the raw bits come from the
        // prefix tables (whole words strictly between wlo and whi) and the two clipped end words; every effective exclusion then
        // takes one off the count and its own term off the hash (linear, wave.hpp) — no word is rebuilt with exclusions cleared.
        const int32_t *PC = nullptr;
        const uint64_t *PH = nullptr;
        const int32_t *SEL = nullptr, *RK = nullptr;  // the row's inverse tables (Snap::sel / ::rk)
        uint32_t xmask = 0;             // slots of sx[] that are distinct candidates inside [start, end)
        int32_t sx[kInlineExcl];        // the exclusions in ascending position order (LONG only)
        int32_t rkx[kInlineExcl];       // per slot: the candidate's number in the row's numbering (pc's), -1 if it is none / outside [start, end)
        uint64_t r_lo = 0, r_hi = 0;    // the raw end words clipped to [start, end)
        int n_rlo = 0, pbase = 0;
        int g0 = 0;                     // the number (pc's numbering) of the first candidate bit at or behind `start`
        if (LONG && (is_long || lc)) {  // (lc: the record of a short list is taken by the same formulas — prefix differences hold for any range)
            const size_t row = (size_t)(has_pm ? 1 : 0) * S.T + type;
            const size_t tb = row * (size_t)(W + 1);
            PC = S.pc + tb;
            PH = S.ph + tb;
            SEL = S.sel + row * (size_t)W * 64;
            RK = S.rk + row * (size_t)W * 64;
            auto raw = [&](int w) {
                uint64_t v = L.E[w];
                if (has_pm) v &= Pm[w];
                return v;
            };
            sort_excl(L.ex, sx);
            // everything the corrections read is fetched here, together (the slots are independent of one another): one load
            // latency instead of one per exclusion — and ONE load per exclusion (rk) instead of its word, its preference word and
            // its prefix count
            uint64_t amx[kInlineExcl];  // the slots' word multipliers (the hash correction below), fetched beside their ranks
#pragma unroll
            for (int i = 0; i < kInlineExcl; i++) {
                const int e = sx[i];
                const bool in = e >= start && e < end;  // (the padding, INT32_MAX, is not)
                rkx[i] = in ? RK[e] : -1;
                amx[i] = in ? S.amul[e >> 6] : 0ull;
            }
            r_lo = raw(wlo) & ((~0ull) << (start & 63));                 // start lies in wlo or is the first bit of wlo + 1
            if ((start >> 6) != wlo) r_lo = 0;
            r_hi = raw(whi) & ((end & 63) ? bits_below(end) : ~0ull);    // end - 1 lies in whi
            n_rlo = __popcll((unsigned long long)r_lo);
            pbase = PC[wlo + 1];
            g0 = pbase - n_rlo;
            ccount = 1 + n_rlo + __popcll((unsigned long long)r_hi) + (PC[whi] - pbase);
            const uint64_t m_lo = audit_mul((uint64_t)wlo);
            hsum = (m_lo << (bestpos & 63)) + r_lo * m_lo + audit_term(r_hi, (uint64_t)whi) + (PH[whi] - PH[wlo + 1]);
#pragma unroll
            for (int i = 0; i < kInlineExcl; i++) {
                const int e = sx[i];
                const bool dup = i > 0 && sx[i - 1] == e;  // the same pod twice among the exclusions (tried and loaded, say)
                const bool bit = !dup && rkx[i] >= 0;      // rkx is -1 outside [start, end) and for a position that is no candidate
                xmask |= (uint32_t)bit << i;
                if (bit) hsum -= amx[i] << (e & 63);
            }
            ccount -= __popc(xmask);
            if (lc) {
                LongMemo &M = lc->m;
                M.b_rem = b_rem;
                M.b_lru = b_lru;
                M.best_is_full = best_is_full;
                M.b_rpm = b_rpm;
                M.best_idx = best_idx;
                M.has_pm = has_pm;
                M.best0 = best0;
                M.bestpos = bestpos;
                M.e_rpm = e_rpm;
                M.sbk = self_break;
                LongVar &V = M.v[lc->nsb ? 1 : 0];
                V.valid = 1;
                V.end = end;
                V.ccount = ccount;
                V.g0 = g0;
                V.hsum = hsum;
                lc->ok = 1;
                return kLaneDone;
            }
        } else {
            for (int w = wlo; w <= whi; w++) {
                const uint64_t v = cand(w);
                ccount += __popcll((unsigned long long)v);
                hsum += term(v, w);
            }
        }
        PHASE(4);  // count + hash
        int remaining = ccount;
        bool null0 = false, null_s = false, null_o = false;
        if (ccount >= 2) {  // rpm filter, :4951-4980 (quirks B#2/B#3: three rpm classes)
            const int n_others = ccount - 1 - (self_in_c ? 1 : 0);
            int32_t mn = b_rpm;
            if (self_in_c && e_rpm < mn) mn = e_rpm;
            if (n_others > 0 && f_rpm < mn) mn = f_rpm;
            RpmRule rule;
            rule.init(age_of(r.last_used, A.now), mn);
            null0 = rule.nulls(b_rpm);
            null_s = self_in_c && rule.nulls(e_rpm);
            null_o = n_others > 0 && rule.nulls(f_rpm);
            remaining = ccount - (null0 ? 1 : 0) - (null_s ? 1 : 0) - (null_o ? n_others : 0);
        }
        PHASE(5);  // rpm rule
        const int index = remaining <= 1 ? 0 : (int)(((uint64_t)r.pick * (uint64_t)(uint32_t)remaining) >> 32);
        int cpos = kNoPos;
        if (remaining >= 1) {
            const int bw = bestpos >> 6, sw = self_in_c ? (selfpos >> 6) : -1;
            auto surv = [&](int w) {  // the word's candidates that the rpm filter left in
                uint64_t v = cand(w);
                uint64_t special = 0;
                if (w == bw) special |= 1ull << (bestpos & 63);
                if (w == sw) special |= 1ull << (selfpos & 63);
                if (null_o) v &= special;
                if (null0 && w == bw) v &= ~(1ull << (bestpos & 63));
                if (null_s && w == sw) v &= ~(1ull << (selfpos & 63));
                return v;
            };
            if (LONG && is_long) {
                if (null_o) {  // only the best and the self entry can be left, in that order
                    int k = index;
                    if (!null0) {
                        if (k == 0) cpos = bestpos;
                        k--;
                    }
                    if (cpos == kNoPos && self_in_c && !null_s && k == 0) cpos = selfpos;
                } else {
                    // The shortlist in order: the best instance (raw rank 0), then the raw candidate bits of [start, end).  REMOVED
                    // from it: the best if the rpm rule nulls it, the caller's own entry if the rule nulls it, the effective
                    // exclusions.  t = the raw rank of the index-th survivor: every removed entry at or before it pushes it one
                    // further (one ascending pass).  The search over the prefix counts is then the plain one.
                    // raw rank of a candidate at or behind `start` = its number in the row's numbering - g0 + 1 (the best is rank 0)
                    int rho_s = 0;
                    if (null_s) rho_s = 1 + RK[selfpos] - g0;  // (self_in_c: the caller's entry is a candidate inside [start, end))
                    bool self_pending = null_s;
                    int t = index + (null0 ? 1 : 0);
#pragma unroll
                    for (int i = 0; i < kInlineExcl; i++) {
                        if (self_pending && selfpos < sx[i]) {
                            if (rho_s <= t) t++;
                            self_pending = false;
                        }
                        if ((xmask >> i) & 1u) {
                            if (1 + rkx[i] - g0 <= t) t++;
                        }
                    }
                    if (self_pending && rho_s <= t) t++;
                    // the t-th entry of the shortlist: the best itself, or candidate number g0 + t - 1 of the row — one lookup (round 4:
                    // a binary search over the prefix counts, eight dependent loads, and a select inside the word it found)
                    cpos = t == 0 ? bestpos : SEL[g0 + t - 1];
                }
            } else {
                int running = 0;
                for (int w = wlo; w <= whi; w++) {
                    const uint64_t v = surv(w);
                    const int c = __popcll((unsigned long long)v);
                    if (index < running + c) {
                        cpos = w * 64 + select_kth_bit(v, index - running);
                        break;
                    }
                    running += c;
                }
            }
        }
        o.best = best_idx;
        o.n_candidates = ccount;
        o.hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
        if (cpos != kNoPos) {
            o.chosen = S.orig[cpos];
            if (!favour && cpos == selfpos) o.chosen = MMP_SELF;  // :4989-4991
        }
        PHASE(6);  // survivor select + translation
    } while (false);
    return fb ? (VIEW ? kLaneIncomplete : (far && !LONG && S.pc ? kLaneLong : kLaneWave)) : kLaneDone;
}

// getNext on the type's head window (TypeWin): Ws = the windows in LDS; scr = this lane's column of the
// workgroup's scratch — word j of the window's eligibility words, the request's exclusions cleared, at
// scr[j * kPlaceBlock]; the candidate words (eligible and preferred-if-the-type-prefers, :4905) are that column
// and-ed with the window's preference words where they are read (a second column for them cost 12 KB of LDS per
// workgroup and an LDS atomic per exclusion).  Step for step lane_decide_r's simple case; kLaneHeadMiss whenever the window cannot
// answer — the caller then runs lane_decide_r on the same resolved request.
// Shape of the code, from measurements (tools/phase_clock.py, SQ counters): a wavefront of this kernel is alone or
// nearly alone on its SIMD, so what it pays for is every dependent LDS round trip (~0.1 us) and every instruction of
// its one serial stream.  Hence: everything the first steps need (header, words, the likeliest row) is fetched in
// one round; the break scans (:4909-4927), the candidate count and the audit hash are ONE walk over the candidate
// words (a trip is one LDS read; one or two trips); the candidate words are parked for the final select; the rpm
// rule is one threshold.  (A variant with 4 or 6 words per bitmap in registers and no loops at all was measured
// too: 1059 instead of 744 VALU instructions per wavefront, 8.6 / 9.9 us per launch instead of 7.5.)
// VIEW: S is a pod-axis shard's view of its slice (windows built over the view by the sharded commit): positions are local, the
// audit hash takes the GLOBAL word index, and a shortlist that reaches the end of a slice with more slices behind it is not the
// window's to answer (kLaneHeadMiss; lane_decide_r<true> then reports kLaneIncomplete).
// What build_memo_body takes out of lane_decide_win<…, MEMO> (see TypeMemo): the walk's result for a request WITHOUT
// exclusions or a caller's entry in reach, with the fresh-row break (MM.java:4913-4922) forced to `nsb`.
struct MemoCap {
    int nsb;
    int best0, bestpos, end, wlo, whi, ccount;
    uint64_t hsum;
    int64_t b_rem, b_lru;
    int32_t b_rpm, best_idx, e_rpm;
    bool best_is_full;
    bool sbk;  // the break rule of the caller's own entry (curInst = bestEntry's row) as the walk computed it for a caller that is not the best instance
};
template <bool VIEW = false, bool MEMO = false, int SCR = kPlaceBlock>  // SCR: lanes per scratch column
__device__ __forceinline__ int lane_decide_win(const Snap &S, const PlaceArgs &A, const ResolvedReq &r, const TypeWin *Ws,
                                               uint64_t *scr, mmp_place_out &o, MemoCap *mc = nullptr)
{
    PHASE_T0();
    if (r.type < 0 || r.type >= kWinLds || r.n_excl > kInlineExcl || A.force_wave) return kLaneHeadMiss;
    const TypeWin &Wn = Ws[r.type];
    uint64_t *scrE = scr;  // one column per lane: the window's eligibility words, the request's exclusions cleared
    // round 1: header, the words of both bitmaps, the first row
    const int4 hdr = *reinterpret_cast<const int4 *>(&Wn);  // valid, w0, nw, flags
    {
        uint64_t e6[kWinWords];
#pragma unroll
        for (int j = 0; j < kWinWords; j++) e6[j] = Wn.E[j];  // words beyond nw are zero in the record
#pragma unroll
        for (int j = 0; j < kWinWords; j++) scrE[j * SCR] = e6[j];
    }
    WinRow r0 = Wn.rowsE[0];
    if (!hdr.x) return kLaneHeadMiss;
    const int P = S.P;
    const int w0 = hdr.y, nw = hdr.z;
    const int win_lo = w0 * 64;
    const int win_end = (w0 + nw) * 64 < P ? (w0 + nw) * 64 : P;  // the window answers for positions [win_lo, win_end)
    // the request's exclusions (CacheMissExcludeSet, :4740-4743): the model's, then the late-bound ones of the request
    auto clear_at = [&](int e) {
        if (__ballot(e >= 0) == 0) return;  // a slot no decision of this wavefront uses (most of the ten): one compare
        const int j = (e >> 6) - w0;
        if (e >= 0 && j >= 0 && j < nw) {
            const unsigned long long m = ~(1ull << (e & 63));
            atomicAnd((unsigned long long *)&scrE[j * SCR], m);
        }
    };
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) clear_at(r.excl_pos[i]);
#pragma unroll
    for (int i = 0; i < kLateExtra; i++) clear_at(r.late_pos[i]);  // -1 unless late-bound
    auto ew = [&](int w) { return scrE[(w - w0) * SCR]; };  // w0 <= w < w0 + nw
    auto dw = [&](int w) { return scrE[(w - w0) * SCR] & Wn.Pm[w - w0]; };  // ... that may be chosen (preference mask)
    // first set bit of f(w) at a position in [from, to) of the window (to <= win_end); kNoPos if none
    auto first_in = [&](auto f, int from, int to) {
        if (from >= to) return (int)kNoPos;
        const int ws = from >> 6, wl = (to - 1) >> 6;
        for (int w = ws; w <= wl; w++) {
            uint64_t v = f(w);
            if (w == ws) v &= (~0ull) << (from & 63);
            if (w == wl && (to & 63)) v &= (1ull << (to & 63)) - 1ull;
            if (v) return w * 64 + (__ffsll((unsigned long long)v) - 1);
        }
        return (int)kNoPos;
    };
    // row index of a window position: the number of (unmasked) bits of `words` below it
    auto rank_in = [&](const uint64_t *words, int pos, bool and_pm) {
        const int jb = (pos >> 6) - w0;
        int k = 0;
        for (int j = 0; j <= jb; j++) {
            uint64_t v = words[j];
            if (and_pm) v &= Wn.Pm[j];
            if (j == jb) v &= (1ull << (pos & 63)) - 1ull;
            k += __popcll((unsigned long long)v);
        }
        return k;
    };
    const int best0 = first_in(ew, win_lo, win_end);
    if (best0 == kNoPos) return kLaneHeadMiss;  // null, or the retry without excludeReplicaSets (:4797-4804)
    // every eligible position before best0 is excluded; none unless an exclusion removed the type's first instance
    if (best0 != (r0.pos & kWinPosMask)) {
        const int i0 = rank_in(Wn.E, best0, false);
        if (i0 >= kWinRows) return kLaneHeadMiss;
        r0 = Wn.rowsE[i0];
    }
    PHASE(1);  // first eligible pod
    const int selfpos = r.selfpos;
    const bool favour = (r.flags & MMP_REQ_FAVOUR_SELF) != 0;
    const int64_t f_lru = r.fresh_lru, f_rem = r.f_rem;
    const int32_t f_rpm = r.fresh_rpm;
    const int64_t e_lru = r0.lru, e_rem = r0.rem;
    const int32_t e_rpm = r0.rpm;
    bool us = best0 == selfpos;
    int64_t b_lru = us ? f_lru : e_lru, b_rem = us ? f_rem : e_rem;
    int32_t b_cnt = us ? r.fresh_count : r0.cnt, b_rpm = us ? f_rpm : e_rpm;
    const bool best_is_full = b_rem < S.min_space;  // :4811, never recomputed
    int bestpos = best0;
    int32_t best_idx = r0.orig;
    if ((hdr.w & 2) && !(r0.pos & kWinPosPreferred)) {
        if (best_is_full) return kLaneHeadMiss;  // case (b): per-candidate rpm, the wave path
        // case (a): the first preferred pod, provided no full pod comes before it.  The type's precomputed answer (rowsP[0],
        // no full eligible instance before it: build_wins_kernel) holds unless this request excludes that very instance;
        // the scans run only in wavefronts where some decision needs them.
        WinRow rq = Wn.rowsP[0];
        int q1 = rq.pos & kWinPosMask;
        const bool pre = !(hdr.w & 4) && rq.pos >= 0 && q1 > best0 && q1 < win_end && ((ew(q1 >> 6) >> (q1 & 63)) & 1ull);
        if (!pre) {
            q1 = first_in(dw, best0 + 1, win_end);
            if (q1 == kNoPos) return kLaneHeadMiss;  // none in the window (or none at all: the replay list)
            auto ewf = [&](int w) { return ew(w) & Wn.F[w - w0]; };
            if (first_in(ewf, best0 + 1, q1) != kNoPos) return kLaneHeadMiss;
            const int iq = rank_in(Wn.E, q1, true);
            if (iq >= kWinRows) return kLaneHeadMiss;
            rq = Wn.rowsP[iq];
        }
        bestpos = q1;
        b_lru = rq.lru;
        b_rem = rq.rem;
        b_cnt = rq.cnt;
        b_rpm = rq.rpm;
        best_idx = rq.orig;
        us = q1 == selfpos;
    }
    PHASE(2);  // best row + preference step
    o.chosen = MMP_NONE;
    o.best = best_idx;
    o.n_candidates = 0;
    o.hash = 0;
    if (us && favour) {  // :4891-4895
        o.chosen = MMP_SELF;
        return kLaneDone;
    }
    const int64_t oldest = b_lru;
    bool ns_break, self_break;
    int from_t = kNoPos;  // first window position whose count breaks the list (:4925-4926); none in full mode
    if (best_is_full) {
        const int64_t rel = age_of(oldest, A.now) / 10;
        const int64_t d1 = jsub64(f_lru, oldest), d2 = jsub64(e_lru, oldest);
        ns_break = d1 > 45000LL && d1 > rel;  // :4913-4917
        self_break = d2 > 45000LL && d2 > rel;
    } else {
        const int64_t q = b_rem >> 2;  // :4922
        ns_break = f_rem < S.min_space || f_rem < q;
        self_break = e_rem < S.min_space || e_rem < q;
        const int32_t thr = (int32_t)((uint32_t)b_cnt + (uint32_t)(b_cnt >> 2));  // :4926
        const int64_t T = thr < kGeBase - 1 ? (int64_t)kGeBase : (int64_t)thr + 1;  // count >= 10 && count > thr
        if (T >= kGeBase + kGeRows) return kLaneHeadMiss;
        from_t = win_lo + (int)Wn.ct[T - kGeBase];  // counts do not decrease along the window
    }
    if (MEMO) {  // the type's shortlist for either outcome of the fresh-row test; no caller's entry in reach
        mc->sbk = self_break;
        mc->e_rpm = e_rpm;
        ns_break = mc->nsb != 0;
        self_break = false;
    }
    // ONE walk over the candidate words from the best position on: the first position that ends the list — the first
    // candidate that is not the caller's own entry when the fresh-row rule fires, the caller's own entry when its rule
    // fires, the first candidate whose count is past the threshold — and, up to it, the candidates' count and audit
    // hash; the clipped words are parked in the eligibility column for the select below.
    const int start = bestpos + 1;
    const int wlo = bestpos >> 6, wend = (win_end - 1) >> 6;
    const int lo_t = from_t > start ? from_t : start;
    int end = kNoPos, whi = wlo, ccount = 0;
    bool self_in_c = false;
    uint64_t hsum = 0;
    for (int w = wlo; w <= wend; w++) {
        const uint64_t d = dw(w);
        uint64_t v = d;
        if (w == wlo) v &= (start & 63) ? (~0ull) << (start & 63) : ((start >> 6) == wlo ? ~0ull : 0ull);  // positions >= start
        const bool self_here = (selfpos >> 6) == w && ((v >> (selfpos & 63)) & 1ull);
        int e_w = kNoPos;
        if (ns_break) {
            const uint64_t x = self_here ? v & ~(1ull << (selfpos & 63)) : v;
            if (x) e_w = w * 64 + (__ffsll((unsigned long long)x) - 1);
        }
        if (self_here && self_break && selfpos < e_w) e_w = selfpos;
        if ((lo_t >> 6) <= w) {  // (from_t == kNoPos in full mode: never)
            const uint64_t y = (lo_t >> 6) == w ? v & ((~0ull) << (lo_t & 63)) : v;
            if (y) {
                const int pc = w * 64 + (__ffsll((unsigned long long)y) - 1);
                e_w = pc < e_w ? pc : e_w;
            }
        }
        if (e_w != kNoPos) {
            end = e_w;
            v &= (1ull << (e_w & 63)) - 1ull;  // e_w lies in this word
        }
        if (self_here && (end == kNoPos || selfpos < end)) self_in_c = true;
        if (w == wlo) v |= 1ull << (bestpos & 63);
        scrE[(w - w0) * SCR] = v;
        whi = w;
        ccount += __popcll((unsigned long long)v);
        hsum += audit_term(v, (uint64_t)((VIEW ? S.w_base : 0) + w));
        if (end != kNoPos) break;
    }
    if (end == kNoPos && (win_end < P || (VIEW && S.more_after))) return kLaneHeadMiss;  // the list runs past the window (or the slice)
    PHASE(3);  // break scans + count + audit hash
    if (MEMO) {
        mc->best0 = best0;
        mc->bestpos = bestpos;
        mc->end = end;
        mc->wlo = wlo;
        mc->whi = whi;
        mc->ccount = ccount;
        mc->hsum = hsum;
        mc->b_rem = b_rem;
        mc->b_lru = b_lru;
        mc->b_rpm = b_rpm;
        mc->best_idx = best_idx;
        mc->best_is_full = best_is_full;
        return kLaneDone;
    }
    if (self_in_c && favour) {  // :4931-4933
        o.chosen = MMP_SELF;
        return kLaneDone;
    }
    PHASE(4);
    int remaining = ccount;
    bool null0 = false, null_s = false, null_o = false;
    if (ccount >= 2) {  // rpm filter, :4951-4980 (quirks B#2/B#3: three rpm classes)
        const int n_others = ccount - 1 - (self_in_c ? 1 : 0);
        int32_t mn = b_rpm;
        if (self_in_c && e_rpm < mn) mn = e_rpm;
        if (n_others > 0 && f_rpm < mn) mn = f_rpm;
        RpmRule rule;
        rule.init(age_of(r.last_used, A.now), mn);
        const int32_t lim = rule.limit();
        null0 = b_rpm >= 100 && b_rpm > lim;
        null_s = self_in_c && e_rpm >= 100 && e_rpm > lim;
        null_o = n_others > 0 && f_rpm >= 100 && f_rpm > lim;
        remaining = ccount - (null0 ? 1 : 0) - (null_s ? 1 : 0) - (null_o ? n_others : 0);
    }
    PHASE(5);  // rpm rule
    const int index = remaining <= 1 ? 0 : (int)(((uint64_t)r.pick * (uint64_t)(uint32_t)remaining) >> 32);
    int cpos = kNoPos;
    if (remaining >= 1) {
        const int sw = self_in_c ? (selfpos >> 6) : -1;
        int running = 0;
        for (int w = wlo; w <= whi; w++) {
            uint64_t v = scrE[(w - w0) * SCR];  // the word's candidates; those the rpm filter left in:
            uint64_t special = 0;
            if (w == wlo) special |= 1ull << (bestpos & 63);
            if (w == sw) special |= 1ull << (selfpos & 63);
            if (null_o) v &= special;
            if (null0 && w == wlo) v &= ~(1ull << (bestpos & 63));
            if (null_s && w == sw) v &= ~(1ull << (selfpos & 63));
            const int c = __popcll((unsigned long long)v);
            if (index < running + c) {
                cpos = w * 64 + select_kth_bit(v, index - running);
                break;
            }
            running += c;
        }
    }
    o.n_candidates = ccount;
    o.hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
    if (cpos != kNoPos) {
        o.chosen = cpos == bestpos ? best_idx : S.orig[cpos];
        if (!favour && cpos == selfpos) o.chosen = MMP_SELF;  // :4989-4991
    }
    PHASE(6);  // survivor select + translation
    return kLaneDone;
}

// (MMP_PLACE_ONE_NOINLINE: experiment builds, tools/r6 — the general path as a function of its own)
#ifdef MMP_PLACE_ONE_NOINLINE
#define MMP_PLACE_ONE_ATTR __attribute__((noinline))
#else
#define MMP_PLACE_ONE_ATTR __forceinline__
#endif
template <int FORM = kReq64>
__device__ MMP_PLACE_ONE_ATTR void place_one(const Snap &S, const PlaceArgs &A, int d, uint64_t *ew, uint64_t *fw,
                                          const mmp_place_caller &C = mmp_place_caller{})
{
    const int lane = lane_id();
    const int P = S.P, W = S.W;
    mmp_place_out *out = &A.outs[d];
    const mmp_place_req rq = fetch_req<FORM>(A, C, d);
    if (rq.model < 0 || rq.model >= A.n_models) {
        write_out(out, MMP_NONE, -1, 0, 0);
        return;
    }
    const mmp_model_row m = A.models[rq.model];
    int type = m.type;
    if (type < 0 || type >= S.T) type = 0;
    const int32_t *ents = A.ent_pod + m.ent_off;
    const int32_t n_ents = m.n_loaded + m.n_failed;
    const int32_t *extra = A.extra + rq.extra_off;

    const int selfpos = (rq.self_pod >= 0 && rq.self_pod < P) ? S.pos_of[rq.self_pod] : -1;
    const bool favour = (rq.flags & MMP_REQ_FAVOUR_SELF) != 0;

    // filter(...) and the retry without excludeReplicaSets, MM.java:4793-4805
    stage_eligible(S, S.elig + (size_t)type * W, ew, ents, n_ents, extra, rq.n_extra);
    int best0 = first_set_from(ew, nullptr, 0, W);
    if (best0 == kNoPos && S.any_rs) {
        stage_eligible(S, S.elig_nors + (size_t)type * W, ew, ents, n_ents, extra, rq.n_extra);
        best0 = first_set_from(ew, nullptr, 0, W);
    }
    if (best0 == kNoPos) {
        write_out(out, MMP_NONE, -1, 0, 0);
        return;
    }

    // the caller's getFreshInstanceRecord()
    const int64_t f_lru = rq.fresh_lru;
    const int64_t f_rem = remaining_of(rq.fresh_capacity, rq.fresh_used);
    const int32_t f_rpm = rq.fresh_rpm;

    // bestEntry.getValue(): the snapshot row of the first eligible pod
    const int64_t e_lru = S.lru[best0], e_rem = S.rem[best0];
    const int32_t e_cnt = S.cnt[best0], e_rpm = S.rpm[best0];

    bool us = (best0 == selfpos);                    // :4808
    int64_t b_lru = us ? f_lru : e_lru;              // bestInst, :4810
    int64_t b_rem = us ? f_rem : e_rem;
    int32_t b_cnt = us ? rq.fresh_count : e_cnt;
    int32_t b_rpm = us ? f_rpm : e_rpm;
    const bool best_is_full = b_rem < S.min_space;   // :4811, never recomputed (quirk B#14)

    const uint64_t *Pm = S.has_pref[type] ? S.pref + (size_t)type * W : nullptr;  // :4817
    int bestpos = best0;
    const uint64_t *Dm = Pm;  // "treat preference as required" mask of the simple loop (:4905)
    int limit = P;            // exclusive end of the iterator the simple loop walks
    bool mode_b = false;

    if (Pm && !test_bit(Pm, best0)) {  // !simpleCase, :4822-4823
        if (!best_is_full) {
            // case (a) :4828-4852 — first preferred pod before the first full pod
            const int q1 = first_set_from(ew, Pm, best0 + 1, W);
            const int q2 = first_set_from(ew, S.fullw, best0 + 1, W);
            if (q1 != kNoPos && q1 <= q2) {
                bestpos = q1;
                b_lru = S.lru[q1];
                b_rem = S.rem[q1];
                b_cnt = S.cnt[q1];
                b_rpm = S.rpm[q1];
                us = (q1 == selfpos);
            } else {
                Dm = nullptr;  // prefer = null, replay list = eligible pods before q2
                limit = q2 < P ? q2 : P;
            }
        } else {
            // case (b) :4853-4887
            const int q3 = first_lru_break(ew, best0 + 1, W, S.lru, b_lru, 120000LL, age_of(b_lru, A.now) / 4);
            const int lim = q3 < P ? q3 : P;
            const int qp = first_set_from(ew, Pm, best0 + 1, W);
            limit = lim;
            if (qp < lim)
                mode_b = true;
            else
                Dm = nullptr;
        }
    }

    const int64_t ago = age_of(rq.last_used, A.now);  // :4951
    int wlo, whi, ccount;
    uint64_t hsum;
    RpmRule rule;
    int remaining;

    if (mode_b) {
        // only preferred pods inside the age window are candidates; best0 is not (:4867-4877)
        const int start = best0 + 1;
        if (selfpos >= start && selfpos < limit && test_bit(ew, selfpos) && test_bit(Pm, selfpos) && favour) {
            write_out(out, MMP_NONE, S.orig[best0], 0, 0);  // :4871-4873 return null
            return;
        }
        wlo = start >> 6;
        whi = (limit - 1) >> 6;
        int cc = 0, mn = INT32_MAX;
        uint64_t h = 0;
        for (int base = wlo; base <= whi; base += 64) {
            const int w = base + lane;
            uint64_t v = 0;
            if (w <= whi) {
                v = clip_word(ew[w] & Pm[w], w, start, limit);
                fw[w] = v;
            }
            cc += __popcll((unsigned long long)v);
            h += audit_term(v, (uint64_t)w);
            for (uint64_t t = v; t; t &= t - 1) {
                const int32_t r = S.rpm[w * 64 + (__ffsll((unsigned long long)t) - 1)];
                mn = r < mn ? r : mn;
            }
        }
        ccount = wave_sum_i32(cc);
        hsum = wave_sum_u64(h);
        mn = wave_min_i32(mn);
        remaining = ccount;
        if (ccount >= 2) {
            rule.init(ago, mn);
            if (rule.active) {
                wave_sync();
                int nulled = 0;
                for (int base = wlo; base <= whi; base += 64) {
                    const int w = base + lane;
                    if (w <= whi) {
                        uint64_t v = fw[w], keep = v;
                        for (uint64_t t = v; t; t &= t - 1) {
                            const int bit = __ffsll((unsigned long long)t) - 1;
                            if (rule.nulls(S.rpm[w * 64 + bit])) {
                                keep &= ~(1ull << bit);
                                nulled++;
                            }
                        }
                        fw[w] = keep;
                    }
                }
                remaining = ccount - wave_sum_i32(nulled);
            }
        }
        wave_sync();
    } else {
        // simple case :4890-4938
        if (us && favour) {
            write_out(out, MMP_SELF, S.orig[bestpos], 0, 0);  // :4891-4895
            return;
        }
        const int64_t oldest = b_lru;
        bool ns_break, self_break;  // break predicate for non-self pods (curInst = caller's fresh
                                    // row) and for the self pod (curInst = bestEntry's row), :4909
        if (best_is_full) {
            const int64_t rel = age_of(oldest, A.now) / 10;
            const int64_t d1 = jsub64(f_lru, oldest), d2 = jsub64(e_lru, oldest);
            ns_break = d1 > 45000LL && d1 > rel;  // :4913-4917
            self_break = d2 > 45000LL && d2 > rel;
        } else {
            const int64_t q = b_rem >> 2;  // :4922 ("half" in the comment, quarter in the code)
            ns_break = f_rem < S.min_space || f_rem < q;
            self_break = e_rem < S.min_space || e_rem < q;
        }
        const int start = bestpos + 1;
        const bool self_in_d = selfpos >= start && selfpos < limit && test_bit(ew, selfpos) &&
                               (!Dm || test_bit(Dm, selfpos));
        int end = limit;
        if (ns_break) {
            int p1 = first_set_from(ew, Dm, start, W);
            if (p1 == selfpos) p1 = first_set_from(ew, Dm, selfpos + 1, W);
            end = p1 < end ? p1 : end;
        }
        if (self_in_d && self_break) end = selfpos < end ? selfpos : end;
        if (!best_is_full) {
            const int32_t thr = (int32_t)((uint32_t)b_cnt + (uint32_t)(b_cnt >> 2));  // :4926
            const int64_t Tg = thr < kGeBase - 1 ? (int64_t)kGeBase : (int64_t)thr + 1;  // count >= 10 && count > thr
            const int pc = (S.ge && Tg < kGeBase + kGeRows) ? first_ge_break(ew, Dm, S.ge + (size_t)(Tg - kGeBase) * W, start, end)
                                                           : first_count_break(ew, Dm, start, end, S.cnt, thr);
            end = pc < end ? pc : end;
        }
        const bool self_in_c = self_in_d && selfpos < end;
        if (self_in_c && favour) {
            write_out(out, MMP_SELF, S.orig[bestpos], 0, 0);  // :4931-4933
            return;
        }
        // candidates = {best} ∪ D∩[start,end)
        wlo = bestpos >> 6;
        whi = end > start ? (end - 1) >> 6 : wlo;
        // count + hash the shortlist words: the non-empty ones are normally one or two, visited with
        // scalar code; a trip with many takes the wave-wide reductions
        int cc = 0, cc_s = 0;
        uint64_t h = 0, h_s = 0;
        bool dense = false;
        for (int base = wlo; base <= whi; base += 64) {
            const int w = base + lane;
            uint64_t v = 0;
            if (w <= whi) {
                v = ew[w];
                if (Dm) v &= Dm[w];
                v = clip_word(v, w, start, end);
                if (w == wlo) v |= 1ull << (bestpos & 63);
                fw[w] = v;
            }
            uint64_t nz = __ballot(v != 0);
            if (__popcll((unsigned long long)nz) <= 8) {
                while (nz) {
                    const int l = __ffsll((unsigned long long)nz) - 1;
                    nz &= nz - 1;
                    const uint64_t vv = readlane_u64(v, l);
                    cc_s += __popcll((unsigned long long)vv);
                    h_s += audit_term(vv, (uint64_t)(base + l));
                }
            } else {
                dense = true;
                cc += __popcll((unsigned long long)v);
                h += audit_term(v, (uint64_t)w);
            }
        }
        ccount = cc_s;
        hsum = h_s;
        if (dense) {
            ccount += wave_sum_i32(cc);
            hsum += wave_sum_u64(h);
        }
        remaining = ccount;
        wave_sync();
        if (ccount >= 2) {
            // instReqLoad: bestInst.rpm for the best, bestEntry's rpm for a self entry,
            // the caller's fresh rpm for every other pod (quirks B#2, B#3)
            const int n_others = ccount - 1 - (self_in_c ? 1 : 0);
            int32_t mn = b_rpm;
            if (self_in_c && e_rpm < mn) mn = e_rpm;
            if (n_others > 0 && f_rpm < mn) mn = f_rpm;
            rule.init(ago, mn);
            const bool null0 = rule.nulls(b_rpm);
            const bool null_s = self_in_c && rule.nulls(e_rpm);
            const bool null_o = n_others > 0 && rule.nulls(f_rpm);
            if (null0 || null_s || null_o) {
                remaining = ccount - (null0 ? 1 : 0) - (null_s ? 1 : 0) - (null_o ? n_others : 0);
                const int bw = bestpos >> 6, sw = self_in_c ? (selfpos >> 6) : -1;
                for (int base = wlo; base <= whi; base += 64) {
                    const int w = base + lane;
                    if (w <= whi) {
                        uint64_t v = fw[w];
                        uint64_t special = 0;
                        if (w == bw) special |= 1ull << (bestpos & 63);
                        if (w == sw) special |= 1ull << (selfpos & 63);
                        if (null_o) v &= special;
                        if (null0 && w == bw) v &= ~(1ull << (bestpos & 63));
                        if (null_s && w == sw) v &= ~(1ull << (selfpos & 63));
                        fw[w] = v;
                    }
                }
                wave_sync();
            }
        }
    }

    const int32_t best_idx = S.orig[bestpos];
    if (ccount == 0) {  // :4941-4943
        write_out(out, MMP_NONE, best_idx, 0, 0);
        return;
    }
    const uint32_t hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
    // :4981-4986 — index-th non-null candidate
    const int index = remaining <= 1 ? 0 : (int)(((uint64_t)rq.pick * (uint64_t)(uint32_t)remaining) >> 32);
    const int cpos = remaining >= 1 ? select_in_range(fw, wlo, whi, index) : kNoPos;
    int32_t chosen = MMP_NONE;
    if (cpos != kNoPos) {
        chosen = S.orig[cpos];
        if (!favour && cpos == selfpos) chosen = MMP_SELF;  // :4989-4991
    }
    write_out(out, chosen, best_idx, ccount, hash);
}

// LDS per workgroup: kPlaceWaves × 2 bitmaps × Wpad words.
// The load-target kernel: 256 decisions per workgroup, one per lane (lane_decide); the decisions that
// leave the common shape are collected in LDS and then taken one wavefront at a time by the general
// path (place_one) inside the same launch.  LDS: kPlaceWaves × 2 bitmaps × wpad words for that path.
// WITH_LONG: the kernel carries the phase that decides whole-table shortlists on single lanes through the
// prefix tables (lane_decide<…, LONG>).  That phase needs 108 VGPRs against 94 (4 instead of 5 wavefronts per
// SIMD: -5 % on a lone 100k launch, -14 % saturated), so it is compiled into a kernel of its own which the host
// launches only for snapshots in which (nearly) every instance is full — the only regime that produces such
// shortlists in number; otherwise they take the wave path.
// LDS of place_block: static = the lists; dynamic = max(windows + per-lane scratch of the lane phase, the wave path's
// tiles) — the host checks the sum against the device's per-workgroup limit
// windows are staged in whole 1 KB chunks, as many as the snapshot has type rows (at most kWinLds)
__host__ __device__ constexpr int win_lds_bytes(int type_rows)
{
    return (((type_rows < kWinLds ? type_rows : kWinLds) * (int)sizeof(TypeWin) + 1023) / 1024) * 1024;
}
constexpr int kWinLdsBytes = win_lds_bytes(kWinLds);
constexpr int kLaneScratchBytes = kWinWords * kPlaceBlock * 8;  // one column of kWinWords words per lane
__host__ __device__ constexpr int place_lane_lds(int type_rows) { return win_lds_bytes(type_rows) + kLaneScratchBytes; }
constexpr int kPlaceLaneLds = kWinLdsBytes + kLaneScratchBytes;  // the most the lane phase needs: windows + scratch
// the barrier-free kernels (place_block<..., NOBAR>): per wavefront max(its 64 lanes' scratch columns, its general path's two tiles), 16-byte aligned
__host__ __device__ constexpr int place_wave_lds(int wpad)
{
    return (((kWinWords * 64 * 8 > 2 * wpad * 8 ? kWinWords * 64 * 8 : 2 * wpad * 8) + 15) / 16) * 16;
}
constexpr int kPlaceStaticLds = 2 * kPlaceBlock * 4 + 64 + 256;                // the lists (+ place_single_kernel's request)
// bytes of the long path's per-type tables when they are staged in LDS: elig + pref ([T][W] words each), pc + nz ([2][T][W + 1] ints each)
__host__ __device__ constexpr size_t long_tables_bytes(int T, int W) { return (size_t)T * W * 16 + (size_t)4 * T * (W + 1) * 4; }
// A request against its type's recorded shortlists (see TypeMemo), a lane per request, the tables read from (L1-resident) global
// memory: true = decided, the result row written.  false: the request's own positions would change the walk itself (or: more
// exclusions than the check sees, no recorded list for the type) — the ordinary path decides.
// `rows`: the snapshot's records — S.memo, or the wavefront's copy of it in LDS.
// The result row of a request the check decides, as a NON-TEMPORAL 16-byte store: nobody on the device reads the rows again, and 12.8 MB
// of them per 800k batch otherwise take lines of L2 from the tables every request gathers from.  Measured (tools/r6/exp30.sh, one visit,
// alternating builds): the first launch alone 19.06 -> 18.36 us per 800k rows, a split call on four streams 13.12 -> 12.2 us (61 -> 65.5 G
// decisions/s).  The same for the request LOADS is a loss (21.8 / 17.4 us): they are read once, but in 64-byte pieces a lane.  The lane
// phase of place_block stores its rows the same way (800k rows through place_batch_kernel 25.6 -> 24.6 us, the full cluster 41.8 -> 40.8 us,
// 100k launches unchanged: tools/r6/exp31.sh).
__device__ __forceinline__ void store_out_streaming(mmp_place_out *p, const mmp_place_out &o)
{
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    static_assert(sizeof(mmp_place_out) == 16, "one 16-byte store");
    __builtin_nontemporal_store(*reinterpret_cast<const v4i_ *>(&o), reinterpret_cast<v4i_ *>(p));
}
template <int FORM>
__device__ __forceinline__ bool memo_try(const Snap &S, const PlaceArgs &A, const mmp_place_req &rq, int d, const TypeMemo *rows)
{
    // first level: the model's word, the caller's position, the positions of the request's own exclusions — side by side
    const bool m_ok = (uint32_t)rq.model < (uint32_t)A.n_models;
    uint32_t mw = (uint32_t)A.mtw[m_ok ? rq.model : 0];
    if (!m_ok) mw = kMtwNone;
    const bool s_ok = (uint32_t)rq.self_pod < (uint32_t)S.P;
    int32_t sp = S.pos_of[s_ok ? rq.self_pod : 0];
    if (!s_ok) sp = -1;
    int32_t xp[kLateExtra];
#pragma unroll
    for (int j = 0; j < kLateExtra; j++) xp[j] = -1;
    const bool x_ok = (uint32_t)rq.n_extra <= (uint32_t)kLateExtra && !(A.extra_bound != 0 && bad_extra_range(A, rq));
    if (__ballot(x_ok && rq.n_extra > 0)) {  // (wave-uniform) the request's own exclusions: their positions
        // Branch-free: all four pool reads in flight together, then all four position gathers (a lane without a j-th exclusion reads a
        // harmless word instead) — a load under a branch of its own is waited for before the next one is issued, and with one request
        // in twenty carrying 1-3 exclusions nearly every wavefront walked them as six dependent levels.
        int32_t ep[kLateExtra];
#pragma unroll
        for (int j = 0; j < kLateExtra; j++) {
            const int32_t *pa = (x_ok && j < rq.n_extra) ? A.extra + rq.extra_off + j : S.pos_of;
            ep[j] = *pa;
        }
#pragma unroll
        for (int j = 0; j < kLateExtra; j++) {
            const bool on = x_ok && j < rq.n_extra && (uint32_t)ep[j] < (uint32_t)S.P;
            const int32_t v = S.pos_of[on ? ep[j] : 0];
            xp[j] = on ? v : -1;
        }
    }
    // second level: the type's record — the best row AND both lists' headers at once (one level of the chain instead of two; the lean
    // kernel has the registers)
    const int type = (int)(mw & kMtwTypeMask);
    const bool t_ok = type < kWinLds && x_ok;
    const TypeMemo *Mp = &rows[type < kWinLds ? type : 0];
    const int64_t b_rem = Mp->b_rem, b_lru = Mp->b_lru;
    const int4 bh = *reinterpret_cast<const int4 *>(&Mp->best_is_full);  // best_is_full, b_rpm, best_idx, plain
    const int4 v0h = *reinterpret_cast<const int4 *>(&Mp->v[0]), v1h = *reinterpret_cast<const int4 *>(&Mp->v[1]);  // valid, lo, hi, ccount
    const uint64_t hs0 = Mp->v[0].hsum, hs1 = Mp->v[1].hsum;
    const int32_t full = bh.x, b_rpm = bh.y, best_idx = bh.z;
    // the fresh-row test, MM.java:4913-4922 (as lane_decide_win has it); the full-mode form only in wavefronts that hold such a type
    const int64_t f_rem = remaining_of(rq.fresh_capacity, rq.fresh_used);
    bool nsb = f_rem < S.min_space || f_rem < (b_rem >> 2);
    if (__ballot(full != 0)) {  // (wave-uniform)
        const int64_t rel = age_of(b_lru, A.now) / 10;
        const int64_t d1 = jsub64(rq.fresh_lru, b_lru);
        if (full) nsb = d1 > 45000LL && d1 > rel;
    }
    const int4 vh = nsb ? v1h : v0h;
    uint64_t hsum = nsb ? hs1 : hs0;
    const int lo = vh.y;
    const uint32_t len = (uint32_t)(vh.z - lo);
    bool miss = !t_ok || !vh.x || (mw & (nsb ? kMtwTail1 : kMtwTail)) != 0;
    // own positions inside [lo, hi): the caller, the request's exclusions; the model's come classified in its word
    bool x_in = false;
#pragma unroll
    for (int j = 0; j < kLateExtra; j++) x_in |= (uint32_t)(xp[j] - lo) < len;  // (-1 - lo wraps far beyond len)
    const bool sp_in = (uint32_t)(sp - lo) < len;
    const uint32_t moff = nsb ? 0u : (mw >> kMtwOffShift);  // (list 1 is the best instance alone: nothing to take out of it)
    const bool favour = (rq.flags & MMP_REQ_FAVOUR_SELF) != 0;
    int ccount = vh.w;
    int r1 = 0, r2 = 0;      // candidate numbers of the (at most two) excluded candidates, 0: none
    int ks = 0;              // the caller's candidate number when its own entry is a candidate (self_in)
    bool self_in = false, self_best = false;
    int32_t e_rpm = 0;
#ifdef MMP_XP_NOOWN  // (experiment builds only, tools/r6: what the own-positions block costs)
    miss |= x_in || sp_in || moff != 0;
#endif
    const bool own = !miss && (x_in || sp_in || moff != 0);
    if (__ballot(own)) {  // (wave-uniform: a third of the wavefronts of a batch of request rows, C3)
        if (own) {
            const int4 mt = *reinterpret_cast<const int4 *>(&Mp->w0);  // w0, bestpos, best0, e_rpm
            const int32_t sbk = Mp->sbk;
            const int wl = mt.x * 64;
            // what each own position is to the list: one lookup apiece, all in flight together — in the row itself for the first
            // kMemoNear positions from the first eligible instance on (nearly always), else in the window-wide table
            const int16_t *RK = S.memo_rk + type * kMemoCand;
            // (all seven row reads first, then — hardly ever — the window-wide table for the positions the row does not reach: kept
            // apart, or the compiler merges a row read and a table read into one FLAT load through a selected pointer)
            const int o1 = (int)(moff & kMtwOffMask) - 1, o2 = (int)((moff >> 9) & kMtwOffMask) - 1;
            int qp[kLateExtra + 3];  // the own positions that lie inside [lo, hi), -1: none
            qp[0] = sp_in ? sp : -1;
            qp[1] = o1 >= 0 ? wl + o1 : -1;
            qp[2] = o2 >= 0 ? wl + o2 : -1;
#pragma unroll
            for (int j = 0; j < kLateExtra; j++) qp[3 + j] = (uint32_t)(xp[j] - lo) < len ? xp[j] : -1;
            int qr[kLateExtra + 3];
            bool far = false;
#pragma unroll
            for (int j = 0; j < kLateExtra + 3; j++) {
                const int rel = qp[j] - lo;  // (lo = the type's first eligible instance for either list)
                const bool near = qp[j] >= 0 && rel < kMemoNear;
                qr[j] = (int)Mp->rk64[near ? rel : 0];
                if (!near) qr[j] = kRkNone;
                far |= qp[j] >= 0 && rel >= kMemoNear;
            }
            if (__ballot(far)) {
#pragma unroll
                for (int j = 0; j < kLateExtra + 3; j++) {
                    const bool f = qp[j] >= 0 && qp[j] - lo >= kMemoNear;
                    int g = (int)RK[f ? qp[j] - wl : 0];
                    asm volatile("" : "+v"(g));
                    if (f) qr[j] = g;
                }
            }
            const int srk = qr[0], m1 = qr[1], m2 = qr[2];
            int xr[kLateExtra];
#pragma unroll
            for (int j = 0; j < kLateExtra; j++) xr[j] = qr[3 + j];
            const int end_v = vh.z - 1;  // the instance that ends the list (list 1 depends on it)
            // an exclusion at a list position: the best / first eligible instance -> another walk; candidate k -> taken out
            auto take = [&](int rk, int pos) {
                if (rk == 0 || rk == kRkFirst || (nsb && pos == end_v))
                    miss = true;
                else if (!nsb && rk > 0 && rk != r1 && rk != r2) {
                    if (r1 == 0)
                        r1 = rk;
                    else if (r2 == 0)
                        r2 = rk;
                    else
                        miss = true;
                    hsum -= audit_mul((uint64_t)(pos >> 6)) << (pos & 63);
                }
            };
            take(m1, wl + o1);
            take(m2, wl + o2);
#pragma unroll
            for (int j = 0; j < kLateExtra; j++) take(xr[j], xp[j]);
            // The caller's own entry.  As the best instance: ABORT_REQUEST with favourSelf (:4891-4895), else its fresh row replaces
            // the best row in every rule: the ordinary path.  As candidate k with the fresh-row break off: one more candidate of the
            // SAME list (the break rules compare the best row with itself where no preference step was taken; where one was, `sbk`
            // says whether the caller's own break fires); what changes is favourSelf (:4931), the rpm rule's classes (:4951-4980: the
            // caller's entry carries the snapshot rpm of the FIRST eligible instance, "the others" the fresh one) and that choosing it
            // means ABORT_REQUEST (:4989).  With the fresh-row break on, the caller next in line behind the best instance stays in the
            // list (the break skips it): the ordinary path; further back it is not reached.
            if (sp_in) {
                if (srk == 0) {
                    if (favour)
                        self_best = true;
                    else
                        miss = true;
                } else if (srk == kRkFirst)
                    miss = true;
                else if (nsb) {
                    if (sp == end_v) miss = true;
                } else if (srk > 0 && srk != r1 && srk != r2) {  // (an excluded caller is no candidate)
                    if (sbk)
                        miss = true;
                    else {
                        self_in = true;
                        ks = srk;
                        e_rpm = mt.w;
                    }
                }
            }
            ccount -= (r1 != 0) + (r2 != 0);
        }
    }
    if (miss) return false;
    mmp_place_out o;
    o.best = best_idx;
    if (self_best || (self_in && favour)) {  // :4891-4895, :4931-4933
        o.chosen = MMP_SELF;
        o.n_candidates = 0;
        o.hash = 0;
        store_out_streaming(A.outs + d, o);
        return true;
    }
    // rpm filter, :4951-4980: the best instance, the caller's entry, "the others" (the fresh rpm, quirks B#2/B#3)
    const int32_t f_rpm = rq.fresh_rpm;
    const int n_others = ccount - 1 - (self_in ? 1 : 0);
    int32_t mn = b_rpm;
    if (self_in && e_rpm < mn) mn = e_rpm;
    if (n_others > 0 && f_rpm < mn) mn = f_rpm;
    RpmRule rule;
    rule.init(age_of(rq.last_used, A.now), mn);
    const int32_t lim = rule.limit();
    const bool null0 = ccount >= 2 && b_rpm >= 100 && b_rpm > lim;
    const bool null_s = ccount >= 2 && self_in && e_rpm >= 100 && e_rpm > lim;
    const bool null_o = ccount >= 2 && n_others > 0 && f_rpm >= 100 && f_rpm > lim;
    const int remaining = ccount - (null0 ? 1 : 0) - (null_s ? 1 : 0) - (null_o ? n_others : 0);
    const int index = remaining <= 1 ? 0 : (int)(((uint64_t)rq.pick * (uint64_t)(uint32_t)remaining) >> 32);
    // candidate number of the index-th survivor
    int k;
    if (null_o)
        k = (!null0 && index == 0) ? 0 : ks;  // survivors: the best instance, then the caller's entry (whichever the rule left in)
    else {
        k = index + (null0 ? 1 : 0);
        if (__ballot(null_s || r1 != 0)) {  // (wave-uniform) numbers taken out of the list, ascending: each one at or before k pushes it on
            int a = null_s ? ks : INT32_MAX, b = r1 ? r1 : INT32_MAX, c = r2 ? r2 : INT32_MAX, t;
            if (a > b) { t = a; a = b; b = t; }
            if (b > c) { t = b; b = c; c = t; }
            if (a > b) { t = a; a = b; b = t; }
            if (a <= k) k++;
            if (b <= k) k++;
            if (c <= k) k++;
        }
    }
    o.chosen = MMP_NONE;
    if (remaining >= 1) {
#ifdef MMP_XP_NOCAND  // (experiment builds only: the last gather of the chain left out — wrong results)
        o.chosen = best_idx + k;
#else
        o.chosen = Mp->cand64[k < kMemoNear ? k : 0];
        if (k >= kMemoNear) {  // (see rk_of)
            int32_t g = S.memo_cand[type * kMemoCand + k];
            asm volatile("" : "+v"(g));
            o.chosen = g;
        }
        if (k == 0) o.chosen = best_idx;
#endif
        if (self_in && k == ks) o.chosen = MMP_SELF;  // :4989-4991
    }
    o.n_candidates = ccount;
    o.hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
    store_out_streaming(A.outs + d, o);
    return true;
}

// A resolved request (exclusions merged) against its type's recorded long walk (see LongMemo): true = decided, `o` filled (the caller
// stores it).  false: one of the request's own positions steers the walk — an exclusion or the caller inside [best0, bestpos], the
// instance that ends the list excluded (or, with the fresh-row break on, the caller: that scan skips it), the caller's own break rule
// cutting the list short — or there is no record: lane_decide_r<..., LONG> walks.  Behind the record the steps are the prefix-table
// phase's own: corrections for the excluded candidates, the rpm rule's three classes, the index-th survivor by one `sel` lookup.
__device__ __forceinline__ bool long_memo_try(const Snap &S, const PlaceArgs &A, const ResolvedReq &r, mmp_place_out &o)
{
    bool miss = r.type < 0 || r.n_excl > kInlineExcl || A.force_wave;
#ifdef MMP_PHASE_CLOCK
    if (miss) PHASE_WHY(1);
#endif
    const LongMemo *Mp = S.lmemo + (size_t)(r.type < 0 ? 0 : r.type) * kLongLevels;
    int64_t b_rem = Mp->b_rem, b_lru = Mp->b_lru;
    int4 h1 = *reinterpret_cast<const int4 *>(&Mp->best_is_full);  // best_is_full, b_rpm, best_idx, has_pm
    int4 h2 = *reinterpret_cast<const int4 *>(&Mp->best0);         // best0, bestpos, e_rpm, sbk
    int4 v0 = *reinterpret_cast<const int4 *>(&Mp->v[0]), v1 = *reinterpret_cast<const int4 *>(&Mp->v[1]);  // valid, end, ccount, g0
    uint64_t hs0 = Mp->v[0].hsum, hs1 = Mp->v[1].hsum;
    const int32_t nb1 = Mp[1].best0;  // the type's second eligible instance (fetched beside the record)
    auto excluded = [&](int p) {
        bool x = false;
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++) x |= r.excl_pos[i] == p;
        return x;
    };
    // the type's first eligible instance(s) excluded: the record of that many levels down (wave-uniform test: hardly ever)
    int lvl = 0;
    if (excluded(h2.x)) lvl = excluded(nb1) ? 2 : 1;
    if (__ballot(lvl != 0)) {
        if (lvl != 0) {
            const LongMemo *Lp = Mp + lvl;
            b_rem = Lp->b_rem;
            b_lru = Lp->b_lru;
            h1 = *reinterpret_cast<const int4 *>(&Lp->best_is_full);
            h2 = *reinterpret_cast<const int4 *>(&Lp->best0);
            v0 = *reinterpret_cast<const int4 *>(&Lp->v[0]);
            v1 = *reinterpret_cast<const int4 *>(&Lp->v[1]);
            hs0 = Lp->v[0].hsum;
            hs1 = Lp->v[1].hsum;
        }
    }
    const int P = S.P, W = S.W;
    const int best0 = h2.x, bestpos = h2.y;
    const int selfpos = r.selfpos;
    const bool favour = (r.flags & MMP_REQ_FAVOUR_SELF) != 0;
    // The caller IS the best instance (no preference step taken): ABORT_REQUEST with favourSelf (:4891-4895); else its fresh record
    // replaces the best row in every rule (:4811-4830) — a full caller (fresh record below minSpace) puts the walk into the LRU-window
    // mode with d1 = 0: no fresh-row break, no count break, its own entry in front of the range: the list of bit 0 where nothing ended
    // that one either, with the caller's fresh rpm as the best instance's
    const bool self_best = selfpos >= 0 && selfpos == best0 && best0 == bestpos && (v0.x | v1.x) != 0;
    const bool abort_now = !miss && self_best && favour;
    bool nsb;  // :4913-4922, as lane_decide_r has it
    if (h1.x) {
        const int64_t rel = age_of(b_lru, A.now) / 10;
        const int64_t d1 = jsub64(r.fresh_lru, b_lru);
        nsb = d1 > 45000LL && d1 > rel;
    } else
        nsb = r.f_rem < S.min_space || r.f_rem < (b_rem >> 2);
    if (self_best) nsb = false;
    const int4 vh = nsb ? v1 : v0;
    uint64_t hsum = nsb ? hs1 : hs0;
    const int end = vh.y, start = bestpos + 1, g0 = vh.w;
    const int32_t b_rpm = self_best ? r.fresh_rpm : h1.y, best_idx = h1.z, e_rpm = h2.z, f_rpm = r.fresh_rpm;
    bool self_excl = false;
    if (!abort_now) {
        if (!vh.x) {
            if (!miss) PHASE_WHY(2);
            miss = true;
        }
        if (self_best && !(r.f_rem < S.min_space && end == P)) {
            if (!miss) PHASE_WHY(4);
            miss = true;
        }
        bool hit = self_best || (!(selfpos >= best0 && selfpos <= bestpos) && !(nsb && selfpos == end));
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++) {
            const int e = r.excl_pos[i];
            hit &= !(e >= best0 && e <= bestpos) && e != end;
            self_excl |= e == selfpos;
        }
        if (!hit) {
            if (!miss) PHASE_WHY(8);
            miss = true;
        }
        // the caller's own break rule (a type-level fact, `sbk`) would end the list at the caller's position if that is a candidate:
        // another list.  (Whether it is a candidate is one more lookup; such types are left to the walk for every caller in range.)
        if (h2.w && selfpos >= start && selfpos < end && !self_excl) {
            if (!miss) PHASE_WHY(16);
            miss = true;
        }
    }
    // (Measured and not kept: the whole wavefront walking as soon as ONE of its requests must — a walk with 64 lanes active is slower
    // than the check plus a walk with one: the slowest wavefront of a 100k launch 12.3 -> 13.8 us, and that wavefront IS the launch;
    // 800k requests 40.3 -> 43.1 us.  tools/r6/wave_timeline.py, profiles/r6/long_records.txt)
    if (miss) return false;
    if (abort_now) {
        o.chosen = MMP_SELF;
        o.best = h1.z;
        o.n_candidates = 0;
        o.hash = 0;
        return true;
    }
    const size_t row = (size_t)(h1.w ? 1 : 0) * S.T + r.type;
    const int32_t *SEL = S.sel + row * (size_t)W * 64, *RK = S.rk + row * (size_t)W * 64;
    int32_t sx[kInlineExcl], rkx[kInlineExcl];
    uint64_t amx[kInlineExcl];
    sort_excl(r.excl_pos, sx);
    // everything the corrections read, and what the caller's own position is to the row, fetched together
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int e = sx[i];
        const bool in = e >= start && e < end;  // (the padding, INT32_MAX, is not)
        rkx[i] = in ? RK[e] : -1;
        amx[i] = in ? S.amul[e >> 6] : 0ull;
    }
    const bool s_rng = selfpos >= start && selfpos < P;
    int32_t srk = RK[s_rng ? selfpos : 0];
    if (!s_rng) srk = -1;
    const bool self_in_d = srk >= 0 && !(self_excl && selfpos >= 0);
    const bool self_in_c = self_in_d && selfpos < end;
    o.chosen = MMP_NONE;
    o.best = best_idx;
    o.n_candidates = 0;
    o.hash = 0;
    if (self_in_c && favour) {  // :4931-4933
        o.chosen = MMP_SELF;
        return true;
    }
    uint32_t xmask = 0;  // slots of sx[] that are distinct candidates inside [start, end)
#pragma unroll
    for (int i = 0; i < kInlineExcl; i++) {
        const int e = sx[i];
        const bool dup = i > 0 && sx[i - 1] == e;
        const bool bit = !dup && rkx[i] >= 0;
        xmask |= (uint32_t)bit << i;
        if (bit) hsum -= amx[i] << (e & 63);
    }
    const int ccount = vh.z - __popc(xmask);
    int remaining = ccount;
    bool null0 = false, null_s = false, null_o = false;
    int n_others = 0;
    if (ccount >= 2) {  // rpm filter, :4951-4980 (quirks B#2/B#3: three rpm classes)
        n_others = ccount - 1 - (self_in_c ? 1 : 0);
        int32_t mn = b_rpm;
        if (self_in_c && e_rpm < mn) mn = e_rpm;
        if (n_others > 0 && f_rpm < mn) mn = f_rpm;
        RpmRule rule;
        rule.init(age_of(r.last_used, A.now), mn);
        null0 = rule.nulls(b_rpm);
        null_s = self_in_c && rule.nulls(e_rpm);
        null_o = n_others > 0 && rule.nulls(f_rpm);
        remaining = ccount - (null0 ? 1 : 0) - (null_s ? 1 : 0) - (null_o ? n_others : 0);
    }
    const int index = remaining <= 1 ? 0 : (int)(((uint64_t)r.pick * (uint64_t)(uint32_t)remaining) >> 32);
    int cpos = kNoPos;
    if (remaining >= 1) {
        if (null_o) {  // only the best and the self entry can be left, in that order
            int k = index;
            if (!null0) {
                if (k == 0) cpos = bestpos;
                k--;
            }
            if (cpos == kNoPos && self_in_c && !null_s && k == 0) cpos = selfpos;
        } else {
            // t = the raw rank of the index-th survivor: every removed entry at or before it pushes it one further (one ascending pass)
            const int rho_s = null_s ? 1 + srk - g0 : 0;
            bool self_pending = null_s;
            int t = index + (null0 ? 1 : 0);
#pragma unroll
            for (int i = 0; i < kInlineExcl; i++) {
                if (self_pending && selfpos < sx[i]) {
                    if (rho_s <= t) t++;
                    self_pending = false;
                }
                if ((xmask >> i) & 1u) {
                    if (1 + rkx[i] - g0 <= t) t++;
                }
            }
            if (self_pending && rho_s <= t) t++;
            cpos = t == 0 ? bestpos : SEL[g0 + t - 1];
        }
    }
    o.n_candidates = ccount;
    o.hash = (uint32_t)(hsum ^ (hsum >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
    if (cpos != kNoPos) {
        o.chosen = S.orig[cpos];
        if (!favour && cpos == selfpos) o.chosen = MMP_SELF;  // :4989-4991
    }
    return true;
}

// commit (the last blocks of build_sel_memo_kernel): one wavefront per type row, lanes 0 / 1 = the fresh-row bit (see LongMemo); reads
// the prefix tables, not sel / rk (a request without exclusions looks nothing up in them), so it runs beside their construction
__device__ __forceinline__ void build_long_memo_body(int t, const Snap &S, LongMemo *__restrict__ out)
{
    __shared__ LongCap caps[2 * kLongLevels];
    const int lane = lane_id();
    if (lane < 2 * kLongLevels) {
        const int level = lane >> 1;
        PlaceArgs A{};
        ResolvedReq r{};
        r.type = t;
        r.selfpos = -1;
        r.n_late = -1;
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++) r.excl_pos[i] = -1;
#pragma unroll
        for (int i = 0; i < kLateExtra; i++) r.late_pos[i] = -1;
        // level l: the type's first l eligible instances excluded
        bool bad = false;
        int p = -1;
        const uint64_t *E = S.elig + (size_t)t * S.W;
#pragma unroll
        for (int j = 0; j < kLongLevels - 1; j++) {
            if (j < level && !bad) {
                bool far = false;
                const int q = lane_first([&](int w) { return E[w]; }, p + 1, S.P, far, S.nz, (size_t)t * (size_t)(S.W + 1));
                if (far || q == kNoPos)
                    bad = true;
                else {
                    r.excl_pos[j] = q;
                    p = q;
                }
            }
        }
        r.n_excl = level;
        LongCap lc{};
        lc.nsb = lane & 1;
        if (!bad) {
            mmp_place_out o;
            const int code = lane_decide_r<false, true>(S, A, r, o, BLds{}, &lc);
            if (code != kLaneDone) lc.ok = 0;
        }
        caps[lane] = lc;
    }
    __syncthreads();
    if (lane < kLongLevels) {
        const LongCap &c0 = caps[2 * lane], &c1 = caps[2 * lane + 1];
        LongMemo M = (c0.ok ? c0 : c1).m;  // the head fields do not depend on the bit
        M.v[0] = c0.ok ? c0.m.v[0] : LongVar{};
        M.v[1] = c1.ok ? c1.m.v[1] : LongVar{};
        if (!c0.ok && !c1.ok) {
            M = LongMemo{};
            M.best0 = M.bestpos = kNoPos;  // (never a request's own position)
        }
        out[(size_t)t * kLongLevels + lane] = M;
    }
}

// NOBAR (the kernels with the shortlist check in front: place_batch_m_kernel, place_batch_c_m_kernel): no workgroup barrier at all —
// the head windows are read from global memory (a few L1-resident rows) instead of being staged in LDS by the workgroup, and a
// wavefront keeps the general path's list for itself.  A wavefront whose requests the shortlists all cover then neither waits for
// the staging nor, at the end, for the lane phase of the workgroup's other wavefronts (with the barriers the check bought nothing
// for request rows: 27.4 against 25.9 us; without them 24.7).  The kernels WITHOUT the check keep staging + barrier: for them the
// windows in LDS are the faster read (800k rows: 26.0 against 26.5 us, four streams 17.75 against 18.4).
// LIST (the tail launch of the split form, place_tail_body): the workgroup's decisions come as a list — this lane's is `d_list`
// (-1: none) — and the function may be called again by the same workgroup (it ends behind a barrier then).
template <bool WITH_LONG, int FORM = kReq64, bool MEMO = false, bool NOBAR = false, bool LIST = false>
__device__ __forceinline__ void place_block(const Snap &S, const PlaceArgs &A, int32_t wpad, unsigned char *smem,
                                            uint32_t *done_blocks = nullptr, const mmp_place_caller &C = mmp_place_caller{}, int d_list = -1)
{
    static_assert(!(LIST && NOBAR), "the list form keeps the workgroup's barriers");
    __shared__ int32_t fb_list[kPlaceBlock], lr_list[kPlaceBlock];
    __shared__ int32_t fb_n, lr_n;
    // The staged windows and the per-lane scratch of the lane phase, and the wave path's bitmap tiles (kPlaceWaves x 2
    // bitmaps x wpad words) of the phase behind it, share ONE dynamic LDS region (smem; the host sizes it for the
    // larger of the two, place_lane_lds(T) / the tiles): a 50k-instance table's tiles are 50 KB, and next to 41 KB of
    // windows + scratch they left room for one workgroup per CU (C4, round 2: 86 us per 1M decisions).
    // NOBAR: nothing orders one wavefront's lane phase before another's general path, so every wavefront has a region of its OWN
    // (place_wave_lds(wpad) bytes: its lanes' scratch columns, and — the lane phase being over by then — its general path's two tiles).
    TypeWin *s_wins = reinterpret_cast<TypeWin *>(smem);
    const int wave_nb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *smem_w = smem + (size_t)wave_nb * place_wave_lds(wpad);
    uint64_t *s_scr = NOBAR ? reinterpret_cast<uint64_t *>(smem_w)
                            : reinterpret_cast<uint64_t *>(smem + win_lds_bytes(S.T));  // per lane: the window's eligibility words with the request's exclusions cleared
    if (threadIdx.x == 0) fb_n = lr_n = 0;
    const int d = LIST ? d_list : (int)(blockIdx.x * kPlaceBlock + threadIdx.x);
    PHASE_T0();
    PHASE_ABS(14);
    // The windows are fetched beside the request (they depend on nothing) and parked in LDS while the
    // request -> model row chain is in flight; the barrier below is the one the lists needed anyway.
    const bool use_wins = A.wins != nullptr;  // wave-uniform
    const TypeWin *Ws = NOBAR ? A.wins : s_wins;
    // global -> LDS directly (global_load_lds_dwordx4: no staging registers), one 1 KB chunk per wavefront and
    // trip: lane l of the wavefront that takes chunk c moves bytes [1024 c + 16 l, +16).  The window table is
    // allocated for kWinLds rows, so whole chunks are always in bounds.
    constexpr int kWinBytes = (int)sizeof(TypeWin);
    if (use_wins && !NOBAR) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int chunks = ((S.T < kWinLds ? S.T : kWinLds) * kWinBytes + 1023) >> 10;
        const char *src = reinterpret_cast<const char *>(A.wins);
        char *dst = reinterpret_cast<char *>(s_wins);
        for (int c = wave; c < chunks; c += kPlaceWaves)
            __builtin_amdgcn_global_load_lds(src + (size_t)c * 1024 + lane_id() * 16,
                                             (__attribute__((address_space(3))) void *)(dst + c * 1024), 16, 0, 0);
    }
    // The workgroup barrier sits right behind the request fetch — the one load every decision needs first.  Loads
    // return in order, so the LDS-bound copies issued before it have landed when it has; and nothing issued AFTER the
    // barrier (the model row, the caller's position, the late-bound exclusions) has to be drained for it.
    // bounded calls: a request whose exclusions lie outside the declared pool is answered here and takes no further part
    bool live = LIST ? d >= 0 : d < A.n;
    mmp_place_req rq{};
    if (live) rq = fetch_req<FORM>(A, C, d);
    if (A.extra_bound != 0) {  // (wave-uniform)
        if (live && bad_extra_range(A, rq)) {
            live = false;
            mmp_place_out bo;
            bo.chosen = MMP_NONE;
            bo.best = MMP_BAD_REQUEST;
            bo.n_candidates = 0;
            bo.hash = 0;
            A.outs[d] = bo;
        }
    }
    // The long path on a full cluster reads, per decision, ~57 words of tables every decision shares — the type's eligibility /
    // preference rows (the scans), the prefix counts (8 binary-search steps), the next-non-empty-word rows: 20 KB on C3.  Staged
    // here (beside the request fetch, before the barrier) for launches that fill the chip, where they relieve L2 (-5 % at 800k;
    // a 100k launch is not faster with them: the host decides, PlaceArgs::long_first).
    Snap Sl = S;
    if (WITH_LONG && !NOBAR && A.long_first > 1) {  // wave-uniform (the barrier-free instantiation is launched without staged tables)
        unsigned char *tb = smem + (A.long_first - 1);
        const int nE = S.T * S.W, nPC = 2 * S.T * (S.W + 1);
        uint64_t *lE = reinterpret_cast<uint64_t *>(tb), *lP = lE + nE;
        int32_t *lpc = reinterpret_cast<int32_t *>(lP + nE), *lnz = lpc + nPC;
        for (int i = threadIdx.x; i < nE; i += kPlaceBlock) {
            lE[i] = S.elig[i];
            lP[i] = S.pref[i];
        }
        for (int i = threadIdx.x; i < nPC; i += kPlaceBlock) {
            lpc[i] = S.pc[i];
            lnz[i] = S.nz[i];
        }
        Sl.elig = lE;
        Sl.pref = lP;
        Sl.pc = lpc;
        Sl.nz = lnz;
    }
    // case (b) on a full cluster (long kernel): the snapshot's whole-window tables (BSlot)
    BLds Bt{};  // (by value, n_slots == 0 = none: a pointer to it kept the struct in scratch memory — 40 bytes written per decision)
    if (WITH_LONG && A.n_bslots > 0) {  // wave-uniform
        Bt.slots = A.bslots;
        Bt.n_slots = A.n_bslots < kBSlots ? A.n_bslots : kBSlots;
        Bt.W = S.W;
        Bt.surv = A.bsurv;
        Bt.pcs = A.bpcs;
        Bt.launch = A.bwin;
    }
    if (!NOBAR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // the type's recorded shortlist first (TypeMemo): a wavefront whose requests are all covered is done here
    if (MEMO && live && memo_try<FORM>(S, A, rq, d, S.memo)) live = false;
    ResolvedReq r;
    int wcode = 0;  // (NOBAR) this lane's decision is left to the general path
    if (live) r = resolve_req<false, true>(S, A, rq);
    PHASE(0);  // request + model row resolved
    if (live) {
        mmp_place_out o;
        int code = kLaneHeadMiss;
        if (WITH_LONG && A.long_first) {  // (wave-uniform) a full cluster: nearly every decision would end in the long phase anyway
            merge_late_extras(r);
            // the type's recorded walk first (LongMemo): a wavefront whose requests it all answers skips the scans
            const bool rec = S.lmemo != nullptr && long_memo_try(S, A, r, o);  // (S.lmemo: wave-uniform)
            code = kLaneDone;
            if (!rec) code = lane_decide_r<false, true>(Sl, A, r, o, Bt);
        } else {
            if (use_wins)
                code = NOBAR ? lane_decide_win<false, false, 64>(S, A, r, Ws, s_scr + lane_id(), o)
                             : lane_decide_win(S, A, r, Ws, s_scr + threadIdx.x, o);
            if (code == kLaneHeadMiss) {
                merge_late_extras(r);
                code = lane_decide_r<false>(S, A, r, o, Bt);
            }
        }
        if (WITH_LONG && NOBAR && (code == kLaneLong || code == kLaneCaseB)) {  // the prefix-table phase at once, in this lane
            code = lane_decide<false, true, FORM>(Sl, A, d, o, Bt, C);
            if (code == kLaneDone)
                store_out_streaming(A.outs + d, o);
            else
                wcode = 1;
        } else if (WITH_LONG && (code == kLaneLong || code == kLaneCaseB))
            lr_list[atomicAdd(&lr_n, 1)] = d;
        else if (code != kLaneDone) {
            if (NOBAR)
                wcode = 1;
            else
                fb_list[atomicAdd(&fb_n, 1)] = d;
        } else
            store_out_streaming(A.outs + d, o);
    }
    PHASE(8);  // the whole lane phase of this wavefront (incl. the result store)
    if (NOBAR) {  // the general path for this wavefront's own leftovers, a decision at a time
        const uint64_t fb = __ballot(wcode != 0);
        if (fb) {
            const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            int32_t *mine = fb_list + wave * 64;
            if (wcode) mine[__popcll((unsigned long long)(fb & ((1ull << lane_id()) - 1ull)))] = d;
            wave_sync();
            uint64_t *ew = reinterpret_cast<uint64_t *>(smem_w);  // (this wavefront's own region: its lane phase is over)
            uint64_t *fw = ew + wpad;
            const int nf = __popcll((unsigned long long)fb);
            for (int i = 0; i < nf; i++) {
                const int fd = __builtin_amdgcn_readfirstlane(mine[i]);
                place_one<FORM>(S, A, fd, ew, fw, C);
                wave_sync();
            }
        }
        PHASE(10);  // the general path
        PHASE_COUNT(11, 1);  // wavefronts
        PHASE_ABS(15);
        announce_done(DoneFlag{A.done_flag, done_blocks, A.done_seq});  // (latency-slot launches of the barrier-free long kernel; a no-op otherwise)
        return;
    }
    __syncthreads();
    PHASE(9);  // waiting for the workgroup's other wavefronts
    // the decisions whose shortlist spans many words: again one lane each, this time through the prefix tables
    // (a phase of its own so that its registers do not count against the common path above)
    if (WITH_LONG && lr_n != 0) {
        const int nlr = lr_n;
        if ((int)threadIdx.x < nlr) {
            const int ld = lr_list[threadIdx.x];
            mmp_place_out o;
            if (lane_decide<false, true, FORM>(Sl, A, ld, o, Bt, C) != kLaneDone)
                fb_list[atomicAdd(&fb_n, 1)] = ld;
            else
                store_out_streaming(A.outs + ld, o);
        }
        __syncthreads();
    }
    const int nfb = fb_n;
    if (nfb != 0) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        uint64_t *ew = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * 2 * wpad;
        uint64_t *fw = ew + wpad;
        for (int i = wave; i < nfb; i += kPlaceWaves) {
            const int fd = __builtin_amdgcn_readfirstlane(fb_list[i]);
            place_one<FORM>(S, A, fd, ew, fw, C);
            wave_sync();
            PHASE_COUNT(12, 1);  // decisions the wave path took
        }
    }
    PHASE(10);  // long phase + wave path
    PHASE_COUNT(11, 1);  // wavefronts
    PHASE_ABS(15);
    announce_done(DoneFlag{A.done_flag, done_blocks, A.done_seq});
    if (LIST) __syncthreads();  // the next call resets the lists
}

// 6 wavefronts per SIMD (80 VGPRs, 100 bytes of spill per lane) instead of the 5 the unconstrained allocation (95) gives:
// measured -3.4 % per 800k-decision launch, -5 % per step on two streams; 7 the same, 8 (64 VGPRs) +7 %.
#ifndef MMP_PLACE_EU
#define MMP_PLACE_EU 6
#endif
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_PLACE_EU, MMP_PLACE_EU))) void place_batch_kernel(Snap S, PlaceArgs A, int32_t wpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<false>(S, A, wpad, smem);
}

// the same with the long-shortlist phase (see place_block): 144 VGPRs, 3 wavefronts per SIMD
// (launches below kLongDenseFrom decisions: the per-type tables are never staged in LDS for them, so the workgroup needs no barrier —
// place_block<..., NOBAR>: no window staging either, which the full-cluster path does not read)
__global__ __launch_bounds__(kPlaceBlock) void place_batch_long_kernel(Snap S, PlaceArgs A, int32_t wpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<true, kReq64, false, true>(S, A, wpad, smem);
}

// ... for launches that put more than three wavefronts on a SIMD: 128 VGPRs (88 instead of 48 bytes of spill per lane), 4
// per SIMD.  Measured on the full cluster (C3): 800k decisions per launch 78.5 -> 71.2 us, 100k (1.5 wavefronts per SIMD:
// occupancy is not what limits it) 20.3 -> 21.0 us — hence two instantiations, chosen by the launch's size.
constexpr int kLongDenseFrom = 3 * 4 * 256 * 64;  // decisions from which a launch fills 3 wavefronts on each of the 1024 SIMDs
// (Measured and not kept, round 4: case (b) as a phase of its own in these launches — collected per workgroup, one wavefront
// instead of four runs it — 49.3 against 48.9 us per 800k; five wavefronts per SIMD — 96 VGPRs, 244 bytes of spill — 105 us.)
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void place_batch_long4_kernel(Snap S, PlaceArgs A, int32_t wpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<true>(S, A, wpad, smem);
}

// The single-caller form (mmp_place_batch_c): the same three kernels on 24-byte request rows, the caller's side in the arguments
// (measured, round 5, 800k decisions of one caller per launch: 4 / 5 / 6 / 7 / 8 wavefronts per SIMD 23.7 / 21.1 / 20.5 / 19.3 / 19.7 us —
// with a third of the request bytes a wavefront waits less for memory and one more per SIMD fills the issue slots)
#ifndef MMP_C_WAVES
#define MMP_C_WAVES 7
#endif
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_C_WAVES, MMP_C_WAVES))) void place_batch_c_kernel(Snap S, PlaceArgs A, int32_t wpad,
                                                                                                            mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<false, kReqC>(S, A, wpad, smem, nullptr, C);
}
// ---- per-type shortlists: build (commit) and use (see TypeMemo) ------------------------------------------------------------------
// One wavefront per type row: lanes 0 / 1 run lane_decide_win on the type's window for a request without exclusions and without a
// caller's entry, the fresh-row break off / on; then the wavefront writes the candidates' pod indices in shortlist order.
// (the last blocks of build_sel_memo_kernel: one launch builds sel / rk and the shortlists — both need only what level 2 left)
__device__ __forceinline__ void build_memo_body(int t, const Snap &S, const TypeWin *__restrict__ wins, TypeMemo *__restrict__ memo,
                                                int32_t *__restrict__ cand, int16_t *__restrict__ rkt)
{
    constexpr int kScr = 2;  // scratch columns: the two lanes that decide
    __shared__ uint64_t scr[kWinWords * kScr];
    __shared__ MemoCap caps[2];
    __shared__ int ok[2];
    const int lane = lane_id();
    if (lane < 2) {
        PlaceArgs A{};
        ResolvedReq r{};
        r.type = t;
        r.selfpos = -1;
        r.n_late = -1;
#pragma unroll
        for (int i = 0; i < kInlineExcl; i++) r.excl_pos[i] = -1;
#pragma unroll
        for (int i = 0; i < kLateExtra; i++) r.late_pos[i] = -1;
        MemoCap mc{};
        mc.nsb = lane;
        mmp_place_out o;
        const int code = lane_decide_win<false, true, kScr>(S, A, r, wins, scr + lane, o, &mc);
        caps[lane] = mc;
        ok[lane] = code == kLaneDone && mc.ccount <= kMemoCand;
    }
    __syncthreads();
    const int w0 = wins[t].w0;
    const bool any = ok[0] || ok[1];
    const MemoCap &c0 = caps[ok[0] ? 0 : 1];  // the best row does not depend on the bit
    if (lane == 0) {
        TypeMemoHead M{};
        M.b_rem = c0.b_rem;
        M.b_lru = c0.b_lru;
        M.best_is_full = c0.best_is_full;
        M.b_rpm = c0.b_rpm;
        M.best_idx = c0.best_idx;
        M.plain = c0.bestpos == c0.best0;
        M.w0 = w0;
        M.bestpos = c0.bestpos;
        M.best0 = c0.best0;
        M.e_rpm = c0.e_rpm;
        M.sbk = c0.sbk ? 1 : 0;
        M.end0 = caps[0].end;
        for (int v = 0; v < 2; v++) {
            const MemoCap &c = caps[v];
            M.v[v].valid = ok[v];
            M.v[v].lo = c.best0;
            M.v[v].hi = c.end == kNoPos ? S.P : c.end + 1;  // the instance that ends the list is part of what the answer depends on
            M.v[v].ccount = c.ccount;
            M.v[v].hsum = c.hsum;
            M.v[v].hash = (uint32_t)(c.hsum ^ (c.hsum >> 32));
        }
        static_cast<TypeMemoHead &>(memo[t]) = M;
    }
    __shared__ int32_t near_c[kMemoNear];  // the row's own copy of the list's head (TypeMemo::cand64 / ::rk64), gathered here first
    __shared__ int32_t near_r[kMemoNear];
    near_c[lane] = 0;
    near_r[lane] = kRkNone;
    wave_sync();
    // list 0: the candidates' pod indices in shortlist order, and what every position of the window is to the list
    int32_t *out = cand + (size_t)t * kMemoCand;
    int16_t *rk = rkt + (size_t)t * kMemoCand;
    int running = 0;
    for (int j = 0; j < kWinWords; j++) {
        const int w = w0 + j, pos = w * 64 + lane;
        // the walk parked the clipped candidate words here (best bit set)
        const uint64_t word = (ok[0] && w >= caps[0].wlo && w <= caps[0].whi) ? scr[j * kScr] : 0ull;
        const bool bit = (word >> lane) & 1ull;
        const int k = running + __popcll((unsigned long long)(word & ((1ull << lane) - 1ull)));
        int v = kRkNone;
        if (bit) {
            v = k;  // (the best instance is the list's first bit: k == 0)
            const int32_t pod = S.orig[pos];
            out[k] = pod;
            if (k < kMemoNear) near_c[k] = pod;
        }
        if (any && pos == c0.bestpos) v = 0;
        if (any && pos == c0.best0 && c0.best0 != c0.bestpos) v = kRkFirst;
        rk[j * 64 + lane] = (int16_t)v;
        if (any && pos >= c0.best0 && pos - c0.best0 < kMemoNear) near_r[pos - c0.best0] = v;
        running += __popcll((unsigned long long)word);
    }
    wave_sync();
    memo[t].cand64[lane] = near_c[lane];
    memo[t].rk64[lane] = (int16_t)near_r[lane];
}
__global__ __launch_bounds__(64) void build_sel_memo_kernel(Snap S, int32_t *__restrict__ sel, int32_t *__restrict__ rk, const TypeWin *__restrict__ wins,
                                                            TypeMemo *__restrict__ memo, int32_t *__restrict__ cand, int16_t *__restrict__ memo_rk,
                                                            LongMemo *__restrict__ lmemo)
{
    const int n_sel = sel ? 2 * S.T * S.W : 0;  // (sel == null: the inverse tables are not built for this snapshot, kSelMaxBytes)
    const int n_memo = S.T < kWinLds ? S.T : kWinLds;
    if ((int)blockIdx.x < n_sel)
        build_sel_body((int)blockIdx.x, S, sel, rk);
    else if ((int)blockIdx.x < n_sel + n_memo)
        build_memo_body((int)blockIdx.x - n_sel, S, wins, memo, cand, memo_rk);
    else
        build_long_memo_body((int)blockIdx.x - n_sel - n_memo, S, lmemo);  // (launched with T more blocks when lmemo != null)
}

// The window kernels with the recorded shortlists in front (place_block<..., MEMO, NOBAR>; see TypeMemo): a wavefront whose 64 requests
// are all covered is done after ~150 instructions and no barrier.  One caller per batch (place_batch_c_m_kernel) means one position of
// "self" for the whole batch, so a request leaves the shortlist only through its model's instances or its own exclusions (C3: 0.65 % for a
// caller with room, two wavefronts in three skip the lane phase; a FULL caller's fresh-row test fires, its shortlists are the best
// instance alone and nearly every wavefront is covered).  Request rows (place_batch_m_kernel) bring a caller per request — 1.3 % leave,
// every second wavefront runs the lane phase behind the check.  Measured, C3, per launch on one stream / on four (tools/r5/memo_sweep.py,
// profiles/r5/shortlist_experiments): 800k of one caller with room 19.9 -> 17.7 us / 12.6 -> 11.1 us (72 G decisions/s), of a full caller
// 20.0 -> 15.3 us / 12.7 -> 9.3 us (86 G decisions/s); 800k rows 25.9 -> 24.7 us / 17.7 -> 16.7 us; 1.6 M rows 43.4 -> 41.2 / 35.2 -> 32.6; 400k
// rows 15.8 -> 13.55 / 8.8 -> 8.4; at 200k the two meet and below the check is the longer chain — hence kMemoFrom.  7 wavefronts per SIMD
// for both (rows at 6: 25.0 / 17.7 us).
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_C_WAVES, MMP_C_WAVES))) void place_batch_c_m_kernel(Snap S, PlaceArgs A, int32_t wpad,
                                                                                                              mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<false, kReqC, true, true>(S, A, wpad, smem, nullptr, C);
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void place_batch_m_kernel(Snap S, PlaceArgs A, int32_t wpad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<false, kReq64, true, true>(S, A, wpad, smem);
}
// decisions from which a batch takes the kernel with the shortlists in front (measured, C3, one stream / four streams, per launch:
// rows 200k 10.2 -> 10.1 / 5.8 -> 5.95 us, 400k 15.8 -> 13.55 / 8.8 -> 8.4; one caller 200k 9.15 -> 8.4 / 4.96 -> 5.04 (a full caller), 400k
// 12.7 -> 11.35 / 6.9 -> 6.7 (a caller with room); below, the check is the longer chain — a 100k launch of rows: 7.75 -> 8.6 us)
constexpr int kMemoFrom = 4 * 1024 * 64;      // request rows
constexpr int kMemoFromC = 3 * 1024 * 64;     // the single-caller form
// decisions from which a batch is split into place_memo_kernel + place_tail_kernel.  Measured, C3, round-robin on four streams with a
// hardware queue each, per call (tools/r6/split_sweep.py, profiles/r6/split_sweeps.txt): request rows 300k 5.7 -> 7.3 us, 400k 7.3 -> 7.2,
// 800k 15.8 -> 12.7 (63 G decisions/s); one caller 400k 5.9 -> 7.1, 800k 11.1 -> 8.1 (98 G decisions/s); C4, 1M rows 27.5 -> 15.7 (the one-launch
// kernels are bound by their LDS there).  A pair of launches has a floor of ~7 us; on ONE stream the tail is not hidden (800k rows: 26.2 -> 27.9 us).
constexpr int kSplitFrom = 6 * 1024 * 64;
constexpr int kSplitFromC = 8 * 1024 * 64;

// ---- the split form (round 6): a launch that does nothing but the shortlist check, and a dense tail --------------------------------
// place_batch_m_kernel carries the check AND the ordinary path — ~150 instructions in front of ~1000, one register allocation (72 VGPRs
// + scratch), 25 KB of LDS per workgroup — so the requests the records cover pay for the ones they do not.  Here the two are two
// launches.  place_memo_kernel: a lane per request, memo_try and nothing else — the records in LDS (a copy per wavefront), no barrier,
// no scratch, 8 wavefronts per SIMD.  A wavefront with requests the check did not decide (one in sixteen, C3) reserves room for them
// in one of kRestLists lists — wavefront w in list w mod 64: ONE returning atomic on a counter that ~1/64 of those wavefronts share
// (a launch-wide list behind ONE counter was 16 ns per atomic in sequence, profiles/r5/shortlist_experiments v3) — and leaves their
// indices there.  place_tail_kernel, queued right behind it on the same stream (so it runs beside the next batch's first launch, given a
// second stream): workgroup g takes lists g, g + G, ..., reads their counts and entries — two dependent loads, no scan (summing a word
// per first-launch wavefront took a tail workgroup 8 us) — decides them with the ordinary place_block, windows, lists, general path
// and all (place_block<..., LIST>), and zeroes its counters for the stream's next batch.  With the check answering 99.8 % of a batch
// of request rows (C3) the tail is a thousand requests; a batch the records do not fit (every request with a position inside its list)
// still comes out right — the tail then loops — and the host stops splitting when a tail reports more than 1/32 of its batch
// (mmp_ctx::split_off).
constexpr int kRestLists = 64;
constexpr int kRestCntStride = 32;  // ints between two counters: a 128-byte line each
// ints of the stream's buffer for a first launch of n_words wavefronts: the counters, then kRestLists lists of rest_list_cap entries
__host__ __device__ constexpr int rest_list_cap(int n_words) { return ((n_words + kRestLists - 1) / kRestLists) * 64; }
__host__ __device__ constexpr size_t rest_buffer_ints(int n_words) { return (size_t)kRestLists * kRestCntStride + (size_t)kRestLists * rest_list_cap(n_words); }

// the undecided requests of this wavefront (`todo` lanes): room in its list — ONE returning atomic — then their indices
__device__ __forceinline__ void rest_append(int32_t *__restrict__ rest, int32_t cap, int d, bool todo)
{
    const uint64_t mk = __ballot(todo);
    if (mk && rest) {  // (wave-uniform; rest == null: diagnostics, nobody will decide them)
        const int l = __builtin_amdgcn_readfirstlane(d >> 6) & (kRestLists - 1);
        const int first = __ffsll((unsigned long long)mk) - 1;
        int base = 0;
        if (lane_id() == first)
            base = __hip_atomic_fetch_add(rest + l * kRestCntStride, __popcll((unsigned long long)mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = readlane_i32(base, first);
        if (todo) rest[kRestLists * kRestCntStride + (size_t)l * cap + base + __popcll((unsigned long long)(mk & ((1ull << lane_id()) - 1ull)))] = d;
    }
}

template <int FORM>
__device__ __forceinline__ void place_memo_body(const Snap &S, const PlaceArgs &A, int32_t *__restrict__ rest, int32_t cap, const mmp_place_caller &C,
                                                unsigned char *smem)
{
    // The types' records (TypeMemo rows: 512 bytes a type) into LDS, a copy per WAVEFRONT — global -> LDS directly, issued before the
    // request fetch and landed when that has (loads return in order), no barrier: the check then reads its type's record, the head of
    // its candidate list and what its own positions are to the list from LDS instead of as three more dependent levels of (L1-resident)
    // global gathers — a record is 80 bytes per lane, more than the request itself.
    const int stage = memo_stage_bytes(S.T);
    unsigned char *mine = smem + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * stage;
    {
        const char *src = reinterpret_cast<const char *>(S.memo);
        for (int c = 0; c < stage; c += 1024)
            __builtin_amdgcn_global_load_lds(src + c + lane_id() * 16, (__attribute__((address_space(3))) void *)(mine + c), 16, 0, 0);
    }
    const int d = blockIdx.x * kPlaceBlock + threadIdx.x;
    bool live = d < A.n;
    mmp_place_req rq{};
    if (live) rq = fetch_req<FORM>(A, C, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (A.extra_bound != 0) {  // (wave-uniform) bounded calls: as place_block answers them
        if (live && bad_extra_range(A, rq)) {
            live = false;
            mmp_place_out bo;
            bo.chosen = MMP_NONE;
            bo.best = MMP_BAD_REQUEST;
            bo.n_candidates = 0;
            bo.hash = 0;
            A.outs[d] = bo;
        }
    }
    bool todo = false;
#ifdef MMP_XP_STREAM  // (experiment builds only: the batch streamed in and out, nothing decided — the floor of a kernel of this form)
    if (live) {
        mmp_place_out o;
        o.chosen = rq.model ^ rq.self_pod;
        o.best = (int32_t)(rq.pick ^ rq.flags) + rq.n_extra + rq.extra_off;
        o.n_candidates = (int32_t)(rq.last_used ^ rq.fresh_lru);
        o.hash = (uint32_t)(rq.fresh_capacity ^ rq.fresh_used) + (uint32_t)rq.fresh_count + (uint32_t)rq.fresh_rpm;
        A.outs[d] = o;
    }
#else
    if (live) todo = !memo_try<FORM>(S, A, rq, d, reinterpret_cast<const TypeMemo *>(mine));
#endif
    rest_append(rest, cap, d, todo);
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void place_memo_kernel(Snap S, PlaceArgs A, int32_t *__restrict__ rest, int32_t cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_memo_body<kReq64>(S, A, rest, cap, mmp_place_caller{}, smem);
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void place_memo_c_kernel(Snap S, PlaceArgs A, int32_t *__restrict__ rest, int32_t cap,
                                                                                                           mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_memo_body<kReqC>(S, A, rest, cap, C, smem);
}

// `report` (may be null): workgroup 0 leaves {undecided requests, n} there (pinned host memory: the host reads the pair some launches
// later, never waits for it)
template <int FORM, bool WITH_LONG = false>
__device__ __forceinline__ void place_tail_body(const Snap &S, const PlaceArgs &A, int32_t wpad, unsigned char *smem, int32_t *__restrict__ rest, int32_t cap,
                                                int32_t *report, const mmp_place_caller &C)
{
    __shared__ int32_t t_cnt[kRestLists + 1];  // this workgroup's lists: running totals
    // A tail wavefront is ONE long dependent chain on a SIMD it shares with up to eight wavefronts of the next batch's first launch:
    // it goes first whenever it can issue
    __builtin_amdgcn_s_setprio(3);
    PHASE_T0();
    const int tid = threadIdx.x;
    const int G = gridDim.x, g = blockIdx.x;
    if (g >= kRestLists) return;
    const int mine = (kRestLists - g + G - 1) / G;  // lists g, g + G, ...
    if (tid < 64) {
        const int c = tid < mine ? rest[(g + tid * G) * kRestCntStride] : 0;
        const int incl = wave_incl_scan_i32(c);
        t_cnt[tid + 1] = incl;
        if (tid == 0) t_cnt[0] = 0;
        if (report && g == 0) {  // (every list's count: read here, at the very start — the other workgroups zero theirs at their end)
            const int all = wave_sum_i32(rest[tid * kRestCntStride]);
            if (tid == 0) {
                __hip_atomic_store(report, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(report + 1, A.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    const int total = t_cnt[mine];
    PHASE(13);  // (tail) the counts read
    for (int base = 0; base < total; base += kPlaceBlock) {  // (workgroup-uniform)
        int d = -1;
        const int i = base + tid;
        if (i < total) {
            int k = 0;
            while (t_cnt[k + 1] <= i) k++;  // (a handful of lists per workgroup)
            d = rest[kRestLists * kRestCntStride + (size_t)(g + k * G) * cap + (i - t_cnt[k])];
        }
        PHASE(14);  // (tail) this pass's entries read
        place_block<WITH_LONG, FORM, false, false, true>(S, A, wpad, smem, nullptr, C, d);
    }
    if (tid < mine) rest[(g + tid * G) * kRestCntStride] = 0;  // for the stream's next batch (ordered behind this launch)
}
// (5 wavefronts per SIMD = 96 VGPRs: a tail wavefront then fits on a SIMD that holds eight wavefronts of another stream's first launch —
// 8 x 48 + 96 <= 512 registers; with the 154 the compiler takes unasked, the tail's workgroups waited for a compute unit to drain)
#ifndef MMP_TAIL_EU
#define MMP_TAIL_EU 5
#endif
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_TAIL_EU, MMP_TAIL_EU))) void place_tail_kernel(Snap S, PlaceArgs A, int32_t wpad, int32_t *__restrict__ rest, int32_t cap, int32_t *report)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_tail_body<kReq64>(S, A, wpad, smem, rest, cap, report, mmp_place_caller{});
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_TAIL_EU, MMP_TAIL_EU))) void place_tail_c_kernel(Snap S, PlaceArgs A, int32_t wpad, int32_t *__restrict__ rest, int32_t cap, int32_t *report,
                                                                   mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_tail_body<kReqC>(S, A, wpad, smem, rest, cap, report, C);
}

// ---- the split form on a full cluster (round 6): the recorded long walks in a launch of their own ------------------------------------
// place_batch_long_kernel carries the record check (long_memo_try) AND the walk it replaces: 139 VGPRs, three wavefronts per SIMD, for
// requests of which one in a hundred thousand needs the walk.  From kLongSplitFrom requests on the check runs alone — request ->
// {registry row with the model's positions, caller position, late exclusions} -> record -> rk / amul lookups -> sel -> orig — and leaves
// what it cannot answer in the stream's lists for place_long_tail_kernel (place_block<WITH_LONG, ..., LIST>: the walk, case (b), the
// general path), exactly as place_memo_kernel / place_tail_kernel do for the head windows.
template <int FORM>
__device__ __forceinline__ void place_long_memo_body(const Snap &S, const PlaceArgs &A, int32_t *__restrict__ rest, int32_t cap, const mmp_place_caller &C)
{
    const int d = blockIdx.x * kPlaceBlock + threadIdx.x;
    bool live = d < A.n;
    mmp_place_req rq{};
    if (live) rq = fetch_req<FORM>(A, C, d);
    if (A.extra_bound != 0) {  // (wave-uniform) bounded calls: as place_block answers them
        if (live && bad_extra_range(A, rq)) {
            live = false;
            mmp_place_out bo;
            bo.chosen = MMP_NONE;
            bo.best = MMP_BAD_REQUEST;
            bo.n_candidates = 0;
            bo.hash = 0;
            A.outs[d] = bo;
        }
    }
    bool todo = false;
    if (live) {
        ResolvedReq r = resolve_req<false, true>(S, A, rq);
        merge_late_extras(r);
        mmp_place_out o;
        if (long_memo_try(S, A, r, o))
            A.outs[d] = o;
        else
            todo = true;
    }
    rest_append(rest, cap, d, todo);
}
#ifndef MMP_LONG_MEMO_EU
#define MMP_LONG_MEMO_EU 6
#endif
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_LONG_MEMO_EU, MMP_LONG_MEMO_EU))) void place_long_memo_kernel(Snap S, PlaceArgs A, int32_t *__restrict__ rest, int32_t cap)
{
    place_long_memo_body<kReq64>(S, A, rest, cap, mmp_place_caller{});
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(MMP_LONG_MEMO_EU, MMP_LONG_MEMO_EU))) void place_long_memo_c_kernel(Snap S, PlaceArgs A, int32_t *__restrict__ rest, int32_t cap,
                                                                                                                          mmp_place_caller C)
{
    place_long_memo_body<kReqC>(S, A, rest, cap, C);
}
__global__ __launch_bounds__(kPlaceBlock) void place_long_tail_kernel(Snap S, PlaceArgs A, int32_t wpad, int32_t *__restrict__ rest, int32_t cap, int32_t *report)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_tail_body<kReq64, true>(S, A, wpad, smem, rest, cap, report, mmp_place_caller{});
}
__global__ __launch_bounds__(kPlaceBlock) void place_long_tail_c_kernel(Snap S, PlaceArgs A, int32_t wpad, int32_t *__restrict__ rest, int32_t cap, int32_t *report, mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_tail_body<kReqC, true>(S, A, wpad, smem, rest, cap, report, C);
}
// Requests from which a full-cluster batch is split: NEVER by default.  Measured (tools/r6/long_split_sweep.sh, profiles/r6/long_records.txt,
// C3 full cluster, 800k requests): the first launch alone 33.1 us against 40.3 us for the one-launch kernel, but its tail — the walk,
// 194 VGPRs — is 14.8 us: 47.9 us per call on one stream, 30.2 us on four (one launch: 30.1).  MMP_LONG_SPLIT_FROM=n switches it on.
constexpr int kLongSplitFrom = INT32_MAX;

__global__ __launch_bounds__(kPlaceBlock) void place_batch_long_c_kernel(Snap S, PlaceArgs A, int32_t wpad, mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<true, kReqC, false, true>(S, A, wpad, smem, nullptr, C);
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void place_batch_long4_c_kernel(Snap S, PlaceArgs A, int32_t wpad,
                                                                                                                  mmp_place_caller C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<true, kReqC>(S, A, wpad, smem, nullptr, C);
}

// The latency path's launches of more than one workgroup: done_blocks = device counter of finished
// workgroups (left at zero); the last one to finish announces completion (PlaceArgs::done_flag).  A kernel of
// its own so that the throughput kernel's argument block stays as small as it was (measured: the launch
// path of this runtime is sensitive to it).
// (WITH_LONG: on the latency path occupancy is nothing — a handful of workgroups — and a shortlist that spans the table must not
// fall to the wave path: one wavefront sweeping thousands of candidates took 40-60 us per single decision on a churned C3 fleet,
// the whole of the 49 us p99 under churn of round 3; through the prefix tables it is a few us.)
__global__ __launch_bounds__(kPlaceBlock) void place_batch_flag_kernel(Snap S, PlaceArgs A, int32_t wpad, uint32_t *done_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    place_block<true>(S, A, wpad, smem, done_blocks);
}

// One decision whose request rides in the kernel arguments (the latency path's n = 1 call without extra
// exclusions): the kernel does not have to fetch the request from pinned host memory over the fabric.
__global__ __launch_bounds__(kPlaceBlock) void place_single_kernel(Snap S, PlaceArgs A, int32_t wpad, mmp_place_req rq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ mmp_place_req srq;
    if (threadIdx.x == 0) srq = rq;
    __syncthreads();
    A.reqs = &srq;
    A.n = 1;
    place_block<true>(S, A, wpad, smem);
}


// ---- the resident decision kernel (single requests) ----------------------------------------------------------
// mmp_place_batch(n = 1) as a kernel launch costs a launch: 6.5 us before the first instruction of the decision
// (tools/micro/doorbell.hip: launch + completion flag round trip), and one launch per request per thread.  This
// kernel stays resident instead — ONE wavefront, lane l serving ring slot l: the host writes a request into pinned
// memory and then its tag; the lane that polls the slot sees the tag, decides (the same lane_decide_win /
// lane_decide_r as the batch kernel, windows staged in LDS once), writes the result row and then the tag back.  64
// request threads are served concurrently, none of them launches anything.
//   * The slot is read in ONE sweep (request + tag, 80 bytes) and the decision starts on it at once; a second read of the
//     request, issued after the sweep that showed the new tag has returned, is compared before the result is published:
//     PCIe may serve the four pieces of a sweep in any order, so a sweep can pair the new tag with stale request bytes —
//     the second read cannot (the host wrote the tag last).  Equal: publish.  Different: decide again on the second read.
//   * The kernel holds ONE snapshot (its launch arguments).  Whoever publishes another or rewrites what it reads raises
//     `stop` and waits for its stream (quiesce_decisions); the next request launches a new one.  It also leaves by
//     itself after idle_ticks without a request (so that a host that died leaves nothing behind) and reports `exited`.
//   * Requests it cannot decide alone (more than kLateExtra + 6 exclusions never come here; the wave path: case (b), the
//     replay list, the replica-set retry ...) are answered with status kResidentPunt and the host takes the launch path.
constexpr int kResidentSlots = 64;
constexpr uint32_t kResidentPunt = 0x80000000u;  // or-ed into the tag written back: "decide this one with a launch"
constexpr int kResidentNowBits = 44;             // now_ms travels in the low bits of the bell (2^44 ms = 557 years)
// 16 bytes of pinned host memory in one system-scope load (the builtin atomics stop at 8)
__device__ __forceinline__ void load16_sys(const void *p, uint64_t &lo, uint64_t &hi)
{
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    lo = ((uint64_t)v.y << 32) | v.x;
    hi = ((uint64_t)v.w << 32) | v.z;
}
struct __attribute__((aligned(16))) ResidentSlot {  // pinned host memory, device-mapped
    mmp_place_req req;   // 64 B, written first
    uint64_t bell;       // written last (release): sequence number << 44 | now_ms & (2^44 - 1) — ONE word, so that the lane
                         // gets the request's clock with the tag; it answers bells whose sequence it has not answered yet
    uint64_t pad[7];     // 128 B per slot; the device never writes here
};
static_assert(sizeof(ResidentSlot) == 128, "ResidentSlot is 128 bytes");
// The answers live in lines of their own (the device never writes into a line a lane polls).
struct __attribute__((aligned(64))) ResidentAnswer {  // pinned host memory, device-mapped; the host never writes here
    mmp_place_out out;   // written by the device ...
    uint32_t done;       // ... then the sequence number (| kResidentPunt) it answers
    uint32_t pad[11];
};
static_assert(sizeof(ResidentAnswer) == 64, "ResidentAnswer is 64 bytes");
struct ResidentCtl {  // pinned host memory
    uint32_t stop;    // host -> device: the generation that has to leave
    uint32_t exited;  // device -> host: the generation that has left (a late store of an earlier one cannot pass for the current)
    uint32_t pad[14];
};

__global__ __launch_bounds__(64) void place_resident_kernel(Snap S, PlaceArgs A, ResidentSlot *slots, ResidentAnswer *answers,
                                                             ResidentCtl *ctl, long long idle_ticks, uint32_t generation)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TypeWin *s_wins = reinterpret_cast<TypeWin *>(smem);
    uint64_t *s_scr = reinterpret_cast<uint64_t *>(smem + kWinLdsBytes);  // column stride: kPlaceBlock lanes (the first 64 used)
    const int lane = lane_id();
    const bool use_wins = A.wins != nullptr;
    if (use_wins) {
        const int chunks = ((S.T < kWinLds ? S.T : kWinLds) * (int)sizeof(TypeWin) + 1023) >> 10;
        const char *src = reinterpret_cast<const char *>(A.wins);
        char *dst = reinterpret_cast<char *>(s_wins);
        for (int c = 0; c < chunks; c++)
            __builtin_amdgcn_global_load_lds(src + (size_t)c * 1024 + lane * 16,
                                             (__attribute__((address_space(3))) void *)(dst + c * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wave_sync();
    ResidentSlot *slot = &slots[lane];
    ResidentAnswer *ans = &answers[lane];
    uint32_t seen = __hip_atomic_load(&ans->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~kResidentPunt;
    long long last = wall_clock64();
    A.n = 1;
    const uint64_t *q = reinterpret_cast<const uint64_t *>(&slot->req);
    for (;;) {
        // system-scope loads: they go to the host's memory every time (a plain or non-temporal load of pinned memory may be
        // served from the device's caches: the wavefront then polls a stale copy forever — observed).  One sweep = the request
        // and the bell, nine 8-byte reads in flight.
        uint64_t w[8], bell;
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bell = __hip_atomic_load(&slot->bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the stop word rides in EVERY sweep (one more read in flight, the same address for all lanes): with request threads
        // ringing back to back no sweep is ever empty, and a commit that waits for this kernel holds the state lock
        const bool leave = __ballot(__hip_atomic_load(&ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == generation) != 0;  // wave-uniform
        const uint32_t tag = (uint32_t)(bell >> kResidentNowBits);
        const bool fresh = tag != seen;
        if (__ballot(fresh)) {
            if (fresh) {
                mmp_place_req rq;
                __builtin_memcpy(&rq, w, sizeof rq);
                // the confirming read: issued now, i.e. after the sweep that showed the new bell has returned; it is consumed
                // only when the decision is ready, so the two overlap
                uint64_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                mmp_place_out o;
                int code = kLaneWave;
                PlaceArgs Al = A;  // this lane's request carries its own now_ms (the bell word: read atomically with the tag)
                Al.now = (int64_t)(bell & ((1ull << kResidentNowBits) - 1ull));
                for (int pass = 0; pass < 2; pass++) {
                    if (rq.n_extra == 0) {  // (exclusions of the request itself ride the launch path: the pool is not mapped here)
                        ResolvedReq r = resolve_req<false, false>(S, Al, rq);
                        code = kLaneHeadMiss;
                        if (use_wins) code = lane_decide_win(S, Al, r, s_wins, s_scr + lane, o);
                        if (code == kLaneHeadMiss) code = lane_decide_r<false>(S, Al, r, o);
                    }
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < 8; k++) same = same && v[k] == w[k];
                    if (same) break;
#pragma unroll
                    for (int k = 0; k < 8; k++) w[k] = v[k];  // the sweep was torn: the confirming read is the request
                    __builtin_memcpy(&rq, w, sizeof rq);
                }
                uint32_t answer = tag;
                if (code == kLaneDone) {
                    uint64_t ov[2];
                    __builtin_memcpy(ov, &o, sizeof ov);
                    uint64_t *op = reinterpret_cast<uint64_t *>(&ans->out);
                    __hip_atomic_store(op, ov[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(op + 1, ov[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else
                    answer |= kResidentPunt;
                __hip_atomic_store(&ans->done, answer, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                seen = tag;
            }
            last = wall_clock64();
            if (leave) break;  // this sweep's requests are answered; whoever rings now sees `exited` and relaunches
        } else {
            // nobody rang: leave when told to, or when idle for long
            if (leave || wall_clock64() - last > idle_ticks) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    // a request that arrived between the last sweep and here stays unanswered: the host sees `exited` and relaunches
    if (lane == 0) __hip_atomic_store(&ctl->exited, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace mmp
