/*
 * mmplace.h — C ABI of libmmplace: the MI355X (gfx950) placement / eviction
 * solver that replaces the bodies of ModelMesh's instance-selection hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, never
 * throws and never aborts the process.  Return value: 0 (MMP_OK) or a negative
 * MMP_E* code; text via mmp_last_error().  There is NO CPU fallback: if no HIP
 * device is usable mmp_create() fails with MMP_ENODEVICE.
 *
 * Each function cites the reference interface it replaces.  "MM.java" =
 * src/main/java/com/ibm/watson/modelmesh/ModelMesh.java of kserve/modelmesh.
 * The JNI / Java binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Pod and model indices are dense ints chosen by the caller (the Java side
 * interns instance ids; `id_order` carries String.compareTo order).
 */
#ifndef MMPLACE_H
#define MMPLACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMP_ABI_VERSION 3 /* 3: latency-based rebalancers (mmp_*_plan_conc), the single-caller request form, bounded device-pointer calls */

/* return codes */
#define MMP_OK 0
#define MMP_EINVAL (-1)     /* bad argument                                    */
#define MMP_ENODEVICE (-2)  /* no usable HIP device (never falls back to CPU)  */
#define MMP_EHIP (-3)       /* a HIP runtime call failed                       */
#define MMP_EORDER (-4)     /* PLACEMENT_ORDER is not a total order on rows    */
#define MMP_ESTATE (-5)     /* call sequence error (e.g. place before commit)  */
#define MMP_ENOMEM (-6)

/* `chosen` conventions — LoadBalancer.getNext return values:
 *   null                         -> MMP_NONE   (MM.java:4796,4803,4872,4942)
 *   LoadBalancer.ABORT_REQUEST   -> MMP_SELF   (MM.java:4384,4894,4932,4990)
 *   a ServiceInstanceInfo        -> its pod index */
#define MMP_NONE (-1)
#define MMP_SELF (-2)

typedef struct mmp_ctx mmp_ctx;

/* Solver parameters that are inputs, not constants. */
typedef struct {
    int32_t device;           /* HIP device ordinal                             */
    int32_t reserved0;
    int64_t min_space_units;  /* MM.java:765-771 (see mmp_min_space_units)      */
    int64_t min_churn_age_ms; /* MM.java:697                                    */
} mmp_config;

/* One InstanceRecord (InstanceRecord.java:37-69) — 64 bytes.
 * flags: bit0 shuttingDown (row is treated as deleted, MM.java:1462-1464),
 *        bit1 live = present in the litelinks instance map (MM.java:4765),
 *        bit2 tombstone (slot unused; keeps indices stable across removals). */
#define MMP_POD_SHUTTING_DOWN 1u
#define MMP_POD_LIVE 2u
#define MMP_POD_TOMBSTONE 4u
typedef struct {
    int64_t lru_time; /* Long.MAX_VALUE when empty */
    int64_t capacity; /* 8 KiB units, ModelLoader.java:37 */
    int64_t used;
    int64_t version;  /* instanceVersion */
    int32_t count;
    int32_t loading_threads;
    int32_t loading_in_progress;
    int32_t rpm;
    uint32_t id_order;   /* dense rank of the id under String.compareTo  */
    int32_t replica_set; /* interned id.substring(0,6); -1 if |id| < 7   */
    uint32_t flags;
    uint32_t reserved;
} mmp_pod_row;

/* One ModelRecord (ModelRecord.java:61-114) — 24 bytes + 12 bytes per entry.
 * Entries [ent_off, ent_off+n_loaded) are instanceIds in TreeMap (id) order,
 * followed by n_failed loadFailedInstanceIds. */
typedef struct {
    int32_t type;     /* interned model type; ignored when no type table */
    int32_t ent_off;
    int32_t n_loaded;
    int32_t n_failed;
    int64_t last_used;
} mmp_model_row;

/* One CacheMissForwardingLB.getNext call (MM.java:4776) — 64 bytes.
 * flags bit0 = exclude.favourSelf (MM.java:4781).  The fresh_* fields are the
 * caller's getFreshInstanceRecord() (MM.java:5369-5386; its rpm is 0 there). */
#define MMP_REQ_FAVOUR_SELF 1u
typedef struct {
    int32_t model;      /* row in the model table                            */
    int32_t self_pod;   /* caller's pod index, -1 if not in the table        */
    uint32_t flags;
    uint32_t pick;      /* replaces ThreadLocalRandom: index=(pick*n)>>32    */
    int64_t last_used;  /* exclude.lastUsedTime (MM.java:4736,4951)          */
    int32_t extra_off;  /* tried-this-request ∪ explicit excludes (pool idx) */
    int32_t n_extra;
    int64_t fresh_lru;
    int64_t fresh_capacity;
    int64_t fresh_used;
    int32_t fresh_count;
    int32_t fresh_rpm;
} mmp_place_req;

/* The single-caller form of a batch of load-target decisions.  `self` and getFreshInstanceRecord() (MM.java:5369-5386) belong
 * to the CALLING INSTANCE, not to the request, and every batch the reference itself produces is issued by one instance: the rate
 * task (:5636), the janitor (:6110), the reaper (:6616), preShutdown (:6959), an instance's own request threads.  The caller's
 * side travels once per call (mmp_place_caller, 40 bytes), a decision is 24 bytes instead of 64: mmp_place_batch_c.  The
 * decision is the one of mmp_place_req {model, caller.self_pod, caller.flags, pick, last_used, extra_off, n_extra, caller.fresh_*}. */
typedef struct {
    int32_t self_pod; /* the calling instance, -1 if not in the table */
    uint32_t flags;   /* MMP_REQ_FAVOUR_SELF                          */
    int64_t fresh_lru;
    int64_t fresh_capacity;
    int64_t fresh_used;
    int32_t fresh_count;
    int32_t fresh_rpm;
} mmp_place_caller;
typedef struct {
    int32_t model;
    uint32_t pick;
    int64_t last_used;
    int32_t extra_off;
    int32_t n_extra;
} mmp_place_req_c;

/* mmp_place_out::best of a decision whose request the library refused to follow (bounded device-pointer calls: an exclusion
 * range outside the pool the caller declared); its chosen is MMP_NONE, n_candidates and hash are 0. */
#define MMP_BAD_REQUEST (-3)

/* 16 bytes per decision. On early returns (no eligible pod, or the immediate
 * "choose self" returns at MM.java:4872,4894,4932) n_candidates = hash = 0. */
typedef struct {
    int32_t chosen;       /* pod index | MMP_NONE | MMP_SELF                 */
    int32_t best;         /* final bestIid's pod index, -1 if none           */
    int32_t n_candidates; /* candidates.size() before the rpm filter         */
    uint32_t hash;        /* shortlist bitmap hash (audit)      */
} mmp_place_out;

/* One ForwardingLB.getNext call (MM.java:4315) — cache-hit routing, 48 bytes.
 * The reference reads litelinks' per-instance counters only for the instances that hold a copy of the model
 * (si.getInUseCount() / si.getLastUsedTime(), MM.java:4356,4360, inside the loop over filteredInstances): the
 * request brings exactly those — [cnt_off, cnt_off + n_cnt) of the call's mmp_serve_counter pool, one entry per
 * copy whose instance litelinks lists (siMap.get(iid) != null, :4343).  A copy WITHOUT an entry is a copy litelinks
 * does not list: it is skipped as at :4344-4347.  Nothing the size of the instance table crosses the boundary. */
#define MMP_SERVE_EXCLUDE_SELF 1u
#define MMP_SERVE_PREFER_SELF 2u
typedef struct {
    int32_t pod;       /* the instance (pod index)                         */
    int32_t in_use;    /* ServiceInstance.getInUseCount(), MM.java:4356    */
    int64_t last_used; /* ServiceInstance.getLastUsedTime(), MM.java:4360  */
} mmp_serve_counter;
typedef struct {
    int32_t model;
    int32_t self_pod;
    uint32_t flags;
    int32_t local_in_flight;     /* localInvokesInFlight (MM.java:4303)         */
    int64_t last_invoke_time;    /* lastInvokeTime (MM.java:4304)               */
    int64_t assume_completed_ms; /* TimeStats.assumeCompletedAfterMillis        */
    int32_t excl_off;            /* (pod,loadStart) pairs already tried + keyExcludes */
    int32_t n_excl;
    int32_t cnt_off;             /* this request's counters in the pool               */
    int32_t n_cnt;
} mmp_serve_req;

typedef struct {
    int32_t chosen; /* pod index | MMP_NONE | MMP_SELF */
    int32_t pad;
    int64_t chosen_load_start;
} mmp_serve_out;

/* ClusterStats (MM.java:1570-1591, InstanceSetStatsTracker.java:53-92). */
typedef struct {
    int64_t total_capacity;
    int64_t total_free;
    int64_t global_lru; /* Long.MAX_VALUE if none */
    int32_t instance_count;
    int32_t model_copy_count;
} mmp_stats;

/* One eviction evaluation on one pod's cache (clhm/ConcurrentLinkedHashMap.java
 * :329-352,590-652; clhm/LinkedDeque.java:243-288): insert an entry of `weight`
 * stamped `last_used` (0 = now) into the time-ordered deque, then evict from
 * the head while weightedSize > capacity. */
typedef struct {
    int32_t cache;      /* which per-pod cache segment                        */
    int32_t weight;     /* weight of the incoming entry (or weight delta)     */
    int64_t last_used;  /* 0 means now (clhm :1357-1360)                      */
} mmp_evict_req;

typedef struct {
    int32_t insert_pos;    /* deque position the new node is linked at         */
    int32_t n_victims;     /* nodes polled from the head                       */
    int32_t self_evicted;  /* 1 if the new node itself was among the victims   */
    int32_t pad;
    int64_t weighted_size; /* after eviction                                   */
    int64_t oldest_time;   /* oldestTime() after eviction, -1 if empty         */
} mmp_evict_out;

/* Stateful per-pod caches (rows a12 + a13): the clhm operations and the ModelCacheUnloadBufManager
 * methods, replayed in caller order per cache by mmp_cache_replay.  Keys are interned model ids;
 * MMP_UNLOADBUF_KEY is the manager's pinned "___UNLOADBUF" pseudo-entry
 * (ModelCacheUnloadBufManager.java:42,88), which a managed cache must contain when it is loaded. */
#define MMP_UNLOADBUF_KEY (-1000000)
#define MMP_COP_PUT_IF_ABSENT 0           /* clhm putIfAbsent(key, w=arg, time)  :804-834; result 1 = inserted      */
#define MMP_COP_GET 1                     /* clhm get(key, time)                 :726-733; result = found           */
#define MMP_COP_UPDATE_WEIGHT 2           /* replace / replaceQuietly, new weight = arg, time -1 = quiet  :902-985   */
#define MMP_COP_REMOVE 3                  /* clhm remove(key)                    :861-870                           */
#define MMP_COP_UBM_INSERT_NEW_ENTRY 4    /* insertNewEntry(key, w=arg, time)    ModelCacheUnloadBufManager :130-145 */
#define MMP_COP_UBM_ADJUST_SPACE_REQUEST 5 /* adjustNewEntrySpaceRequest(increase=arg, key)              :152-166   */
#define MMP_COP_UBM_SPACE_IS_READY 6      /* cacheSpaceIsReady(required=arg)     :395-402 (read only)               */
#define MMP_COP_UBM_CLAIM_SPACE 7         /* claimRequestedSpaceIfReady(required=arg)                    :190-202   */
#define MMP_COP_UBM_ADJUST_AFTER_LOAD 8   /* adjustWeightAfterLoad(delta=arg, key)                       :224-246   */
#define MMP_COP_UBM_UNLOAD_COMPLETE 9     /* unloadComplete(weight=arg, success=flag)                    :318-338   */
#define MMP_COP_UBM_REMOVE_ENTRY 10       /* removeEntry(key) + entryRemoved; result = weight or -1      :281-316   */
#define MMP_COP_UBM_DISCARD_FAILED 11     /* discardFailedEntry(weight=arg)                              :343-349   */
#define MMP_COP_UBM_INSERT_FAILED_PLACEHOLDER 12 /* insertFailedPlaceholderEntry(key, w=arg, time)       :250-274   */
typedef struct {
    int32_t cache;
    int32_t op;
    int32_t key;
    int32_t arg;
    int64_t time; /* lastUsed: 0 = now (clhm :1357-1360) */
    int32_t flag;
    int32_t reserved;
} mmp_cache_op;

typedef struct {
    int32_t result;
    int32_t n_evicted;     /* entries this operation evicted (listener order)            */
    int32_t evicted_off;   /* their keys start here in the call's evicted_keys array      */
    int32_t buffer_weight; /* getUnloadBufferWeight() after the operation (0: unmanaged)  */
    int64_t weighted_size; /* runtimeCache.weightedSize() after the operation             */
    int64_t oldest_time;   /* runtimeCache.oldestTime(), -1 if empty                      */
} mmp_cache_op_out;

/* ModelCacheUnloadBufManager fields (:57-79); reserved < 0 = this cache has no manager. */
typedef struct {
    int32_t reserved;        /* unloadsReservedSizeUnits   */
    int32_t total_unloading; /* totalUnloadingWeight       */
    int64_t total_occupancy; /* totalModelCacheOccupancy   */
    int32_t cache_deficit;   /* cacheDeficit               */
    int32_t pad;
} mmp_ubm_state;

/* The request-level guards invokeModel evaluates around the two selections
 * (SURVEY.md §8 rows a10, a11, a14, a20), batched.  One struct carries the
 * scalar inputs of all of them; unused groups may be left zero. */
#define MMP_GATE_FAVOUR_SELF_FOR_HITS 1u /* favourSelfForHits, MM.java:3606             */
#define MMP_GATE_HAVE_CACHE_ENTRY 2u     /* getFromCache(...) != null, MM.java:3607-3612  */
#define MMP_GATE_ENTRY_DONE 4u           /* cacheEntry.isDone(), MM.java:3613             */
#define MMP_GATE_WE_CREATED_ENTRY 8u     /* weCreatedCacheEntry, MM.java:5186             */
#define MMP_GATE_ENTRY_FAILED 16u        /* ce.isFailed(), MM.java:2886                   */
#define MMP_GATE_HAVE_SIZE_HINT 32u      /* tas.known_size present, MM.java:5160          */
#define MMP_GATE_PUBLISH_FORCE 64u       /* publishInstanceRecord(force, ...), MM.java:5388 */
#define MMP_GATE_PRE_SHUTDOWN 128u
#define MMP_GATE_FRESH_SHUTTING_DOWN 256u
typedef struct {
    int32_t model;
    int32_t self_pod;
    uint32_t flags;
    int32_t excl_off, n_excl;         /* cache-hit (pod,loadStart) excludes filtering the copies */
    int32_t explicit_off, n_explicit; /* explicitExcludes / load-target filter members (pod idx)  */
    int32_t size_hint;                /* tas.known_size                                          */
    int64_t last_used_time;
    int64_t cache_capacity;           /* runtimeCache.capacity()                                 */
    int64_t cache_weighted_size;      /* runtimeCache.weightedSize()                             */
    int64_t cache_oldest_time;        /* runtimeCache.oldestTime(), -1 if empty                  */
    int32_t loader_predicted;         /* ce.loaderPredictedWeight()                              */
    int32_t loading_count;            /* loadingCount.get()                                      */
    int32_t weight_predict_cutoff;    /* loadingThreads + loadingThreads/3, MM.java:5013         */
    int32_t reserved;
    int64_t loaded_time;              /* registry load time of the evicted copy, <0 if absent    */
    int64_t load_timeout_ms;
    int64_t fresh_lru, fresh_capacity, fresh_used; /* getFreshInstanceRecord(), MM.java:5369; fresh_lru may be
                                                      runtimeCache.oldestTime() as it is: -1 (empty cache) is read as
                                                      Long.MAX_VALUE by the publish rule (:5423-5425)         */
    int32_t fresh_count, fresh_loading_threads, fresh_in_progress, fresh_rpm;
    int64_t last_published;           /* lastPublished, MM.java:5387                             */
} mmp_gate_req;

#define MMP_GATE_GO_LOCAL 1u            /* serve the hit locally, MM.java:3603-3626               */
#define MMP_GATE_FAILURES_BREACHED 2u   /* checkLoadFailureCount would throw, MM.java:4607-4627   */
#define MMP_GATE_LOCATIONS_BREACHED 4u  /* checkLoadLocationCount would throw, MM.java:4590-4604  */
#define MMP_GATE_LOCAL_NOT_ALLOWED 8u   /* throwIfLocalLoadNotAllowed would throw, MM.java:4003    */
#define MMP_GATE_CHURN_REJECT 16u       /* "Cache churn threshold exceeded", MM.java:3870-3884     */
#define MMP_GATE_EARLY_REJECT 32u       /* loadLocal aborts before inserting, MM.java:5185-5197    */
#define MMP_GATE_RELOAD_ELSEWHERE 64u   /* onEviction re-places the model, MM.java:2895,2919-2920  */
#define MMP_GATE_SHOULD_PUBLISH 128u    /* publishInstanceRecord writes an update, MM.java:5397-5468 */
typedef struct {
    uint32_t bits;
    int32_t initial_size; /* signed initialSize of loadLocal (negative = average-based), MM.java:5158-5179 */
} mmp_gate_out;

/* Leader "reaper" proactive-load plan. */
typedef struct {
    int32_t size_estimate; /* sizeEstimate, MM.java:6622-6629                           */
    int32_t free_count;    /* freeSpaceProactiveLoadCount, MM.java:6651                  */
    int32_t total_count;   /* totalProactiveLoadCount, MM.java:6655                      */
    int32_t n_candidates;  /* models passing the registry rule MM.java:6574-6577         */
    int32_t n_selected;    /* ensureLoadedInternal calls the Java would make             */
    int32_t error;         /* 1: sizeEstimate == 0 (the Java throws ArithmeticException) */
    int64_t space_to_fill; /* MM.java:6633-6650                                          */
    int64_t cutoff;        /* proactiveLastUsedCutoff, MM.java:6662-6664                 */
} mmp_proactive_info;

/* One local CacheEntry as the rebalancers see it (rows a15, a16, a21) — 56 bytes. */
#define MMP_CE_FAILED 1u /* ce == null || ce.isFailed() */
typedef struct {
    int32_t model;                 /* registry row, -1 if registry.get(modelId) == null */
    int32_t weight;                /* ce.getWeight()                                    */
    int64_t last_used;             /* cache last-used time                              */
    int64_t interval_count;        /* getAndResetIntervalCount() / getIntervalCount()   */
    int64_t last_heavy_time;       /* ce.getLastHeavyTime()                             */
    int64_t last_unload_time;      /* mr.getLastUnloadTime()                            */
    int32_t earlier_use_iteration; /* ce.earlierUseIteration                            */
    int32_t last_used_iteration;   /* ce.lastUsedIteration                              */
    uint32_t flags;
    int32_t reserved;
} mmp_cache_entry;

/* MaxConcCacheEntry (MM.java:2641-2797): what a mesh that runs with limitModelConcurrency == true keeps per loaded model — one
 * row per cache entry, in the order of the entries.  The rate task evaluates getRpmScaleThreshold(true) on it (the row's own
 * threshold replaces scale_up_rpm_threshold: "latency-based" scaling, MM.java:5677, :5702-5707), the janitor
 * getRpmScaleThreshold(false) and queuedRequestCount() (:6294-6305). */
#define MMP_CONC_COUNT_BITS 21 /* MM.java:2653: completed invocations in the low bits of countAndTimeSum, the sum of their
                                  durations in 1/10 ms above */
typedef struct {
    int64_t count_and_time_sum; /* countAndTimeSum.sum()                      */
    int64_t prior_sum;          /* priorSum                                   */
    int32_t prior_count;        /* priorCount                                 */
    int32_t max_conc;           /* maxConc                                    */
    int32_t queued_requests;    /* queuedRequestCount() (the janitor, :6303)  */
    int32_t reserved;
} mmp_conc_entry;
typedef struct {
    int32_t threshold;       /* getRpmScaleThreshold(true) as the task got it; 0 for an entry it skipped before the call (:5697) */
    int32_t reset;           /* 1: the call took countAndTimeSum.sumThenReset() (:2771): the caller resets its adder and stores: */
    int64_t new_prior_sum;   /*    priorSum   (unchanged when reset == 0)                                                        */
    int32_t new_prior_count; /*    priorCount                                                                                    */
    int32_t reserved;
} mmp_conc_out;
typedef struct {
    int64_t dynamic_rpm_scale_constant; /* 600 000 * percentage / 100, MM.java:370, :732                           */
    double average_model_parallelism;   /* the task's field going into this run (:5634; 1.0 before the first run)   */
} mmp_conc_params;
typedef struct {
    double average_model_parallelism; /* after this run: max(1.0, (double) modelParallelismSum / entries), :5815-5818 — the
                                         input value when the run returned early.  The path's only floating-point value: a sum
                                         of ints, one IEEE division, one comparison — compared EXACTLY with the reference's    */
    int32_t exclude_set_rpms;         /* (int) (900.0 * averageModelParallelism) of the INPUT value: getExcludeSet's threshold, :5836 */
    int32_t model_parallelism_sum;    /* modelParallelismSum, :5706                                                          */
} mmp_conc_result;

/* rateTrackingTask (MM.java:5636-5832).  mmp_scaleup_plan: limitModelConcurrency == false; mmp_scaleup_plan_conc: true. */
typedef struct {
    int32_t self_pod;
    int32_t iteration_counter;
    int32_t second_copy_max_age_iters; /* MM.java:5621 */
    int32_t second_copy_min_age_iters; /* MM.java:5622 */
    int32_t scale_up_rpm_threshold;    /* MM.java:240  */
    int32_t our_rpm;                   /* invokeCounter.getBusyness(), MM.java:5837 */
    int64_t now;
    int64_t last_check_time;
    int64_t rate_check_interval_ms;        /* MM.java:238  */
    int64_t second_copy_lru_threshold_ms;  /* MM.java:5628 */
    int64_t assume_completed_ms;           /* loadingTimeStats(type).assumeCompletedAfterMillis() */
} mmp_scaleup_params;
#define MMP_SCALE_NONE 0
#define MMP_SCALE_SECOND_COPY 1 /* ensureLoadedInternalAsync(id, lastTime, w, excludeThisInstance, 0), MM.java:5755 */
#define MMP_SCALE_UP 2          /* ensureLoadedInternalAsync(id, now+20s, w, exclude, copies-1), MM.java:5805     */
typedef struct {
    int32_t action;
    int32_t copies;    /* copiesToLoad */
    int64_t timestamp; /* lastUsedTime to pass to the load-target decision */
    int32_t new_i1, new_i2; /* updated earlierUseIteration / lastUsedIteration */
    int32_t heavy;     /* ce.setLastHeavyTime(now) */
    int32_t rpm;
} mmp_scaleup_out;

/* janitor scale-down (MM.java:6110-6145, removeModelCopies :6197-6310). */
typedef struct {
    int32_t self_pod;
    int32_t shutting_down;
    int64_t now;
    int64_t last_check_time;
    int64_t rate_check_interval_ms;
    int64_t adjusted_cache_capacity; /* getAdjustedCacheCapacity(), MM.java:6117 */
    int32_t scale_up_rpm_threshold;
    int32_t reserved;
} mmp_scaledown_params;

/* ---- lifecycle --------------------------------------------------------- */
int mmp_abi_version(void);
int mmp_create(const mmp_config *cfg, mmp_ctx **out);
void mmp_destroy(mmp_ctx *ctx);
/* Text of the CALLING THREAD's last failure in this library (any context; ctx may be NULL).  The pointer stays
 * valid until the same thread fails again: concurrent callers never see each other's message. */
const char *mmp_last_error(mmp_ctx *ctx);
/* 1 = hip (single device). There is no host backend. */
int mmp_backend(mmp_ctx *ctx);

/* MM.java:765-771 */
int64_t mmp_min_space_units(int32_t default_model_size_units, int32_t loading_threads,
                            int64_t capacity_units, int have_unload_manager);

/* ---- snapshot: what clusterState/registry/typeConstraints hold --------- */
/* Replace the whole instance table (MM.java:332 clusterState, fed by
 * handleInstanceTableChange MM.java:1455). Host pointer, copied. */
int mmp_pods_load(mmp_ctx *ctx, const mmp_pod_row *rows, int32_t n_pods);
/* Upsert / delete single rows by index (ENTRY_ADDED/UPDATED/DELETED,
 * MM.java:1476-1542). idx[i] may equal the current pod count to append. */
int mmp_pods_upsert(mmp_ctx *ctx, const int32_t *idx, const mmp_pod_row *rows, int32_t n);
int mmp_pods_remove(mmp_ctx *ctx, const int32_t *idx, int32_t n);
/* TypeConstraintManager.getCandidateInstances / getPreferredInstances
 * (TypeConstraintManager.java:242-251) as bitmaps over pod index, row-major
 * [n_types][ceil(n_pods/64)] uint64 words, bit p%64 of word p/64.
 * has_allowed[t]==0 / has_prefer[t]==0 mean the Java returned null.
 * n_types==0 means typeConstraints==null. */
int mmp_types_load(mmp_ctx *ctx, int32_t n_types, const uint64_t *allowed, const uint64_t *prefer,
                   const uint8_t *has_allowed, const uint8_t *has_prefer);
/* Rebuild the per-type instance sets from labels ON THE DEVICE (row a18): labels are interned to
 * bits; pod_labels[p] = the instance's label set, required[t]/preferred[t] = the type's
 * requiredLabels / preferredLabels (TypeConstraintManager.java:337-447, :478-486, :680-747).
 * Installs n_types+1 type rows for the next commit: row n_types is the row for model types that
 * are not in the config (no constraint, defaultPreferredInstances). Optional outputs (may be
 * NULL) return the computed tables in the mmp_types_load format with n_types+1 rows. */
int mmp_types_from_labels(mmp_ctx *ctx, int32_t n_types, const uint64_t *required, const uint64_t *preferred,
                          const uint64_t *pod_labels, uint64_t *allowed_out, uint64_t *prefer_out,
                          uint8_t *has_allowed_out, uint8_t *has_prefer_out);
/* UpgradeTracker.getLikelyReplacedReplicaSets (UpgradeTracker.java:78). */
int mmp_replaced_rs_load(mmp_ctx *ctx, const int32_t *replica_sets, int32_t n);
/* UpgradeTracker as a stateful part of the context (row a19; UpgradeTracker.java:85-200): the Java
 * instance-table listener forwards its three calls (MM.java:1532,1553,1563).  labels_key = identity of
 * the record's labels array as the reference's HashMap<String[],..> sees it (0 for NO_LABELS),
 * replica_set = interned id.substring(0,6) or -1 if |id| < 7.  The resulting replica-set list replaces
 * the one given to mmp_replaced_rs_load and takes effect at the next commit. */
int mmp_upgrade_instance_added(mmp_ctx *ctx, int64_t labels_key, int32_t replica_set, int64_t start_time, int64_t now_ms);
int mmp_upgrade_instance_removed(mmp_ctx *ctx, int64_t labels_key, int32_t replica_set, int64_t now_ms);
int mmp_upgrade_housekeeping(mmp_ctx *ctx, int64_t now_ms);
/* getLikelyReplacedReplicaSets(): up to max entries (replica set, expiry); *n_out = entries in the map. */
int mmp_upgrade_replaced(mmp_ctx *ctx, int32_t *rs_out, int64_t *expiry_out, int32_t max, int32_t *n_out);
/* The model registry view (MM.java:308). ent_pod / ent_time have n_entries items. */
int mmp_models_load(mmp_ctx *ctx, const mmp_model_row *rows, int32_t n_models,
                    const int32_t *ent_pod, const int64_t *ent_time, int32_t n_entries);
/* Registry events (the registry's KV listener, MM.java:628; ModelRecord is replaced as a whole on every
 * change): rows[i] replaces model idx[i] (idx[i] == current model count appends); rows[i].ent_off indexes
 * the ent_pod / ent_time arrays of THIS call.  A deleted record is upserted as an empty row.  When a model
 * appears twice the last row wins.  O(rows + entries) per call: the new entries are appended to an
 * arena, the rows (and their resolved exclusion positions) are rewritten in place; the arena is squeezed
 * when it holds more garbage than live entries.  Takes effect immediately (no commit needed: the
 * registry is not part of the snapshot). */
int mmp_models_upsert(mmp_ctx *ctx, const int32_t *idx, const mmp_model_row *rows, int32_t n, const int32_t *ent_pod,
                      const int64_t *ent_time, int32_t n_entries);
/* Rank pods by PLACEMENT_ORDER (MM.java:4646-4703) on the device and publish
 * the new immutable snapshot. MMP_EORDER if the comparator is inconsistent.
 * Wait-free for the latency path: the snapshot is built beside the published one and published with a
 * pointer swap; concurrent calls that ride the latency slots — mmp_place_batch up to 4096 decisions,
 * mmp_gate_batch up to 1820 requests, mmp_evict_batch up to 2048 evaluations — and mmp_place_batch_dev
 * launches keep answering for the published snapshot until then.  Calls that stage through the context's
 * batch stream (larger host-pointer batches, mmp_serve_batch, the plan calls, mmp_cache_replay) share that
 * stream and its scratch with the commit and therefore queue behind a running one; so do loaders of the
 * commit's inputs and other commits. */
int mmp_snapshot_commit(mmp_ctx *ctx);
/* handleInstanceTableChange delivers ONE InstanceRecord per event (MM.java:1455-1568): when at most 16 rows were written
 * (mmp_pods_upsert / _remove / _ingest_json) since the published snapshot and PLACEMENT_ORDER is a total order on the table
 * before and after, the commit re-ranks by insertion — the unchanged rows keep their relative order, the changed ones are
 * placed by binary search with the literal comparator — instead of sorting; the result is the same snapshot.
 * *n_commits_out = commits that took that path on this context (diagnostics; MMP_NO_DELTA=1 in the environment disables it). */
int mmp_delta_commits(mmp_ctx *ctx, int64_t *n_commits_out);
/* The per-type SHORTLISTS of the published snapshot (diagnostics).  What getNext's walk (MM.java:4806-4947: first eligible instance,
 * preference step, the three breaks, count) yields depends on the request only through positions of its own — its exclusions, the
 * calling instance — that lie INSIDE the shortlist, and through one bit, the fresh-row test of :4913-4922.  commit records, per type row
 * (the first 12) and per value of that bit, the shortlist of a request that has no position of its own in reach; large single-caller
 * batches (mmp_place_batch_c / _c_dev) decide a request from it after checking exactly that, and every other request by the ordinary
 * path in the same launch (results identical either way; MMP_NO_MEMO=1 in the environment keeps every request on the ordinary path).
 * rows[2 * t + bit] = {valid, lo, hi, n_candidates}: the list holds for requests without a position in [lo, hi).
 * *n_rows_out = 2 * min(type rows, 12); rows beyond cap_rows are not written. */
typedef struct {
    int32_t valid;
    int32_t lo, hi;
    int32_t n_candidates;
} mmp_shortlist_row;
int mmp_shortlists(mmp_ctx *ctx, mmp_shortlist_row *rows, int32_t cap_rows, int32_t *n_rows_out);
/* The same for the LONG shortlists of a cluster whose instances are (nearly) all full (diagnostics; ModelMesh.java:4880-4991 through the
 * prefix tables): there a shortlist spans the table and a request always has positions of its own inside it, which the prefix-table
 * path treats as corrections (count, audit hash, the pick's rank) of a list it never builds.  The walk itself — first eligible
 * instance, preference step, break scans — depends on the request only when an exclusion or the calling instance sits on a position
 * that steers it; commit records it per type row (every row) and per value of the fresh-row bit, and a request none of whose own
 * positions is one of those is decided from the record (results identical; MMP_NO_LONG_MEMO=1 in the environment: no records).
 * rows[2 * t + bit] = {valid, lo = the type's first eligible position, hi = where the list ends, n_candidates}; *n_rows_out = 2 * type
 * rows (0: no records for this snapshot). */
int mmp_long_shortlists(mmp_ctx *ctx, mmp_shortlist_row *rows, int32_t cap_rows, int32_t *n_rows_out);
/* Large batches (request rows from 393 216 decisions, the single-caller form from 524 288; host-pointer and device-pointer calls alike)
 * are decided by TWO launches on the call's stream: the first checks every request against the shortlists above and decides what they
 * cover — on the bench configuration 99.97 % — the second, a few workgroups, decides the rest by the ordinary path (results identical
 * either way; MMP_NO_SPLIT=1 in the environment keeps such batches in one launch).  The second launch is hidden behind the first launch of
 * the next batch only if that one runs on ANOTHER hardware queue: a host that issues batches from several streams should give each a queue
 * of its own (GPU_MAX_HW_QUEUES >= its streams + the library's: 8 for four).  A second launch that finds more than 1/32 of its batch left
 * switches the split off until the next commit.  Diagnostics: *n_split_out = batches issued that way on this
 * context, *off_out = 1 while the split is switched off (either may be null). */
int mmp_split_batches(mmp_ctx *ctx, int64_t *n_split_out, int32_t *off_out);
/* clusterState iteration order (the `getCacheState` dump, MM.java:5552-5608).
 * order_out has room for n_pods ints; *n_out = rows actually in the set. */
int mmp_get_order(mmp_ctx *ctx, int32_t *order_out, int32_t *n_out);
/* ClusterStats of the committed snapshot (MM.java:1570-1591). */
int mmp_cluster_stats(mmp_ctx *ctx, mmp_stats *out);
/* With type constraints the mesh does not use the cluster-wide stats everywhere (TypeConstraintManager): the
 * instances are partitioned by their ProhibitedTypeSet — the constrained types they cannot host — each
 * partition has its own stats (InstanceSetStatsTracker), typeSetStats(type) is the sum over the partitions
 * that can host the type (MM.java:1432-1439: loadLocal sizing :5169, the onEviction reload rule :2918, the
 * scale-up task :5691) and instanceSetStats() is the partition of this instance (:1446-1448: scale-down
 * :6228).  The library rebuilds all of them at commit and uses them in mmp_gate_batch, mmp_scaleup_plan,
 * mmp_scaledown_plan; these calls read them back.  A partition's lru is the cluster-wide minimum (the Java
 * re-accumulates it over ALL instances on every event, MM.java:1515-1542).  Without type constraints there are
 * no partitions and every type's stats are the cluster's. */
int mmp_type_stats(mmp_ctx *ctx, int32_t type, mmp_stats *out);
int mmp_partition_count(mmp_ctx *ctx, int32_t *n_out);
/* prohibited_out (may be NULL with max_words 0): the partition's prohibited types as a bitset over type rows */
int mmp_partition_stats(mmp_ctx *ctx, int32_t partition, mmp_stats *out, uint64_t *prohibited_out, int32_t max_words);
/* partition of every pod slot (-1: not in the table); *n_out = pod slots */
int mmp_pod_partitions(mmp_ctx *ctx, int32_t *partition_out, int32_t max_pods, int32_t *n_out);

/* ---- decisions --------------------------------------------------------- */
/* n load-target decisions = n × CacheMissForwardingLB.getNext (MM.java:4776-5005).
 * extra_pool: pod indices referenced by reqs[i].extra_off/n_extra. Host pointers. */
int mmp_place_batch(mmp_ctx *ctx, const mmp_place_req *reqs, int32_t n, const int32_t *extra_pool,
                    int32_t n_extra_pool, int64_t now_ms, mmp_place_out *outs);
/* Same, with every buffer already in device memory and launched on `stream`
 * (a hipStream_t, NULL = default) without synchronising.  The library remembers every stream it was handed:
 * before it rewrites state such a launch may still be reading (the second commit after it, registry loads
 * and events, cache-table loads) it waits for those streams as it waits for its own.  A stream must
 * therefore stay valid until mmp_stream_retire() or mmp_destroy(). */
int mmp_place_batch_dev(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_extra_pool,
                        int64_t now_ms, void *d_outs, void *stream);
/* mmp_place_batch_dev with the pool's length (entries): a request whose exclusion range [extra_off, extra_off + n_extra) does not
 * lie inside the pool is not followed — its result row is {MMP_NONE, MMP_BAD_REQUEST, 0, 0} — instead of being read wherever it
 * points (mmp_place_batch_dev cannot check: it is not told the length).  d_extra_pool may be NULL when n_extra_pool == 0. */
int mmp_place_batch_dev2(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_extra_pool, int32_t n_extra_pool,
                         int64_t now_ms, void *d_outs, void *stream);
/* The single-caller form (mmp_place_caller + mmp_place_req_c, above): host pointers / device pointers.  `caller` is host memory
 * in both (it rides in the kernel's arguments).  The device-pointer call is bounded like mmp_place_batch_dev2.  Results are
 * bit-identical to the same decisions as mmp_place_req rows. */
int mmp_place_batch_c(mmp_ctx *ctx, const mmp_place_caller *caller, const mmp_place_req_c *reqs, int32_t n,
                      const int32_t *extra_pool, int32_t n_extra_pool, int64_t now_ms, mmp_place_out *outs);
int mmp_place_batch_c_dev(mmp_ctx *ctx, const mmp_place_caller *caller, const void *d_reqs, int32_t n, const void *d_extra_pool,
                          int32_t n_extra_pool, int64_t now_ms, void *d_outs, void *stream);
/* k request arrays decided by ONE launch: the same as k calls of mmp_place_batch_dev on `stream` (array i: n[i] requests at
 * d_reqs[i], its own exclusion pool d_extra_pool[i] — the array may be NULL when no request carries extras —, results to
 * d_outs[i]), for a host that holds many batches the size of one request set: a 100k-decision launch lasts an empty launch + one
 * dependent chain (7.8 us, 0.18 of the HBM peak), eight of them in one launch run at the rate of an 800k batch (25 us instead of
 * 62).  The pointer ARRAYS are host memory and are read before the call returns; at most 16 arrays share a launch (more are
 * split).  Results are bit-identical to the separate calls. */
int mmp_place_multi_dev(mmp_ctx *ctx, int32_t k, const void *const *d_reqs, const int32_t *n, const void *const *d_extra_pool,
                        int64_t now_ms, void *const *d_outs, void *stream);
/* The resident decision kernel.  A single request through mmp_place_batch(n = 1) normally costs a kernel launch
 * (6.5 us before the decision's first instruction, tools/micro/doorbell.hip).  mmp_resident(ctx, 1) — or MMP_RESIDENT=1
 * in the environment of mmp_create — keeps ONE wavefront resident instead: its 64 lanes poll 64 request slots in pinned
 * host memory, so up to 64 request threads are decided concurrently and none of them launches anything; a request is
 * a 64-byte store plus a tag, the answer a 16-byte row plus the tag.  It serves requests WITHOUT exclusions of their
 * own (n_extra = 0; the others, and the rare shapes that need the wave path, take the launch path transparently), holds
 * the published snapshot (a commit, registry event or cache-table load stops it and the next request starts a new one)
 * and leaves the GPU by itself after MMP_RESIDENT_IDLE_MS (default 50) without a request.  Results are bit-identical.
 * Two guards: eight hand-backs to the launch path in a row send the next 4096 single requests there directly (a table on
 * which most decisions need the wave path), and three answers in a row slower than 20 ms from a kernel that was not
 * restarted meanwhile switch the resident path off for the context (mmp_last_error says so). */
int mmp_resident(mmp_ctx *ctx, int enable);
int mmp_resident_stats(mmp_ctx *ctx, uint64_t *launches, uint64_t *served, uint64_t *punted);
/* Submission threads.  One host thread spends ~3 us in HIP's launch path per kernel — more than a 100k-decision batch
 * takes the GPU when several are in flight.  mmp_issue_threads(ctx, n) starts n helper threads (they spin: use them for
 * bursts) and mmp_place_batch_dev then only validates, appends a descriptor to the ring of the helper that owns the
 * stream (launches on one stream keep their order) and returns; the helper captures the published snapshot and launches.
 * mmp_issue_flush returns once everything submitted so far has been handed to the HIP stream (first launch error, if
 * any); call it before synchronising the streams.  n = 0 stops the helpers. */
int mmp_issue_threads(mmp_ctx *ctx, int32_t n);
int mmp_issue_flush(mmp_ctx *ctx);
/* Forget a caller-owned stream (waits for what was enqueued on it first); call before destroying a stream
 * that was passed to a *_dev entry point. */
int mmp_stream_retire(mmp_ctx *ctx, void *stream);

/* n serve-target decisions = n × ForwardingLB.getNext (MM.java:4315-4392).
 * counters: the requests' mmp_serve_counter entries (see mmp_serve_req); excl_pod / excl_time: pairs referenced by
 * excl_off.  O(copies) per request on both sides of the boundary; calls of up to 1024 requests (8192 counters, 4096
 * exclusion pairs) ride the latency slots like mmp_place_batch's. */
int mmp_serve_batch(mmp_ctx *ctx, const mmp_serve_req *reqs, int32_t n, const mmp_serve_counter *counters,
                    int32_t n_counters, const int32_t *excl_pod, const int64_t *excl_time, int32_t n_excl_pool,
                    int64_t now_ms, mmp_serve_out *outs);

/* Per-pod cache segments for eviction: seg_off has n_caches+1 entries; entry i
 * of a segment is the i-th node of that cache's evictionDeque (oldest first). */
int mmp_caches_load(mmp_ctx *ctx, int32_t n_caches, const int32_t *seg_off, const int64_t *last_used,
                    const int32_t *weight, const int64_t *capacity);
int mmp_evict_batch(mmp_ctx *ctx, const mmp_evict_req *reqs, int32_t n, int64_t now_ms,
                    mmp_evict_out *outs);

/* Stateful caches: entry i of cache c's segment is the i-th node of its evictionDeque (oldest first)
 * with its key; ubm may be NULL (no cache is managed). */
int mmp_caches_load_keyed(mmp_ctx *ctx, int32_t n_caches, const int32_t *seg_off, const int64_t *last_used,
                          const int32_t *weight, const int32_t *key, const int64_t *capacity, const mmp_ubm_state *ubm);
/* Apply n_ops operations (each cache's operations in the order given) to the stateful caches.
 * evicted_keys has room for max_evicted keys; *n_evicted_slots = slots the call used (outs index into
 * it).  MMP_EINVAL if one cache's entries + inserts exceed the 2048-slot tile or max_evicted is too small. */
int mmp_cache_replay(mmp_ctx *ctx, const mmp_cache_op *ops, int32_t n_ops, int64_t now_ms, mmp_cache_op_out *outs,
                     int32_t *evicted_keys, int32_t max_evicted, int32_t *n_evicted_slots);
/* Read one cache back (deque order). */
int mmp_cache_read(mmp_ctx *ctx, int32_t cache, int32_t max_entries, int64_t *last_used, int32_t *weight, int32_t *key,
                   int32_t *n_out, int64_t *capacity, int64_t *weighted_size, mmp_ubm_state *ubm);

/* n guard evaluations (rows a10/a11/a14/a20). in_use_failure_expiry_ms = IN_USE_LOAD_FAILURE_EXPIRY_MS
 * (MM.java:221). excl_pod/excl_time and explicit_pool are the pools the requests index. */
int mmp_gate_batch(mmp_ctx *ctx, const mmp_gate_req *reqs, int32_t n, const int32_t *excl_pod,
                   const int64_t *excl_time, int32_t n_excl_pool, const int32_t *explicit_pool,
                   int32_t n_explicit_pool, int64_t now_ms, int64_t in_use_failure_expiry_ms,
                   mmp_gate_out *outs);
/* The cache-MISS route of one request: the request guards (as mmp_gate_batch) AND the load target (as mmp_place_batch) of the
 * same model in ONE call — invokeModel evaluates the guards and then asks CacheMissForwardingLB.getNext (MM.java:3603-3626,
 * :4590-4627, :4776).  greqs[i] and preqs[i] name the same model; the gate pools (excl_pod / excl_time / explicit_pool) and the
 * load target's extra_pool are separate, as in the two calls.  Up to 256 requests ride one latency slot: the two kernels are
 * enqueued behind one another on the slot's stream and the call waits ONCE (one request: p50 ~14 us against 23 us for the two
 * calls); larger batches are the two calls.  Rows are bit-identical to the two calls'. */
int mmp_miss_batch(mmp_ctx *ctx, const mmp_gate_req *gate_reqs, const mmp_place_req *place_reqs, int32_t n, const int32_t *excl_pod,
                   const int64_t *excl_time, int32_t n_excl, const int32_t *explicit_pool, int32_t n_explicit,
                   const int32_t *extra_pool, int32_t n_extra_pool, int64_t now_ms, int64_t in_use_expiry_ms, mmp_gate_out *gate_outs,
                   mmp_place_out *place_outs);
/* The cache-hit route of invokeModel in one call and ONE launch: request i's guards (gate_reqs[i], as mmp_gate_batch) and its
 * serve target among the model's copies (serve_reqs[i], as mmp_serve_batch; serve_reqs[i].model == gate_reqs[i].model).  The two
 * request arrays index the SAME (excl_pod, excl_time) pool — cacheHitExcludeTl's MapFilteringSet is one object for goLocal and
 * for ForwardingLB.getNext (MM.java:3634, :4316) — and the serve requests their counters as in mmp_serve_batch.  Results equal
 * those of the two separate calls. */
int mmp_route_batch(mmp_ctx *ctx, const mmp_gate_req *gate_reqs, const mmp_serve_req *serve_reqs, int32_t n,
                    const mmp_serve_counter *counters, int32_t n_counters, const int32_t *excl_pod, const int64_t *excl_time,
                    int32_t n_excl, const int32_t *explicit_pool, int32_t n_explicit, int64_t now_ms,
                    int64_t in_use_failure_expiry_ms, mmp_gate_out *gate_outs, mmp_serve_out *serve_outs);

/* triggerProactiveLoadsForInstanceSubset (MM.java:6616-6747, excludeTypes == null) over the
 * committed snapshot and the loaded model table (models in registry iteration order): which
 * unloaded models the leader would proactively load, most recently used first. The caller then
 * feeds them to mmp_place_batch with last_used = out_last_used[i] (MM.java:6727). */
int mmp_proactive_plan(mmp_ctx *ctx, int32_t default_model_size_units, int64_t now_ms, int32_t max_out,
                       int32_t *out_model, int64_t *out_last_used, mmp_proactive_info *info);
/* The same plan for ONE instance partition: with type constraints the reaper calls
 * triggerProactiveLoadsForInstanceSubset once per ProhibitedTypeSet partition (MM.java:6473-6488) with that
 * partition's stats, its instances for the free-space budget, and its prohibited types excluded from the
 * candidates; skip_models = the models already triggered for an earlier partition of the same run
 * (allCandidates.set(index, null), :6724).  partition = -1: the whole cluster (typeConstraints == null),
 * which is what mmp_proactive_plan does.  The candidate rule (pruneModelRegistry, :6459-6462, :6574-6577)
 * always uses the cluster-wide stats. */
int mmp_proactive_plan_subset(mmp_ctx *ctx, int32_t partition, const int32_t *skip_models, int32_t n_skip,
                              int32_t default_model_size_units, int64_t now_ms, int32_t max_out, int32_t *out_model,
                              int64_t *out_last_used, mmp_proactive_info *info);

/* a15: entries = usedSinceLastRun (runtimeCache.descendingMapWithCutoff(lastTime)) in iteration order.
 * overloaded_out has one byte per pod = membership in getExcludeSet() (MM.java:5835-5856); for
 * MMP_SCALE_UP rows the caller passes those pods as extra excludes of the load-target decisions
 * (the Java builds the set lazily, so it is only meaningful when some row is MMP_SCALE_UP).
 * *skipped = 1 when the task returns before looking at the entries (MM.java:5646, :5658). */
int mmp_scaleup_plan(mmp_ctx *ctx, const mmp_cache_entry *entries, int32_t n, const mmp_scaleup_params *params,
                     mmp_scaleup_out *outs, uint8_t *overloaded_out, int32_t *skipped);
/* a15 with limitModelConcurrency == true (MM.java:5677: latencyBased): conc[i] = the MaxConcCacheEntry state of entries[i].  Per
 * entry the threshold is mcce.getRpmScaleThreshold(true) (:5704, :2766-2796; params->scale_up_rpm_threshold is what that
 * returns for an entry without enough samples, :2781), heavyRpms three quarters of it, and maxConc adds to modelParallelismSum;
 * getExcludeSet() uses (int) (900.0 * averageModelParallelism) of the PREVIOUS run (:5836).  conc_outs[i] = the threshold and the
 * counter reset the call made; *result = the task's averageModelParallelism afterwards. */
int mmp_scaleup_plan_conc(mmp_ctx *ctx, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                          const mmp_scaleup_params *params, const mmp_conc_params *conc_params, mmp_scaleup_out *outs,
                          mmp_conc_out *conc_outs, uint8_t *overloaded_out, int32_t *skipped, mmp_conc_result *result);
/* a16: entries = scaleCopiesCandidates, oldest first; removed_out[i] = this copy is removed. */
int mmp_scaledown_plan(mmp_ctx *ctx, const mmp_cache_entry *entries, int32_t n,
                       const mmp_scaledown_params *params, uint8_t *removed_out);
/* a16 with MaxConcCacheEntry entries (MM.java:6294-6305): the threshold of a model with three or more copies is
 * mcce.getRpmScaleThreshold(false), and a copy with more than one queued request stays. */
int mmp_scaledown_plan_conc(mmp_ctx *ctx, const mmp_cache_entry *entries, const mmp_conc_entry *conc, int32_t n,
                            const mmp_scaledown_params *params, int64_t dynamic_rpm_scale_constant, uint8_t *removed_out);
/* a21: entries = runtimeCache.descendingLruMap() (MRU first). action_out[i] = 1:
 * triggerNewModelCopyElsewhere (MM.java:6913-6928) is issued with lastUsedTime = entry.last_used
 * and excludes = current holders ∪ self; wait_out[i] = 1: shutdown waits for it (CUTOFF_AGE_MS). */
int mmp_migration_plan(mmp_ctx *ctx, const mmp_cache_entry *entries, int32_t n, int32_t self_pod, int64_t now_ms,
                       int64_t cutoff_age_ms, uint8_t *action_out, uint8_t *wait_out);

/* ---- ingestion of the KV-store wire format ---------------------------------------
 * The instance table and the registry live in etcd / ZooKeeper as Jackson JSON values
 * (MM.java:346 INST_REC_SERIALIZER, :628 registry view; InstanceRecord.java:37-69,
 * ModelRecord.java:61-114).  These entry points take the raw values and parse them on the device. */
/* Define the pod index space from the instance ids (the KV keys): computes id_order (rank under
 * String.compareTo; ids must be ASCII) and replica_set (interned id.substring(0,6), -1 if |id| < 7,
 * MM.java:4769) for every pod, and the id -> pod table used to resolve ModelRecord.instanceIds keys.
 * Rows not yet ingested are absent (tombstones).  Optional outputs may be NULL. */
int mmp_pod_ids_load(mmp_ctx *ctx, const char *ids, const int32_t *id_off, int32_t n_pods, uint32_t *id_order_out,
                     int32_t *replica_set_out);
/* n InstanceRecord JSON values, value i = buf[off[i], off[i+1]) for pod pod_idx[i]; live[i] != 0 marks
 * the instance as present in the litelinks registry (MM.java:4765).  Equivalent to mmp_pods_upsert with
 * rows parsed from the JSON.  status_out[i] = 1 for a malformed value (that row is left unchanged);
 * start_time_out[i] = InstanceRecord.startTime (input of mmp_upgrade_instance_added). */
int mmp_pods_ingest_json(mmp_ctx *ctx, const char *buf, const int64_t *off, int32_t n, const int32_t *pod_idx,
                         const uint8_t *live, int64_t *start_time_out, int32_t *status_out);
/* Names of the model types in type-table order (ModelRecord "type"); a name not listed maps to
 * unknown_type (the extra row mmp_types_from_labels installs), an absent / null type to the index of
 * "NLCLASSIFIER" (ModelRecord.DEFAULT_TYPE, ModelRecord.java:121-133) if listed, else unknown_type. */
int mmp_type_names_load(mmp_ctx *ctx, const char *names, const int32_t *name_off, int32_t n_types, int32_t unknown_type);
/* Replace the registry view (like mmp_models_load) from n_models ModelRecord JSON values; model i =
 * value i.  Ids that are not in the pod table become entries with pod -1 (they still count as copies).
 * last_unload_out[i] = "lul" (ModelRecord.lastUnloadTime, an input of the scale-down plan). */
int mmp_models_ingest_json(mmp_ctx *ctx, const char *buf, const int64_t *off, int32_t n_models, int64_t *last_unload_out,
                           int32_t *status_out);
/* Read the staged instance table / the loaded registry view back (tests, diagnostics). */
int mmp_pods_get(mmp_ctx *ctx, mmp_pod_row *rows_out, int32_t max_rows, int32_t *n_out);
int mmp_models_get(mmp_ctx *ctx, mmp_model_row *rows_out, int32_t max_models, int32_t *ent_pod_out, int64_t *ent_time_out,
                   int32_t max_entries, int32_t *n_models_out, int32_t *n_entries_out);

/* ---- pod-axis sharding across the GPUs of one node ---------------------------
 * There is no reference counterpart: the reference walks clusterState (MM.java:4763) on one JVM
 * thread.  Here shard g of G owns a contiguous range of PLACEMENT_ORDER positions (the words
 * [g*ceil(W/G), ...) of every rank-ordered bitmap and column); every shard sees the whole request
 * batch, and one CacheMissForwardingLB.getNext (MM.java:4776-5005) is evaluated as six local scans
 * with an all-reduce of a small per-decision int64 vector after each (MIN, except phase 5 = SUM).
 * The library launches the kernels; the HOST performs the collectives between phases (RCCL
 * all-reduce over xGMI: modelmesh_amd/dist.py does it with torch.distributed; a Java host would
 * call ncclAllReduce on the same device buffers).  Results are bit-identical to mmp_place_batch.
 *
 *   mmp_shard_configure(ctx, g, G)        once, before the first commit (G = 1 is allowed)
 *   commit:  mmp_shard_rank_dev(ctx, d_rank)   -> all-reduce SUM of int32 d_rank[n_pods]
 *            mmp_shard_commit_dev(ctx, d_rank)     (both synchronise the context's stream)
 *   batch:   for phase in 1..6: mmp_shard_place_phase_dev(...); all-reduce d_xchg[phase-1]
 *            mmp_shard_place_phase_dev(phase 7) writes d_outs on every shard
 * d_xchg[k] (k = 0..5) are device int64 buffers of n * mmp_shard_xchg_slots(k+1, G) elements. */
int mmp_shard_configure(mmp_ctx *ctx, int32_t shard, int32_t n_shards);
int32_t mmp_shard_xchg_slots(int32_t phase, int32_t n_shards);
int32_t mmp_shard_xchg_is_sum(int32_t phase); /* 1: SUM, 0: MIN */
/* PLACEMENT_ORDER ranks (MM.java:4646-4703) of this shard's slice of the pod table against all
 * pods; other entries of d_rank (device int32[n_pods]) are zeroed. */
int mmp_shard_rank_dev(mmp_ctx *ctx, void *d_rank);
int mmp_shard_commit_dev(mmp_ctx *ctx, const void *d_rank);
int mmp_shard_place_phase_dev(mmp_ctx *ctx, int32_t phase, const void *d_reqs, int32_t n, const void *d_extra_pool,
                              int64_t now_ms, void *const *d_xchg, void *d_outs, void *stream);

/* The speculative single-exchange form in front of the phases above (csrc/shard_kernels.hpp).  The head of
 * PLACEMENT_ORDER decides almost every request, so each shard first runs the complete lane-per-decision
 * getNext on its own slice and publishes, per decision, mmp_shard_fast_slots() int64 words: INT64_MAX = "no
 * eligible pod in my slice", otherwise its result keyed by the shard number, with an "incomplete" bit when
 * the shortlist runs off the end of the slice or the decision needs the general path.  After ONE
 * all-reduce(MIN) of d_xf[n * slots] the lowest shard holding an eligible pod has won every slot:
 *
 *   mmp_shard_place_fast_dev(ctx, d_reqs, n, d_extra, now, d_xf, stream)        -> all-reduce MIN d_xf
 *   mmp_shard_place_fast_finish_dev(..., &n_rest, &d_rest_reqs, &d_rest_outs)      writes the decided rows of
 *        d_outs, counts the undecided requests (the count reaches the host in pinned memory; `stream` is
 *        synchronised) and, only when there are any, compacts them (same order on every shard) into
 *        library-owned device buffers
 *   if n_rest: phases 1..7 of mmp_shard_place_phase_dev on (d_rest_reqs, n_rest, d_rest_outs), then
 *        mmp_shard_place_fast_scatter_dev(ctx, n_rest, d_outs, stream)             rows back into d_outs
 *
 * One batch in flight per shard context (the rest buffers belong to the context).  Results are
 * bit-identical to mmp_place_batch. */
int32_t mmp_shard_fast_slots(void);
int mmp_shard_place_fast_dev(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_extra_pool, int64_t now_ms,
                             void *d_xf, void *stream);
int mmp_shard_place_fast_finish_dev(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_xf, void *d_outs,
                                    void *stream, int32_t *n_rest_out, void **d_rest_reqs_out, void **d_rest_outs_out);
int mmp_shard_place_fast_scatter_dev(mmp_ctx *ctx, int32_t n_rest, void *d_outs, void *stream);

/* ---- the pod-axis group with RCCL inside the boundary (north star: "the pod axis shards naturally across the 8
 * GPUs of one node with an RCCL allreduce over xGMI of per-shard best-candidate scores") -------------------------
 * One context per GPU (one process per GPU, or one thread per context).  The calls above leave the collectives to
 * the host (a torch.distributed process group in modelmesh_amd/dist.py); the calls below run them themselves, on
 * the context's own stream, through librccl (bound at run time), so that a host that holds no RCCL handles — the
 * Java mesh — reaches the multi-GPU layout through host pointers alone:
 *   rank 0:      mmp_shard_unique_id(id)                     128 bytes, handed to the other ranks out of band
 *                                                            (the mesh's KV store, litelinks, a file ...)
 *   every rank:  mmp_shard_group_init(ctx, id, rank, world)  ncclCommInitRank + mmp_shard_configure(rank, world)
 *                load the instance table / types / registry as for an unsharded context
 *                mmp_shard_commit(ctx)                       collective: rank slice -> ncclAllReduce(SUM) -> scatter
 *                mmp_shard_place_batch(ctx, reqs, n, ...)    collective: every rank passes the SAME batch and gets
 *                                                            the same result rows (bit-identical to mmp_place_batch)
 * A batch is: place_shard_fast_kernel (the slice's head windows + resolved registry rows, as the unsharded kernel) ->
 * ncclAllReduce(MIN, 2 int64 per decision) -> decided rows + the COUNT of the undecided rest, which the host reads
 * (identical on every shard: the words are the reduced ones) -> only when it is not zero: compaction, the six exchange
 * phases (5 x MIN, 1 x SUM) over exactly those rows, scatter (*n_rest_out = how many took the six phases).  (Round 2
 * kept the count on the device and always ran the six phases over a fixed-capacity sub-batch: seven launches and six
 * collectives per batch that mostly found no rows.)  unique_id may be NULL for world == 1: a group of one shard
 * without a communicator (no RCCL needed). */
#define MMP_SHARD_UNIQUE_ID_BYTES 128
/* mmp_shard_place_batch_async_dev: the same batch WITHOUT the host synchronisation at its end — the fast kernel, the
 * all-reduce and the finish kernel are enqueued on the context's stream and the call returns.  The batch is completed (its
 * rest count read, the six phases run if there is a rest) by the NEXT group call on the context — another batch, a commit —
 * or by mmp_shard_wait, which also returns the rest count.  Until then d_reqs / d_extra_pool / d_outs must stay valid and
 * d_outs must not be read: its rows are defined only once mmp_shard_wait (or a synchronous group call) has returned — those
 * synchronise the context's stream; the rest count the library polls in between carries no ordering for the rows.  Every shard of the group must issue the same sequence of calls.  (One shard, 100k decisions:
 * 26 us per batch with the synchronisation, 14 us without: the device's own time.  The exchange words, flags and count of a
 * batch live in one of two slots, so the batch before is completed AFTER this one has been enqueued.) */
int mmp_shard_unique_id(void *id_out);
/* A host that moves the exchange words itself (another transport than RCCL; several shards driven from one process)
 * installs a callback BEFORE mmp_shard_group_init and passes unique_id = NULL there.  The callback must all-reduce
 * `count` elements at device pointer `dev_buf` in place across the group — elem64: 0 = int32, 1 = int64; op_min:
 * 0 = SUM, 1 = MIN — ordered after everything already queued on `stream` (the context's hipStream_t), and return 0. */
typedef int (*mmp_exchange_fn)(void *user, void *dev_buf, int64_t count, int32_t elem64, int32_t op_min, void *stream);
int mmp_shard_group_set_exchange(mmp_ctx *ctx, mmp_exchange_fn fn, void *user);
int mmp_shard_group_init(mmp_ctx *ctx, const void *unique_id, int32_t rank, int32_t world);
int mmp_shard_group_destroy(mmp_ctx *ctx);
int mmp_shard_commit(mmp_ctx *ctx);
int mmp_shard_place_batch(mmp_ctx *ctx, const mmp_place_req *reqs, int32_t n, const int32_t *extra_pool,
                          int32_t n_extra_pool, int64_t now_ms, mmp_place_out *outs, int32_t *n_rest_out);
int mmp_shard_place_batch_dev(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_extra_pool, int64_t now_ms,
                              void *d_outs, int32_t *n_rest_out);
int mmp_shard_place_batch_async_dev(mmp_ctx *ctx, const void *d_reqs, int32_t n, const void *d_extra_pool, int64_t now_ms,
                                    void *d_outs);
int mmp_shard_wait(mmp_ctx *ctx, int32_t *n_rest_out);

/* Wait for everything queued on the context's own stream. */
int mmp_sync(mmp_ctx *ctx);

/* Operator metrics (the role of ModelMesh's Metrics timers around these decisions, Metrics.java):
 * with profiling enabled every host-pointer entry point brackets its kernels (not its staging copies)
 * with HIP events on the context's stream; mmp_last_kernel_ms returns the device time of the most
 * recent such call, or a negative value if none was recorded. */
int mmp_profile(mmp_ctx *ctx, int enable);
double mmp_last_kernel_ms(mmp_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MMPLACE_H */
