/*
 * mm_oracle.h — CPU restatement of ModelMesh's placement / eviction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP library in
 * modelmesh_amd/csrc; it is never linked into, called from or shipped with the
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Every function cites the reference file:line it follows.  "MM.java" is
 * src/main/java/com/ibm/watson/modelmesh/ModelMesh.java in kserve/modelmesh.
 *
 * Pinning status:
 *   - eviction / capacity arithmetic: pinned against the reference's own
 *     known-answer tests (EvictionsModelMeshTest.java:36-125,
 *     ModelMeshEvictionsTest.java:156-280) — see tests/test_oracle_kat.py.
 *   - PLACEMENT_ORDER, the filter, the load-target shortlist, the rpm filter, the
 *     serve-target choice, ClusterStats: no reference test names them and no JVM
 *     exists here; PINNED TO THE REFERENCE'S OWN TEXT since round 3 — the Java
 *     method bodies, cut from /root/reference at build time, run compiled against
 *     stand-ins (oracle/ref_harness -> tests/golden/ref_getnext.npz) and
 *     tests/test_ref_vectors.py holds this restatement to those vectors.  A second,
 *     independently written restatement (oracle/py_oracle.py) must agree too.
 *
 * All arithmetic is Java semantics: int64/int32 two's-complement wrap, '/'
 * truncating toward zero, '>>' arithmetic.
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NONE (-1) /* getNext returned null                     */
#define ORC_SELF (-2) /* getNext returned LoadBalancer.ABORT_REQUEST */

/* InstanceRecord.java:37-69 — one pod row.  Strings are interned by the
 * caller: id_order is the dense rank of the instance id under
 * String.compareTo (UTF-16 code-unit order), replica_set is the interned
 * id.substring(0,6) or -1 when id.length() < 7 (MM.java:4769-4770). */
typedef struct {
    int64_t lru_time; /* Long.MAX_VALUE when empty */
    int64_t capacity;
    int64_t used;
    int64_t start_time;
    int64_t version;
    int32_t count;
    int32_t loading_threads;
    int32_t loading_in_progress;
    int32_t rpm;
    int32_t shutting_down;
    uint32_t id_order;
    int32_t replica_set;
    int32_t pad_;
} orc_pod;

/* A view of everything CacheMissForwardingLB.getNext reads that is not
 * per-request: clusterState (MM.java:332,774), typeConstraints
 * (TypeConstraintManager.java:242-251), the litelinks live-instance map
 * (MM.java:4778) and UpgradeTracker's replaced replica sets (MM.java:4792). */
typedef struct {
    const orc_pod *pods;  /* n_rows rows, indexed by pod index                     */
    int32_t n_pods;       /* rows in clusterState = entries of order[]             */
    const int32_t *order; /* pod indices in clusterState iteration order           */
    int64_t min_space_units;
    int64_t min_churn_age_ms;
    int32_t n_types;
    /* per type: byte-per-pod membership; a NULL row means "null set"
     * (no constraint / no preference) exactly as in the Java. */
    const uint8_t *const *allowed;
    const uint8_t *const *prefer;
    const uint8_t *live; /* siMap.containsKey(iid) */
    const int32_t *replaced_rs;
    int32_t n_replaced_rs;
    int32_t n_rows;       /* size of pods[] (>= n_pods: absent rows keep their index) */
} orc_snapshot;

typedef struct {
    int32_t type;          /* exclude.modelType                              */
    int32_t self;          /* index of instanceId in pods[], -1 if absent    */
    int32_t favour_self;   /* exclude.favourSelf                             */
    uint32_t pick;         /* replaces ThreadLocalRandom: idx=(pick*n)>>32   */
    int64_t last_used;     /* exclude.lastUsedTime                           */
    int64_t now;           /* replaces currentTimeMillis()                   */
    const int32_t *loaded; /* mr.getInstanceIds().keySet()                   */
    int32_t n_loaded;
    const int32_t *failed; /* mr.getLoadFailedInstanceIds().keySet()         */
    int32_t n_failed;
    const int32_t *extra;  /* the HashSet itself ∪ explicit                  */
    int32_t n_extra;
    orc_pod fresh;         /* getFreshInstanceRecord() of the caller         */
} orc_place_req;

typedef struct {
    int32_t chosen;       /* pod index | ORC_NONE | ORC_SELF */
    int32_t best;         /* final bestIid's pod index, -1 if none           */
    int32_t n_candidates; /* candidates.size(); 0 on early returns           */
    int32_t n_remaining;  /* after the rpm filter                            */
    uint32_t hash;        /* shortlist hash, see orc_shortlist_hash          */
} orc_place_out;

/* MM.java:765-771 */
int64_t orc_min_space_units(int32_t default_model_size_units, int32_t loading_threads,
                            int64_t cap_units, int have_unload_manager);
/* InstanceRecord.java:203-205 */
int64_t orc_remaining(const orc_pod *p);
/* MM.java:4640-4642 */
int orc_is_full(int64_t avail, int64_t min_space_units);
/* MM.java:4646-4703 (ids unique ⇒ location/zone/labels never reached) */
int orc_placement_compare(const orc_pod *a, const orc_pod *b, int64_t min_space_units,
                          int64_t min_churn_age_ms);
/* Sorted iteration order of the ConcurrentSkipListSet (MM.java:774).
 * Returns 0, or -1 if the comparator is not a consistent total order on
 * these rows (SURVEY.md §7 "Transitivity"). */
int orc_sort_pods(const orc_pod *pods, int32_t n, int64_t min_space_units,
                  int64_t min_churn_age_ms, int32_t *order_out);

/* MM.java:4776-5005.  cand_out (optional, capacity n_pods) receives the
 * shortlist before the rpm filter. */
int orc_place(const orc_snapshot *s, const orc_place_req *r, orc_place_out *o, int32_t *cand_out);

/* The same algorithm without the checker's audit machinery (no per-call allocation, no pos_of rebuild,
 * no shortlist hash: o->hash = 0): the CPU port bench.py times as cpu_baseline.  scratch = 4 * (n_pods + 1)
 * ints owned by the caller. */
int orc_place_lean(const orc_snapshot *s, const orc_place_req *r, orc_place_out *o, int32_t *scratch);

/* Commutative hash of the shortlist as a bitmap over clusterState positions. */
uint32_t orc_shortlist_hash(const int32_t *pos_of, const int32_t *cand, int32_t n,
                            int32_t n_remaining, int32_t n_pods);

/* ---------- serve-target selection, MM.java:4315-4392 ------------------- */
typedef struct {
    int32_t self;
    int32_t exclude_self; /* filtered.excludeSelf */
    int32_t prefer_self;  /* filtered.preferSelf  */
    int32_t n_copies;
    const int32_t *copy_pod;    /* filteredInstances keys, in TreeMap (id) order */
    const int64_t *copy_loaded; /* values: load start time                       */
    int64_t now;
    int64_t assume_completed_ms; /* loadingTimeStats(type).assumeCompletedAfterMillis() */
    int32_t local_in_flight;     /* localInvokesInFlight.get() */
    int32_t pad_;
    int64_t last_invoke_time;    /* lastInvokeTime             */
} orc_serve_req;

/* live: siMap membership; in_use/last_used: ServiceInstance.getInUseCount /
 * getLastUsedTime per pod (MM.java:4356,4360). */
int32_t orc_serve(const orc_serve_req *r, const uint8_t *live, const int32_t *in_use,
                  const int64_t *last_used, int64_t *chosen_ts);

/* ---------- cluster stats, InstanceSetStatsTracker.java:53-92 ------------ */
typedef struct {
    int64_t total_capacity, total_free, global_lru;
    int32_t instance_count, model_copy_count;
} orc_cluster_stats;
void orc_cluster_stats_of(const orc_pod *pods, int32_t n, int64_t min_space_units,
                          orc_cluster_stats *out);

/* ---------- batch driver over flat tables (mm_oracle_batch.c) ------------ */
typedef struct {
    int32_t type, ent_off, n_loaded, n_failed;
    int64_t last_used;
} orc_flat_model;
typedef struct {
    int32_t model, self_pod;
    uint32_t flags, pick;
    int64_t last_used;
    int32_t extra_off, n_extra;
    int64_t fresh_lru, fresh_capacity, fresh_used;
    int32_t fresh_count, fresh_rpm;
} orc_flat_req;
typedef struct {
    int32_t chosen, best, n_candidates;
    uint32_t hash;
} orc_flat_out;
/* n decisions split over n_threads pthreads; lat_ns (optional) gets the wall
 * time of every single orc_place call. */
int orc_place_batch(const orc_snapshot *snap, const orc_flat_model *models, const int32_t *ent_pod,
                    const orc_flat_req *reqs, const int32_t *extra, int32_t n, int64_t now,
                    orc_flat_out *outs, int32_t n_threads, double *lat_ns);

/* cpu_baseline: a persistent pool of n_threads workers (created once, parked on a condition variable between
 * batches) running orc_place_lean over dynamically dealt chunks of the request table. */
typedef struct orc_pool orc_pool;
orc_pool *orc_pool_create(int32_t n_threads, int32_t max_pods);
void orc_pool_destroy(orc_pool *p);
int orc_pool_place(orc_pool *p, const orc_snapshot *snap, const orc_flat_model *models, const int32_t *ent_pod,
                   const orc_flat_req *reqs, const int32_t *extra, int32_t n, int64_t now, orc_flat_out *outs,
                   double *lat_ns);

/* ---------- local LRU cache + eviction (clhm) --------------------------- */
typedef struct {
    int64_t last_used;
    int32_t weight;
    int32_t key;
} orc_node;

/* nodes[] is the evictionDeque, oldest (head) first. */
typedef struct {
    orc_node *nodes;
    int32_t n, cap_nodes;
    int64_t weighted_size;
    int64_t capacity;
    int64_t oldest_time; /* the map's oldestTime FIELD (clhm :1120): stored by updateOldestTime() only (afterWrite :444,
                            tryToDrainBuffers :463) — setCapacity (:305-316) evicts without storing it */
} orc_cache;

void orc_cache_init(orc_cache *c, int64_t capacity);
void orc_cache_free(orc_cache *c);
int32_t orc_cache_find(const orc_cache *c, int32_t key);
int32_t orc_cache_put_if_absent(orc_cache *c, int32_t key, int32_t weight, int64_t last_used,
                                int64_t now, int32_t *victims, int32_t max_victims,
                                int32_t *insert_pos);
int orc_cache_get(orc_cache *c, int32_t key, int64_t last_used, int64_t now);
int32_t orc_cache_update_weight(orc_cache *c, int32_t key, int32_t new_weight, int64_t new_time,
                                int64_t now, int32_t *victims, int32_t max_victims);
int orc_cache_remove(orc_cache *c, int32_t key);
int64_t orc_cache_oldest_time(const orc_cache *c);

typedef struct {
    int32_t insert_pos, n_victims, self_evicted, pad_;
    int64_t weighted_size, oldest_time;
} orc_evict_result;
void orc_evict_eval(const int64_t *lu, const int32_t *wt, int32_t n, int64_t capacity, int32_t weight,
                    int64_t last_used, int64_t now, orc_evict_result *out);

/* ---------- request-level guards (mm_gates_oracle.c) ---------------------- */
int orc_go_local(const int32_t *copy_pod, const int64_t *copy_loaded, int32_t n, int32_t self,
                 int favour_self_for_hits, int have_cache_entry, int entry_done, int64_t now);
int orc_load_failures_breached(const int64_t *fail_time, int32_t n, int64_t now, int64_t in_use_expiry_ms);
int orc_load_locations_breached(const int32_t *loaded_pod, int32_t n, const int32_t *explicit_excl,
                                int32_t n_explicit, const uint8_t *in_table);
int orc_churn_reject(int64_t min_churn_age_ms, int64_t min_space_units, int64_t cache_capacity,
                     int64_t cache_weighted_size, int64_t cache_oldest_time, int64_t now);
int32_t orc_load_local_initial_size(int have_size_hint, int32_t size_hint, int32_t loading_count,
                                    int32_t weight_predict_cutoff, int32_t loader_predicted,
                                    const orc_cluster_stats *stats, int we_created_entry, int64_t last_used_time,
                                    int64_t cache_capacity, int64_t cache_weighted_size, int64_t cache_oldest_time,
                                    int *reject);
int orc_reload_elsewhere(int entry_failed, int64_t loaded_time, int64_t load_timeout_ms, int64_t now,
                         const orc_cluster_stats *stats);
int orc_should_publish(const orc_pod *cur, const orc_pod *fresh, int64_t now, int64_t last_published,
                       int force, int pre_shutdown, int64_t min_space_units);

/* ---------- batch rebalancers (mm_rebalance_oracle.c) --------------------- */
typedef struct {
    int32_t size_estimate; /* sizeEstimate, MM.java:6622-6629             */
    int32_t free_count;    /* freeSpaceProactiveLoadCount, :6651           */
    int32_t total_count;   /* totalProactiveLoadCount, :6655               */
    int32_t n_candidates;  /* |proactiveLoadCandidates|, :6574-6577        */
    int32_t n_selected;    /* ensureLoadedInternal calls made, :6709-6734  */
    int32_t error;         /* 1: sizeEstimate == 0 (ArithmeticException)   */
    int64_t space_to_fill; /* after the /2, :6650                          */
    int64_t cutoff;        /* proactiveLastUsedCutoff, :6662-6664          */
} orc_proactive_info;
int32_t orc_proactive_plan(const orc_pod *pods, int32_t n_pods, const orc_cluster_stats *global,
                           const orc_cluster_stats *stats, const uint8_t *in_subset, const uint64_t *prohibited,
                           int32_t n_types, const uint8_t *skip, const orc_flat_model *models, int32_t n_models,
                           int32_t default_model_size_units, int64_t now, int32_t *out_model, int64_t *out_last_used,
                           int32_t max_out, orc_proactive_info *info);

/* one local CacheEntry as the rebalancers see it */
typedef struct {
    int32_t model;                 /* registry row, -1 if registry.get(modelId) == null */
    int32_t weight;                /* ce.getWeight()                                    */
    int64_t last_used;             /* cache last-used time of the entry                 */
    int64_t interval_count;        /* getAndResetIntervalCount() / getIntervalCount()   */
    int64_t last_heavy_time;       /* ce.getLastHeavyTime()                             */
    int64_t last_unload_time;      /* mr.getLastUnloadTime()                            */
    int32_t earlier_use_iteration; /* ce.earlierUseIteration                            */
    int32_t last_used_iteration;   /* ce.lastUsedIteration                              */
    uint32_t flags;                /* bit0 failed / gone                                */
    int32_t pad_;
} orc_cache_entry;
typedef struct {
    int32_t self_pod, iteration_counter, second_copy_max_age_iters, second_copy_min_age_iters;
    int32_t scale_up_rpm_threshold, our_rpm;
    int64_t now, last_check_time, rate_check_interval_ms, second_copy_lru_threshold_ms, assume_completed_ms;
} orc_scaleup_params;
typedef struct {
    int32_t action; /* 0 none, 1 second copy (exclude self, lastUsed = lastCheckTime), 2 scale up */
    int32_t copies; /* copiesToLoad                                                                */
    int64_t timestamp;
    int32_t new_i1, new_i2;
    int32_t heavy; /* ce.setLastHeavyTime(now) */
    int32_t rpm;
} orc_scaleup_out;
int orc_scaleup_plan(const orc_pod *pods, int32_t n_pods, const int32_t *order, int32_t n_order,
                     const orc_cluster_stats *stats, const orc_cluster_stats *type_stats, int32_t t_rows, int has_tc,
                     const orc_flat_model *models, const int32_t *ent_pod,
                     const int64_t *ent_time, const orc_cache_entry *entries, int32_t n, const orc_scaleup_params *p,
                     orc_scaleup_out *outs, uint8_t *overloaded_out);
/* limitModelConcurrency == true: MaxConcCacheEntry (MM.java:2641-2797), one row per cache entry */
#define ORC_CONC_COUNT_BITS 21 /* MM.java:2653 */
typedef struct {
    int64_t count_and_time_sum; /* countAndTimeSum.sum() */
    int64_t prior_sum;          /* priorSum */
    int32_t prior_count;        /* priorCount */
    int32_t max_conc;           /* maxConc */
    int32_t queued_requests;    /* queuedRequestCount() */
    int32_t pad_;
} orc_conc_entry;
typedef struct {
    int32_t threshold; /* getRpmScaleThreshold(true) as evaluated, 0 if the entry was skipped before the call */
    int32_t reset;     /* the call took sumThenReset() */
    int64_t new_prior_sum;
    int32_t new_prior_count, pad_;
} orc_conc_out;
typedef struct {
    int64_t dynamic_rpm_scale_constant; /* MM.java:370 */
    double average_model_parallelism;   /* the task's field before the run, MM.java:5634 */
} orc_conc_params;
typedef struct {
    double average_model_parallelism; /* after the run, MM.java:5815-5818 */
    int32_t exclude_set_rpms;         /* (int) (900.0 * averageModelParallelism), MM.java:5836 */
    int32_t model_parallelism_sum;    /* MM.java:5706 */
} orc_conc_result;
/* MaxConcCacheEntry.getRpmScaleThreshold, MM.java:2766-2796 */
int32_t orc_rpm_scale_threshold(const orc_conc_entry *m, int and_reset, int32_t scale_up_rpm_threshold, int64_t dyn_const, orc_conc_out *o);
/* the rate task with limitModelConcurrency == true (latencyBased, MM.java:5677) */
int orc_scaleup_plan_conc(const orc_pod *pods, int32_t n_pods, const int32_t *order, int32_t n_order,
                          const orc_cluster_stats *stats, const orc_cluster_stats *type_stats, int32_t t_rows, int has_tc,
                          const orc_flat_model *models, const int32_t *ent_pod,
                          const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                          const orc_scaleup_params *p, const orc_conc_params *cp, orc_scaleup_out *outs, orc_conc_out *conc_outs,
                          uint8_t *overloaded_out, orc_conc_result *result);
typedef struct {
    int32_t self_pod, shutting_down;
    int64_t now, last_check_time, rate_check_interval_ms, adjusted_cache_capacity;
    int32_t scale_up_rpm_threshold, pad_;
} orc_scaledown_params;
/* the janitor with MaxConcCacheEntry entries (MM.java:6294-6305) */
void orc_scaledown_plan_conc(const orc_pod *pods, const int32_t *pos_of, const uint8_t *in_table,
                             const orc_cluster_stats *stats, const orc_flat_model *models, const int32_t *ent_pod,
                             const int64_t *ent_time, const orc_cache_entry *entries, const orc_conc_entry *conc, int32_t n,
                             const orc_scaledown_params *p, int64_t dyn_const, uint8_t *removed_out);
void orc_scaledown_plan(const orc_pod *pods, const int32_t *pos_of, const uint8_t *in_table,
                        const orc_cluster_stats *stats, const orc_flat_model *models, const int32_t *ent_pod,
                        const int64_t *ent_time, const orc_cache_entry *entries, int32_t n,
                        const orc_scaledown_params *p, uint8_t *removed_out);
void orc_migration_plan(const orc_flat_model *models, const int32_t *ent_pod, const orc_cache_entry *entries,
                        int32_t n, int32_t self_pod, int64_t now, int64_t cutoff_age_ms, uint8_t *action_out,
                        uint8_t *wait_out);

/* ---------- unload-buffer accounting (ModelCacheUnloadBufManager.java) -- */
#define ORC_UBM_MAX_EVICTED 1024
typedef struct {
    orc_cache *cache;
    int32_t reserved;         /* unloadsReservedSizeUnits */
    int32_t total_unloading;  /* totalUnloadingWeight     */
    int64_t total_occupancy;  /* totalModelCacheOccupancy */
    int32_t cache_deficit;    /* cacheDeficit             */
    int32_t n_evicted;
    int32_t evicted[ORC_UBM_MAX_EVICTED]; /* keys in eviction order */
} orc_ubm;
void orc_ubm_init(orc_ubm *u, orc_cache *cache, int32_t reserved, int64_t now);
int32_t orc_ubm_buffer_weight(const orc_ubm *u);
int orc_ubm_insert_new_entry(orc_ubm *u, int32_t key, int32_t weight, int64_t last_used, int64_t now);
void orc_ubm_adjust_new_entry_space_request(orc_ubm *u, int32_t increase, int32_t key, int64_t now);
int orc_ubm_cache_space_is_ready(const orc_ubm *u, int32_t required);
int orc_ubm_claim_requested_space_if_ready(orc_ubm *u, int32_t required, int64_t now);
void orc_ubm_adjust_weight_after_load(orc_ubm *u, int32_t delta, int32_t key, int64_t now);
void orc_ubm_unload_complete(orc_ubm *u, int32_t weight, int success, int64_t now);
int32_t orc_ubm_remove_entry(orc_ubm *u, int32_t key, int64_t now);
void orc_ubm_discard_failed_entry(orc_ubm *u, int32_t weight, int64_t now);
int orc_ubm_insert_failed_placeholder_entry(orc_ubm *u, int32_t key, int32_t weight, int64_t last_used, int64_t now);

#ifdef __cplusplus
}
#endif
#endif
