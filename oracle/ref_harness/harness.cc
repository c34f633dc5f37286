// harness.cc — runs the REFERENCE'S OWN method bodies (extracted by extract.py into oracle/_ref/gen/*.inc, compiled
// against the stand-ins of javastub.hpp) over fleets written by make_ref_vectors.py, and writes what they decided.
//
// TEST INFRASTRUCTURE.  This file supplies only what surrounds those bodies in ModelMesh.java: the declarations the
// bodies sit in (each cites the line of the Java declaration it stands for), the fields of the enclosing ModelMesh
// instance as globals, and I/O.  It contains no placement logic: which instance is chosen, in which order instances
// are visited, what is filtered — all of that is `#include`d reference text.
//
// usage: ref_harness <input.bin> <output.bin>     (format: make_ref_vectors.py)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "javastub.hpp"
#include "../../include/mmplace.h"

std::vector<std::shared_ptr<void>> g_lists_created;

// ---- TypeConstraintManager.ProhibitedTypeSet (TypeConstraintManager.java:295-333): a sorted array of type names with
// contains() and equals() ------------------------------------------------------------------------------------------------
class ProhibitedTypeSet {
    std::shared_ptr<std::vector<std::string>> p;  // sorted

public:
    ProhibitedTypeSet() {}
    ProhibitedTypeSet(std::nullptr_t) {}
    explicit ProhibitedTypeSet(std::vector<std::string> types) : p(std::make_shared<std::vector<std::string>>(std::move(types))) { std::sort(p->begin(), p->end()); }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    template <class L, class = decltype(std::declval<L>().size()), class = decltype(std::declval<L>().get(0))>
    explicit ProhibitedTypeSet(const L &types) : p(std::make_shared<std::vector<std::string>>())  // ProhibitedTypeSet(Collection<String>), :299-303
    {
        for (int i = 0; i < types.size(); i++) p->push_back(types.get(i).str());
        std::sort(p->begin(), p->end());
    }
    int size() const { return (int)p->size(); }                                                              // :305-307
    const std::vector<std::string> &types() const { return *p; }
    boolean contains(const String &type) const { return std::binary_search(p->begin(), p->end(), type.str()); }  // :309-311
    boolean equals(const ProhibitedTypeSet &o) const { return o.p && *p == *o.p; }                                  // :314-317
};

// ---- com.ibm.watson.modelmesh.InstanceRecord (InstanceRecord.java:37-69): a record with getters -------------------
class InstanceRecord {
    struct Rep {
        long lruTime, capacity, used, instanceVersion;
        int count, loadingThreads, loadingInProgress, reqsPerMinute;
        boolean shuttingDown;
        StringArray labels;
    };
    std::shared_ptr<Rep> p;

public:
    ProhibitedTypeSet prohibitedTypes;  // InstanceRecord.java: transient, set by the TypeConstraintManager (only row a17 reads it here)
    InstanceRecord() {}
    InstanceRecord(long lru, long cap, long used_, long vers, int cnt, int lt, int lip, int rpm, boolean sd)
        : p(std::make_shared<Rep>(Rep{lru, cap, used_, vers, cnt, lt, lip, rpm, sd, StringArray()})) {}
    // InstanceRecord.java:97-109 (startTime, version, location, zone, labels, lruTime, count, capacity, used, lThreads, lInProg, shuttingDown)
    InstanceRecord(long, long vers, const String &, const String &, const StringArray &, long lru, int cnt, long cap, long used_, int lt, int lip,
                   boolean sd)
        : p(std::make_shared<Rep>(Rep{lru, cap, used_, vers, cnt, lt, lip, 0, sd, StringArray()})) {}
    InstanceRecord(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    long getUsed() const { return p->used; }
    long a19_startTime = 0;  // InstanceRecord.startTime (row a19 reads it; the record's other users never do)
    long getStartTime() const { return a19_startTime; }
    bool operator==(const InstanceRecord &o) const { return p == o.p; }  // Java `==` on references: identity
    long getLruTime() const { return p->lruTime; }
    long getCapacity() const { return p->capacity; }
    long getInstanceVersion() const { return p->instanceVersion; }
    int getCount() const { return p->count; }
    int getLoadingThreads() const { return p->loadingThreads; }
    int getLoadingInProgress() const { return p->loadingInProgress; }
    int getReqsPerMinute() const { return p->reqsPerMinute; }
    boolean isShuttingDown() const { return p->shuttingDown; }
    String getLocation() const { return null; }
    String getZone() const { return null; }
    StringArray getLabels() const { return p->labels; }
    void a18_setLabels(const StringArray &l) const { p->labels = l; }
    long getRemaining() const  // InstanceRecord.java:203
    {
        const long capacity = p->capacity, used = p->used;
#include "../_ref/gen/getRemaining_body.inc"
    }
};

// ---- litelinks ServiceInstanceInfo / ServiceInstance: what the LBs read of them (MM.java:4356, :4360) --------------
class ServiceInstanceInfo {
public:
    struct Rep { String id; int pod, inUse; long lastUsed; };
    std::shared_ptr<Rep> p;
    ServiceInstanceInfo() {}
    ServiceInstanceInfo(std::nullptr_t) {}
    ServiceInstanceInfo(const String &id, int pod, int in_use, long last_used) : p(std::make_shared<Rep>(Rep{id, pod, in_use, last_used})) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    String getInstanceId() const { return p->id; }
};
class ServiceInstance {
    ServiceInstanceInfo s;

public:
    ServiceInstance(const ServiceInstanceInfo &i) : s(i) {}  // the Java cast (ServiceInstance<?>) sii
    int getInUseCount() const { return s.p->inUse; }
    long getLastUsedTime() const { return s.p->lastUsed; }
};
// what getNext returns: null, LoadBalancer.ABORT_REQUEST, or the chosen ServiceInstanceInfo
struct Object {
    int kind = 0;  // 0 null, 1 ABORT_REQUEST, 2 an instance
    ServiceInstanceInfo sii;
    Object() {}
    Object(std::nullptr_t) {}
    Object(const ServiceInstanceInfo &s) : kind(s == null ? 0 : 2), sii(s) {}
    static Object abort_request() { Object o; o.kind = 1; return o; }
};
typedef Object T;  // <T> T getNext(...)
struct ObjectArray {};
static const struct { Object ABORT_REQUEST = Object::abort_request(); } LoadBalancer;

// ---- fields of the enclosing ModelMesh instance, set per fleet / per request by main() --------------------------------
static long minSpaceUnits, minChurnAgeMs;           // MM.java:765-771, :697
static String instanceId;                           // MM.java:338
// MMP_REF_SEND_DEST=1 (make_ref_vectors.py's second pass over two cases): the mesh tells the chosen instance its own id through the
// request context (:4997-4999, :4386-4388) — a side effect behind the decision; the pass must give the same decisions
static boolean sendDestinationId = getenv("MMP_REF_SEND_DEST") != nullptr;
static long g_now;
static long currentTimeMillis() { return g_now; }
static InstanceRecord g_fresh;
static InstanceRecord getFreshInstanceRecord() { return g_fresh; }  // MM.java:5369 (the request carries the row)
static uint32_t g_pick;
static const struct {
    struct R { int nextInt(int n) const { return (int)(((uint64_t)g_pick * (uint64_t)(uint32_t)n) >> 32); } };
    R current() const { return R(); }
} ThreadLocalRandom;  // :4981: the pick is an input of every restatement (SURVEY B#9)
static const struct {
    void warn(const String &) const {}
    void info(const String &) const {}
    void debug(const String &) const {}
    boolean isDebugEnabled() const { return true; }  // the reference's debug statements run too (they build a message and discard it)
} logger;
static long nanoTime() { return 0; }
static const String CACHE_MISS_EXCLUDES_KEY("tas.cm_excludes"), DEST_INST_ID_KEY("tas.dest_iid");
struct ThreadContextT { int getCurrentContext() const { return 0; } };
static const ThreadContextT ThreadContext;
static Map<String, String> ensureContextMapIsMutable(int) { return Map<String, String>::make(); }

static std::shared_ptr<std::vector<Entry<String, InstanceRecord>>> g_cluster;  // clusterState, kept in PLACEMENT_ORDER
static const struct {
    Iterator<Entry<String, InstanceRecord>> iterator() const { return iterate(g_cluster); }
    std::vector<Entry<String, InstanceRecord>>::const_iterator begin() const { return g_cluster->begin(); }
    std::vector<Entry<String, InstanceRecord>>::const_iterator end() const { return g_cluster->end(); }
} clusterState;
static Map<String, ServiceInstanceInfo> g_simap;
static Map<String, ServiceInstanceInfo> getMap(const ObjectArray &) { return g_simap; }  // MM.java:3301

// TypeConstraintManager.getCandidateInstances / getPreferredInstances (TypeConstraintManager.java:242-251): per type a
// set of instance ids or null; `typeConstraints` itself is null without a type table (MM.java:4789)
class TypeConstraintsT {
public:
    std::shared_ptr<std::map<std::string, std::pair<Set<String>, Set<String>>>> p;
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    Set<String> getCandidateInstances(const String &type) const { return (*p)[type.str()].first; }
    Set<String> getPreferredInstances(const String &type) const { return (*p)[type.str()].second; }
};
static TypeConstraintsT typeConstraints;
static ObjectLongMap<String> g_replaced;
static const struct { ObjectLongMap<String> getLikelyReplacedReplicaSets() const { return g_replaced; } } upgradeTracker;

// ---- MM.java:4640 isFull, :4162 age ------------------------------------------------------------------------------
static boolean isFull(long availableUnits)
{
#include "../_ref/gen/isFull_body.inc"
}
static long age(long timeMillis)
{
#include "../_ref/gen/age_body.inc"
}
// Utils.java:25 STRING_ARRAY_COMP
static const struct {
    struct {
        int operator()(const StringArray &l1, const StringArray &l2) const
        {
#include "../_ref/gen/string_array_comp_body.inc"
        }
    } STRING_ARRAY_COMP;
} Utils;
// MM.java:4751-4753
#include "../_ref/gen/lb_constants.inc"

// ---- MM.java:4646 PLACEMENT_ORDER = new Comparator<>() { public int compare(e1, e2) { ... } } -----------------------
static int placement_order_compare(Entry<String, InstanceRecord> e1, Entry<String, InstanceRecord> e2)
{
#include "../_ref/gen/placement_order_compare_body.inc"
}

// ---- MM.java:4717 CacheMissExcludeSet extends HashSet<String> -------------------------------------------------------
struct CacheMissExcludeSet {
    Set<String> self_ = Set<String>::make();  // the HashSet part
    String modelType;
    boolean favourSelf = false;
    Set<String> loaded = Set<String>::make(), failed = Set<String>::make();
    Collection<String> explicit_;
    long lastUsedTime = 0;
    boolean contains(const String &s) const { return self_.contains(s); }
    boolean isEmpty() const { return self_.isEmpty(); }
    boolean add(const String &s) const { return self_.add(s); }
    boolean isExcluded(String instanceId) const  // :4740
    {
#include "../_ref/gen/isExcluded_body.inc"
    }
};
static CacheMissExcludeSet g_exclude;
static const struct { CacheMissExcludeSet get() const { return g_exclude; } } cacheMissExcludeTl;  // :4755
static const struct { String join(const CacheMissExcludeSet &) const { return String(""); } } COMMA_JOIN;

// ---- MM.java:4757 class CacheMissForwardingLB extends IdBasedLoadBalancer ----------------------------------------------
struct CacheMissForwardingLB {
    // :4760
    Iterator<Entry<String, InstanceRecord>> filter(Set<String> constrainTo, CacheMissExcludeSet exclude, Map<String, ServiceInstanceInfo> siMap,
                                                   ObjectLongMap<String> excludeReplicaSets)
    {
#include "../_ref/gen/filter_body.inc"
    }
    // :4776 public <T> T getNext(Object[] sis, String method, Object[] args)
    T getNext(ObjectArray sis, String method, ObjectArray args)
    {
#include "../_ref/gen/cachemiss_getNext_body.inc"
    }
};

// ---- MM.java:4265 MapFilteringSet<K, V> extends HashMap<Entry<K, V>, Boolean> implements Predicate<Entry<K, V>> -------
template <class K, class V> struct MapFilteringSet {
    boolean excludeSelf = false, preferSelf = false;
    String modelType;
    std::shared_ptr<std::vector<std::pair<std::string, long>>> tried = std::make_shared<std::vector<std::pair<std::string, long>>>();
    Map<K, V> map_;
    Collection<K> keyExcludes;
    boolean containsKey(const Entry<K, V> &e) const
    {
        for (auto &t : *tried)
            if (t.first == e.getKey().str() && t.second == (long)e.getValue()) return true;
        return false;
    }
    boolean apply(Entry<K, V> input) const  // :4281
    {
#include "../_ref/gen/mapfilteringset_apply_body.inc"
    }
    boolean add(const K &key, const V &value) const  // :4285 put(immutableEntry(key, value), TRUE) == null
    {
        tried->emplace_back(key.str(), (long)value);
        return true;
    }
    Map<K, V> map() const { return map_; }
    Map<K, V> filteredMap(const Map<K, V> &source)  // :4293 Maps.filterEntries(source, this)
    {
        map_ = Map<K, V>::make();
        for (auto &e : source.entrySet())
            if (apply(e)) map_.put(e.getKey(), e.getValue());
        return map_;
    }
};
static MapFilteringSet<String, Long> g_filtered;
static const struct { MapFilteringSet<String, Long> get() const { return g_filtered; } } cacheHitExcludeTl;  // :4307
static int g_local_in_flight;
static const struct { int get() const { return g_local_in_flight; } } localInvokesInFlight;  // :4303
static long lastInvokeTime;                                                                     // :4304
static long g_assume_completed;
struct TimeStatsT { long assumeCompletedAfterMillis() const { return g_assume_completed; } };
static TimeStatsT loadingTimeStats(const String &) { return TimeStatsT(); }  // TimeStats.java:50-68: an input here

// ---- MM.java:4309 class ForwardingLB extends IdBasedLoadBalancer ----------------------------------------------------------
struct ForwardingLB {
    T getNext(ObjectArray sis, String method, ObjectArray args)  // :4315
    {
#include "../_ref/gen/forwarding_getNext_body.inc"
    }
};

// ================================ the request-level guards ==============================
// Each function below is the Java declaration around one extracted fragment of invokeModel / loadLocal / onEviction /
// publishInstanceRecord; the locals and fields the fragment reads are declared from the request (mmp_gate_req).
struct TException {
    bool isnull = true;
    TException() {}
    TException(std::nullptr_t) {}
    bool operator!=(std::nullptr_t) const { return !isnull; }
};
struct ModelLoadException : TException {
    ModelLoadException() {}
    ModelLoadException(std::nullptr_t) {}
    ModelLoadException(const String &, const String &, long, std::nullptr_t) { isnull = false; }
};
static ModelLoadException newModelLoadException(const String &, long, std::nullptr_t) { return ModelLoadException(String(""), null, 0L, null); }
static TException newInternalException(const String &, std::nullptr_t) { TException t; t.isnull = false; return t; }
struct ClusterStats {  // MM.java:1570-1590
    long totalCapacity = 0, totalFree = 0, globalLru = 0;
    int instanceCount = 0, modelCopyCount = 0;
    ClusterStats() {}
    ClusterStats(long cap, long free_, long lru, int n, int copies) : totalCapacity(cap), totalFree(free_), globalLru(lru), instanceCount(n), modelCopyCount(copies) {}
};
static std::vector<ClusterStats> g_tstats;
static ClusterStats clusterStats;  // MM.java:1570 (the cluster-wide stats)
static ClusterStats typeSetStats(const String &type)  // :1432 — "t<row>"; "t-1": an entry without a registry record (no type: the cluster's stats)
{
    const long row = std::stol(type.str().substr(1));
    return row < 0 ? clusterStats : g_tstats.at(std::min<size_t>((size_t)row, g_tstats.size() - 1));
}

// ModelRecord (ModelRecord.java:61-114): type + the two id -> time maps
struct ModelRecord {
    String type;
    Map<String, Long> instanceIds = Map<String, Long>::make(), failed = Map<String, Long>::make();
    bool isnull = false;
    ModelRecord() {}
    ModelRecord(std::nullptr_t) : isnull(true) {}
    bool operator!=(std::nullptr_t) const { return !isnull; }
    bool operator==(std::nullptr_t) const { return isnull; }
    boolean loadFailedInInstance(const String &iid) const { return failed.containsKey(iid); }  // ModelRecord.java:208
    long lastUnloadTime = 0, lastUsed = 0;
    long getLastUnloadTime() const { return lastUnloadTime; }
    long getLastUsed() const { return lastUsed; }
    String getType() const { return type; }
    Map<String, Long> getInstanceIds() const { return instanceIds; }
    Map<String, Long> getLoadFailedInstanceIds() const { return failed; }
    boolean hasLoadFailure() const { return !failed.isEmpty(); }  // ModelRecord.java:203
    String getLoadFailureMessage(const String &) const { return String(""); }
};
struct IntField {  // an assignable int field of an object with reference semantics (ce.earlierUseIteration = i2)
    std::shared_ptr<int> p = std::make_shared<int>(0);
    operator int() const { return *p; }
    const IntField &operator=(int v) const { *p = v; return *this; }
};
struct ConcState {  // MaxConcCacheEntry's own fields (see below)
    long countAndTimeSum = 0, priorSum = 0;
    int priorCount = 0, maxConc = 1, queued = 0;
    int resets = 0, lastThreshold = 0;  // observed: sumThenReset() calls, what getRpmScaleThreshold returned last
};
struct CacheEntry {  // what the fragments ask of a CacheEntry<?>
    std::shared_ptr<ConcState> conc;  // non-null: the object is a MaxConcCacheEntry
    bool isnull = true, done = false, failed = false;
    int predicted = 0;
    String modelInfoType;
    CacheEntry() {}
    CacheEntry(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
    boolean isDone() const { return done; }
    boolean isFailed() const { return failed; }
    int loaderPredictedWeight() const { return predicted; }
    void remove() const {}
    struct MI { String t, serviceType; String getServiceType() const { return t; } } modelInfo;
    // the rate-tracking fields (CacheEntry's usage slices, MM.java:5729-5741)
    IntField earlierUseIteration, lastUsedIteration;
    std::shared_ptr<long> heavy = std::make_shared<long>(0);  // lastHeavyTime as set by this run (0: untouched)
    long intervalCount = 0;
    int weight = 0;
    long getAndResetIntervalCount() const { return intervalCount; }
    long getIntervalCount() const { return intervalCount; }
    long lastHeavyTimeIn = 0;
    long getLastHeavyTime() const { return lastHeavyTimeIn; }
    long getRpm(long timeSinceLastCheck) const  // MM.java:1738
    {
#include "../_ref/gen/getRpm_body.inc"
    }
    void setLastHeavyTime(long t) const { *heavy = t; }
    int getWeight() const { return weight; }
};
// MaxConcCacheEntry (MM.java:2641-2797), limitModelConcurrency == true: the cache entries of such a mesh ARE MaxConcCacheEntry objects.
// The state the rate task and the janitor read lives in the CacheEntry stand-in (ConcState, shared by the handle copies); this
// class is the `(MaxConcCacheEntry<?>) ce` view of it, and getRpmScaleThreshold's body is the reference's text.
static int scaleUpRpmThreshold;                 // MM.java:240 / :369 (read by getRpmScaleThreshold, :2781, and the rate task)
static long dynamicRpmScaleConstant;            // MM.java:370
struct LongAdderStub {  // java.util.concurrent.atomic.LongAdder, single-threaded
    std::shared_ptr<ConcState> st;
    long sum() const { return st->countAndTimeSum; }
    long sumThenReset() const { const long v = st->countAndTimeSum; st->countAndTimeSum = 0; st->resets++; return v; }
};
struct MaxConcCacheEntry {
#include "../_ref/gen/mcce_count_bits.inc"
#include "../_ref/gen/mcce_count_mask.inc"
    std::shared_ptr<ConcState> st;
    int maxConc = 1;
    LongAdderStub countAndTimeSum;
    int &priorCount;
    long &priorSum;
    static ConcState &dummy() { static ConcState d; return d; }
    MaxConcCacheEntry(const CacheEntry &ce) : st(ce.conc), maxConc(ce.conc ? ce.conc->maxConc : 1), countAndTimeSum{ce.conc},
                                              priorCount(ce.conc ? ce.conc->priorCount : dummy().priorCount),
                                              priorSum(ce.conc ? ce.conc->priorSum : dummy().priorSum) {}
    MaxConcCacheEntry(std::nullptr_t) : priorCount(dummy().priorCount), priorSum(dummy().priorSum) {}
    bool operator==(std::nullptr_t) const { return !st; }
    bool operator!=(std::nullptr_t) const { return (bool)st; }
    int getRpmScaleThreshold(boolean andReset)  // :2766
    {
        const int rc = getRpmScaleThreshold_(andReset);
        st->lastThreshold = rc;
        return rc;
    }
    int getRpmScaleThreshold_(boolean andReset)
    {
#include "../_ref/gen/mcce_getRpmScaleThreshold_body.inc"
    }
    int queuedRequestCount() const { return st->queued; }  // :2746 (limiter.getQueueLength(): an input)
};
static boolean instanceof_MaxConcCacheEntry(const CacheEntry &ce) { return (bool)ce.conc; }
static const mmp_gate_req *g_gq;  // the request whose fragment is running
static CacheEntry g_cache_entry;
static CacheEntry getFromCache(const String &, long) { return g_cache_entry; }  // MM.java:3609
static const struct {
    long capacity() const { return g_cap; }
    long weightedSize() const { return g_wsize; }
    long oldestTime() const { return g_oldest; }
    int size() const { return g_size; }
    mutable long g_cap = 0, g_wsize = 0, g_oldest = 0;
    mutable int g_size = 0;
} runtimeCache;
static std::vector<uint8_t> g_in_table;
static std::unordered_map<std::string, int> g_pod_of;
static std::vector<InstanceRecord> g_table_rec;  // the table's record per pod (null: not in the table)
static const struct {
    boolean contains(const String &id) const { auto it = g_pod_of.find(id.str()); return it != g_pod_of.end() && g_in_table[it->second]; }
    InstanceRecord getOrStrongIfAbsent(const String &id) const
    {
        auto it = g_pod_of.find(id.str());
        return it == g_pod_of.end() ? InstanceRecord(null) : g_table_rec[it->second];
    }
    InstanceRecord get(const String &id) const { return getOrStrongIfAbsent(id); }  // instanceInfo.get(iid): null when not in the table
    int keyIterable() const { return 0; }
} instanceInfo;
static const struct { String toString(int) const { return String(""); } } Iterables;
static long IN_USE_LOAD_FAILURE_EXPIRY_MS;  // MM.java:221 (a parameter: LOAD_FAILURE_EXPIRY_MS / 2)
#include "../_ref/gen/failure_constants.inc"
#include "../_ref/gen/publish_constants.inc"

static long oldest_of(Map<String, Long> instances)  // MM.java:4166 oldest()
{
#include "../_ref/gen/oldest_body.inc"
}
static String getMostRecent(Map<String, Long> instances)  // :4629
{
#include "../_ref/gen/getMostRecent_body.inc"
}
// invokeModel, the cache-hit branch (:3599-3626): is the hit served locally?
static boolean frag_goLocal(Map<String, Long> filteredInstances, boolean favourSelfForHits, String modelId, long lastUsedTime)
{
    CacheEntry cacheEntry = null;
    boolean result = false;
#include "../_ref/gen/goLocal_fragment.inc"
        result = goLocal;
    }
    return result;
}
static void checkLoadLocationCount(ModelRecord mr, Collection<String> explicitExcludes, TException internalFailureSeen)  // :4589
{
#include "../_ref/gen/checkLoadLocationCount_body.inc"
}
static void checkLoadFailureCount(ModelRecord mr, ModelLoadException loadFailureSeen)  // :4607
{
#include "../_ref/gen/checkLoadFailureCount_body.inc"
}
static void throwIfLocalLoadNotAllowed(String modelId, boolean externalReq, ModelRecord mr, CacheMissExcludeSet *loadTargetFilter_,
                                       ModelLoadException loadFailureSeen, TException internalFailureSeen)  // :4003
{
    struct { CacheMissExcludeSet *p; bool operator!=(std::nullptr_t) const { return p != nullptr; } boolean isExcluded(const String &s) const { return p->isExcluded(s); } } loadTargetFilter{loadTargetFilter_};
#include "../_ref/gen/throwIfLocalLoadNotAllowed_body.inc"
}
static void frag_churn(String modelId)  // invokeModel :3872-3884
{
#include "../_ref/gen/churn_fragment.inc"
}
static int loadingThreads;
static const struct { int get() const { return g_v; } mutable int g_v = 0; } loadingCount;
static int weightPredictCutoff()  // :5013 — the request carries the value (mmp_gate_req::weight_predict_cutoff); the body is compiled for the record
{
    if (g_gq) return g_gq->weight_predict_cutoff;
#include "../_ref/gen/weightPredictCutoff_body.inc"
}
static const String KNOWN_SIZE_CXT_KEY("tas.known_size");
static void logCacheFallthru(const String &, long, long, long, int) {}
static int g_initial_size;
// loadLocal :5159-5197: the initial size of the new entry, and the early reject (returns null)
static CacheEntry frag_loadLocal_sizing(CacheEntry ce, ModelRecord mr, Map<String, String> contextMap, boolean weCreatedCacheEntry, long lastUsedTime,
                                        String modelId, long now)
{
#include "../_ref/gen/loadLocal_sizing_fragment.inc"
    g_initial_size = initialSize;
    return ce;
}
static long loadTimeoutMs;
static ModelRecord g_registry_mr;
static const struct { ModelRecord get(const String &) const { return g_registry_mr; } } registry;
// onEviction :2886-2897 and :2918-2920: is the evicted model re-placed elsewhere?
static boolean frag_onEviction(CacheEntry ce, String key, long now)
{
    boolean reload = false;
#include "../_ref/gen/onEviction_attempt_fragment.inc"
    if (attemptReload) {
#include "../_ref/gen/onEviction_cluster_fragment.inc"
            reload = true;
        }
    }
    return reload;
}
static boolean loadingChange(InstanceRecord curRec, int loadInProg)  // :5536
{
#include "../_ref/gen/loadingChange_body.inc"
}
static boolean loadChange(int curRecRpms, int rpms)  // :5546
{
#include "../_ref/gen/loadChange_body.inc"
}
static boolean shuttingDown;
static long lastPublished, instanceStartTime, longVersion;
static String instanceLocation, instanceZone;
static StringArray instanceLabels;
static const struct { int getBusyness() const { return g_v; } mutable int g_v = 0; } invokeCounter;
static boolean g_publish;
// publishInstanceRecord :5395-5468 (without :5409-5422, see extract.py): does the record get re-published?
static void frag_publish(boolean force, boolean preShutdown)
{
    boolean isShuttingDown = shuttingDown;  // :5392
    g_publish = false;
#include "../_ref/gen/publish_fragment_a.inc"
#include "../_ref/gen/publish_fragment_b.inc"
            g_publish = true;  // :5470 the setters and the KV put follow
            return;
        }
        g_publish = true;  // a new record was created (:5436): it is written
        return;
    }
}

// =================================== a15: rateTrackingTask, the scale-up planner (MM.java:5619-5871) =========================
static long lastCheckTime, RATE_CHECK_INTERVAL_MS;      // :5617, :238
static int iterationCounter, secondCopyMaxAgeIters, secondCopyMinAgeIters, secondCopyLruThresholdMillis;
static boolean limitModelConcurrency = false;
static double averageModelParallelism = 1.0;
// (unloadManager != null whenever the runtime can unload — the default; removeUnloadBufferEntry drops the unload-buffer's own entry,
// which is not a model and which the inputs here never contain)
static const struct { bool operator!=(std::nullptr_t) const { return true; } void removeUnloadBufferEntry(const Map<String, CacheEntry> &) const {} } unloadManager;
static Map<String, CacheEntry> g_used_since_last_run;
static const struct RuntimeCacheScale { Map<String, CacheEntry> descendingMapWithCutoff(long) const { return g_used_since_last_run; } } runtimeCacheScale;
static List<String> excludeThisInstance;
struct AsyncLoad { std::string model; long ts; int weight; int extra; std::vector<std::string> exclude; };
static std::vector<AsyncLoad> g_async_loads;
static void ensureLoadedInternalAsync(const String &modelId, long ts, int weight, const List<String> &exclude, int extraCopies)  // :6940
{
    AsyncLoad a{modelId.str(), ts, weight, extraCopies, {}};
    for (int i = 0; i < exclude.size(); i++) a.exclude.push_back(exclude.get(i).str());
    g_async_loads.push_back(a);
}
static std::map<std::string, ModelRecord> g_registry_all;
static const struct { ModelRecord get(const String &id) const { auto it = g_registry_all.find(id.str()); return it == g_registry_all.end() ? ModelRecord(null) : it->second; } } registryAll;
static boolean loadedSince(ModelRecord mr, long recentLoadCutoff, String ignoreInstance)  // :5860
{
#include "../_ref/gen/loadedSince_body.inc"
}
static Set<String> g_exclude_set_seen;
static Set<String> getExcludeSet_()  // :5835
{
#include "../_ref/gen/getExcludeSet_body.inc"
}
static Set<String> getExcludeSet()
{
    g_exclude_set_seen = getExcludeSet_();
    return g_exclude_set_seen;
}
// Runnable.run of rateTrackingTask (:5636): prologue, then the loop body per entry of usedSinceLastRun
static void rateTrackingTask_run()
{
    const auto &runtimeCache = runtimeCacheScale;  // (the fragment's `runtimeCache` is the scale-up view here)
    const auto &registry = registryAll;
#include "../_ref/gen/ratetask_prologue_a.inc"
#include "../_ref/gen/ratetask_prologue_b.inc"
    for (Entry<String, CacheEntry> ent : usedSinceLastRun.entrySet()) {  // :5686 (the try / catch / finally around the body: plumbing)
#include "../_ref/gen/ratetask_loop_body.inc"
    }
#include "../_ref/gen/ratetask_epilogue.inc"
}

// ================================ a16: the janitor's scale-down of model copies (MM.java:6110-6335) ===========================
#include "../_ref/gen/second_copy_remove_constant.inc"
static ClusterStats g_instance_set_stats;
static ClusterStats instanceSetStats() { return g_instance_set_stats; }  // :1446 (this instance's partition: an input)
static const struct { template <class K, class V> Entry<K, V> immutableEntry(const K &k, const V &v) const { return Entry<K, V>(k, v); } } Maps;
static const struct { int compare(Entry<String, InstanceRecord> a, Entry<String, InstanceRecord> b) const { return placement_order_compare(a, b); } } PLACEMENT_ORDER;
static std::vector<std::string> g_removed_local;
static boolean removeLocalModelCopyAsync(const String &modelId, const ModelRecord &, const CacheEntry &, long)  // :6345
{
    g_removed_local.push_back(modelId.str());
    return true;
}
static boolean removeSecondModelCopy(String modelId, ModelRecord mr, CacheEntry ce, long lastUsed, Entry<String, InstanceRecord> otherValidInstance)  // :6314
{
#include "../_ref/gen/removeSecondModelCopy_body.inc"
}
static boolean removeModelCopies(String modelId, ModelRecord mr, CacheEntry ce, long lastUsed, long nowMillis, boolean canRemove)  // :6197
{
#include "../_ref/gen/removeModelCopies_body.inc"
}
// an element of scaleCopiesCandidates' key arrays, (String) / (ModelRecord) / (CacheEntry<?>) cast by the loop (:6123-6126)
struct AnyRef {
    String s;
    ModelRecord mr;
    CacheEntry ce;
    operator String() const { return s; }
    operator ModelRecord() const { return mr; }
    operator CacheEntry() const { return ce; }
};
struct ObjectArr3 {
    std::shared_ptr<std::vector<AnyRef>> p = std::make_shared<std::vector<AnyRef>>(3);
    const AnyRef &operator[](int i) const { return (*p)[i]; }
};
struct Exception {};
static std::vector<Entry<ObjectArr3, Long>> g_scale_candidates;
static long g_adjusted_capacity;
static long getAdjustedCacheCapacity() { return g_adjusted_capacity; }  // :5363
static void janitor_scaledown(long now)
{
    const struct { void error(const String &, const Exception &) const {} void info(const String &) const {} } logger;
    String modelId = null;
    const std::vector<Entry<ObjectArr3, Long>> &scaleCopiesCandidates = g_scale_candidates;
#include "../_ref/gen/janitor_scaledown_fragment.inc"
}

// ======================= a5: the instance-table listener (MM.java:1455-1568) and InstanceSetStatsTracker ======================
static boolean isFull(long availableUnits);
struct LongPredicate { boolean test(long v) const { return isFull(v); } };  // this::isFull, MM.java:1451
class InstanceSetStatsTracker {  // InstanceSetStatsTracker.java:31-93; the method bodies are the reference's text
    bool isnull = false;

public:
    LongPredicate isFull;
    ProhibitedTypeSet prohibitedTypesSet;
    long totalCapacity = 0, totalFree = 0, lru = Long::MAX_VALUE;
    int count = 0, modelCount = 0;
    ClusterStats currentStats = ClusterStats(0L, 0L, Long::MAX_VALUE, 0, 0);  // EMPTY_STATS, :33
    InstanceSetStatsTracker() {}
    InstanceSetStatsTracker(std::nullptr_t) : isnull(true) {}
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
    void resetLru()
    {
#include "../_ref/gen/ist_resetLru_body.inc"
    }
    void addLru(long lru)
    {
#include "../_ref/gen/ist_addLru_body.inc"
    }
    void add(const String &iid, const InstanceRecord &ir)
    {
#include "../_ref/gen/ist_add_body.inc"
    }
    boolean remove(const String &iid, const InstanceRecord &ir)
    {
#include "../_ref/gen/ist_remove_body.inc"
    }
    ClusterStats update()
    {
#include "../_ref/gen/ist_update_body.inc"
    }
};
enum EventType { ENTRY_ADDED, ENTRY_UPDATED, ENTRY_DELETED };
// clusterState = new ConcurrentSkipListSet<>(PLACEMENT_ORDER) (MM.java:774) as the listener uses it: add() refuses an element
// that compares equal to one in the set, the iterator runs in the comparator's order and can remove
struct SortedClusterState {
    std::shared_ptr<std::vector<Entry<String, InstanceRecord>>> v = std::make_shared<std::vector<Entry<String, InstanceRecord>>>();
    boolean add(const Entry<String, InstanceRecord> &e) const;
    Iterator<Entry<String, InstanceRecord>> iterator() const
    {
        auto vv = v;
        auto i = std::make_shared<size_t>(0);
        return Iterator<Entry<String, InstanceRecord>>([vv, i] { return *i < vv->size(); }, [vv, i] { return (*vv)[(*i)++]; },
                                                       [vv, i] { vv->erase(vv->begin() + (long)--*i); });
    }
};
static InstanceSetStatsTracker clusterStatsTracker;  // :1451
static int changeCounter;
static int g_upgrade_added, g_upgrade_removed, g_housekeepings, g_republish;
static void handleInstanceTableChange(const SortedClusterState &clusterState, EventType type, const String &key, InstanceRecord record);

// ======================= a21: preShutdown's migration loop (MM.java:6998-7046; triggerNewModelCopyElsewhere :6913-6928) ===========
namespace a21 {
#include "../_ref/gen/preshutdown_cutoff_constant.inc"
template <class X> struct Future { X v; };
static const struct { template <class F> auto submit(F f) const -> Future<decltype(f())> { return Future<decltype(f())>{f()}; } } taskPool;  // the pool's thread: here, at once
enum class Status { LOADING, LOADING_FAILED, LOADED };
struct StatusInfo { Status s; Status getStatus() const { return s; } String getErrorMessages() const { return null; } };
struct ShutdownCacheEntry : ::CacheEntry {  // + what the loop reads (the entry's removal and abort state: the cache's business)
    long loadTimestamp = 0, loadCompleteTimestamp = 0;
    // MMP_REF_MIGRATION_FAULTS=1 (make_ref_vectors.py's second pass over the preShutdown cases): every third entry's load was
    // aborted (the loop deregisters it, :7027-7030), every fourth triggered copy comes back LOADING_FAILED (:7033-7036: logged, not
    // waited for) — outcomes of the cache and of the remote call, not inputs of the plan
    int ordinal = 0;
    boolean isAborted() const { return g_faults && ordinal % 3 == 0; }
    static bool g_faults;
    using ::CacheEntry::CacheEntry;
};
bool ShutdownCacheEntry::g_faults = getenv("MMP_REF_MIGRATION_FAULTS") != nullptr;
static std::map<std::string, ShutdownCacheEntry> g_cache;  // runtimeCache.getQuietly
static const struct {
    ShutdownCacheEntry getQuietly(const String &id) const { auto it = g_cache.find(id.str()); return it == g_cache.end() ? ShutdownCacheEntry(null) : it->second; }
    long getLastUsedTime(const String &) const { return -1L; }  // (only asked when the descending map carried 0: an entry gone meanwhile)
} runtimeCache;
static std::vector<std::string> g_triggered;
static StatusInfo triggerNewModelCopyElsewhere(const String &modelId, const ModelRecord &, long, int)
{
    g_triggered.push_back(modelId.str());
    return StatusInfo{ShutdownCacheEntry::g_faults && g_triggered.size() % 4 == 0 ? Status::LOADING_FAILED : Status::LOADING};
}
static void deregisterModelAsync(const String &, long, long, long) {}
static Set<String> ConcurrentHashMap_newKeySet() { return Set<String>::make(); }
static const struct {
    void warn(const String &) const {}
    void warn(const String &, const TException &) const {}
} logger;
// -> per cache entry (in the map's order): bit 0 = a copy elsewhere was triggered, bit 1 = the shutdown waits for it
static std::vector<uint8_t> migration(const std::vector<Entry<String, Long>> &entries)
{
    const struct { const std::vector<Entry<String, Long>> &e; const std::vector<Entry<String, Long>> &entrySet() const { return e; } int size() const { return (int)e.size(); } } cacheEntries{entries};
    const auto &registry = registryAll;
    g_triggered.clear();
#define CacheEntry ShutdownCacheEntry  /* `CacheEntry<?> ce = runtimeCache.getQuietly(modelId)` */
#include "../_ref/gen/preshutdown_migration_fragment.inc"
#undef CacheEntry
    std::vector<uint8_t> out(entries.size(), 0);
    std::map<std::string, size_t> at;
    for (size_t i = 0; i < entries.size(); i++) at[entries[i].getKey().str()] = i;
    for (const auto &id : g_triggered) out[at[id]] |= 1;
    for (int i = 0; i < waitFor.size(); i++) {
        const Entry<String, Long> w = waitFor.get(i).v;
        if (w != null) out[at[w.getKey().str()]] |= 2;
    }
    return out;
}
}  // namespace a21

// ======================= a18: TypeConstraintManager (TypeConstraintManager.java:264-270, :337-451, :478-486, :557-567, :680-747) ===============
// The static computation — what typeMappingsUpdated (:602-667) assembles from these pieces for a given clusterState and
// configuration: per type the allowed / configured-preferred instances (fromInstanceSet + instanceMatches), per instance its
// ProhibitedTypeSet (getInstanceSetStats :561-567) hence the partitions, the instance scores and the inferred preferred
// instances (refreshPerTypeInstanceSets :684-723, inferPreferredInstances), the per-type subset stats (candidateSubsetStats) and
// the partitions' order (PARTITION_STATS_COMP).  The glue between the pieces is the harness's (the maps of trackers by labels /
// by ProhibitedTypeSet, :575-580, :655-664); the incremental path (updateInstance, updateInstanceSet) is not run.
namespace a18 {
struct TrackerObj {
    ::InstanceSetStatsTracker sums;  // add() / addLru() / update(): the reference's text (row a5)
    ClusterStats currentStats = ClusterStats(0L, 0L, Long::MAX_VALUE, 0, 0);
    ProhibitedTypeSet prohibitedTypesSet;
    int id = -1;
};
struct InstanceSetStatsTracker {  // a reference to one tracker
    std::shared_ptr<TrackerObj> p;
    InstanceSetStatsTracker() {}
    InstanceSetStatsTracker(std::nullptr_t) {}
    TrackerObj *operator->() const { return p.get(); }
    bool operator==(const InstanceSetStatsTracker &o) const { return p == o.p; }
};
struct TrackerArray {  // InstanceSetStatsTracker[] (nullable)
    std::shared_ptr<std::vector<InstanceSetStatsTracker>> p;
    TrackerArray() {}
    TrackerArray(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return !p; }
    int length() const { return (int)p->size(); }
    const InstanceSetStatsTracker &operator[](int i) const { return (*p)[i]; }
    std::vector<InstanceSetStatsTracker>::const_iterator begin() const { return p->begin(); }
    std::vector<InstanceSetStatsTracker>::const_iterator end() const { return p->end(); }
};
struct TrackerSet {  // Set<InstanceSetStatsTracker> statSet = new HashSet<>(...) (:706): by identity
    std::vector<InstanceSetStatsTracker> v;
    void clear() { v.clear(); }
    void add(const InstanceSetStatsTracker &t) { if (std::find(v.begin(), v.end(), t) == v.end()) v.push_back(t); }
};
struct NullableStats {  // ClusterStats or null (candidateSubsetStats: null = "use the cluster's stats")
    bool isnull = true;
    ClusterStats v;
    NullableStats(std::nullptr_t) {}
    NullableStats(const ClusterStats &c) : isnull(false), v(c) {}
};
template <class X> using Predicate = std::function<bool(const X &)>;
template <class X> struct Stream {
    std::vector<X> v;
    boolean allMatch(const Predicate<X> &f) const { for (auto &x : v) if (!f(x)) return false; return true; }
    boolean anyMatch(const Predicate<X> &f) const { for (auto &x : v) if (f(x)) return true; return false; }
};
static const struct {
    int binarySearch(const StringArray &a, const String &key) const  // java.util.Arrays.binarySearch on a sorted array
    {
        int lo = 0, hi = a.length() - 1;
        while (lo <= hi) {
            const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
            const int c = a[mid].compareTo(key);
            if (c < 0) lo = mid + 1; else if (c > 0) hi = mid - 1; else return mid;
        }
        return -(lo + 1);
    }
    Stream<String> stream(const StringArray &a) const { Stream<String> s; for (int i = 0; i < a.length(); i++) s.v.push_back(a[i]); return s; }
    String toString(const StringArray &) const { return String(""); }
} Arrays;
struct ImmutableSetBuilder {  // com.google.common.collect.ImmutableSet.Builder<String> (nullable)
    Set<String> s;
    ImmutableSetBuilder() {}
    ImmutableSetBuilder(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return s == null; }
    bool operator!=(std::nullptr_t) const { return s != null; }
    void add(const String &x) const { s.add(x); }
    Set<String> build() const { return s; }
};
static const struct {
    ImmutableSetBuilder builder() const { ImmutableSetBuilder b; b.s = Set<String>::make(); return b; }
    Set<String> copyOf(const Set<String> &src) const { Set<String> c = Set<String>::make(); for (auto &x : src) c.add(x); return c; }
} ImmutableSet;
template <class T> using ArrayList = List<T>;
template <class K> struct ObjectIntPair { K k; int v; K getOne() const { return k; } int getTwo() const { return v; } };
template <class K> struct ObjectIntMap {  // org.eclipse.collections ObjectIntMap<String> (iteration: by key; the result is a SET of ids)
    std::shared_ptr<std::map<std::string, int>> p = std::make_shared<std::map<std::string, int>>();
    void put(const K &k, int v) const { (*p)[k.str()] = v; }
    boolean containsKey(const K &k) const { return p->count(k.str()) != 0; }
    void addToValue(const K &k, int d) const { (*p)[k.str()] += d; }
    std::vector<ObjectIntPair<K>> keyValuesView() const { std::vector<ObjectIntPair<K>> o; for (auto &e : *p) o.push_back({String(e.first), e.second}); return o; }
};
template <class K> using MutableObjectIntMap = ObjectIntMap<K>;
static ObjectIntMap<String> ObjectIntHashMap_new(int) { return ObjectIntMap<String>(); }
static const struct { void warn(const String &) const {} void info(const String &) const {} } logger;

static boolean instanceMatches(StringArray instanceLabels, StringArray typeLabels, boolean matchAll)  // :478
{
#include "../_ref/gen/tcm_instanceMatches_body.inc"
}
struct ModelTypeConstraints {  // :337-355: immutable; the fields the fragments read
    bool isnull = true;
    StringArray requiredLabels, preferredLabels;
    Set<String> allowedInstances, preferredInstances, configuredPreferredInstances;
    TrackerArray instanceSetStats;
    ModelTypeConstraints() {}
    ModelTypeConstraints(std::nullptr_t) {}
    ModelTypeConstraints(const StringArray &req, const StringArray &pref, const Set<String> &allowed, const Set<String> &configured,
                         const TrackerArray &stats, const Set<String> &resolved)  // :377-388
        : isnull(false), requiredLabels(req), preferredLabels(pref), allowedInstances(allowed), preferredInstances(resolved),
          configuredPreferredInstances(configured), instanceSetStats(stats) {}
    boolean allowedOnInstance(const String &iid) const
    {
#include "../_ref/gen/tcm_allowedOnInstance_body.inc"
    }
    NullableStats candidateSubsetStats() const
    {
#include "../_ref/gen/tcm_candidateSubsetStats_body.inc"
    }
    // updateInstanceSetStats (:394-413) returns this or a copy that differs in instanceSetStats / preferredInstances: the copy
    ModelTypeConstraints updateInstanceSetStats(std::nullptr_t, const Set<String> &newInferredPreferred) const
    {
        return ModelTypeConstraints(requiredLabels, preferredLabels, allowedInstances, configuredPreferredInstances, TrackerArray(null), newInferredPreferred);
    }
    ModelTypeConstraints updateInstanceSetStats(const TrackerSet &newStats, const Set<String> &newInferredPreferred) const
    {
        TrackerArray a;
        a.p = std::make_shared<std::vector<InstanceSetStatsTracker>>(newStats.v);
        return ModelTypeConstraints(requiredLabels, preferredLabels, allowedInstances, configuredPreferredInstances, a, newInferredPreferred);
    }
};
static ModelTypeConstraints fromInstanceSet(StringArray requiredLabels, StringArray preferredLabels, const std::vector<Entry<String, InstanceRecord>> &instances,
                                            String typeName, TrackerArray instanceSetStats)  // :416
{
#include "../_ref/gen/tcm_fromInstanceSet_body.inc"
}
static Set<String> inferPreferredInstances(ObjectIntMap<String> instanceScores, Set<String> include)  // :727
{
#include "../_ref/gen/tcm_inferPreferredInstances_body.inc"
}
struct MtcMap {  // Map<String, ModelTypeConstraints> whose entries write through (ent.setValue, :709, :720)
    std::vector<Entry<String, ModelTypeConstraints>> es;
    const std::vector<Entry<String, ModelTypeConstraints>> &entrySet() const { return es; }
    int size() const { return (int)es.size(); }
};
static ProhibitedTypeSet prohibitedTypesOf(const String &iid, const MtcMap &tcMap)  // getInstanceSetStats, :561-567
{
#include "../_ref/gen/tcm_prohibited_types_fragment.inc"
    return pts;
}
struct PtsMap {  // Map<ProhibitedTypeSet, InstanceSetStatsTracker> ptsToInstanceSetStats (:132), in creation order
    std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>> es;
    const std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>> &entrySet() const { return es; }
    int size() const { return (int)es.size(); }
};
static Set<String> g_defaultPreferred;
static void refreshPerTypeInstanceSets(MtcMap &mtcMap, const PtsMap &ptsToInstanceSetStats)  // :680-725 (the logging of :700-704 left out)
{
    const struct { int size() const { return (int)g_cluster->size(); }
                   std::vector<Entry<String, InstanceRecord>>::const_iterator begin() const { return g_cluster->begin(); }
                   std::vector<Entry<String, InstanceRecord>>::const_iterator end() const { return g_cluster->end(); } } clusterState;
#include "../_ref/gen/tcm_scores_fragment.inc"
    g_defaultPreferred = defaultPreferred;
    TrackerSet statSet;  // :706
#include "../_ref/gen/tcm_per_type_fragment.inc"
}
static int partition_stats_comp(const InstanceSetStatsTracker &isst1, const InstanceSetStatsTracker &isst2)  // :264
{
#include "../_ref/gen/tcm_partition_stats_comp_body.inc"
}
}  // namespace a18

// ======================= a19: UpgradeTracker (UpgradeTracker.java:45-201) ============================================================
namespace a19 {
#include "../_ref/gen/upgrade_constants.inc"
struct LongField {  // an assignable long field of an object with reference semantics
    std::shared_ptr<long> p;
    explicit LongField(long v = 0) : p(std::make_shared<long>(v)) {}
    operator long() const { return *p; }
    const LongField &operator=(long v) const { *p = v; return *this; }
};
struct CountField {
    std::shared_ptr<int> p = std::make_shared<int>(0);
    operator int() const { return *p; }
    int operator--(int) const { return (*p)--; }
    int operator++(int) const { return (*p)++; }
};
struct ReplicaSetStats {  // :52-57
    bool isnull = true;
    CountField size;
    LongField earliestStartTime{Long::MAX_VALUE}, latestStartTime, lastChangeTime;
    ReplicaSetStats() {}
    ReplicaSetStats(std::nullptr_t) {}
    static ReplicaSetStats make() { ReplicaSetStats r; r.isnull = false; return r; }
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
};
template <class X> struct Optional { X v; X get() const { return v; } };
template <class X> struct StreamOf {
    std::vector<X> v;
    template <class C> Optional<X> max(C cmp) const  // Stream.max == reduce(BinaryOperator.maxBy(cmp)): of equals, the first
    {
        X best = v[0];
        for (size_t i = 1; i < v.size(); i++)
            if (!(cmp(best, v[i]) >= 0)) best = v[i];
        return Optional<X>{best};
    }
    template <class P> StreamOf filter(P pred) const
    {
        StreamOf out;
        for (auto &x : v)
            if (pred(x)) out.v.push_back(x);
        return out;
    }
    struct Keys {
        std::vector<String> k;
        Set<String> collect_toSet() const
        {
            Set<String> s = Set<String>::make();
            for (auto &x : k) s.add(x);
            return s;
        }
    };
    Keys map_getKey() const
    {
        Keys out;
        for (auto &x : v) out.k.push_back(x.getKey());
        return out;
    }
};
template <class X> struct CollectionOf { std::vector<X> v; StreamOf<X> stream() const { return StreamOf<X>{v}; }
                                         typename std::vector<X>::const_iterator begin() const { return v.begin(); }
                                         typename std::vector<X>::const_iterator end() const { return v.end(); } };
// PerTypeLabelStats extends HashMap<String, ReplicaSetStats> (:59-64).  Iteration order: by key here (a HashMap's is by hash;
// the one place the order could matter is Stream.max among replica sets with EQUAL earliestStartTime)
struct PerTypeLabelStats {
    std::shared_ptr<std::map<std::string, ReplicaSetStats>> p;
    PerTypeLabelStats() {}
    PerTypeLabelStats(std::nullptr_t) {}
    static PerTypeLabelStats make() { PerTypeLabelStats m; m.p = std::make_shared<std::map<std::string, ReplicaSetStats>>(); return m; }
    bool operator==(std::nullptr_t) const { return !p; }
    ReplicaSetStats get(const String &k) const { auto it = p->find(k.str()); return it == p->end() ? ReplicaSetStats(null) : it->second; }
    void put(const String &k, const ReplicaSetStats &v) const { (*p)[k.str()] = v; }
    void remove(const String &k) const { p->erase(k.str()); }
    int size() const { return (int)p->size(); }
    CollectionOf<ReplicaSetStats> values() const { CollectionOf<ReplicaSetStats> c; for (auto &e : *p) c.v.push_back(e.second); return c; }
    CollectionOf<Entry<String, ReplicaSetStats>> entrySet() const
    {
        CollectionOf<Entry<String, ReplicaSetStats>> c;
        for (auto &e : *p) c.v.push_back(Entry<String, ReplicaSetStats>(String(e.first), e.second));
        return c;
    }
};
// Map<String[], PerTypeLabelStats> upgradeTracker = new HashMap<>(1) (:67): arrays hash by IDENTITY — the key is the labels
// array object (here: the label-set id the event carries; records with the same label set share one interned array)
static long g_labels_key;
struct LabelsKey { long id; };
static struct {
    std::map<long, PerTypeLabelStats> m;
    PerTypeLabelStats get(const LabelsKey &k) { auto it = m.find(k.id); return it == m.end() ? PerTypeLabelStats(null) : it->second; }
    void put(const LabelsKey &k, const PerTypeLabelStats &v) { m[k.id] = v; }
} upgradeTracker;
static ObjectLongMap<String> likelyReplacedReplicaSets;  // :71 (= ObjectLongMaps.immutable.empty())
struct Ir {  // what the tracker reads of an InstanceRecord
    long startTime;
    LabelsKey getLabels() const { return LabelsKey{g_labels_key}; }
    long getStartTime() const { return startTime; }
};
static void instanceRemoved(String iid, Ir ir)
{
#include "../_ref/gen/upgrade_instanceRemoved_body.inc"
}
static void instanceAdded(String iid, Ir ir)
{
#include "../_ref/gen/upgrade_instanceAdded_body.inc"
}
static void doHousekeeping()
{
#include "../_ref/gen/upgrade_doHousekeeping_body.inc"
}
}  // namespace a19

// ======================= a17: the leader's reaper — proactive loading (MM.java:6456-6490, :6574-6577, :6616-6747) ===============
struct ModelToLoad {  // :6393-6409
    String modelId;
    long lastUsed;
    int index;
    ModelToLoad(const String &m, long l, int i) : modelId(m), lastUsed(l), index(i) {}
    int compareTo(const ModelToLoad &m) const
    {
#include "../_ref/gen/modeltoload_compareTo_body.inc"
    }
};
// java.util.TreeSet under the element's own compareTo: add() of an element that compares equal to one in the set is a no-op
template <class X> class NavigableSet {
    struct Less { bool operator()(const X &a, const X &b) const { return a.compareTo(b) < 0; } };
    typedef std::set<X, Less> Rep;
    std::shared_ptr<Rep> p;

public:
    NavigableSet() {}
    NavigableSet(std::nullptr_t) {}
    NavigableSet &operator=(const Set<String> &) { p = std::make_shared<Rep>(); return *this; }  // `toLoad = new TreeSet<>()` (extract.py: TreeSet_new())
    bool operator==(std::nullptr_t) const { return !p; }
    int size() const { return (int)p->size(); }
    boolean isEmpty() const { return p->empty(); }
    const X &last() const { return *p->rbegin(); }
    boolean add(const X &x) const { return p->insert(x).second; }
    void pollLast() const { p->erase(std::prev(p->end())); }
    typename Rep::const_iterator begin() const { return p->begin(); }
    typename Rep::const_iterator end() const { return p->end(); }
};
struct InterruptedException {};
struct Phaser {
    explicit Phaser(int) {}
    int register_() const { return 0; }
    int arrive() const { return 0; }
    void awaitAdvanceInterruptibly(int) const {}
};
static const struct { template <class F> void execute(F f) const { f(); } } taskPool;  // the pool's thread: here, at once
static const struct { void sleep(long) const {} } Thread;
static long msSince(long) { return 0; }
static String readableTime(long) { return String(""); }
static boolean DISABLE_PROACTIVE_LOADING = false;
static int defaultModelSizeUnits;
struct ProactiveCall { std::string model; long ts; int partition; };
static std::vector<ProactiveCall> g_proactive_calls;
static int g_proactive_partition;
static void ensureLoadedInternal(const String &modelId, long lastUsedTime, int, std::nullptr_t, int, boolean) { g_proactive_calls.push_back({modelId.str(), lastUsedTime, g_proactive_partition}); }  // :6727
static void triggerProactiveLoadsForInstanceSubset(ClusterStats stats, List<Entry<String, ModelRecord>> allCandidates, ProhibitedTypeSet excludeTypes)
{
    const struct {
        void warn(const String &) const {}
        void warn(const String &, const Exception &) const {}
        void info(const String &) const {}
        void debug(const String &) const {}
        boolean isDebugEnabled() const { return true; }  // the reference's debug statements run too (they build a message and discard it)
    } logger;
    g_proactive_partition++;
#include "../_ref/gen/triggerProactiveLoads_body.inc"
}
struct ReaperTypeConstraints {
    std::shared_ptr<std::vector<InstanceSetStatsTracker>> p;
    bool operator==(std::nullptr_t) const { return !p; }
    const std::vector<InstanceSetStatsTracker> &getPartitionStats() const { return *p; }
};
// the reaper's run as far as proactive loading goes: candidate collection over the registry (in its iteration order), then the
// dispatch per instance subset
static void reaper_proactive(const std::vector<Entry<String, ModelRecord>> &registryIterable, const ReaperTypeConstraints &typeConstraints)
{
#include "../_ref/gen/reaper_candidates_prologue.inc"
    for (const Entry<String, ModelRecord> &ent : registryIterable) {
        ModelRecord mr = ent.getValue();
        Map<String, Long> insts = mr.getInstanceIds(), failInsts = mr.getLoadFailedInstanceIds();  // :6542-6543
#include "../_ref/gen/reaper_candidate_rule.inc"
    }
    if (proactiveLoadCandidates != null && !proactiveLoadCandidates.isEmpty()) {  // :6471
#include "../_ref/gen/reaper_dispatch_fragment.inc"
    }
}

// ---- a5, continued (needs PLACEMENT_ORDER and Maps from above)
boolean SortedClusterState::add(const Entry<String, InstanceRecord> &e) const
{
    auto it = std::lower_bound(v->begin(), v->end(), e, [](const Entry<String, InstanceRecord> &a, const Entry<String, InstanceRecord> &b) { return placement_order_compare(a, b) < 0; });
    if (it != v->end() && placement_order_compare(*it, e) == 0) return false;
    v->insert(it, e);
    return true;
}
static void handleInstanceTableChange(const SortedClusterState &clusterState, EventType type, const String &key, InstanceRecord record)
{
    // typeConstraints == null here (the subset-stats branches compile against these and are not run: rows a18 / f-4)
    const struct {
        bool operator!=(std::nullptr_t) const { return false; }
        InstanceSetStatsTracker getStatsForLabels(const StringArray &) const { return null; }
        InstanceSetStatsTracker instanceAdded(const String &, const StringArray &, boolean) const { return null; }
        void instanceRemoved(const String &, const StringArray &) const {}
    } typeConstraints;
    const struct {
        void instanceRemoved(const String &, const InstanceRecord &) const { g_upgrade_removed++; }
        void instanceAdded(const String &, const InstanceRecord &) const { g_upgrade_added++; }
        void doHousekeeping() const { g_housekeepings++; }
    } upgradeTracker;
    struct LeaderElection {
        LeaderElection() {}
        LeaderElection(std::nullptr_t) {}
        bool operator!=(std::nullptr_t) const { return false; }
        boolean isLeader() const { return false; }
    };
    const LeaderElection leaderLatch;
    const struct { void remove(const String &) const {} } missings;
    const struct { boolean equals(const String &a, const String &b) const { return a == null ? b == null : (b != null && a.equals(b)); } } Objects;  // java.util.Objects.equals
    auto publishInstanceRecordAsync = [] { g_republish++; };
    const struct {
        void warn(const String &) const {}
        void debug(const String &) const {}
        boolean isDebugEnabled() const { return true; }  // the reference's debug statements run too (they build a message and discard it)
    } logger;
#include "../_ref/gen/handleInstanceTableChange_body.inc"
}

// ======================================================== I/O ===============================================================
// The audit hash of a shortlist (machinery of THIS repository, not of the reference): a function of the set of
// rank positions in the shortlist and of the count that survived the rpm filter — computed here from the reference's own
// candidate list and the reference's own clusterState order, so that a fixture row is 16 bytes instead of a ragged list.
static uint64_t audit_mul(uint64_t word)  // csrc/wave.hpp: the audit hash is linear in the candidate bits
{
    uint64_t x = (word + 1ull) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return x | 1ull;
}
static uint32_t shortlist_hash(const std::vector<int32_t> &pos_of, const int32_t *cand, int32_t n, int32_t remaining, int64_t n_pods)
{
    uint64_t h = 0;
    std::vector<uint8_t> seen((size_t)n_pods + 1, 0);  // (a set of positions: an instance listed twice counts once)
    for (int32_t i = 0; i < n; i++) {
        const int32_t p = pos_of[cand[i]];
        if (seen[(size_t)p]) continue;
        seen[(size_t)p] = 1;
        h += audit_mul((uint64_t)(p >> 6)) << (p & 63);
    }
    return (uint32_t)(h ^ (h >> 32)) ^ ((uint32_t)remaining * 0x9E3779B1u);
}

template <class X> static std::vector<X> rd(FILE *f, size_t n)
{
    std::vector<X> v(n);
    if (n && fread(v.data(), sizeof(X), n, f) != n) { fprintf(stderr, "ref_harness: short read\n"); exit(2); }
    return v;
}
template <class X> static void wr(FILE *f, const std::vector<X> &v)
{
    if (!v.empty() && fwrite(v.data(), sizeof(X), v.size(), f) != v.size()) { fprintf(stderr, "ref_harness: short write\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: ref_harness <input.bin> <output.bin>\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    auto magic = rd<char>(f, 8);
    if (memcmp(magic.data(), "MMREF1\0\0", 8) != 0) { fprintf(stderr, "ref_harness: bad magic\n"); return 1; }
    auto H = rd<int64_t>(f, 16);
    const int64_t P = H[0], M = H[1], n_ent = H[2], Tn = H[3], W = H[4], n_repl = H[5], n_req = H[6], n_extra = H[7];
    minSpaceUnits = H[8];
    minChurnAgeMs = H[9];
    g_now = H[10];
    const int64_t n_serve = H[11], n_sexcl = H[12], n_gate = H[13], n_gexcl = H[14], n_gexpl = H[15];
    auto pods = rd<mmp_pod_row>(f, P);
    auto idbuf = rd<char>(f, P * 16);
    auto models = rd<mmp_model_row>(f, M);
    auto ent_pod = rd<int32_t>(f, n_ent);
    auto ent_time = rd<int64_t>(f, n_ent);
    auto has_allowed = rd<uint8_t>(f, Tn), has_prefer = rd<uint8_t>(f, Tn);
    auto allowed = rd<uint64_t>(f, Tn * W), prefer = rd<uint64_t>(f, Tn * W);
    auto replaced = rd<int32_t>(f, n_repl);
    auto reqs = rd<mmp_place_req>(f, n_req);
    auto extra = rd<int32_t>(f, n_extra);
    auto sreqs = rd<mmp_serve_req>(f, n_serve);
    auto in_use = rd<int32_t>(f, n_serve ? P : 0);
    auto last_used = rd<int64_t>(f, n_serve ? P : 0);
    auto sx_pod = rd<int32_t>(f, n_sexcl);
    auto sx_time = rd<int64_t>(f, n_sexcl);
    auto greqs = rd<mmp_gate_req>(f, n_gate);
    auto gx_pod = rd<int32_t>(f, n_gexcl);
    auto gx_time = rd<int64_t>(f, n_gexcl);
    auto gexplicit = rd<int32_t>(f, n_gexpl);
    auto gparams = rd<int64_t>(f, n_gate ? 1 : 0);  // IN_USE_LOAD_FAILURE_EXPIRY_MS
    struct StatsRow { int64_t total_capacity, total_free, global_lru; int32_t instance_count, model_copy_count; };
    auto tstats = rd<StatsRow>(f, n_gate ? std::max<int64_t>(Tn, 1) : 0);  // typeSetStats(type) per type row: an input here (rows a5 / a18)
    // a15: one run of the rate-tracking task: params, the entries of usedSinceLastRun in its order, cluster + per-type stats
    auto n_scale_v = rd<int64_t>(f, 1);
    const int64_t n_scale = n_scale_v[0];
    auto sparams = rd<mmp_scaleup_params>(f, n_scale >= 0 ? 1 : 0);
    auto sentries = rd<mmp_cache_entry>(f, n_scale > 0 ? n_scale : 0);
    auto sstats = rd<StatsRow>(f, n_scale >= 0 ? 1 + std::max<int64_t>(Tn, 1) : 0);  // [0] = clusterStats, then typeSetStats rows
    // a16: one janitor pass over scaleCopiesCandidates (oldest first): params, entries, instanceSetStats()
    auto n_sd_v = rd<int64_t>(f, 1);
    const int64_t n_sd = n_sd_v[0];
    auto dparams = rd<mmp_scaledown_params>(f, n_sd >= 0 ? 1 : 0);
    auto dentries = rd<mmp_cache_entry>(f, n_sd > 0 ? n_sd : 0);
    auto dstats = rd<StatsRow>(f, n_sd >= 0 ? 1 : 0);
    // a17: one reaper run: defaultModelSizeUnits, clusterStats, then per instance partition (0 partitions: typeConstraints == null)
    // its stats (InstanceSetStatsTracker.currentStats: an input, rows a5 / a18) and prohibited type rows, then the partition of every instance
    auto n_pro_v = rd<int64_t>(f, 1);
    const int64_t n_pro = n_pro_v[0];
    auto pro_units = rd<int64_t>(f, n_pro >= 0 ? 1 : 0);
    auto pro_gstats = rd<StatsRow>(f, n_pro >= 0 ? 1 : 0);
    std::vector<StatsRow> pro_pstats;
    std::vector<std::vector<int32_t>> pro_types;
    for (int64_t k = 0; k < n_pro; k++) {
        pro_pstats.push_back(rd<StatsRow>(f, 1)[0]);
        const int64_t nt = rd<int64_t>(f, 1)[0];
        pro_types.push_back(rd<int32_t>(f, (size_t)nt));
    }
    auto pro_pod_part = rd<int32_t>(f, n_pro > 0 ? (size_t)P : 0);
    // a5: a stream of instance-table listener events from an empty table: (type, instance, its new record); the cluster's stats
    // and clusterState's order are written every `ck` events and at the end
    struct TableEvent { int32_t type, pod; mmp_pod_row row; };
    auto n_ev_v = rd<int64_t>(f, 1);
    const int64_t n_ev = n_ev_v[0];
    auto ev_ck = rd<int64_t>(f, n_ev >= 0 ? 1 : 0);
    auto events = rd<TableEvent>(f, n_ev > 0 ? (size_t)n_ev : 0);
    // a19: a stream of UpgradeTracker calls: kind 0 instanceAdded, 1 instanceRemoved, 2 doHousekeeping; the likely-replaced map is
    // written after every call
    struct UpgradeEvent { int32_t kind, replica_set; int64_t labels_key, start_time, now; };
    auto n_up_v = rd<int64_t>(f, 1);
    const int64_t n_up = n_up_v[0];
    auto upevents = rd<UpgradeEvent>(f, n_up > 0 ? (size_t)n_up : 0);
    // a18: a type-constraint configuration over the instance table: T types, the label bitset of every instance and the required /
    // preferred label bitsets of every type (bit i = label "l<i, two digits>")
    auto n_tc_v = rd<int64_t>(f, 1);
    const int64_t n_tc = n_tc_v[0];
    auto tc_pod_bits = rd<uint64_t>(f, n_tc >= 0 ? (size_t)P : 0);
    auto tc_req_bits = rd<uint64_t>(f, n_tc > 0 ? (size_t)n_tc : 0);
    auto tc_pref_bits = rd<uint64_t>(f, n_tc > 0 ? (size_t)n_tc : 0);
    // a21: preShutdown of one instance: its cache entries in descendingLruMap() order
    auto n_mig_v = rd<int64_t>(f, 1);
    const int64_t n_mig = n_mig_v[0];
    auto mig_hdr = rd<int64_t>(f, n_mig >= 0 ? 2 : 0);  // self instance, now
    auto mig_entries = rd<mmp_cache_entry>(f, n_mig > 0 ? (size_t)n_mig : 0);
    // optional trailer (absent in the inputs of rounds 3-4, whose digests therefore stand): limitModelConcurrency == true — the
    // MaxConcCacheEntry row of every cache entry of the a15 / a16 section of this input, dynamicRpmScaleConstant, and the rate
    // task's averageModelParallelism going into the run
    int64_t n_conc = -1;
    std::vector<mmp_conc_params> conc_params;
    std::vector<mmp_conc_entry> conc_rows;
    {
        int64_t v = 0;
        if (fread(&v, sizeof v, 1, f) == 1) {
            n_conc = v;
            conc_params = rd<mmp_conc_params>(f, 1);
            conc_rows = rd<mmp_conc_entry>(f, (size_t)n_conc);
        }
    }
    fclose(f);
    auto conc_state = [&](int64_t e) {
        auto st = std::make_shared<ConcState>();
        const mmp_conc_entry &m = conc_rows[(size_t)e];
        st->countAndTimeSum = m.count_and_time_sum;
        st->priorSum = m.prior_sum;
        st->priorCount = m.prior_count;
        st->maxConc = m.max_conc;
        st->queued = m.queued_requests;
        return st;
    };

    std::vector<String> ids(P);
    std::unordered_map<std::string, int> pod_of;
    for (int64_t i = 0; i < P; i++) {
        ids[i] = String(std::string(&idbuf[i * 16], strnlen(&idbuf[i * 16], 16)));
        pod_of[ids[i].str()] = (int)i;
    }
    // the instance table as the KV listener leaves it: shutting-down / absent records are not in clusterState
    // (MM.java:1462-1464); litelinks' instance list (sis) = the live ones
    g_cluster = std::make_shared<std::vector<Entry<String, InstanceRecord>>>();
    g_simap = Map<String, ServiceInstanceInfo>::make();
    for (int64_t i = 0; i < P; i++) {
        const mmp_pod_row &r = pods[i];
        if (!(r.flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE)))
            g_cluster->emplace_back(ids[i], InstanceRecord(r.lru_time, r.capacity, r.used, r.version, r.count, r.loading_threads,
                                                            r.loading_in_progress, r.rpm, false));
        if (r.flags & MMP_POD_LIVE)
            g_simap.put(ids[i], ServiceInstanceInfo(ids[i], (int)i, n_serve ? in_use[i] : 0, n_serve ? last_used[i] : 0));
    }
    // clusterState = new ConcurrentSkipListSet<>(PLACEMENT_ORDER), MM.java:774: a sorted set under the reference's comparator
    std::stable_sort(g_cluster->begin(), g_cluster->end(),
                     [](const Entry<String, InstanceRecord> &a, const Entry<String, InstanceRecord> &b) { return placement_order_compare(a, b) < 0; });
    if (Tn > 0) {
        typeConstraints.p = std::make_shared<std::map<std::string, std::pair<Set<String>, Set<String>>>>();
        for (int64_t t = 0; t < Tn; t++) {
            Set<String> al, pf;
            if (has_allowed[t]) al = Set<String>::make();
            if (has_prefer[t]) pf = Set<String>::make();
            for (int64_t i = 0; i < P; i++) {
                if (has_allowed[t] && ((allowed[t * W + (i >> 6)] >> (i & 63)) & 1ull)) al.add(ids[i]);
                if (has_prefer[t] && ((prefer[t * W + (i >> 6)] >> (i & 63)) & 1ull)) pf.add(ids[i]);
            }
            (*typeConstraints.p)["t" + std::to_string(t)] = {al, pf};
        }
    }
    for (int32_t rs : replaced) {  // UpgradeTracker.getLikelyReplacedReplicaSets(): keys are id.substring(0, 6)
        std::string key = "~rs" + std::to_string(rs);  // a set no instance belongs to still makes the map non-empty
        for (int64_t i = 0; i < P; i++)
            if (pods[i].replica_set == rs && ids[i].length() >= 7) { key = ids[i].str().substr(0, 6); break; }
        (*g_replaced.p)[key] = 0;
    }

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    std::vector<int32_t> order, pos_of(P, -1);
    for (auto &e : *g_cluster) {
        pos_of[pod_of[e.getKey().str()]] = (int32_t)order.size();
        order.push_back(pod_of[e.getKey().str()]);
    }
    std::vector<int64_t> oh = {(int64_t)order.size()};
    wr(o, oh);
    wr(o, order);

    // ---- load-target decisions
    std::vector<int32_t> out(n_req * 4), shortlist;  // per decision: chosen, candidates.size(), survivors of the rpm filter, audit hash
    CacheMissForwardingLB lb;
    for (int64_t d = 0; d < n_req; d++) {
        const mmp_place_req &rq = reqs[d];
        int32_t chosen = MMP_NONE, ccount = 0, remaining = 0;
        uint32_t hash = 0;
        if (rq.model >= 0 && rq.model < M) {  // an unknown model never reaches the LB
            const mmp_model_row &m = models[rq.model];
            g_exclude = CacheMissExcludeSet();
            const int type = (m.type < 0 || m.type >= Tn) ? 0 : m.type;
            g_exclude.modelType = String("t" + std::to_string(type));
            g_exclude.favourSelf = (rq.flags & MMP_REQ_FAVOUR_SELF) != 0;
            g_exclude.lastUsedTime = rq.last_used;
            for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++) {
                const int32_t pod = ent_pod[m.ent_off + k];
                if (pod < 0 || pod >= P) continue;
                (k < m.n_loaded ? g_exclude.loaded : g_exclude.failed).add(ids[pod]);
            }
            for (int32_t k = 0; k < rq.n_extra; k++) {
                const int32_t pod = extra[rq.extra_off + k];
                if (pod >= 0 && pod < P) g_exclude.add(ids[pod]);
            }
            instanceId = (rq.self_pod >= 0 && rq.self_pod < P) ? ids[rq.self_pod] : String("(this instance is not in the table)");
            g_fresh = InstanceRecord(rq.fresh_lru, rq.fresh_capacity, rq.fresh_used, 0, rq.fresh_count, 0, 0, rq.fresh_rpm, false);
            g_pick = rq.pick;
            g_lists_created.clear();
            const Object r = lb.getNext(ObjectArray(), String("applyModel"), ObjectArray());
            chosen = r.kind == 0 ? MMP_NONE : r.kind == 1 ? MMP_SELF : r.sii.p->pod;
            if (!g_lists_created.empty()) {  // the first list getNext creates is `candidates` (:4814)
                auto cand = std::static_pointer_cast<List<String>::Rep>(g_lists_created[0]);
                ccount = (int32_t)cand->added.size();
                shortlist.clear();
                for (auto &s : cand->added) shortlist.push_back(pod_of[s.str()]);
                for (auto &s : cand->v)
                    if (s != null) remaining++;
                hash = shortlist_hash(pos_of, shortlist.data(), ccount, remaining, P);
            }
        }
        out[d * 4 + 0] = chosen;
        out[d * 4 + 1] = ccount;
        out[d * 4 + 2] = remaining;
        out[d * 4 + 3] = (int32_t)hash;
    }
    wr(o, out);

    // ---- serve-target decisions
    std::vector<int64_t> sout(n_serve * 2);
    ForwardingLB flb;
    for (int64_t d = 0; d < n_serve; d++) {
        const mmp_serve_req &rq = sreqs[d];
        int64_t chosen = MMP_NONE, ts = 0;
        if (rq.model >= 0 && rq.model < M) {
            const mmp_model_row &m = models[rq.model];
            g_filtered = MapFilteringSet<String, Long>();
            g_filtered.excludeSelf = (rq.flags & MMP_SERVE_EXCLUDE_SELF) != 0;
            g_filtered.preferSelf = (rq.flags & MMP_SERVE_PREFER_SELF) != 0;
            g_filtered.modelType = String("t0");
            for (int32_t k = 0; k < rq.n_excl; k++) {
                const int32_t pod = sx_pod[rq.excl_off + k];
                if (pod < 0 || pod >= P) continue;
                if (sx_time[rq.excl_off + k] == INT64_MIN) {  // MMP_ANY_TIME: keyExcludes (:4282)
                    if (g_filtered.keyExcludes == null) g_filtered.keyExcludes = Collection<String>::make();
                    g_filtered.keyExcludes.add(ids[pod]);
                } else
                    g_filtered.add(ids[pod], Long(sx_time[rq.excl_off + k]));
            }
            Map<String, Long> source = Map<String, Long>::make();  // ModelRecord.instanceIds (a TreeMap: id order)
            for (int32_t k = 0; k < m.n_loaded; k++) {
                const int32_t pod = ent_pod[m.ent_off + k];
                source.put(pod >= 0 && pod < P ? ids[pod] : String("~unknown-" + std::to_string(k)), Long(ent_time[m.ent_off + k]));
            }
            const size_t tried_before = g_filtered.tried->size();
            g_filtered.filteredMap(source);
            instanceId = (rq.self_pod >= 0 && rq.self_pod < P) ? ids[rq.self_pod] : String("(this instance is not in the table)");
            g_local_in_flight = rq.local_in_flight;
            lastInvokeTime = rq.last_invoke_time;
            g_assume_completed = rq.assume_completed_ms;
            const Object r = flb.getNext(ObjectArray(), String("applyModel"), ObjectArray());
            chosen = r.kind == 0 ? MMP_NONE : r.kind == 1 ? MMP_SELF : r.sii.p->pod;
            if (g_filtered.tried->size() > tried_before) ts = g_filtered.tried->back().second;  // filtered.add(chosenId, chosenTimeStamp), :4389
        }
        sout[d * 2] = chosen;
        sout[d * 2 + 1] = ts;
    }
    wr(o, sout);

    // ---- the request-level guards: bits as in include/mmplace.h (MMP_GATE_*), and loadLocal's signed initial size
    std::vector<int32_t> gout(n_gate * 2);
    if (n_gate) {
        IN_USE_LOAD_FAILURE_EXPIRY_MS = gparams[0];
        for (auto &r : tstats) g_tstats.push_back(ClusterStats{r.total_capacity, r.total_free, r.global_lru, r.instance_count, r.model_copy_count});
        g_pod_of = pod_of;
        g_in_table.assign(P, 0);
        g_table_rec.assign(P, InstanceRecord(null));
        for (int64_t i = 0; i < P; i++) {
            const mmp_pod_row &r = pods[i];
            g_in_table[i] = !(r.flags & MMP_POD_TOMBSTONE);
            if (g_in_table[i]) {
                g_table_rec[i] = InstanceRecord(r.lru_time, r.capacity, r.used, r.version, r.count, r.loading_threads, r.loading_in_progress, r.rpm,
                                                (r.flags & MMP_POD_SHUTTING_DOWN) != 0);
            }
        }
    }
    for (int64_t d = 0; d < n_gate; d++) {
        const mmp_gate_req &q = greqs[d];
        g_gq = &q;
        uint32_t bits = 0;
        const mmp_model_row &m = models[q.model];
        const int type = (m.type < 0 || m.type >= Tn) ? 0 : m.type;
        ModelRecord mr;
        mr.type = String("t" + std::to_string(type));
        for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++) {
            const int32_t pod = ent_pod[m.ent_off + k];
            (k < m.n_loaded ? mr.instanceIds : mr.failed).put(ids[pod], Long(ent_time[m.ent_off + k]));
        }
        instanceId = ids[q.self_pod];
        const String modelId("model");
        // goLocal: filteredInstances = the copies minus the cache-hit excludes (MapFilteringSet, key excludes here)
        {
            Map<String, Long> filtered = Map<String, Long>::make();
            for (auto &e : mr.instanceIds.entrySet()) {
                bool ex = false;
                for (int32_t k = 0; k < q.n_excl; k++)
                    if (ids[gx_pod[q.excl_off + k]].str() == e.getKey().str() &&
                        (gx_time[q.excl_off + k] == INT64_MIN || gx_time[q.excl_off + k] == (long)e.getValue()))
                        ex = true;
                if (!ex) filtered.put(e.getKey(), e.getValue());
            }
            g_cache_entry = CacheEntry();
            if (q.flags & MMP_GATE_HAVE_CACHE_ENTRY) { g_cache_entry.isnull = false; g_cache_entry.done = (q.flags & MMP_GATE_ENTRY_DONE) != 0; }
            if (frag_goLocal(filtered, (q.flags & MMP_GATE_FAVOUR_SELF_FOR_HITS) != 0, modelId, q.last_used_time)) bits |= MMP_GATE_GO_LOCAL;
        }
        Collection<String> explicitExcludes = Collection<String>::make();
        for (int32_t k = 0; k < q.n_explicit; k++) explicitExcludes.add(ids[gexplicit[q.explicit_off + k]]);
        // MMP_REF_EXC_CONTEXT=1 (make_ref_vectors.py's second pass over the guard cases): the request has ALREADY seen a load
        // failure / an internal failure elsewhere, or is not an external request — the Java then rethrows THAT exception instead
        // of a new one (:4019-4031, :4598-4601, :4618-4621).  Which exception flies is not part of the decision (it throws either
        // way): the pass must give the same bits, and executes those lines.
        static const bool exc_ctx = getenv("MMP_REF_EXC_CONTEXT") != nullptr;
        const ModelLoadException lfs = (exc_ctx && (d % 3) != 2) ? ModelLoadException(String("seen before"), null, 0L, null) : ModelLoadException(null);
        const TException ifs = (exc_ctx && (d % 3) != 1) ? newInternalException(String("internal failure seen before"), null) : TException(null);
        try { checkLoadFailureCount(mr, lfs); } catch (const TException &) { bits |= MMP_GATE_FAILURES_BREACHED; }
        try { checkLoadLocationCount(mr, explicitExcludes, ifs); } catch (const TException &) { bits |= MMP_GATE_LOCATIONS_BREACHED; }
        {   // the load-target filter of the request: loaded / failed of the record + the explicit excludes (:4706-4715)
            CacheMissExcludeSet ltf;
            for (auto &e : mr.instanceIds.keySet()) ltf.loaded.add(e);
            for (auto &e : mr.failed.keySet()) ltf.failed.add(e);
            ltf.explicit_ = explicitExcludes;
            try { throwIfLocalLoadNotAllowed(modelId, !(exc_ctx && (d % 5) == 0), mr, &ltf, lfs, ifs); } catch (const TException &) { bits |= MMP_GATE_LOCAL_NOT_ALLOWED; }
        }
        runtimeCache.g_cap = q.cache_capacity;
        runtimeCache.g_wsize = q.cache_weighted_size;
        runtimeCache.g_oldest = q.cache_oldest_time;
        g_fresh = InstanceRecord(q.fresh_lru, q.fresh_capacity, q.fresh_used, 0, q.fresh_count, q.fresh_loading_threads, q.fresh_in_progress, q.fresh_rpm, false);
        try { frag_churn(modelId); } catch (const TException &) { bits |= MMP_GATE_CHURN_REJECT; }
        {
            CacheEntry ce;
            ce.isnull = false;
            ce.predicted = q.loader_predicted;
            Map<String, String> contextMap = Map<String, String>::make();
            if (q.flags & MMP_GATE_HAVE_SIZE_HINT) contextMap.put(KNOWN_SIZE_CXT_KEY, String(std::to_string(q.size_hint)));
            loadingCount.g_v = q.loading_count;
            g_initial_size = 0;
            if (frag_loadLocal_sizing(ce, mr, contextMap, (q.flags & MMP_GATE_WE_CREATED_ENTRY) != 0, q.last_used_time, modelId, g_now) == null)
                bits |= MMP_GATE_EARLY_REJECT;
            else
                gout[d * 2 + 1] = g_initial_size;
        }
        {
            CacheEntry ce;
            ce.isnull = false;
            ce.failed = (q.flags & MMP_GATE_ENTRY_FAILED) != 0;
            ce.modelInfo.t = mr.type;
            g_registry_mr = ModelRecord();  // the registry's view for onEviction: our own load time only
            g_registry_mr.type = mr.type;
            if (q.loaded_time >= 0) g_registry_mr.instanceIds.put(instanceId, Long(q.loaded_time));
            loadTimeoutMs = q.load_timeout_ms;
            if (frag_onEviction(ce, modelId, g_now)) bits |= MMP_GATE_RELOAD_ELSEWHERE;
        }
        {
            runtimeCache.g_cap = q.fresh_capacity;  // the values getFreshInstanceRecord() publishes (after the unload-buffer adjustment)
            runtimeCache.g_wsize = q.fresh_used;
            runtimeCache.g_oldest = q.fresh_lru;
            runtimeCache.g_size = q.fresh_count;
            loadingThreads = q.fresh_loading_threads;
            loadingCount.g_v = q.fresh_in_progress;
            invokeCounter.g_v = q.fresh_rpm;
            shuttingDown = (q.flags & MMP_GATE_FRESH_SHUTTING_DOWN) != 0;
            lastPublished = q.last_published;
            frag_publish((q.flags & MMP_GATE_PUBLISH_FORCE) != 0, (q.flags & MMP_GATE_PRE_SHUTDOWN) != 0);
            if (g_publish) bits |= MMP_GATE_SHOULD_PUBLISH;
        }
        g_gq = nullptr;
        gout[d * 2] = (int32_t)bits;
    }
    wr(o, gout);

    // ---- a15: the scale-up plan of one rateTrackingTask run.  Per entry: action (0 none, 1 second copy, 2 scale-up), copies,
    // timestamp passed to the load, earlierUseIteration / lastUsedIteration afterwards, lastHeavyTime touched; then getExcludeSet()
    if (n_scale >= 0) {
        const mmp_scaleup_params &sp = sparams[0];
        clusterStats = ClusterStats{sstats[0].total_capacity, sstats[0].total_free, sstats[0].global_lru, sstats[0].instance_count, sstats[0].model_copy_count};
        g_tstats.clear();
        for (size_t i = 1; i < sstats.size(); i++)
            g_tstats.push_back(ClusterStats{sstats[i].total_capacity, sstats[i].total_free, sstats[i].global_lru, sstats[i].instance_count, sstats[i].model_copy_count});
        instanceId = (sp.self_pod >= 0 && sp.self_pod < P) ? ids[sp.self_pod] : String("(this instance is not in the table)");
        g_now = sp.now;
        lastCheckTime = sp.last_check_time;
        RATE_CHECK_INTERVAL_MS = sp.rate_check_interval_ms;
        iterationCounter = sp.iteration_counter;
        secondCopyMaxAgeIters = sp.second_copy_max_age_iters;
        secondCopyMinAgeIters = sp.second_copy_min_age_iters;
        secondCopyLruThresholdMillis = (int)sp.second_copy_lru_threshold_ms;
        scaleUpRpmThreshold = sp.scale_up_rpm_threshold;
        limitModelConcurrency = n_conc >= 0;
        averageModelParallelism = 1.0;
        if (n_conc >= 0) {
            if (n_conc != n_scale) { fprintf(stderr, "ref_harness: %lld MaxConcCacheEntry rows for %lld cache entries\n", (long long)n_conc, (long long)n_scale); return 2; }
            dynamicRpmScaleConstant = conc_params[0].dynamic_rpm_scale_constant;
            averageModelParallelism = conc_params[0].average_model_parallelism;
        }
        invokeCounter.g_v = sp.our_rpm;
        g_assume_completed = sp.assume_completed_ms;
        excludeThisInstance = ArrayList_new();
        excludeThisInstance.add(instanceId);
        g_used_since_last_run = Map<String, CacheEntry>::make();
        g_registry_all.clear();
        std::vector<CacheEntry> ces;
        for (int64_t e = 0; e < n_scale; e++) {
            const mmp_cache_entry &x = sentries[e];
            CacheEntry ce;
            ce.isnull = false;
            ce.intervalCount = x.interval_count;
            ce.weight = x.weight;
            ce.earlierUseIteration = x.earlier_use_iteration;
            ce.lastUsedIteration = x.last_used_iteration;
            if (n_conc >= 0) ce.conc = conc_state(e);
            char key[32];
            snprintf(key, sizeof key, "m%09lld", (long long)e);  // the map iterates in key order = the entries' order
            int type = -1;
            if (x.model >= 0 && x.model < M) {
                const mmp_model_row &m = models[x.model];
                type = (m.type < 0 || m.type >= Tn) ? 0 : m.type;
                ModelRecord mr;
                mr.type = String("t" + std::to_string(type));
                for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++) {
                    const int32_t pod = ent_pod[m.ent_off + k];
                    (k < m.n_loaded ? mr.instanceIds : mr.failed).put(ids[pod], Long(ent_time[m.ent_off + k]));
                }
                g_registry_all[key] = mr;
            }
            ce.modelInfo.serviceType = String("t" + std::to_string(type));
            ce.modelInfo.t = ce.modelInfo.serviceType;
            g_used_since_last_run.put(String(key), ce);
            ces.push_back(ce);
        }
        g_async_loads.clear();
        g_exclude_set_seen = Set<String>(null);
        rateTrackingTask_run();
        std::vector<int64_t> so(n_scale * 6, 0);
        for (int64_t e = 0; e < n_scale; e++) {
            so[e * 6 + 3] = (int)ces[e].earlierUseIteration;
            so[e * 6 + 4] = (int)ces[e].lastUsedIteration;
            so[e * 6 + 5] = *ces[e].heavy != 0;
        }
        for (auto &a : g_async_loads) {
            const int64_t e = std::stoll(a.model.substr(1));
            so[e * 6 + 0] = a.ts == sp.last_check_time && a.extra == 0 && a.exclude.size() == 1 ? 1 : 2;
            so[e * 6 + 1] = a.extra + 1;
            so[e * 6 + 2] = a.ts;
        }
        std::vector<uint8_t> ov(P, 0);
        std::vector<int64_t> flag = {g_exclude_set_seen != null ? 1 : 0};  // was getExcludeSet() called at all (it is lazy, :5771)
        if (g_exclude_set_seen != null)
            for (auto &id : g_exclude_set_seen) ov[pod_of[id.str()]] = 1;
        wr(o, so);
        wr(o, flag);
        wr(o, ov);
        if (n_conc >= 0) {  // per entry: what getRpmScaleThreshold(true) returned (0: never called), whether it reset the adder, priorSum /
                            // priorCount afterwards; then the bits of averageModelParallelism after the run
            std::vector<int64_t> co(n_scale * 4 + 1, 0);
            for (int64_t e = 0; e < n_scale; e++) {
                const ConcState &st = *ces[e].conc;
                co[e * 4 + 0] = st.lastThreshold;
                co[e * 4 + 1] = st.resets;
                co[e * 4 + 2] = st.priorSum;
                co[e * 4 + 3] = st.priorCount;
            }
            static_assert(sizeof(double) == sizeof(int64_t), "the double travels as its bits");
            memcpy(&co[n_scale * 4], &averageModelParallelism, sizeof(double));
            wr(o, co);
        }
    }

    // ---- a16: which local copies the janitor removes (removeLocalModelCopyAsync calls), per candidate
    if (n_sd >= 0) {
        const mmp_scaledown_params &dp = dparams[0];
        g_instance_set_stats = ClusterStats{dstats[0].total_capacity, dstats[0].total_free, dstats[0].global_lru, dstats[0].instance_count, dstats[0].model_copy_count};
        instanceId = (dp.self_pod >= 0 && dp.self_pod < P) ? ids[dp.self_pod] : String("(this instance is not in the table)");
        g_now = dp.now;
        shuttingDown = dp.shutting_down != 0;
        lastCheckTime = dp.last_check_time;
        RATE_CHECK_INTERVAL_MS = dp.rate_check_interval_ms;
        scaleUpRpmThreshold = dp.scale_up_rpm_threshold;
        if (n_conc >= 0) {
            if (n_conc != n_sd) { fprintf(stderr, "ref_harness: %lld MaxConcCacheEntry rows for %lld candidates\n", (long long)n_conc, (long long)n_sd); return 2; }
            dynamicRpmScaleConstant = conc_params[0].dynamic_rpm_scale_constant;
        }
        g_adjusted_capacity = dp.adjusted_cache_capacity;
        g_pod_of = pod_of;
        g_in_table.assign(P, 0);
        g_table_rec.assign(P, InstanceRecord(null));
        for (int64_t i = 0; i < P; i++) {
            const mmp_pod_row &r = pods[i];
            g_in_table[i] = !(r.flags & MMP_POD_TOMBSTONE);
            if (g_in_table[i])
                g_table_rec[i] = InstanceRecord(r.lru_time, r.capacity, r.used, r.version, r.count, r.loading_threads, r.loading_in_progress, r.rpm,
                                                (r.flags & MMP_POD_SHUTTING_DOWN) != 0);
        }
        g_scale_candidates.clear();
        g_removed_local.clear();
        for (int64_t e = 0; e < n_sd; e++) {
            const mmp_cache_entry &x = dentries[e];
            if (x.model < 0 || x.model >= M) continue;  // no ModelRecord: the janitor never lists it (MM.java:6076-6090)
            const mmp_model_row &m = models[x.model];
            ModelRecord mr;
            mr.type = String("t0");
            mr.lastUnloadTime = x.last_unload_time;
            for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++) {
                const int32_t pod = ent_pod[m.ent_off + k];
                (k < m.n_loaded ? mr.instanceIds : mr.failed).put(ids[pod], Long(ent_time[m.ent_off + k]));
            }
            CacheEntry ce;
            ce.isnull = false;
            ce.intervalCount = x.interval_count;
            ce.weight = x.weight;
            ce.lastHeavyTimeIn = x.last_heavy_time;
            if (n_conc >= 0) ce.conc = conc_state(e);
            char key[32];
            snprintf(key, sizeof key, "m%09lld", (long long)e);
            ObjectArr3 arr;
            (*arr.p)[0].s = String(key);
            (*arr.p)[1].mr = mr;
            (*arr.p)[2].ce = ce;
            g_scale_candidates.emplace_back(arr, Long(x.last_used));
        }
        janitor_scaledown(dp.now);
        std::vector<uint8_t> removed(n_sd, 0);
        for (auto &id : g_removed_local) removed[std::stoll(id.substr(1))] = 1;
        wr(o, removed);
    }
    // ---- a17: the ensureLoadedInternal calls of one reaper run, in call order: (registry row, lastUsed, subset number)
    if (n_pro >= 0) {
        auto stats_of = [](const StatsRow &r) { return ClusterStats{r.total_capacity, r.total_free, r.global_lru, r.instance_count, r.model_copy_count}; };
        defaultModelSizeUnits = (int)pro_units[0];
        clusterStats = stats_of(pro_gstats[0]);
        auto type_name = [](int t) { return "t" + std::to_string(t); };
        ReaperTypeConstraints tc;
        std::vector<ProhibitedTypeSet> sets;
        if (n_pro > 0) {
            tc.p = std::make_shared<std::vector<InstanceSetStatsTracker>>();
            for (int64_t k = 0; k < n_pro; k++) {
                std::vector<std::string> names;
                for (int32_t t : pro_types[k]) names.push_back(type_name(t));
                sets.push_back(ProhibitedTypeSet(names));
                InstanceSetStatsTracker isst;
                isst.currentStats = stats_of(pro_pstats[k]);
                isst.prohibitedTypesSet = sets.back();
                tc.p->push_back(isst);
            }
            // clusterState with every record's prohibitedTypes set (TypeConstraintManager does this as records arrive)
            auto withsets = std::make_shared<std::vector<Entry<String, InstanceRecord>>>();
            for (const auto &e : *g_cluster) {
                InstanceRecord ir = e.getValue();
                const int32_t part = pro_pod_part[pod_of[e.getKey().str()]];
                // an instance outside every partition carries a set no partition equals
                ir.prohibitedTypes = part >= 0 ? ProhibitedTypeSet(std::vector<std::string>(1, "")) : ProhibitedTypeSet(std::vector<std::string>(1, "\x01none"));
                if (part >= 0) ir.prohibitedTypes = sets[part];
                withsets->push_back(Entry<String, InstanceRecord>(e.getKey(), ir));
            }
            g_cluster = withsets;
        }
        std::vector<Entry<String, ModelRecord>> reg;
        for (int64_t mi = 0; mi < M; mi++) {
            const mmp_model_row &m = models[mi];
            ModelRecord mr;
            mr.type = String(type_name((m.type < 0 || m.type >= Tn) ? 0 : m.type));
            mr.lastUsed = m.last_used;
            for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++)
                (k < m.n_loaded ? mr.instanceIds : mr.failed).put(ids[ent_pod[m.ent_off + k]], Long(ent_time[m.ent_off + k]));
            reg.push_back(Entry<String, ModelRecord>(String("m" + std::to_string(mi)), mr));
        }
        g_proactive_calls.clear();
        g_proactive_partition = -1;
        reaper_proactive(reg, tc);
        std::vector<int64_t> po;
        po.push_back((int64_t)g_proactive_calls.size());
        for (const auto &c : g_proactive_calls) {
            po.push_back(std::stoll(c.model.substr(1)));
            po.push_back(c.ts);
            po.push_back(c.partition);
        }
        wr(o, po);
    }
    // ---- a5: clusterStats + clusterState after every checkpoint of the event stream
    if (n_ev >= 0) {
        SortedClusterState cs;
        clusterStatsTracker = InstanceSetStatsTracker();
        changeCounter = 0;
        g_upgrade_added = g_upgrade_removed = g_housekeepings = g_republish = 0;
        clusterStats = ClusterStats(0L, 0L, Long::MAX_VALUE, 0, 0);
        std::vector<int64_t> co;
        int64_t n_ck = 0;
        co.push_back(0);
        auto checkpoint = [&] {
            n_ck++;
            co.push_back(clusterStats.totalCapacity);
            co.push_back(clusterStats.totalFree);
            co.push_back(clusterStats.globalLru);
            co.push_back(clusterStats.instanceCount);
            co.push_back(clusterStats.modelCopyCount);
            co.push_back((int64_t)cs.v->size());
            for (const auto &e : *cs.v) co.push_back(pod_of[e.getKey().str()]);
        };
        for (int64_t e = 0; e < n_ev; e++) {
            const TableEvent &x = events[e];
            const mmp_pod_row &r = x.row;
            InstanceRecord rec = x.type == ENTRY_DELETED
                                     ? InstanceRecord(null)
                                     : InstanceRecord(r.lru_time, r.capacity, r.used, r.version, r.count, r.loading_threads, r.loading_in_progress, r.rpm,
                                                      (r.flags & MMP_POD_SHUTTING_DOWN) != 0);
            handleInstanceTableChange(cs, (EventType)x.type, ids[x.pod], rec);
            if ((e + 1) % ev_ck[0] == 0 || e + 1 == n_ev) checkpoint();
        }
        co[0] = n_ck;
        co.push_back(g_upgrade_added);
        co.push_back(g_upgrade_removed);
        co.push_back(g_housekeepings);
        co.push_back(g_republish);
        wr(o, co);
    }
    // ---- a19: getLikelyReplacedReplicaSets() after every call: n, then (replica set, expiry) by replica set
    if (n_up >= 0) {
        a19::upgradeTracker.m.clear();
        a19::likelyReplacedReplicaSets = ObjectLongMap<String>();
        std::vector<int64_t> uo;
        for (int64_t e = 0; e < n_up; e++) {
            const UpgradeEvent &x = upevents[e];
            g_now = x.now;
            a19::g_labels_key = x.labels_key;
            char idb[32];
            if (x.replica_set >= 0)
                snprintf(idb, sizeof idb, "%06d-%03d", x.replica_set, (int)(e % 1000));  // first 6 characters = the replica set (:98, :131)
            else
                snprintf(idb, sizeof idb, "i%d", (int)(e % 1000));  // an id of non-standard format (< 7 characters, :86, :121)
            if (x.kind == 0)
                a19::instanceAdded(String(idb), a19::Ir{x.start_time});
            else if (x.kind == 1)
                a19::instanceRemoved(String(idb), a19::Ir{x.start_time});
            else
                a19::doHousekeeping();
            const auto &m = *a19::likelyReplacedReplicaSets.p;
            uo.push_back((int64_t)m.size());
            for (const auto &kv : m) {
                uo.push_back(std::stoll(kv.first));
                uo.push_back(kv.second);
            }
        }
        wr(o, uo);
    }
    // ---- a18: per type row (T configured types + the row of unconfigured types) has_allowed, has_prefer, the two instance sets as
    // bitmaps over the instance index; every instance's partition; per partition its prohibited type rows and stats; the partitions
    // in the reaper's order; per type candidateSubsetStats
    if (n_tc >= 0) {
        auto labels_of = [](uint64_t bits) {
            StringArray a;
            for (int i = 0; i < 64; i++)
                if ((bits >> i) & 1ull) {
                    char b[8];
                    snprintf(b, sizeof b, "l%02d", i);
                    a.p->push_back(String(b));
                }
            return a;  // sorted: two-digit indices
        };
        auto type_name = [](int64_t t) { char b[16]; snprintf(b, sizeof b, "t%03d", (int)t); return String(b); };
        for (const auto &e : *g_cluster) e.getValue().a18_setLabels(labels_of(tc_pod_bits[pod_of[e.getKey().str()]]));
        a18::MtcMap mtcMap;
        for (int64_t t = 0; t < n_tc; t++)  // typeMappingsUpdated, new types (:641-653)
            mtcMap.es.push_back(Entry<String, a18::ModelTypeConstraints>(
                type_name(t), a18::fromInstanceSet(labels_of(tc_req_bits[t]), labels_of(tc_pref_bits[t]), *g_cluster, type_name(t), a18::TrackerArray(null))));
        // :655-664: every instance into the tracker of its ProhibitedTypeSet, its record marked with the set
        a18::PtsMap ptsMap;
        std::vector<int64_t> pod_part(P, -1);
        auto marked = std::make_shared<std::vector<Entry<String, InstanceRecord>>>();
        for (const auto &e : *g_cluster) {
            const ProhibitedTypeSet pts = a18::prohibitedTypesOf(e.getKey(), mtcMap);
            a18::InstanceSetStatsTracker tr(null);
            for (const auto &pe : ptsMap.es)
                if (pe.getKey().equals(pts)) tr = pe.getValue();
            if (!tr.p) {
                tr.p = std::make_shared<a18::TrackerObj>();
                tr->prohibitedTypesSet = pts;
                tr->id = (int)ptsMap.es.size();
                ptsMap.es.push_back(Entry<ProhibitedTypeSet, a18::InstanceSetStatsTracker>(pts, tr));
            }
            InstanceRecord ir = e.getValue();
            tr->sums.add(e.getKey(), ir);
            ir.prohibitedTypes = tr->prohibitedTypesSet;
            pod_part[pod_of[e.getKey().str()]] = tr->id;
            marked->push_back(Entry<String, InstanceRecord>(e.getKey(), ir));
        }
        g_cluster = marked;
        // the trackers' LRU is not set by typeMappingsUpdated; the listener re-accumulates it over ALL of clusterState on the next
        // table event of a member (MM.java:1515-1542: resetLru, then addLru for every remaining entry) — applied here
        for (const auto &pe : ptsMap.es) {
            a18::InstanceSetStatsTracker tr = pe.getValue();
            tr->sums.resetLru();
            for (const auto &e : *g_cluster) tr->sums.addLru(e.getValue().getLruTime());
            tr->currentStats = tr->sums.update();
        }
        a18::refreshPerTypeInstanceSets(mtcMap, ptsMap);
        const int64_t Wd = (P + 63) / 64;
        std::vector<int64_t> to;
        auto put_set = [&](const Set<String> &st) {
            std::vector<uint64_t> w((size_t)Wd, 0);
            if (st != null)
                for (const auto &id : st) { const int p_ = pod_of[id.str()]; w[p_ >> 6] |= 1ull << (p_ & 63); }
            for (uint64_t x : w) to.push_back((int64_t)x);
        };
        for (int64_t t = 0; t <= n_tc; t++) {
            Set<String> al(null), pf = a18::g_defaultPreferred;  // getCandidateInstances / getPreferredInstances of an unconfigured type (:241-251)
            if (t < n_tc) {
                const a18::ModelTypeConstraints m = mtcMap.es[(size_t)t].getValue();
                al = m.allowedInstances;
                pf = m.preferredInstances;
            }
            to.push_back(al != null);
            to.push_back(pf != null);
            put_set(al);
            put_set(pf);
        }
        for (int64_t v : pod_part) to.push_back(v);
        auto put_stats = [&](const ClusterStats &c) {
            to.push_back(c.totalCapacity); to.push_back(c.totalFree); to.push_back(c.globalLru); to.push_back(c.instanceCount); to.push_back(c.modelCopyCount);
        };
        to.push_back((int64_t)ptsMap.es.size());
        for (const auto &pe : ptsMap.es) {
            const auto &ty = pe.getKey().types();
            to.push_back((int64_t)ty.size());
            for (const auto &nm : ty) to.push_back(std::stoll(nm.substr(1)));
            put_stats(pe.getValue()->currentStats);
        }
        std::vector<a18::InstanceSetStatsTracker> order;  // getPartitionStats (:279-289): the list sorted by PARTITION_STATS_COMP
        for (const auto &pe : ptsMap.es) order.push_back(pe.getValue());
        std::stable_sort(order.begin(), order.end(), [](const a18::InstanceSetStatsTracker &a, const a18::InstanceSetStatsTracker &b) { return a18::partition_stats_comp(a, b) < 0; });
        for (const auto &tr : order) to.push_back(tr->id);
        for (int64_t t = 0; t < n_tc; t++) {  // getTypeSetStats(type) (:228-231)
            const a18::NullableStats ns = mtcMap.es[(size_t)t].getValue().candidateSubsetStats();
            to.push_back(ns.isnull ? 0 : 1);
            put_stats(ns.isnull ? ClusterStats() : ns.v);
        }
        wr(o, to);
    }
    // ---- a21: per cache entry bit 0 = triggerNewModelCopyElsewhere was called, bit 1 = the shutdown waits for that copy
    if (n_mig >= 0) {
        instanceId = ids[mig_hdr[0]];
        g_now = mig_hdr[1];
        g_registry_all.clear();
        a21::g_cache.clear();
        std::vector<Entry<String, Long>> cache;
        for (int64_t e = 0; e < n_mig; e++) {
            const mmp_cache_entry &x = mig_entries[e];
            char key[32];
            snprintf(key, sizeof key, "m%09lld", (long long)e);
            if (x.model >= 0 && x.model < M) {
                const mmp_model_row &m = models[x.model];
                ModelRecord mr;
                for (int32_t k = 0; k < m.n_loaded + m.n_failed; k++)
                    (k < m.n_loaded ? mr.instanceIds : mr.failed).put(ids[ent_pod[m.ent_off + k]], Long(ent_time[m.ent_off + k]));
                g_registry_all[key] = mr;
            }
            a21::ShutdownCacheEntry ce;
            ce.isnull = false;
            ce.failed = (x.flags & MMP_CE_FAILED) != 0;  // ce == null || ce.isFailed()
            ce.weight = x.weight;
            ce.ordinal = (int)e;
            a21::g_cache[key] = ce;
            cache.push_back(Entry<String, Long>(String(key), Long(x.last_used)));
        }
        wr(o, a21::migration(cache));
    }
    fclose(o);
    return 0;
}
