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
static boolean sendDestinationId = false;
static long g_now;
static long currentTimeMillis() { return g_now; }
static InstanceRecord g_fresh;
static InstanceRecord getFreshInstanceRecord() { return g_fresh; }  // MM.java:5369 (the request carries the row)
static uint32_t g_pick;
static const struct {
    struct R { int nextInt(int n) const { return (int)(((uint64_t)g_pick * (uint64_t)(uint32_t)n) >> 32); } };
    R current() const { return R(); }
} ThreadLocalRandom;  // :4981: the pick is an input of every restatement (SURVEY B#9)
static const struct { void warn(const String &) const {} void info(const String &) const {} } logger;
static const String CACHE_MISS_EXCLUDES_KEY("tas.cm_excludes"), DEST_INST_ID_KEY("tas.dest_iid");
struct ThreadContextT { int getCurrentContext() const { return 0; } };
static const ThreadContextT ThreadContext;
static Map<String, String> ensureContextMapIsMutable(int) { return Map<String, String>::make(); }

static std::shared_ptr<std::vector<Entry<String, InstanceRecord>>> g_cluster;  // clusterState, kept in PLACEMENT_ORDER
static const struct { Iterator<Entry<String, InstanceRecord>> iterator() const { return iterate(g_cluster); } } clusterState;
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
struct ClusterStats { long totalCapacity, totalFree, globalLru; int instanceCount, modelCopyCount; };
static std::vector<ClusterStats> g_tstats;
static ClusterStats typeSetStats(const String &type) { return g_tstats.at(std::min<size_t>(std::stoul(type.str().substr(1)), g_tstats.size() - 1)); }

// ModelRecord (ModelRecord.java:61-114): type + the two id -> time maps
struct ModelRecord {
    String type;
    Map<String, Long> instanceIds = Map<String, Long>::make(), failed = Map<String, Long>::make();
    bool isnull = false;
    ModelRecord() {}
    ModelRecord(std::nullptr_t) : isnull(true) {}
    bool operator!=(std::nullptr_t) const { return !isnull; }
    String getType() const { return type; }
    Map<String, Long> getInstanceIds() const { return instanceIds; }
    Map<String, Long> getLoadFailedInstanceIds() const { return failed; }
    boolean hasLoadFailure() const { return !failed.isEmpty(); }  // ModelRecord.java:203
    String getLoadFailureMessage(const String &) const { return String(""); }
};
struct CacheEntry {  // what the fragments ask of a CacheEntry<?>
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
    struct MI { String t; String getServiceType() const { return t; } } modelInfo;
};
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

// ======================================================== I/O ===============================================================
// The audit hash of a shortlist (machinery of THIS repository, not of the reference): a function of the set of
// rank positions in the shortlist and of the count that survived the rpm filter — computed here from the reference's own
// candidate list and the reference's own clusterState order, so that a fixture row is 16 bytes instead of a ragged list.
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static uint32_t shortlist_hash(const std::vector<int32_t> &pos_of, const int32_t *cand, int32_t n, int32_t remaining, int64_t n_pods)
{
    std::vector<uint64_t> bits((size_t)((n_pods + 63) / 64) + 1, 0);
    for (int32_t i = 0; i < n; i++) bits[pos_of[cand[i]] >> 6] |= 1ull << (pos_of[cand[i]] & 63);
    uint64_t h = 0;
    for (size_t w = 0; w < bits.size(); w++)
        if (bits[w]) h += splitmix64(bits[w] ^ (0x9E3779B97F4A7C15ull * (uint64_t)(w + 1)));
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
    fclose(f);

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
        (*g_replaced.p)[String(key)] = 0;
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
        try { checkLoadFailureCount(mr, null); } catch (const TException &) { bits |= MMP_GATE_FAILURES_BREACHED; }
        try { checkLoadLocationCount(mr, explicitExcludes, null); } catch (const TException &) { bits |= MMP_GATE_LOCATIONS_BREACHED; }
        {   // the load-target filter of the request: loaded / failed of the record + the explicit excludes (:4706-4715)
            CacheMissExcludeSet ltf;
            for (auto &e : mr.instanceIds.keySet()) ltf.loaded.add(e);
            for (auto &e : mr.failed.keySet()) ltf.failed.add(e);
            ltf.explicit_ = explicitExcludes;
            try { throwIfLocalLoadNotAllowed(modelId, true, mr, &ltf, null, null); } catch (const TException &) { bits |= MMP_GATE_LOCAL_NOT_ALLOWED; }
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
    fclose(o);
    return 0;
}
