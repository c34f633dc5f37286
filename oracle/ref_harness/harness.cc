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
static const struct { void warn(const char *) const {} } logger;
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
    const int64_t n_serve = H[11], n_sexcl = H[12];
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
    fclose(o);
    return 0;
}
