// tcm_harness.cc — `oracle/_ref/tcm_harness`: the reference's OWN instance-table listener WITH type constraints, run as C++
// (test infrastructure; third binary of oracle/ref_harness, see harness.cc / clhm_harness.cc).
//
// What executes: ModelMesh.handleInstanceTableChange (MM.java:1455-1568) with `typeConstraints != null`, typeSetStats /
// instanceSetStats (:1432-1448), InstanceSetStatsTracker (whole class), PLACEMENT_ORDER / isFull / getRemaining, and
// TypeConstraintManager's INCREMENTAL path: instanceAdded / instanceRemoved / instanceUpdated / getInstanceSetStats /
// refreshPerTypeInstanceSets / inferPreferredInstances / typeMappingsUpdated, ModelTypeConstraints.updateInstance /
// updateInstanceSet / updateInstanceSetStats / fromInstanceSet / candidateSubsetStats and the accessors the mesh reads
// (getCandidateInstances, getPreferredInstances, getTypeSetStats, getLocalInstanceSetStats) — all cut out of /root/reference by
// line range at build time (extract.py: TCMI_RANGES + the ranges shared with harness.cc) and #included below.
// What this file adds: class shells (field declarations, signatures), stand-ins for the JDK / Guava / eclipse-collections
// containers with their documented behaviour, and the driver that feeds an event stream and reads the manager's state.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "javastub.hpp"
#include "../../include/mmplace.h"

std::vector<std::shared_ptr<void>> g_lists_created;
#define JAVA_ASSERT(x) do { if (!(x)) { fprintf(stderr, "tcm_harness: Java assert failed: %s\n", #x); exit(3); } } while (0)
template <class X> static String LOGSTR(const X &) { return String(""); }

// ---- TypeConstraintManager.ProhibitedTypeSet (:295-333): sorted type names, contains / equals / size
class ProhibitedTypeSet {
    std::shared_ptr<std::vector<std::string>> p;

public:
    ProhibitedTypeSet() {}
    ProhibitedTypeSet(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    explicit ProhibitedTypeSet(const List<String> &types) : p(std::make_shared<std::vector<std::string>>())  // :299-303
    {
        for (int i = 0; i < types.size(); i++) p->push_back(types.get(i).str());
        std::sort(p->begin(), p->end());
    }
    int size() const { return (int)p->size(); }
    const std::vector<std::string> &types() const { return *p; }
    boolean contains(const String &type) const { return std::binary_search(p->begin(), p->end(), type.str()); }
    bool same_types(const ProhibitedTypeSet &o) const { return *p == *o.p; }  // equals(), :314-317
};

// ---- InstanceRecord (InstanceRecord.java:37-69): a record with getters; `prohibitedTypes` is a (transient) FIELD of the shared object
class InstanceRecord {
    struct Rep {
        long lruTime, capacity, used, instanceVersion;
        int count, loadingThreads, loadingInProgress, reqsPerMinute;
        boolean shuttingDown;
        StringArray labels;
        ProhibitedTypeSet prohibitedTypes;
    };
    std::shared_ptr<Rep> p;

public:
    struct PtsField {  // `ir.prohibitedTypes = x` writes the object every holder of the record sees
        InstanceRecord *o;
        const PtsField &operator=(const ProhibitedTypeSet &v) const { o->p->prohibitedTypes = v; return *this; }
        int size() const { return o->p->prohibitedTypes.size(); }
        operator ProhibitedTypeSet() const { return o->p->prohibitedTypes; }
    } prohibitedTypes{this};
    InstanceRecord() {}
    InstanceRecord(std::nullptr_t) {}
    InstanceRecord(const InstanceRecord &o) : p(o.p) {}
    InstanceRecord &operator=(const InstanceRecord &o) { p = o.p; return *this; }
    InstanceRecord(long lru, long cap, long used_, long vers, int cnt, int lt, int lip, int rpm, boolean sd, const StringArray &labels)
        : p(std::make_shared<Rep>(Rep{lru, cap, used_, vers, cnt, lt, lip, rpm, sd, labels, ProhibitedTypeSet()})) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const InstanceRecord &o) const { return p == o.p; }
    long getLruTime() const { return p->lruTime; }
    long getCapacity() const { return p->capacity; }
    long getUsed() const { return p->used; }
    long getInstanceVersion() const { return p->instanceVersion; }
    int getCount() const { return p->count; }
    int getLoadingThreads() const { return p->loadingThreads; }
    int getLoadingInProgress() const { return p->loadingInProgress; }
    int getReqsPerMinute() const { return p->reqsPerMinute; }
    boolean isShuttingDown() const { return p->shuttingDown; }
    String getLocation() const { return null; }
    String getZone() const { return null; }
    StringArray getLabels() const { return p->labels; }
    ProhibitedTypeSet pts() const { return p->prohibitedTypes; }
    long getRemaining() const
    {
        const long capacity = p->capacity, used = p->used;
#include "../_ref/gen/getRemaining_body.inc"
    }
};

static long minSpaceUnits, minChurnAgeMs;
static boolean isFull_(long availableUnits)
{
#include "../_ref/gen/isFull_body.inc"
}
struct StringArrayComp {  // Utils.STRING_ARRAY_COMP (Utils.java:25-36)
    int compare(const StringArray &l1, const StringArray &l2) const
    {
#include "../_ref/gen/string_array_comp_body.inc"
    }
    int operator()(const StringArray &a, const StringArray &b) const { return compare(a, b); }
};
static const StringArrayComp STRING_ARRAY_COMP{};
static const struct { StringArrayComp STRING_ARRAY_COMP; } Utils{};
static boolean isFull(long v) { return isFull_(v); }
static int placement_order_compare(Entry<String, InstanceRecord> e1, Entry<String, InstanceRecord> e2)
{
#include "../_ref/gen/placement_order_compare_body.inc"
}
static const struct { int compare(Entry<String, InstanceRecord> a, Entry<String, InstanceRecord> b) const { return placement_order_compare(a, b); } } PLACEMENT_ORDER;

struct ClusterStats {  // MM.java:1570-1590
    long totalCapacity = 0, totalFree = 0, globalLru = 0;
    int instanceCount = 0, modelCopyCount = 0;
    bool isnull = false;
    ClusterStats() {}
    ClusterStats(std::nullptr_t) : isnull(true) {}
    ClusterStats(long c, long f, long l, int n, int m) : totalCapacity(c), totalFree(f), globalLru(l), instanceCount(n), modelCopyCount(m) {}
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
};
static const ClusterStats EMPTY_STATS(0L, 0L, Long::MAX_VALUE, 0, 0);  // InstanceSetStatsTracker.java:33

// ---- InstanceSetStatsTracker (InstanceSetStatsTracker.java:31-93): every method body the reference's; a handle
struct LongPredicate { boolean test(long v) const { return isFull_(v); } };
struct TrackerRep {
    LongPredicate isFull;
    ProhibitedTypeSet prohibitedTypesSet;
    long totalCapacity = 0, totalFree = 0, lru = Long::MAX_VALUE;
    int count = 0, modelCount = 0;
    ClusterStats currentStats = EMPTY_STATS;
    TrackerRep(ProhibitedTypeSet prohibitedTypesSet, LongPredicate isFull)
    {
#include "../_ref/gen/tcmi_ist_ctor_body.inc"
    }
    int getInstanceCount()
    {
#include "../_ref/gen/tcmi_ist_getInstanceCount_body.inc"
    }
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
class InstanceSetStatsTracker {
public:
    std::shared_ptr<TrackerRep> p;
    ProhibitedTypeSet prohibitedTypesSet;  // final
    InstanceSetStatsTracker() {}
    InstanceSetStatsTracker(std::nullptr_t) {}
    InstanceSetStatsTracker(const ProhibitedTypeSet &pts, LongPredicate f) : p(std::make_shared<TrackerRep>(pts, f)), prohibitedTypesSet(pts) {}
    template <class F> InstanceSetStatsTracker(const ProhibitedTypeSet &pts, F) : InstanceSetStatsTracker(pts, LongPredicate()) {}  // (pts, this::isFull)
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const InstanceSetStatsTracker &o) const { return p == o.p; }
    TrackerRep *operator->() const { return p.get(); }
    int getInstanceCount() const { return p->getInstanceCount(); }
    void resetLru() const { p->resetLru(); }
    void addLru(long l) const { p->addLru(l); }
    void add(const String &i, const InstanceRecord &r) const { p->add(i, r); }
    boolean remove(const String &i, const InstanceRecord &r) const { return p->remove(i, r); }
    ClusterStats update() const { return p->update(); }
};
struct TrackerArray {  // InstanceSetStatsTracker[] (nullable)
    std::shared_ptr<std::vector<InstanceSetStatsTracker>> p;
    TrackerArray() {}
    TrackerArray(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    int length() const { return (int)p->size(); }
    const InstanceSetStatsTracker &operator[](int i) const { return (*p)[i]; }
    std::vector<InstanceSetStatsTracker>::const_iterator begin() const { return p->begin(); }
    std::vector<InstanceSetStatsTracker>::const_iterator end() const { return p->end(); }
};
template <> class Set<InstanceSetStatsTracker> {  // HashSet<InstanceSetStatsTracker>: by identity (the class does not override equals)
public:
    std::shared_ptr<std::vector<InstanceSetStatsTracker>> p;
    Set() {}
    Set(std::nullptr_t) {}
    Set(const Set<String> &) : p(std::make_shared<std::vector<InstanceSetStatsTracker>>()) {}  // `new HashSet<>(n)`
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    void clear() const { p->clear(); }
    boolean contains(const InstanceSetStatsTracker &t) const { return std::find(p->begin(), p->end(), t) != p->end(); }
    boolean add(const InstanceSetStatsTracker &t) const { if (contains(t)) return false; p->push_back(t); return true; }
    int size() const { return (int)p->size(); }
    TrackerArray toArray_() const { TrackerArray a; a.p = std::make_shared<std::vector<InstanceSetStatsTracker>>(*p); return a; }
};

// ---- library stand-ins the TypeConstraintManager text names
template <class X> using Predicate = std::function<bool(const X &)>;
template <class X> struct Stream {
    std::vector<X> v;
    boolean allMatch(const Predicate<X> &f) const { for (auto &x : v) if (!f(x)) return false; return true; }
    boolean anyMatch(const Predicate<X> &f) const { for (auto &x : v) if (f(x)) return true; return false; }
};
static const struct {
    int binarySearch(const StringArray &a, const String &key) const
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
    Stream<InstanceSetStatsTracker> stream(const TrackerArray &a) const { Stream<InstanceSetStatsTracker> s; s.v = *a.p; return s; }
    String toString(const StringArray &) const { return String(""); }
    boolean equals(const StringArray &a, const StringArray &b) const
    {
        if (a.length() != b.length()) return false;
        for (int i = 0; i < a.length(); i++) if (!a[i].equals(b[i])) return false;
        return true;
    }
} Arrays;
struct ImmutableSetBuilder {  // ImmutableSet.Builder<String> (nullable)
    Set<String> s;
    ImmutableSetBuilder() {}
    ImmutableSetBuilder(std::nullptr_t) {}
    bool operator==(std::nullptr_t) const { return s == null; }
    bool operator!=(std::nullptr_t) const { return s != null; }
    const ImmutableSetBuilder &add(const String &x) const { s.add(x); return *this; }
    template <class C> const ImmutableSetBuilder &addAll(const C &c) const { for (auto &x : c) s.add(x); return *this; }
    Set<String> build() const { return s; }
};
static const struct {
    ImmutableSetBuilder builder() const { ImmutableSetBuilder b; b.s = Set<String>::make(); return b; }
    ImmutableSetBuilder builderWithExpectedSize(int) const { return builder(); }
    Set<String> copyOf(const Set<String> &src) const { Set<String> c = Set<String>::make(); for (auto &x : src) c.add(x); return c; }
} ImmutableSet;
static const struct {
    template <class P> std::vector<String> filter(const Set<String> &s, P pred) const { std::vector<String> o; for (auto &x : s) if (pred(x)) o.push_back(x); return o; }
} SetsX;
#define Sets SetsX
static const struct {
    // com.google.common.base.Objects.equal / java.util.Objects.equals: null-safe equals(); Set.equals = same elements
    boolean equal(const Set<String> &a, const Set<String> &b) const
    {
        if (a == null || b == null) return a == null && b == null;
        return a.p->size() == b.p->size() && std::equal(a.p->begin(), a.p->end(), b.p->begin(), String::Eq());
    }
    boolean equals(const String &a, const String &b) const { return a == null ? b == null : (b != null && a.equals(b)); }
} Objects;
template <class T> using ArrayList = List<T>;
template <class K> struct ObjectIntPair { K k; int v; K getOne() const { return k; } int getTwo() const { return v; } };
template <class K> struct ObjectIntMap {
    std::shared_ptr<std::map<std::string, int>> p = std::make_shared<std::map<std::string, int>>();
    void put(const K &k, int v) const { (*p)[k.str()] = v; }
    boolean containsKey(const K &k) const { return p->count(k.str()) != 0; }
    void addToValue(const K &k, int d) const { (*p)[k.str()] += d; }
    std::vector<ObjectIntPair<K>> keyValuesView() const { std::vector<ObjectIntPair<K>> o; for (auto &e : *p) o.push_back({String(e.first), e.second}); return o; }
};
template <class K> using MutableObjectIntMap = ObjectIntMap<K>;
static ObjectIntMap<String> ObjectIntHashMap_new(int) { return ObjectIntMap<String>(); }
static boolean g_debug = false;
static const struct {
    void warn(const String &) const {}
    void info(const String &) const {}
    void debug(const String &) const {}
    boolean isDebugEnabled() const { return g_debug; }
} logger;
static boolean empty(const StringArray &a) { return a.length() == 0; }  // Utils.empty
static const StringArray NO_LABELS;

// ---- clusterState = new ConcurrentSkipListSet<>(PLACEMENT_ORDER) (MM.java:774)
struct SortedClusterState {
    std::shared_ptr<std::vector<Entry<String, InstanceRecord>>> v = std::make_shared<std::vector<Entry<String, InstanceRecord>>>();
    boolean add(const Entry<String, InstanceRecord> &e) const
    {
        auto it = std::lower_bound(v->begin(), v->end(), e, [](const Entry<String, InstanceRecord> &a, const Entry<String, InstanceRecord> &b) { return placement_order_compare(a, b) < 0; });
        if (it != v->end() && placement_order_compare(*it, e) == 0) return false;
        v->insert(it, e);
        return true;
    }
    Iterator<Entry<String, InstanceRecord>> iterator() const
    {
        auto vv = v;
        auto i = std::make_shared<size_t>(0);
        return Iterator<Entry<String, InstanceRecord>>([vv, i] { return *i < vv->size(); }, [vv, i] { return (*vv)[(*i)++]; },
                                                       [vv, i] { vv->erase(vv->begin() + (long)--*i); });
    }
    int size() const { return (int)v->size(); }
    // a for-each over the set while the listener's own iterator is positioned in it (instanceRemoved → refreshPerTypeInstanceSets
    // runs inside the listener's loop, before it.remove()): a snapshot, as the skip list's weakly consistent iterator gives
    std::vector<Entry<String, InstanceRecord>> snapshot() const { return *v; }
    std::vector<Entry<String, InstanceRecord>>::const_iterator begin() const { return v->begin(); }
    std::vector<Entry<String, InstanceRecord>>::const_iterator end() const { return v->end(); }
};

// ---- TypeConstraintManager.ModelTypeConstraints (:337-506): immutable; identity matters (`newMtc != mtc`, `return this`)
struct NullableStats {
    bool isnull = true;
    ClusterStats v;
    NullableStats(std::nullptr_t) {}
    NullableStats(const ClusterStats &c) : isnull(c.isnull), v(c) {}
    bool operator!=(std::nullptr_t) const { return !isnull; }
    operator ClusterStats() const { return isnull ? ClusterStats(null) : v; }
};
static boolean instanceMatches(StringArray instanceLabels, StringArray typeLabels, boolean matchAll)
{
#include "../_ref/gen/tcm_instanceMatches_body.inc"
}
static Set<String> updateInstanceSet(String iid, StringArray instanceLabels, StringArray typeLabels, Set<String> instanceSet, boolean matchAll)
{
#include "../_ref/gen/tcmi_updateInstanceSet_body.inc"
}
class ModelTypeConstraints {
    std::shared_ptr<char> id;  // object identity

public:
    StringArray requiredLabels, preferredLabels;
    Set<String> allowedInstances, preferredInstances, configuredPreferredInstances;
    TrackerArray instanceSetStats;
    ModelTypeConstraints() {}
    ModelTypeConstraints(std::nullptr_t) {}
    ModelTypeConstraints(const ModelTypeConstraints *self) { *this = *self; }  // `return this`
    ModelTypeConstraints(StringArray requiredLabels, StringArray preferredLabels, Set<String> allowedInstances, Set<String> configuredPreferredInstances,
                         TrackerArray instanceSetStats, Set<String> resolvedPreferredInstances)
        : id(std::make_shared<char>(0))
    {
#include "../_ref/gen/tcmi_mtc_ctor_body.inc"
    }
    bool operator==(std::nullptr_t) const { return !id; }
    bool operator!=(std::nullptr_t) const { return (bool)id; }
    bool operator==(const ModelTypeConstraints &o) const { return id == o.id; }
    bool operator!=(const ModelTypeConstraints &o) const { return id != o.id; }
    NullableStats candidateSubsetStats() const
    {
#include "../_ref/gen/tcm_candidateSubsetStats_body.inc"
    }
    ModelTypeConstraints updateInstanceSetStats(Set<InstanceSetStatsTracker> newStats, Set<String> newInferredPreferred) const
    {
#include "../_ref/gen/tcmi_updateInstanceSetStats_body.inc"
    }
    static ModelTypeConstraints fromInstanceSet(StringArray requiredLabels, StringArray preferredLabels, const SortedClusterState &instances, String typeName,
                                                TrackerArray instanceSetStats)
    {
#include "../_ref/gen/tcm_fromInstanceSet_body.inc"
    }
    boolean allowedOnInstance(String iid) const
    {
#include "../_ref/gen/tcm_allowedOnInstance_body.inc"
    }
    boolean labelsMatch(StringArray required, StringArray preferred) const
    {
#include "../_ref/gen/tcmi_labelsMatch_body.inc"
    }
    ModelTypeConstraints updateInstance(String iid, StringArray labels) const
    {
#include "../_ref/gen/tcmi_updateInstance_body.inc"
    }
};
// Map<String, ModelTypeConstraints> (HashMap / ImmutableMap): entries write through (ent.setValue, :710, :721); `!=` is identity
template <> class Map<String, ModelTypeConstraints> {
public:
    typedef std::map<std::string, Entry<String, ModelTypeConstraints>> Rep;
    std::shared_ptr<Rep> p;
    Map() {}
    Map(std::nullptr_t) {}
    static Map make() { Map m; m.p = std::make_shared<Rep>(); return m; }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const Map &o) const { return p == o.p; }
    bool operator!=(const Map &o) const { return p != o.p; }
    int size() const { return (int)p->size(); }
    ModelTypeConstraints get(const String &k) const { auto it = p->find(k.str()); return it == p->end() ? ModelTypeConstraints(null) : it->second.getValue(); }
    void put(const String &k, const ModelTypeConstraints &v) const { (*p)[k.str()] = Entry<String, ModelTypeConstraints>(k, v); }
    void remove(const String &k) const { p->erase(k.str()); }
    std::vector<Entry<String, ModelTypeConstraints>> entrySet() const { std::vector<Entry<String, ModelTypeConstraints>> es; for (auto &kv : *p) es.push_back(kv.second); return es; }
    Set<String> keySet() const { Set<String> ks = Set<String>::make(); for (auto &kv : *p) ks.add(String(kv.first)); return ks; }
};
template <class K, class V> using HashMap = Map<K, V>;
static Map<String, ModelTypeConstraints> HashMap_new(const Map<String, ModelTypeConstraints> &src)  // `new HashMap<>(map)`: a copy with entries of its own
{
    Map<String, ModelTypeConstraints> m = Map<String, ModelTypeConstraints>::make();
    for (auto &kv : *src.p) m.put(kv.second.getKey(), kv.second.getValue());
    return m;
}
static const struct { Map<String, ModelTypeConstraints> copyOf(const Map<String, ModelTypeConstraints> &m) const { return HashMap_new(m); } } ImmutableMap;
struct ConfigTypeConstraints {  // :79-102 (normalised by the driver: sorted, deduplicated, disjoint)
    bool isnull = true;
    StringArray required, preferred;
    ConfigTypeConstraints() {}
    ConfigTypeConstraints(std::nullptr_t) {}
    ConfigTypeConstraints(const StringArray &r, const StringArray &f) : isnull(false), required(r), preferred(f) {}
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
    boolean isEmpty() const
    {
#include "../_ref/gen/tcmi_cfg_isEmpty_body.inc"
    }
};
template <> class Map<String, ConfigTypeConstraints> {
public:
    std::shared_ptr<std::map<std::string, ConfigTypeConstraints>> p = std::make_shared<std::map<std::string, ConfigTypeConstraints>>();
    ConfigTypeConstraints remove(const String &k) const
    {
        auto it = p->find(k.str());
        if (it == p->end()) return ConfigTypeConstraints(null);
        ConfigTypeConstraints c = it->second;
        p->erase(it);
        return c;
    }
    void put(const String &k, const ConfigTypeConstraints &v) const { (*p)[k.str()] = v; }
    std::vector<Entry<String, ConfigTypeConstraints>> entrySet() const { std::vector<Entry<String, ConfigTypeConstraints>> es; for (auto &kv : *p) es.emplace_back(String(kv.first), kv.second); return es; }
};

// ---- TypeConstraintManager: fields as declared at :123-152; every method body below is the reference's
#include "../_ref/gen/tcmi_default_type_constant.inc"
struct LabelsMap {  // new TreeMap<>(Utils.STRING_ARRAY_COMP), :134-135
    std::vector<std::pair<StringArray, InstanceSetStatsTracker>> v;
    InstanceSetStatsTracker get(const StringArray &k) const { for (auto &e : v) if (STRING_ARRAY_COMP.compare(e.first, k) == 0) return e.second; return null; }
    void put(const StringArray &k, const InstanceSetStatsTracker &t) { for (auto &e : v) if (STRING_ARRAY_COMP.compare(e.first, k) == 0) { e.second = t; return; } v.emplace_back(k, t); }
    void remove(const StringArray &k) { for (size_t i = 0; i < v.size(); i++) if (STRING_ARRAY_COMP.compare(v[i].first, k) == 0) { v.erase(v.begin() + (long)i); return; } }
    void clear() { v.clear(); }
};
struct PtsMap {  // new HashMap<ProhibitedTypeSet, InstanceSetStatsTracker>(), :137 (equals / hashCode of the key: its types)
    std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>> es;
    InstanceSetStatsTracker get(const ProhibitedTypeSet &k) const { for (auto &e : es) if (e.getKey().same_types(k)) return e.getValue(); return null; }
    void put(const ProhibitedTypeSet &k, const InstanceSetStatsTracker &t) { es.emplace_back(k, t); }
    void remove(const ProhibitedTypeSet &k) { for (size_t i = 0; i < es.size(); i++) if (es[i].getKey().same_types(k)) { es.erase(es.begin() + (long)i); return; } }
    boolean isEmpty() const { return es.empty(); }
    int size() const { return (int)es.size(); }
    void clear() { es.clear(); }
    struct LogStream { LogStream map_log() const { return *this; } String collect_joining() const { return String(""); } };
    struct EntrySet {
        const std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>> *es;
        LogStream stream() const { return LogStream(); }
        std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>>::const_iterator begin() const { return es->begin(); }
        std::vector<Entry<ProhibitedTypeSet, InstanceSetStatsTracker>>::const_iterator end() const { return es->end(); }
    };
    EntrySet entrySet() const { return EntrySet{&es}; }
    struct Values { const PtsMap *m; void forEach_update() const { for (auto &e : m->es) e.getValue().update(); } };
    Values values() const { return Values{this}; }
};
static Set<String> inferPreferredInstances(ObjectIntMap<String> instanceScores, Set<String> include)
{
#include "../_ref/gen/tcm_inferPreferredInstances_body.inc"
}
class TypeConstraintManager {
public:
    SortedClusterState clusterState;
    LongPredicate isFull;
    String localInstanceId;
    LabelsMap labelsToInstanceSetStats;
    PtsMap ptsToInstanceSetStats;
    InstanceSetStatsTracker localInstanceSetStats;
    Map<String, ModelTypeConstraints> typeConstraintsMap = Map<String, ModelTypeConstraints>::make();  // Collections.emptyMap()
    Set<String> defaultPreferredInstances;
    int n_refresh = 0;
    NullableStats getTypeSetStats(String type)
    {
#include "../_ref/gen/tcmi_getTypeSetStats_body.inc"
    }
    ClusterStats getLocalInstanceSetStats()
    {
#include "../_ref/gen/tcmi_getLocalInstanceSetStats_body.inc"
    }
    Set<String> getCandidateInstances(String type)
    {
#include "../_ref/gen/tcmi_getCandidateInstances_body.inc"
    }
    Set<String> getPreferredInstances(String type)
    {
#include "../_ref/gen/tcmi_getPreferredInstances_body.inc"
    }
    ModelTypeConstraints getTypeConstraints(String type)
    {
#include "../_ref/gen/tcmi_getTypeConstraints_body.inc"
    }
    InstanceSetStatsTracker getStatsForLabels(StringArray labels)
    {
#include "../_ref/gen/tcmi_getStatsForLabels_body.inc"
    }
    InstanceSetStatsTracker instanceAdded(String iid, StringArray labels, boolean includedInStats)
    {
#include "../_ref/gen/tcmi_instanceAdded_body.inc"
    }
    void instanceRemoved(String iid, StringArray labels)
    {
#include "../_ref/gen/tcmi_instanceRemoved_body.inc"
    }
    InstanceSetStatsTracker getInstanceSetStats(String iid, StringArray labels, Map<String, ModelTypeConstraints> tcMap)
    {
#include "../_ref/gen/tcmi_getInstanceSetStats_body.inc"
    }
    static Map<String, ModelTypeConstraints> instanceUpdated(String iid, StringArray labels, Map<String, ModelTypeConstraints> mtcMap)
    {
#include "../_ref/gen/tcmi_instanceUpdated_body.inc"
    }
    void typeMappingsUpdated(Map<String, ConfigTypeConstraints> newConfig)
    {
#include "../_ref/gen/tcmi_typeMappingsUpdated_body.inc"
    }
    Map<String, ModelTypeConstraints> refreshPerTypeInstanceSets(Map<String, ModelTypeConstraints> mtcMap)
    {
        n_refresh++;
#include "../_ref/gen/tcmi_refreshPerTypeInstanceSets_body.inc"
    }
};

// ---- ModelMesh: the listener and what it names
enum EventType { ENTRY_ADDED, ENTRY_UPDATED, ENTRY_DELETED };
static TypeConstraintManager *g_tcm;
static const struct {
    bool operator!=(std::nullptr_t) const { return true; }
    bool operator==(std::nullptr_t) const { return false; }
    InstanceSetStatsTracker getStatsForLabels(const StringArray &l) const { return g_tcm->getStatsForLabels(l); }
    InstanceSetStatsTracker instanceAdded(const String &i, const StringArray &l, boolean inc) const { return g_tcm->instanceAdded(i, l, inc); }
    void instanceRemoved(const String &i, const StringArray &l) const { g_tcm->instanceRemoved(i, l); }
    NullableStats getTypeSetStats(const String &t) const { return g_tcm->getTypeSetStats(t); }
    ClusterStats getLocalInstanceSetStats() const { return g_tcm->getLocalInstanceSetStats(); }
} typeConstraints;
static SortedClusterState clusterState;
static InstanceSetStatsTracker clusterStatsTracker;  // new InstanceSetStatsTracker(null, this::isFull), MM.java:1451
static ClusterStats clusterStats;
static int changeCounter;
static String instanceId;
static int g_upgrade_added, g_upgrade_removed, g_housekeepings, g_republish;
static const struct { template <class K, class V> Entry<K, V> immutableEntry(const K &k, const V &v) const { return Entry<K, V>(k, v); } } Maps;
static ClusterStats typeSetStats(String modelType)
{
#include "../_ref/gen/tcmi_typeSetStats_body.inc"
}
static ClusterStats instanceSetStats()
{
#include "../_ref/gen/tcmi_instanceSetStats_body.inc"
}
static void handleInstanceTableChange(EventType type, String key, InstanceRecord record)
{
    const struct {
        void instanceRemoved(const String &, const InstanceRecord &) const { g_upgrade_removed++; }
        void instanceAdded(const String &, const InstanceRecord &) const { g_upgrade_added++; }
        void doHousekeeping() const { g_housekeepings++; }
    } upgradeTracker;
    struct LeaderElection {
        LeaderElection() {}
        bool operator!=(std::nullptr_t) const { return true; }
        boolean isLeader() const { return true; }  // (this instance is the leader: `missings.remove(key)` runs)
    };
    const LeaderElection leaderLatch;
    const struct { void remove(const String &) const {} } missings;
    auto publishInstanceRecordAsync = [] { g_republish++; };
#include "../_ref/gen/handleInstanceTableChange_body.inc"
}

// =================================================== driver + I/O ===========================================================
template <class X> static std::vector<X> rd(FILE *f, size_t n)
{
    std::vector<X> v(n);
    if (n && fread(v.data(), sizeof(X), n, f) != n) { fprintf(stderr, "tcm_harness: short read\n"); exit(2); }
    return v;
}
static StringArray labels_of(uint64_t bits)
{
    StringArray a;
    for (int i = 0; i < 64; i++)
        if ((bits >> i) & 1ull) {
            char b[8];
            snprintf(b, sizeof b, "l%02d", i);
            a.p->push_back(String(b));
        }
    return a;  // sorted (two digits), deduplicated
}
static String type_name(int t) { char b[8]; snprintf(b, sizeof b, "t%02d", t); return String(b); }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: tcm_harness <input.bin> <output.bin>\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    auto magic = rd<char>(f, 8);
    if (memcmp(magic.data(), "MMTCM1\0\0", 8) != 0) { fprintf(stderr, "tcm_harness: bad magic\n"); return 1; }
    // header: P (instance indices), T (configured types), n events, checkpoint period, minSpaceUnits, minChurnAgeMs, local instance, debug
    auto H = rd<int64_t>(f, 8);
    const int64_t P = H[0], T = H[1], n_ev = H[2], ck = H[3];
    minSpaceUnits = H[4];
    minChurnAgeMs = H[5];
    const int64_t local = H[6];
    g_debug = H[7] != 0;
    auto idbuf = rd<char>(f, (size_t)P * 16);
    auto req_bits = rd<uint64_t>(f, (size_t)T), pref_bits = rd<uint64_t>(f, (size_t)T);
    // event: kind 0/1/2 = ENTRY_ADDED / UPDATED / DELETED of instance `pod` with `row` and label bits; kind 3 = the type-constraint
    // configuration changes: type `pod` gets required = labels, preferred = aux (both 0: the type's entry is removed)
    struct Ev { int32_t kind, pod; uint64_t labels, aux; mmp_pod_row row; };
    auto events = rd<Ev>(f, (size_t)n_ev);
    fclose(f);
    std::vector<String> ids;
    std::unordered_map<std::string, int> pod_of;
    for (int64_t p = 0; p < P; p++) {
        ids.push_back(String(std::string(&idbuf[(size_t)p * 16])));
        pod_of[ids.back().str()] = (int)p;
    }
    TypeConstraintManager tcm;
    g_tcm = &tcm;
    tcm.clusterState = clusterState;  // the "live" set of the mesh (shared storage)
    tcm.localInstanceId = local >= 0 ? ids[(size_t)local] : String("");
    instanceId = tcm.localInstanceId;
    clusterStatsTracker = InstanceSetStatsTracker(ProhibitedTypeSet(null), LongPredicate());
    clusterStats = EMPTY_STATS;
    std::vector<std::pair<uint64_t, uint64_t>> cfg((size_t)T);
    auto apply_config = [&] {  // TypeConstraintManager.updateTypeMappings → typeMappingsUpdated(config): the whole configuration, as parsed
        Map<String, ConfigTypeConstraints> m;
        for (int64_t t = 0; t < T; t++)  // (a type configured with no labels at all is in the map too: the manager warns and drops it)
            m.put(type_name((int)t), ConfigTypeConstraints(labels_of(cfg[(size_t)t].first), labels_of(cfg[(size_t)t].second & ~cfg[(size_t)t].first)));
        tcm.typeMappingsUpdated(m);
    };
    for (int64_t t = 0; t < T; t++) cfg[(size_t)t] = {req_bits[(size_t)t], pref_bits[(size_t)t]};
    apply_config();  // at start-up, before any instance event (TypeConstraintManager.get, :157-178)

    const int64_t W = (P + 63) / 64;
    std::vector<int64_t> out;
    int64_t n_ck = 0, e_now = -1;
    out.push_back(0);
    auto push_stats = [&](const ClusterStats &s) {
        out.push_back(s.totalCapacity);
        out.push_back(s.totalFree);
        out.push_back(s.globalLru);
        out.push_back(s.instanceCount);
        out.push_back(s.modelCopyCount);
    };
    auto type_mask = [&](const ProhibitedTypeSet &pts) {
        uint64_t m = 0;
        for (auto &s : pts.types()) m |= 1ull << std::stoi(s.substr(1));
        return (int64_t)m;
    };
    auto push_set = [&](const Set<String> &s) {
        out.push_back(s == null ? 0 : 1);
        std::vector<uint64_t> w((size_t)W, 0);
        if (s != null)
            for (auto &x : s) w[(size_t)(pod_of[x.str()] >> 6)] |= 1ull << (pod_of[x.str()] & 63);
        for (auto x : w) out.push_back((int64_t)x);
    };
    auto checkpoint = [&] {
        n_ck++;
        push_stats(clusterStats);
        out.push_back((int64_t)clusterState.v->size());
        for (const auto &e : *clusterState.v) {
            out.push_back(pod_of[e.getKey().str()]);
            out.push_back(type_mask(e.getValue().pts()));
        }
        for (int64_t t = 0; t <= T; t++) {  // row T: a type without configured constraints
            const String name = t < T ? type_name((int)t) : String("zz");
            push_set(tcm.getCandidateInstances(name));   // what CacheMissForwardingLB.filter constrains to (MM.java:4787)
            push_set(tcm.getPreferredInstances(name));   // :4788
            push_stats(typeSetStats(name));              // MM.java:1432
        }
        std::vector<std::pair<int64_t, const InstanceSetStatsTracker *>> parts;
        for (auto &e : tcm.ptsToInstanceSetStats.es) parts.emplace_back(type_mask(e.getKey()), &e.getValue());
        std::sort(parts.begin(), parts.end(), [](auto &a, auto &b) { return (uint64_t)a.first < (uint64_t)b.first; });
        out.push_back((int64_t)parts.size());
        for (auto &pr : parts) {
            out.push_back(pr.first);
            push_stats((*pr.second)->currentStats);
            out.push_back((*pr.second)->count);
        }
        push_stats(instanceSetStats());  // MM.java:1446: the local instance's partition
        out.push_back(tcm.n_refresh);
        out.push_back(e_now);
    };
    for (int64_t e = 0; e < n_ev; e++) {
        const Ev &x = events[(size_t)e];
        e_now = e;
        if (x.kind == 3) {
            cfg[(size_t)x.pod] = {x.labels, x.aux};
            apply_config();
        } else {
            const mmp_pod_row &r = x.row;
            InstanceRecord rec(r.lru_time, r.capacity, r.used, r.version, r.count, r.loading_threads, r.loading_in_progress, r.rpm,
                               (r.flags & MMP_POD_SHUTTING_DOWN) != 0, labels_of(x.labels));
            handleInstanceTableChange((EventType)x.kind, ids[(size_t)x.pod], rec);
        }
        if ((e + 1) % ck == 0 || e + 1 == n_ev || x.kind == 3) checkpoint();  // (after a configuration change too: there the manager recomputes everything)
    }
    out[0] = n_ck;
    out.push_back(g_upgrade_added);
    out.push_back(g_upgrade_removed);
    out.push_back(g_housekeepings);
    out.push_back(g_republish);
    FILE *g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    fwrite(out.data(), 8, out.size(), g);
    fclose(g);
    return 0;
}
