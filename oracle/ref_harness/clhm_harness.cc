// clhm_harness.cc — `oracle/_ref/clhm_harness`: the reference's OWN local-cache text run as C++ (test infrastructure).
//
// What executes: the method bodies of clhm/ConcurrentLinkedHashMap.java (put / get / remove / replaceQuietly / setCapacity, the
// Add / Update / Removal tasks, evict, the read buffers and their drain, notifyListener, Node.touch), clhm/LinkedDeque.java
// (insert / reposition / unlink / poll) and ModelCacheUnloadBufManager.java (every accounting method), plus the three
// ModelMesh.java fragments between them (CacheEntry.getWeight / updateWeightLocked, newInternalCacheEntry, onEviction's call of
// entryRemoved) — cut out of /root/reference by line range at build time (extract.py: CLHM_RANGES) and #included below.
// What this file adds: the class shells those bodies sit in (field declarations, method signatures — Java's, transcribed),
// stand-ins for java.util.concurrent (AtomicLong, AtomicReference, ReentrantLock, the queues: single-threaded, documented
// semantics) and the driver that maps one mmp_cache_op (include/mmplace.h) onto the call ModelMesh makes for it.  No eviction,
// ordering or accounting rule is written here.
//
// Java objects have reference semantics: every class below is a HANDLE (shared storage, `==` is identity, `null` compares like
// Java's null).  A handle carries copies of the object's FINAL fields (Node.key, WeightedValue.weight/.value,
// CacheEntry.modelId) so that the reference's field reads (`node.key`) compile as they stand.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "javastub.hpp"
#include "../../include/mmplace.h"

std::vector<std::shared_ptr<void>> g_lists_created;  // (javastub.hpp's ArrayList bookkeeping: unused here)

#define JAVA_ASSERT(x) do { if (!(x)) { fprintf(stderr, "clhm_harness: Java assert failed: %s\n", #x); exit(3); } } while (0)
static long g_now;
static long currentTimeMillis() { return g_now; }
static long Thread_currentThread_getId() { return 1; }  // one thread
struct IllegalArgumentException { IllegalArgumentException(const String &) {} };
static void checkArgument(boolean ok) { if (!ok) { fprintf(stderr, "clhm_harness: checkArgument\n"); exit(3); } }
template <class X> static void checkNotNull(const X &x) { if (x == null) { fprintf(stderr, "clhm_harness: checkNotNull\n"); exit(3); } }
static inline long jmax(int a, long b) { return (long)a > b ? (long)a : b; }
// Math.max(0, long) (clhm :672) and Math.max(1L, long): Java widens the int
static const struct {
    int max(int a, int b) const { return a > b ? a : b; }
    long max(long a, long b) const { return a > b ? a : b; }
    long max(int a, long b) const { return jmax(a, b); }
    int min(int a, int b) const { return a < b ? a : b; }
    long min(long a, long b) const { return a < b ? a : b; }
    long min(long a, int b) const { return a < b ? a : (long)b; }
    int abs(int a) const { return a < 0 ? (int)(0u - (unsigned)a) : a; }
    long abs(long a) const { return a < 0 ? (long)(0ul - (unsigned long)a) : a; }
} MathX;
#define Math MathX

// ---- java.util.concurrent stand-ins (one thread: get / lazySet are plain reads and writes)
class AtomicLong {
    std::shared_ptr<long> p = std::make_shared<long>(0);

public:
    AtomicLong() {}
    explicit AtomicLong(long v) { *p = v; }
    long get() const { return *p; }
    void lazySet(long v) const { *p = v; }
};
template <class T> class AtomicReference {
    std::shared_ptr<T> p = std::make_shared<T>();

public:
    AtomicReference() {}
    explicit AtomicReference(const T &v) { *p = v; }
    T get() const { return *p; }
    void lazySet(const T &v) const { *p = v; }
};
struct Condition { void signalAll() const {} };
struct Lock {  // ReentrantLock, uncontended
    void lock() const {}
    void unlock() const {}
    boolean tryLock() const { return true; }
    Condition newCondition() const { return Condition(); }
};
enum DrainStatus { IDLE, PROCESSING };

// ---- ModelMesh.CacheEntry (MM.java:1632-1790): what the cache and the manager ask of it
class ConcurrentLinkedHashMapRef;
class CacheEntry;
struct CacheEntryRep : std::enable_shared_from_this<CacheEntryRep> {
    String modelId;
    int weight;
    int32_t key;  // the driver's interned id
    ConcurrentLinkedHashMapRef *runtimeCache_p;
    CacheEntryRep(const String &id, int w) : modelId(id), weight(w), key(0), runtimeCache_p(nullptr) {}
    int getWeight() const
    {
#include "../_ref/gen/mm_ce_getWeight_body.inc"
    }
    void updateWeightLocked(int newWeight);
};
class CacheEntry {
public:
    std::shared_ptr<CacheEntryRep> p;
    String modelId;  // final
    CacheEntry() {}
    CacheEntry(std::nullptr_t) {}
    CacheEntry(CacheEntryRep *r) : p(r->shared_from_this()), modelId(r->modelId) {}  // `this` handed to a method taking a CacheEntry
    CacheEntry(const String &id, int w) : p(std::make_shared<CacheEntryRep>(id, w)), modelId(id) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const CacheEntry &o) const { return p == o.p; }
    bool operator!=(const CacheEntry &o) const { return p != o.p; }
    boolean equals(const CacheEntry &o) const { return p == o.p; }  // Object.equals: identity (CacheEntry does not override it)
    int getWeight() const { return p->getWeight(); }
    void updateWeightLocked(int w) const { p->updateWeightLocked(w); }
};

typedef String K;
typedef CacheEntry V;

// ---- clhm.WeightedValue (clhm :1296-1332): immutable; compareAndSet on the node compares references
template <class VV> class WeightedValue {
    std::shared_ptr<char> id;

public:
    int weight;  // final
    VV value;    // final
    WeightedValue() : weight(0) {}
    WeightedValue(const VV &value, int weight) : id(std::make_shared<char>(0)), weight(weight), value(value) {}
    bool same(const WeightedValue &o) const { return id == o.id; }
    boolean contains(const VV &o) const
    {
#include "../_ref/gen/clhm_wv_contains_body.inc"
    }
    boolean isAlive() const
    {
#include "../_ref/gen/clhm_wv_isAlive_body.inc"
    }
};

// ---- clhm.Node (clhm :1338-1396) extends AtomicReference<WeightedValue<V>> implements Linked<Node>
class NodeH;
struct NodeRep {
    K key;
    std::shared_ptr<NodeRep> prev, next;
    long lastUsed = 0;
    WeightedValue<V> ref;  // the AtomicReference's value
    void touch(long time)
    {
#include "../_ref/gen/clhm_node_touch_body.inc"
    }
    long getLastUsed() const
    {
#include "../_ref/gen/clhm_node_getLastUsed_body.inc"
    }
    WeightedValue<V> get() const { return ref; }
    V getValue() const
    {
#include "../_ref/gen/clhm_node_getValue_body.inc"
    }
};
class NodeH {
public:
    std::shared_ptr<NodeRep> p;
    K key;  // final
    NodeH() {}
    NodeH(std::nullptr_t) {}
    explicit NodeH(std::shared_ptr<NodeRep> r) : p(std::move(r)) { if (p) key = p->key; }
    NodeH(const K &key, const WeightedValue<V> &weightedValue, long time) : p(std::make_shared<NodeRep>()), key(key)  // clhm :1351-1355
    {
        p->key = key;
        p->ref = weightedValue;
        p->touch(time);
    }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const NodeH &o) const { return p == o.p; }
    bool operator!=(const NodeH &o) const { return p != o.p; }
    void touch(long t) const { p->touch(t); }
    long getLastUsed() const { return p->getLastUsed(); }
    NodeH getPrevious() const { return NodeH(p->prev); }   // clhm :1369-1389: plain field accessors
    void setPrevious(const NodeH &n) const { p->prev = n.p; }
    NodeH getNext() const { return NodeH(p->next); }
    void setNext(const NodeH &n) const { p->next = n.p; }
    WeightedValue<V> get() const { return p->get(); }
    boolean compareAndSet(const WeightedValue<V> &expect, const WeightedValue<V> &update) const
    {
        if (!p->ref.same(expect)) return false;
        p->ref = update;
        return true;
    }
    V getValue() const { return p->getValue(); }
    operator struct ReadRecord() const;  // (the conditional expression of afterRead, :383, needs one static type in C++)
};
template <class A, class B> using Node = NodeH;

// ---- clhm.LinkedDeque<E> (LinkedDeque.java): the bodies are the reference's
template <class E> class LinkedDeque {
public:
    E first, last;
    void linkFirst(E e)
    {
#include "../_ref/gen/deque_linkFirst_body.inc"
    }
    E unlinkFirst()
    {
#include "../_ref/gen/deque_unlinkFirst_body.inc"
    }
    void unlink(E e)
    {
#include "../_ref/gen/deque_unlink_body.inc"
    }
    boolean isEmpty()
    {
#include "../_ref/gen/deque_isEmpty_body.inc"
    }
    boolean contains(E e)
    {
#include "../_ref/gen/deque_contains_body.inc"
    }
    void reposition(E e)
    {
#include "../_ref/gen/deque_reposition_body.inc"
    }
    boolean insert(E e)
    {
#include "../_ref/gen/deque_insert_body.inc"
    }
    E peekFirst()
    {
#include "../_ref/gen/deque_peekFirst_body.inc"
    }
    E poll()
    {
#include "../_ref/gen/deque_poll_body.inc"
    }
    E pollFirst()
    {
#include "../_ref/gen/deque_pollFirst_body.inc"
    }
    boolean remove(E e)
    {
#include "../_ref/gen/deque_remove_body.inc"
    }
};

// ---- the read buffer's records: a Node, or a ReadRecord (clhm :364-380) = node + a non-zero lastUsedTime
struct ReadRecord {
    NodeH n;
    long lu = 0;
    bool plain_node = true;
    ReadRecord() {}
    ReadRecord(const NodeH &node, long lastUsedTime) : n(node), lu(lastUsedTime), plain_node(false) {}
    NodeH node() const { return n; }
    long lastUsedTime() const { return lu; }
};
NodeH::operator ReadRecord() const { ReadRecord r; r.n = *this; return r; }  // a bare Node in the buffer: "now"
struct Object {  // java.lang.Object holding one of the two, or null
    std::shared_ptr<ReadRecord> p;
    Object() {}
    Object(std::nullptr_t) {}
    Object(const ReadRecord &r) : p(std::make_shared<ReadRecord>(r)) {}
    bool operator==(std::nullptr_t) const { return !p; }
    operator NodeH() const { return p->n; }
    operator ReadRecord() const { return *p; }
};
static boolean instanceof_Node(const Object &o) { return o.p->plain_node; }

struct Runnable {
    std::function<void()> f;
    void run() const { f(); }
};
template <class T> struct ConcurrentLinkedQueue {
    std::deque<T> q;
    boolean add(const T &t) { q.push_back(t); return true; }
    T poll()
    {
        if (q.empty()) return T(null);
        T t = q.front();
        q.pop_front();
        return t;
    }
};
struct ConcurrentHashMapKN {  // ConcurrentHashMap<K, Node<K, V>>
    std::map<std::string, NodeH> m;
    NodeH get(const K &k) const { auto it = m.find(k.str()); return it == m.end() ? NodeH(null) : it->second; }
    NodeH putIfAbsent(const K &k, const NodeH &n)
    {
        auto it = m.find(k.str());
        if (it != m.end()) return it->second;
        m[k.str()] = n;
        return NodeH(null);
    }
    NodeH remove(const K &k)
    {
        auto it = m.find(k.str());
        if (it == m.end()) return NodeH(null);
        NodeH n = it->second;
        m.erase(it);
        return n;
    }
    boolean remove(const K &k, const NodeH &n)
    {
        auto it = m.find(k.str());
        if (it == m.end() || it->second != n) return false;
        m.erase(it);
        return true;
    }
};

// the listener: ModelMesh itself (MM.java:2863-2879)
class ModelMesh;
struct EvictionListener {
    ModelMesh *mm = nullptr;
    void onEviction(const K &key, const V &value) const {}  // MM.java:2864: empty
};
template <class A, class B> struct EvictionListenerWithTime {
    ModelMesh *mm;
    EvictionListenerWithTime(const EvictionListener &l) : mm(l.mm) {}
    void onEviction(const K &key, const V &ce, long lastUsed) const;
};
static boolean instanceof_EvictionListenerWithTime(const EvictionListener &) { return true; }  // ModelMesh implements it (MM.java:142)
struct Weigher { int weightOf(const K &, const V &value) const { return value.getWeight(); } };  // MM.java:390: CacheEntry::getWeight

static int NCPU = 1;
static int ceilingNextPowerOfTwo(int x) { int p = 1; while (p < x) p <<= 1; return p; }  // clhm :1252 (1 << (32 - nlz(x - 1)))
#include "../_ref/gen/clhm_constants.inc"
static const long EMPTY_OLDEST_TIME = -1L;  // clhm :1117

// ---- clhm.ConcurrentLinkedHashMap<K, V>: fields as declared at clhm :186-226, constructor :228-260 transcribed
class ConcurrentLinkedHashMap {
public:
    ConcurrentHashMapKN data;
    AtomicLong capacity;
    Weigher weigher;
    Lock evictionLock;
    AtomicLong weightedSize;
    LinkedDeque<Node<K, V>> evictionDeque;
    AtomicReference<DrainStatus> drainStatus{IDLE};
    std::vector<long> readBufferReadCount;
    std::vector<AtomicLong> readBufferWriteCount, readBufferDrainAtWriteCount;
    std::vector<std::vector<AtomicReference<Object>>> readBuffers;
    EvictionListener listener;
    ConcurrentLinkedQueue<Node<K, V>> pendingNotifications;
    long oldestTime_ = EMPTY_OLDEST_TIME;
#define oldestTime oldestTime_ /* Java keeps fields and methods apart (`oldestTime` is both, :1120 / :1125) */

    ConcurrentLinkedHashMap(long cap, ModelMesh *mm) : capacity(Math.min(cap, MAXIMUM_CAPACITY))
    {
        readBufferReadCount.assign(NUMBER_OF_READ_BUFFERS, 0);
        for (int i = 0; i < NUMBER_OF_READ_BUFFERS; i++) {
            readBufferWriteCount.push_back(AtomicLong());
            readBufferDrainAtWriteCount.push_back(AtomicLong());
            readBuffers.emplace_back();
            for (int j = 0; j < READ_BUFFER_SIZE; j++) readBuffers[i].push_back(AtomicReference<Object>());
        }
        listener.mm = mm;
    }
    Lock getEvictionLock() const { return evictionLock; }
    long capacity_() const
    {
#include "../_ref/gen/clhm_capacity_body.inc"
    }
    void setCapacity(long capacity)
    {
#include "../_ref/gen/clhm_setCapacity_body.inc"
    }
    boolean hasOverflowed()
    {
#include "../_ref/gen/clhm_hasOverflowed_body.inc"
    }
    void evict()
    {
#include "../_ref/gen/clhm_evict_body.inc"
    }
    void afterRead(Node<K, V> node, long lastUsed)
    {
#include "../_ref/gen/clhm_afterRead_body.inc"
    }
    static int readBufferIndex()
    {
#include "../_ref/gen/clhm_readBufferIndex_body.inc"
    }
    long recordRead(int bufferIndex, Object record)
    {
#include "../_ref/gen/clhm_recordRead_body.inc"
    }
    void drainOnReadIfNeeded(int bufferIndex, long writeCount)
    {
#include "../_ref/gen/clhm_drainOnReadIfNeeded_body.inc"
    }
    void afterWrite(Runnable task, boolean notify)
    {
#include "../_ref/gen/clhm_afterWrite_body.inc"
    }
    void tryToDrainBuffers()
    {
#include "../_ref/gen/clhm_tryToDrainBuffers_body.inc"
    }
    void drainBuffers()
    {
#include "../_ref/gen/clhm_drainBuffers_body.inc"
    }
    void drainReadBuffer(int bufferIndex)
    {
#include "../_ref/gen/clhm_drainReadBuffer_body.inc"
    }
    void applyRead(Node<K, V> node)
    {
#include "../_ref/gen/clhm_applyRead_body.inc"
    }
    boolean tryToRetire(Node<K, V> node, WeightedValue<V> expect)
    {
#include "../_ref/gen/clhm_tryToRetire_body.inc"
    }
    void makeRetired(Node<K, V> node)
    {
#include "../_ref/gen/clhm_makeRetired_body.inc"
    }
    void makeDead(Node<K, V> node)
    {
#include "../_ref/gen/clhm_makeDead_body.inc"
    }
    void notifyListener()
    {
#include "../_ref/gen/clhm_notifyListener_body.inc"
    }
    // the three inner task classes (clhm :590-652): `new AddTask(node, weight)` captures the enclosing map as Java's inner
    // classes do; run() is the reference's body
    void AddTask_run(Node<K, V> node, int weight)
    {
#include "../_ref/gen/clhm_AddTask_run_body.inc"
    }
    Runnable AddTask(Node<K, V> node, int weight) { return Runnable{[=] { AddTask_run(node, weight); }}; }
    void RemovalTask_run(Node<K, V> node)
    {
#include "../_ref/gen/clhm_RemovalTask_run_body.inc"
    }
    Runnable RemovalTask(Node<K, V> node) { return Runnable{[=] { RemovalTask_run(node); }}; }
    void UpdateTask_run(Node<K, V> node, int weightDifference, long newTime)
    {
#include "../_ref/gen/clhm_UpdateTask_run_body.inc"
    }
    Runnable UpdateTask(Node<K, V> node, int weightDifference, long newTime) { return Runnable{[=] { UpdateTask_run(node, weightDifference, newTime); }}; }
    long weightedSize_() const
    {
#include "../_ref/gen/clhm_weightedSize_body.inc"
    }
    V get(K key, long lastUsed)
    {
#include "../_ref/gen/clhm_get_body.inc"
    }
    V getQuietly(K key)
    {
#include "../_ref/gen/clhm_getQuietly_body.inc"
    }
    V putIfAbsent(K key, V value, long lastUsed)
    {
#include "../_ref/gen/clhm_putIfAbsent3_body.inc"
    }
    V put(K key, V value, long lastUsed, boolean onlyIfAbsent)
    {
#include "../_ref/gen/clhm_put_body.inc"
    }
    V remove(K key)
    {
#include "../_ref/gen/clhm_remove_body.inc"
    }
    boolean remove(K key, V value)
    {
#include "../_ref/gen/clhm_remove2_body.inc"
    }
    boolean replaceQuietly(K key, V oldValue, V newValue)
    {
#include "../_ref/gen/clhm_replaceQuietly_body.inc"
    }
    long oldestTime_get()
    {
#include "../_ref/gen/clhm_oldestTime_body.inc"
    }
    void updateOldestTime()
    {
#include "../_ref/gen/clhm_updateOldestTime_body.inc"
    }
#undef oldestTime
};
// a reference to the map (`runtimeCache` is a field holding one): forwards
class ConcurrentLinkedHashMapRef {
public:
    ConcurrentLinkedHashMap *m = nullptr;
    long capacity() const { return m->capacity_(); }
    long weightedSize() const { return m->weightedSize_(); }
    void setCapacity(long c) const { m->setCapacity(c); }
    V putIfAbsent(const K &k, const V &v, long lu) const { return m->putIfAbsent(k, v, lu); }
    V getQuietly(const K &k) const { return m->getQuietly(k); }
    boolean remove(const K &k, const V &v) const { return m->remove(k, v); }
    boolean replaceQuietly(const K &k, const V &a, const V &b) const { return m->replaceQuietly(k, a, b); }
    Lock getEvictionLock() const { return m->getEvictionLock(); }
};
void CacheEntryRep::updateWeightLocked(int newWeight)
{
    ConcurrentLinkedHashMapRef &runtimeCache = *runtimeCache_p;
#include "../_ref/gen/mm_ce_updateWeightLocked_body.inc"
}

// ---- ModelCacheUnloadBufManager: fields as declared at :41-77, every method body the reference's
static const struct { void warn(const String &) const {} void info(const String &) const {} } logger;
static const long UNIT_SIZE = 8192;  // ModelLoader.java:37 (log messages only)
static String mb(long) { return String(""); }
#include "../_ref/gen/ubm_key_constant.inc"
class ModelCacheUnloadBufManager {
public:
    ConcurrentLinkedHashMapRef runtimeCache;
    CacheEntry UNLOAD_BUFF;
    Lock cacheLock;
    Condition cacheLockCondition;
    int unloadsReservedSizeUnits = 0;
    int totalUnloadingWeight = 0;
    long totalModelCacheOccupancy = 0;
    int cacheDeficit = 0;
    ModelCacheUnloadBufManager(ModelMesh &mm, ConcurrentLinkedHashMapRef cache, int unloadsReservedSizeUnits);
    long getAdjustedCacheCapacity()
    {
#include "../_ref/gen/ubm_getAdjustedCacheCapacity_body.inc"
    }
    int getUnloadBufferWeight()
    {
#include "../_ref/gen/ubm_getUnloadBufferWeight_body.inc"
    }
    CacheEntry insertNewEntry(String modelId, CacheEntry ce, long lastUsedTime)
    {
#include "../_ref/gen/ubm_insertNewEntry_body.inc"
    }
    void adjustNewEntrySpaceRequest(int increase, CacheEntry entry, boolean weakPrediction)
    {
#include "../_ref/gen/ubm_adjustNewEntrySpaceRequest_body.inc"
    }
    boolean claimRequestedSpaceIfReady(int required)
    {
#include "../_ref/gen/ubm_claimRequestedSpaceIfReady_body.inc"
    }
    void adjustWeightAfterLoad(int delta, CacheEntry entry)
    {
#include "../_ref/gen/ubm_adjustWeightAfterLoad_body.inc"
    }
    CacheEntry insertFailedPlaceholderEntry(String modelId, CacheEntry ce, long lastUsedTime)
    {
#include "../_ref/gen/ubm_insertFailedPlaceholderEntry_body.inc"
    }
    int removeEntry(CacheEntry entry)
    {
#include "../_ref/gen/ubm_removeEntry_body.inc"
    }
    void entryRemoved(int weight)
    {
#include "../_ref/gen/ubm_entryRemoved_body.inc"
    }
    void unloadComplete(int weight, boolean success, String modelId)
    {
#include "../_ref/gen/ubm_unloadComplete_body.inc"
    }
    void discardFailedEntry(int weight)
    {
#include "../_ref/gen/ubm_discardFailedEntry_body.inc"
    }
    int cacheRemaining()
    {
#include "../_ref/gen/ubm_cacheRemaining_body.inc"
    }
    void payDownDeficitAndNotifyWaiters(int weight, boolean releaseFromUnloadingWeight, boolean notify)
    {
#include "../_ref/gen/ubm_payDownDeficit_body.inc"
    }
    void adjustTotalModelCacheOccupancy(int delta)
    {
#include "../_ref/gen/ubm_adjustTotalModelCacheOccupancy_body.inc"
    }
    void adjustAggregateUnloadingWeight(int delta)
    {
#include "../_ref/gen/ubm_adjustAggregateUnloadingWeight_body.inc"
    }
    boolean cacheSpaceIsReady(int required)
    {
#include "../_ref/gen/ubm_cacheSpaceIsReady_body.inc"
    }
};
struct UnloadManagerRef {  // the nullable field `unloadManager` (MM.java:401)
    ModelCacheUnloadBufManager *p = nullptr;
    bool operator!=(std::nullptr_t) const { return p != nullptr; }
    void entryRemoved(int w) const { p->entryRemoved(w); }
};

// ---- ModelMesh: the fields and the three fragments that touch the cache
class ModelMesh {
public:
    ConcurrentLinkedHashMapRef runtimeCache;
    UnloadManagerRef unloadManager;
    std::vector<int32_t> evicted;  // the driver's record of the eviction callbacks, in callback order
    CacheEntry newCacheEntry(const String &id, int weight, int32_t key)  // `new CacheEntry(id, weight)` inside ModelMesh: an inner class
    {
        CacheEntry ce(id, weight);
        ce.p->runtimeCache_p = &runtimeCache;
        ce.p->key = key;
        return ce;
    }
    CacheEntry CacheEntry_(const String &id, int weight) { return newCacheEntry(id, weight, MMP_UNLOADBUF_KEY); }
    CacheEntry newInternalCacheEntry(String id, int weight)
    {
#define CacheEntry(a, b) CacheEntry_(a, b) /* the `new CacheEntry(id, weight)` of :1618, an inner-class creation */
#include "../_ref/gen/mm_newInternalCacheEntry_body.inc"
#undef CacheEntry
    }
    void onEviction(String key, CacheEntry ce, long lastUsed)
    {
        evicted.push_back(ce.p->key);
#include "../_ref/gen/mm_onEviction_weight_fragment.inc"
#include "../_ref/gen/mm_onEviction_manager_fragment.inc"
    }
};
template <class A, class B> void EvictionListenerWithTime<A, B>::onEviction(const K &key, const V &ce, long lastUsed) const { mm->onEviction(key, ce, lastUsed); }
ModelCacheUnloadBufManager::ModelCacheUnloadBufManager(ModelMesh &mm, ConcurrentLinkedHashMapRef cache, int unloadsReservedSizeUnits)
{
#include "../_ref/gen/ubm_ctor_body.inc"
}

// =================================================== driver + I/O ===========================================================
template <class X> static std::vector<X> rd(FILE *f, size_t n)
{
    std::vector<X> v(n);
    if (n && fread(v.data(), sizeof(X), n, f) != n) { fprintf(stderr, "clhm_harness: short read\n"); exit(2); }
    return v;
}
template <class X> static void wr(FILE *f, const std::vector<X> &v)
{
    if (!v.empty() && fwrite(v.data(), sizeof(X), v.size(), f) != v.size()) { fprintf(stderr, "clhm_harness: short write\n"); exit(2); }
}
static String key_name(int32_t key) { return key == MMP_UNLOADBUF_KEY ? UNLOAD_BUFFER_CACHE_KEY : String("m" + std::to_string(key)); }

struct World {  // one instance's local cache: MM.java:737-760
    ModelMesh mm;
    std::unique_ptr<ConcurrentLinkedHashMap> cache;
    std::unique_ptr<ModelCacheUnloadBufManager> ubm;
    World(long capacity, int reserved)
    {
        cache.reset(new ConcurrentLinkedHashMap(capacity, &mm));
        mm.runtimeCache.m = cache.get();
        if (reserved >= 0) {
            ubm.reset(new ModelCacheUnloadBufManager(mm, mm.runtimeCache, reserved));
            mm.unloadManager.p = ubm.get();
        }
    }
    CacheEntry existing_or_detached(int32_t key)  // the CacheEntry object the caller holds: the cached one, else one that is not (or no longer) in the cache
    {
        CacheEntry ce = cache->getQuietly(key_name(key));
        return ce != null ? ce : mm.newCacheEntry(key_name(key), 1, key);
    }
};

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: clhm_harness <input.bin> <output.bin>\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    auto magic = rd<char>(f, 8);
    if (memcmp(magic.data(), "MMCLHM1\0", 8) != 0) { fprintf(stderr, "clhm_harness: bad magic\n"); return 1; }
    auto H = rd<int64_t>(f, 3);
    const int64_t n_caches = H[0], n_ops = H[1];
    g_now = H[2];
    auto caps = rd<int64_t>(f, (size_t)n_caches);
    auto reserved = rd<int32_t>(f, (size_t)n_caches);  // < 0: no unload manager (MM.java:745: the loader does not support explicit unloading)
    auto ops = rd<mmp_cache_op>(f, (size_t)n_ops);
    fclose(f);

    std::vector<std::unique_ptr<World>> worlds;
    for (int64_t c = 0; c < n_caches; c++) worlds.emplace_back(new World(caps[c], reserved[c]));
    std::vector<mmp_cache_op_out> outs((size_t)n_ops);
    std::vector<int32_t> evicted_all;
    for (int64_t i = 0; i < n_ops; i++) {
        const mmp_cache_op &op = ops[i];
        World &w = *worlds[op.cache];
        ConcurrentLinkedHashMap &rc = *w.cache;
        ModelCacheUnloadBufManager *um = w.ubm.get();
        w.mm.evicted.clear();
        const String id = key_name(op.key);
        int32_t res = 0;
        if (op.op >= MMP_COP_UBM_INSERT_NEW_ENTRY && !um) { fprintf(stderr, "clhm_harness: manager op on an unmanaged cache\n"); return 2; }
        switch (op.op) {
        case MMP_COP_PUT_IF_ABSENT:  // runtimeCache.putIfAbsent(modelId, ce, lastUsed), as MM.java:3747 / :1620
            res = rc.putIfAbsent(id, w.mm.newCacheEntry(id, op.arg, op.key), op.time) == null ? 1 : 0;
            break;
        case MMP_COP_GET:  // runtimeCache.get(modelId, lastUsed), MM.java:3609
            res = rc.get(id, op.time) != null ? 1 : 0;
            break;
        case MMP_COP_UPDATE_WEIGHT: {  // a present entry's weight becomes arg: quietly (time -1: CacheEntry.updateWeightLocked →
            // replaceQuietly, MM.java:1778-1787) or with a timestamp (put of a replacement value over the same key, clhm :821-857)
            CacheEntry ce = rc.getQuietly(id);
            if (ce == null) { res = 0; break; }
            res = 1;
            if (op.time < 0) ce.updateWeightLocked(op.arg);
            else rc.put(id, w.mm.newCacheEntry(id, op.arg, op.key), op.time, false);
            break;
        }
        case MMP_COP_REMOVE:
            res = rc.remove(id) != null ? 1 : 0;
            break;
        case MMP_COP_UBM_INSERT_NEW_ENTRY:
            res = um->insertNewEntry(id, w.mm.newCacheEntry(id, op.arg, op.key), op.time) == null ? 1 : 0;
            break;
        case MMP_COP_UBM_ADJUST_SPACE_REQUEST: {
            CacheEntry ce = rc.getQuietly(id);
            res = ce != null ? 1 : 0;
            if (ce != null) um->adjustNewEntrySpaceRequest(op.arg, ce, op.flag != 0);
            break;
        }
        case MMP_COP_UBM_SPACE_IS_READY:
            res = um->cacheSpaceIsReady(op.arg) ? 1 : 0;
            break;
        case MMP_COP_UBM_CLAIM_SPACE:
            res = um->claimRequestedSpaceIfReady(op.arg) ? 1 : 0;
            break;
        case MMP_COP_UBM_ADJUST_AFTER_LOAD:
            res = (rc.getQuietly(id) != null || op.arg == 0) ? 1 : 0;  // (the ABI's convention for a void method)
            um->adjustWeightAfterLoad(op.arg, w.existing_or_detached(op.key));
            break;
        case MMP_COP_UBM_UNLOAD_COMPLETE:
            um->unloadComplete(op.arg, op.flag != 0, id);
            res = op.flag != 0 ? 1 : 0;
            break;
        case MMP_COP_UBM_REMOVE_ENTRY:
            res = um->removeEntry(w.existing_or_detached(op.key));
            break;
        case MMP_COP_UBM_DISCARD_FAILED:
            um->discardFailedEntry(op.arg);
            res = 1;
            break;
        case MMP_COP_UBM_INSERT_FAILED_PLACEHOLDER:
            res = um->insertFailedPlaceholderEntry(id, w.mm.newCacheEntry(id, op.arg, op.key), op.time) == null ? 1 : 0;
            break;
        default:
            fprintf(stderr, "clhm_harness: unknown op %d\n", op.op);
            return 2;
        }
        mmp_cache_op_out &o = outs[(size_t)i];
        memset(&o, 0, sizeof o);
        o.result = res;
        o.n_evicted = (int32_t)w.mm.evicted.size();
        o.evicted_off = (int32_t)evicted_all.size();
        evicted_all.insert(evicted_all.end(), w.mm.evicted.begin(), w.mm.evicted.end());
        o.buffer_weight = um ? um->getUnloadBufferWeight() : 0;
        o.weighted_size = w.mm.runtimeCache.weightedSize();
        o.oldest_time = rc.oldestTime_get();
    }

    FILE *g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    wr(g, outs);
    wr(g, std::vector<int64_t>{(int64_t)evicted_all.size()});
    wr(g, evicted_all);
    // the final state of every cache: the eviction deque first → last, the map's capacity / weighted size, the manager's fields
    for (int64_t c = 0; c < n_caches; c++) {
        World &w = *worlds[c];
        std::vector<int32_t> keys, weights;
        std::vector<int64_t> lus;
        for (NodeH n = w.cache->evictionDeque.first; n != null; n = n.getNext()) {
            keys.push_back(n.getValue().p->key);
            weights.push_back(n.get().weight);
            lus.push_back(n.getLastUsed());
        }
        wr(g, std::vector<int64_t>{(int64_t)keys.size(), w.cache->capacity_(), w.cache->weightedSize.get(),
                                   w.ubm ? (int64_t)w.ubm->totalUnloadingWeight : 0, w.ubm ? w.ubm->totalModelCacheOccupancy : 0,
                                   w.ubm ? (int64_t)w.ubm->cacheDeficit : 0, (int64_t)w.cache->data.m.size()});
        wr(g, keys);
        wr(g, weights);
        wr(g, lus);
    }
    fclose(g);
    return 0;
}
