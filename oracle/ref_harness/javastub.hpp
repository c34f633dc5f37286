// javastub.hpp — stand-ins for the Java / Guava / eclipse-collections / litelinks types that the reference's method
// bodies name, so that those bodies (extracted verbatim by extract.py into oracle/_ref/gen/*.inc) compile with g++.
//
// TEST INFRASTRUCTURE (the `oracle/_ref` harness): nothing here knows anything about placement.  Every class below is a
// container, a string, a comparison helper or a record with getters — the semantics are the JDK's / the libraries'
// documented ones (reference semantics for objects: every class is a HANDLE onto shared storage, `null` compares like
// Java's null).  All of ModelMesh's decision logic comes from the reference's own text.
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#define null nullptr
typedef bool boolean;
static_assert(sizeof(long) == 8 && sizeof(int) == 4, "Java long / int (build with -fwrapv: Java arithmetic wraps)");

// (int) of a double: JLS 5.1.3 (NaN -> 0, saturating); (int) of a long: the low 32 bits
static inline int J2I(double d) { return d != d ? 0 : d >= 2147483647.0 ? INT_MAX : d <= -2147483648.0 ? INT_MIN : (int)d; }
static inline int J2I(long v) { return (int)(uint32_t)(uint64_t)v; }
// `x >>> n` on a long (JLS 15.19)
static inline long JUSHR(long x, int n) { return (long)((uint64_t)x >> (n & 63)); }

// java.lang.String (nullable, immutable)
class String {
    std::shared_ptr<const std::string> p;

public:
    String() {}
    String(std::nullptr_t) {}
    String(const char *s) : p(std::make_shared<const std::string>(s)) {}
    String(const std::string &s) : p(std::make_shared<const std::string>(s)) {}
    const std::string &str() const { return *p; }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    boolean equals(const String &o) const { return o.p && *p == *o.p; }
    int compareTo(const String &o) const  // String.compareTo: first differing UTF-16 unit, else the length difference (ids are ASCII)
    {
        const std::string &a = *p, &b = *o.p;
        const size_t n = std::min(a.size(), b.size());
        for (size_t i = 0; i < n; i++)
            if (a[i] != b[i]) return (int)(unsigned char)a[i] - (int)(unsigned char)b[i];
        return (int)a.size() - (int)b.size();
    }
    int length() const { return (int)p->size(); }
    String substring(int a, int b) const { return String(p->substr(a, b - a)); }
    struct Hash { size_t operator()(const String &s) const { return std::hash<std::string>()(s.str()); } };
    struct Eq { bool operator()(const String &a, const String &b) const { return a.str() == b.str(); } };
    struct Less { bool operator()(const String &a, const String &b) const { return a.str() < b.str(); } };
};
// string concatenation builds log and exception messages only: no decision reads them
template <class X> static inline String operator+(const String &, const X &) { return String(""); }
// String[] (non-null here)
struct StringArray {
    std::shared_ptr<std::vector<String>> p = std::make_shared<std::vector<String>>();
    int length() const { return (int)p->size(); }
    const String &operator[](int i) const { return (*p)[i]; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
};

// java.lang.Long / Integer: boxed value as a type argument (Map<String, Long>), static members via `::` (extract.py)
struct Long {
    long v = 0;
    bool isnull = false;
    Long() {}
    Long(long x) : v(x) {}
    Long(std::nullptr_t) : isnull(true) {}  // a null Long (Map.get of an absent key)
    operator long() const { return v; }  // unboxing
    bool operator==(std::nullptr_t) const { return isnull; }
    bool operator!=(std::nullptr_t) const { return !isnull; }
    static constexpr long MAX_VALUE = LONG_MAX;
    static int compare(long a, long b) { return a < b ? -1 : a > b ? 1 : 0; }
};
struct Integer {
    static constexpr int MAX_VALUE = INT_MAX;
    static int parseInt(const String &s) { return (int)std::stol(s.str()); }
};
// java.util.concurrent.TimeUnit.MILLISECONDS.convert(n, unit)
enum TimeUnitT : long { MINUTES = 60000L, DAYS = 86400000L };
static const struct { long convert(long n, TimeUnitT u) const { return n * (long)u; } } MILLISECONDS;
static const struct {
    int max(int a, int b) const { return a > b ? a : b; }
    long max(long a, long b) const { return a > b ? a : b; }
    double max(double a, double b) const { return a > b ? a : b; }
    int min(int a, int b) const { return a < b ? a : b; }
    long min(long a, long b) const { return a < b ? a : b; }
    long min(long a, int b) const { return a < b ? a : (long)b; }  // Java widens the int
    int abs(int a) const { return a < 0 ? (int)(0u - (unsigned)a) : a; }            // Math.abs(MIN_VALUE) == MIN_VALUE
    long abs(long a) const { return a < 0 ? (long)(0ul - (unsigned long)a) : a; }
} Math;

// java.util.Map.Entry
template <class K, class V> class Entry {
    struct Rep { K k; V v; };
    std::shared_ptr<Rep> p;

public:
    Entry() {}
    Entry(std::nullptr_t) {}
    Entry(const K &k, const V &v) : p(std::make_shared<Rep>(Rep{k, v})) {}
    const K &getKey() const { return p->k; }
    const V &getValue() const { return p->v; }
    void setValue(const V &v) const { p->v = v; }  // writes through: the entries a map hands out share their storage with it
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
};

// java.util.Iterator (type-erased)
template <class T> class Iterator {
    struct Rep { std::function<bool()> has; std::function<T()> nxt; std::function<void()> rem; };
    std::shared_ptr<Rep> p;

public:
    Iterator() {}
    Iterator(std::function<bool()> h, std::function<T()> n, std::function<void()> r = nullptr)
        : p(std::make_shared<Rep>(Rep{std::move(h), std::move(n), std::move(r)})) {}
    boolean hasNext() const { return p->has(); }
    T next() const { return p->nxt(); }
    void remove() const { p->rem(); }  // removes the element next() returned last
};
template <class C, class T = typename C::value_type> static Iterator<T> iterate(std::shared_ptr<C> c)
{
    auto it = std::make_shared<typename C::const_iterator>(c->begin());
    return Iterator<T>([c, it] { return *it != c->end(); }, [it] { return *(*it)++; });
}
// com.google.common.collect.Iterators.filter: lazy, keeps the order
static const struct {
    template <class T, class P> Iterator<T> filter(Iterator<T> src, P pred) const
    {
        struct St { Iterator<T> src; P pred; bool have = false; T cur; };
        auto st = std::make_shared<St>(St{src, pred});
        auto advance = [st] {
            while (!st->have && st->src.hasNext()) {
                T t = st->src.next();
                if (st->pred(t)) { st->cur = t; st->have = true; }
            }
            return st->have;
        };
        return Iterator<T>(advance, [st, advance] { advance(); st->have = false; return st->cur; });
    }
} Iterators;

// java.util.List / ArrayList.  Every list remembers what was ever add()ed (the harness reads the shortlist from it:
// getNext nulls entries of `candidates` in place, :4974)
template <class T> class List {
public:
    struct Rep { std::vector<T> v, added; };
    std::shared_ptr<Rep> p;
    List() {}
    List(std::nullptr_t) {}
    explicit List(std::shared_ptr<Rep> r) : p(std::move(r)) {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    boolean add(const T &t) const { p->v.push_back(t); p->added.push_back(t); return true; }
    const T &get(int i) const { return p->v[i]; }
    void set(int i, const T &t) const { p->v[i] = t; }
    int size() const { return (int)p->v.size(); }
    boolean isEmpty() const { return p->v.empty(); }
    Iterator<T> iterator() const
    {
        auto r = p;
        auto i = std::make_shared<size_t>(0);
        return Iterator<T>([r, i] { return *i < r->v.size(); }, [r, i] { return r->v[(*i)++]; });
    }
};
extern std::vector<std::shared_ptr<void>> g_lists_created;  // in creation order, per call (harness.cc clears it)
struct ArrayList_new {  // `new ArrayList<>(n)` / `new ArrayList<>(collection)`: the element type comes from the declaration it initialises
    std::vector<String> init;
    ArrayList_new(int = 0) {}
    template <class C, class = decltype(std::declval<C>().begin())> ArrayList_new(const C &c)
    {
        for (auto &x : c) init.push_back(x);
    }
    template <class T> operator List<T>() const
    {
        auto r = std::make_shared<typename List<T>::Rep>();
        g_lists_created.push_back(r);
        List<T> l(r);
        if constexpr (std::is_same<T, String>::value)
            for (auto &x : init) l.add(x);
        return l;
    }
};
// org.eclipse.collections MutableIntList / IntArrayList
class MutableIntList {
    std::shared_ptr<std::vector<int>> p = std::make_shared<std::vector<int>>();

public:
    boolean add(int v) const { p->push_back(v); return true; }
    int get(int i) const { return (*p)[i]; }
    int min() const { return *std::min_element(p->begin(), p->end()); }
};
static inline MutableIntList IntArrayList_new(int = 0) { return MutableIntList(); }

// java.util.Set<String> / Collection<String> (nullable).  Ordered by the string (a TreeSet; where the Java has a HashSet
// nothing that decides anything depends on its iteration order), iterable with a range-for
template <class T> class Set {
public:
    typedef std::set<String, String::Less> Rep;
    std::shared_ptr<Rep> p;
    Set() {}
    Set(std::nullptr_t) {}
    static Set make() { Set s; s.p = std::make_shared<Rep>(); return s; }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    bool operator==(const Set &o) const { return p == o.p; }  // Java `==` on references: identity
    bool operator!=(const Set &o) const { return p != o.p; }
    boolean contains(const T &t) const { return p->count(t) != 0; }
    boolean isEmpty() const { return p->empty(); }
    boolean add(const T &t) const { return p->insert(t).second; }
    void clear() const { p->clear(); }
    int size() const { return (int)p->size(); }
    Rep::const_iterator begin() const { return p->begin(); }
    Rep::const_iterator end() const { return p->end(); }
};
static inline Set<String> TreeSet_new() { return Set<String>::make(); }
static inline Set<String> HashSet_new(int = 0) { return Set<String>::make(); }  // `new HashSet<>(n)` of strings (iteration order: by string)
// com.google.common.collect.Sets.union (a view in Guava; a copy here: nothing mutates the operands afterwards)
static const struct {
    Set<String> union_(const Set<String> &a, const Set<String> &b) const
    {
        Set<String> u = Set<String>::make();
        for (auto &x : a) u.add(x);
        for (auto &x : b) u.add(x);
        return u;
    }
} Sets;
template <class T> using Collection = Set<T>;
// java.util.Map<String, V>: key-ordered (the one map whose iteration order matters, ModelRecord.instanceIds, is a TreeMap)
template <class K, class V> class Map {
public:
    std::shared_ptr<std::map<String, V, String::Less>> p;
    Map() {}
    Map(std::nullptr_t) {}
    static Map make() { Map m; m.p = std::make_shared<std::map<String, V, String::Less>>(); return m; }
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    int size() const { return (int)p->size(); }
    boolean isEmpty() const { return p->empty(); }
    boolean containsKey(const K &k) const { return p->count(k) != 0; }
    V get(const K &k) const { auto it = p->find(k); return it == p->end() ? V(null) : it->second; }
    V put(const K &k, const V &v) const { auto it = p->find(k); V old = it == p->end() ? V(null) : it->second; (*p)[k] = v; return old; }
    std::vector<Entry<K, V>> entrySet() const
    {
        std::vector<Entry<K, V>> es;
        for (auto &kv : *p) es.emplace_back(kv.first, kv.second);
        return es;
    }
    Set<K> keySet() const
    {
        Set<K> ks = Set<K>::make();
        for (auto &kv : *p) ks.add(kv.first);
        return ks;
    }
    std::vector<V> values() const
    {
        std::vector<V> vs;
        for (auto &kv : *p) vs.push_back(kv.second);
        return vs;
    }
};
// java.util.Collections.min over boxed longs
struct Collections {
    static Set<String> emptySet() { return Set<String>::make(); }
    static Set<String> singleton(const String &x) { Set<String> s = Set<String>::make(); s.add(x); return s; }
    static List<String> singletonList(const String &x) { List<String> l = ArrayList_new(1); l.add(x); return l; }
    static long min(const std::vector<Long> &v)
    {
        long m = v.at(0);
        for (const Long &x : v) m = (long)x < m ? (long)x : m;
        return m;
    }
};
// org.eclipse.collections ObjectLongMap<String> / MutableObjectLongMap / ObjectLongHashMap (one nullable handle class: the
// immutable views are only ever read)
template <class K> class ObjectLongMap {
public:
    typedef std::map<std::string, long> Rep;
    std::shared_ptr<Rep> p = std::make_shared<Rep>();
    ObjectLongMap() {}
    ObjectLongMap(std::nullptr_t) : p() {}
    bool operator==(std::nullptr_t) const { return !p; }
    bool operator!=(std::nullptr_t) const { return (bool)p; }
    boolean isEmpty() const { return p->empty(); }
    boolean containsKey(const K &k) const { return p->count(k.str()) != 0; }
    void put(const K &k, long v) const { (*p)[k.str()] = v; }
    void remove(const K &k) const { p->erase(k.str()); }
    template <class P> boolean anySatisfy(P pred) const
    {
        for (auto &e : *p)
            if (pred(e.second)) return true;
        return false;
    }
    template <class P> ObjectLongMap reject(P pred) const  // a new map without the rejected entries
    {
        ObjectLongMap out;
        for (auto &e : *p)
            if (!pred(String(e.first), e.second)) (*out.p)[e.first] = e.second;
        return out;
    }
};
template <class K> using MutableObjectLongMap = ObjectLongMap<K>;
template <class K> static ObjectLongMap<K> ObjectLongHashMap_new(const ObjectLongMap<K> &src)  // `new ObjectLongHashMap<>(src)`: a copy
{
    ObjectLongMap<K> out;
    *out.p = *src.p;
    return out;
}
static const struct { struct { ObjectLongMap<String> empty() const { return ObjectLongMap<String>(); } } immutable; } ObjectLongMaps;

// com.google.common.collect.ComparisonChain + Ordering.natural().nullsLast()
struct NullsLast {};
static const NullsLast NULLS_LAST;
class ComparisonChainT {
    int r = 0;

public:
    ComparisonChainT start() const { return ComparisonChainT(); }
    ComparisonChainT compare(int a, int b) { if (!r) r = a < b ? -1 : a > b ? 1 : 0; return *this; }
    ComparisonChainT compare(long a, long b) { if (!r) r = a < b ? -1 : a > b ? 1 : 0; return *this; }
    ComparisonChainT compare(const String &a, const String &b) { if (!r) { int c = a.compareTo(b); r = c < 0 ? -1 : c > 0 ? 1 : 0; } return *this; }
    ComparisonChainT compare(const String &a, const String &b, const NullsLast &)
    {
        if (!r) {
            if (a == null || b == null) r = (a == null) ? ((b == null) ? 0 : 1) : -1;
            else { int c = a.compareTo(b); r = c < 0 ? -1 : c > 0 ? 1 : 0; }
        }
        return *this;
    }
    template <class C> ComparisonChainT compare(const StringArray &a, const StringArray &b, const C &cmp)
    {
        if (!r) { int c = cmp(a, b); r = c < 0 ? -1 : c > 0 ? 1 : 0; }
        return *this;
    }
    int result() const { return r; }
};
static const ComparisonChainT ComparisonChain;
