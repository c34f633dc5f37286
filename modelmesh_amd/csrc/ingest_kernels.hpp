// ingest_kernels.hpp — the KV-store wire format straight into the packed tables.
//
// ModelMesh keeps InstanceRecord and ModelRecord values as Jackson JSON in etcd / ZooKeeper
// (MM.java:346 INST_REC_SERIALIZER, :628 registry view).  These kernels parse the raw values on the
// device: one lane per record walks its bytes once, matches the @JsonProperty names
// (InstanceRecord.java:37-69: lruTime,count,cap,used,lThreads,lInProg,rpm,shutdown,startTime,vers,
// loc,zone,labels; ModelRecord.java:61-114: type,encKey,mPath,instanceIds,failedIn,fails,refs,
// autoDel,lu,lul — `instanceIds` has no @JsonProperty and serialises under its bean name) by
// length + FNV-1a hash, and writes the packed row.  Fields Jackson omits because they hold the
// default value come out as 0 / false, exactly like the bean's defaults.  Instance ids inside a
// ModelRecord (the keys of instanceIds / failedIn) are resolved to pod indices through an
// open-addressing table of id hashes built when the ids are loaded.
//
// Pure byte / integer work: no MFMA, bound by the bytes of JSON read once.
#pragma once
#include "snapshot.hpp"

namespace mmp {

__host__ __device__ constexpr uint64_t fnv1a(const char *s, int n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < n; i++) h = (h ^ (uint64_t)(unsigned char)s[i]) * 0x100000001b3ull;
    return h;
}
#define MMP_KEY(lit) fnv1a(lit, (int)sizeof(lit) - 1)

struct JCur {
    const char *p, *e;
    bool bad;
};

__device__ __forceinline__ void j_ws(JCur &c)
{
    while (c.p < c.e && (*c.p == ' ' || *c.p == '\n' || *c.p == '\t' || *c.p == '\r')) c.p++;
}

__device__ __forceinline__ bool j_eat(JCur &c, char ch)
{
    j_ws(c);
    if (c.p < c.e && *c.p == ch) {
        c.p++;
        return true;
    }
    return false;
}

// at the opening quote: hash the raw bytes up to the closing quote (escapes are hashed verbatim)
__device__ __forceinline__ uint64_t j_string_hash(JCur &c)
{
    uint64_t h = 0xcbf29ce484222325ull;
    if (c.p >= c.e || *c.p != '"') {
        c.bad = true;
        return 0;
    }
    c.p++;
    while (c.p < c.e && *c.p != '"') {
        if (*c.p == '\\') {
            h = (h ^ (uint64_t)(unsigned char)*c.p) * 0x100000001b3ull;
            c.p++;
            if (c.p >= c.e) break;
        }
        h = (h ^ (uint64_t)(unsigned char)*c.p) * 0x100000001b3ull;
        c.p++;
    }
    if (c.p >= c.e) {
        c.bad = true;
        return 0;
    }
    c.p++;  // closing quote
    return h;
}

__device__ __forceinline__ void j_skip_value(JCur &c)
{
    j_ws(c);
    if (c.p >= c.e) {
        c.bad = true;
        return;
    }
    if (*c.p == '"') {
        (void)j_string_hash(c);
        return;
    }
    if (*c.p == '{' || *c.p == '[') {
        int depth = 0;
        while (c.p < c.e) {
            const char ch = *c.p;
            if (ch == '"') {
                (void)j_string_hash(c);
                if (c.bad) return;
                continue;
            }
            if (ch == '{' || ch == '[') depth++;
            if (ch == '}' || ch == ']') {
                depth--;
                if (depth == 0) {
                    c.p++;
                    return;
                }
            }
            c.p++;
        }
        c.bad = true;
        return;
    }
    // number / true / false / null
    while (c.p < c.e && *c.p != ',' && *c.p != '}' && *c.p != ']' && *c.p != ' ' && *c.p != '\n' && *c.p != '\t' &&
           *c.p != '\r')
        c.p++;
}

// a Java long / int written by Jackson: optional '-', digits (wraps like Long.parseLong would not — a
// value that does not fit is malformed for these beans)
__device__ __forceinline__ int64_t j_int(JCur &c)
{
    j_ws(c);
    bool neg = false;
    if (c.p < c.e && *c.p == '-') {
        neg = true;
        c.p++;
    }
    if (c.p >= c.e || *c.p < '0' || *c.p > '9') {
        c.bad = true;
        return 0;
    }
    uint64_t v = 0;
    while (c.p < c.e && *c.p >= '0' && *c.p <= '9') {
        v = v * 10u + (uint64_t)(*c.p - '0');
        c.p++;
    }
    if (c.p < c.e && (*c.p == '.' || *c.p == 'e' || *c.p == 'E')) c.bad = true;  // not an integer
    return neg ? (int64_t)(0 - v) : (int64_t)v;
}

__device__ __forceinline__ bool j_bool(JCur &c)
{
    j_ws(c);
    if (c.e - c.p >= 4 && c.p[0] == 't' && c.p[1] == 'r' && c.p[2] == 'u' && c.p[3] == 'e') {
        c.p += 4;
        return true;
    }
    if (c.e - c.p >= 5 && c.p[0] == 'f' && c.p[1] == 'a' && c.p[2] == 'l' && c.p[3] == 's' && c.p[4] == 'e') {
        c.p += 5;
        return false;
    }
    c.bad = true;
    return false;
}

// One InstanceRecord value per lane.  rows[] arrive with id_order / replica_set / flags(LIVE) set by
// the host; every numeric field is (re)written from the JSON.
__global__ void ingest_pods_kernel(const char *__restrict__ buf, const int64_t *__restrict__ off, int32_t n,
                                   mmp_pod_row *__restrict__ rows, int64_t *__restrict__ start_time,
                                   int32_t *__restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    JCur c{buf + off[i], buf + off[i + 1], false};
    mmp_pod_row r = rows[i];
    r.lru_time = r.capacity = r.used = r.version = 0;
    r.count = r.loading_threads = r.loading_in_progress = r.rpm = 0;
    r.flags &= ~MMP_POD_SHUTTING_DOWN;
    int64_t st = 0;
    if (!j_eat(c, '{')) c.bad = true;
    bool first = true;
    while (!c.bad) {
        j_ws(c);
        if (c.p < c.e && *c.p == '}') break;
        if (!first && !j_eat(c, ',')) {
            c.bad = true;
            break;
        }
        first = false;
        j_ws(c);
        const char *k0 = c.p;
        const uint64_t h = j_string_hash(c);
        const int klen = (int)(c.p - k0) - 2;
        if (c.bad || !j_eat(c, ':')) {
            c.bad = true;
            break;
        }
        if (h == MMP_KEY("lruTime") && klen == 7)
            r.lru_time = j_int(c);
        else if (h == MMP_KEY("count") && klen == 5)
            r.count = (int32_t)j_int(c);
        else if (h == MMP_KEY("cap") && klen == 3)
            r.capacity = j_int(c);
        else if (h == MMP_KEY("used") && klen == 4)
            r.used = j_int(c);
        else if (h == MMP_KEY("lThreads") && klen == 8)
            r.loading_threads = (int32_t)j_int(c);
        else if (h == MMP_KEY("lInProg") && klen == 7)
            r.loading_in_progress = (int32_t)j_int(c);
        else if (h == MMP_KEY("rpm") && klen == 3)
            r.rpm = (int32_t)j_int(c);
        else if (h == MMP_KEY("shutdown") && klen == 8) {
            if (j_bool(c)) r.flags |= MMP_POD_SHUTTING_DOWN;
        } else if (h == MMP_KEY("startTime") && klen == 9)
            st = j_int(c);
        else if (h == MMP_KEY("vers") && klen == 4)
            r.version = j_int(c);
        else
            j_skip_value(c);  // loc, zone, labels (interned on the host), anything newer
    }
    status[i] = c.bad ? 1 : 0;
    if (!c.bad) {
        rows[i] = r;
        start_time[i] = st;
    }
}

// open-addressing table of 64-bit string hashes -> small int (instance id -> pod, type name -> type)
struct HashTab {
    const uint64_t *hash;
    const int32_t *val;
    uint32_t mask;  // capacity - 1 (power of two); 0 with hash == nullptr means "empty table"
};

__device__ __forceinline__ int32_t tab_find(const HashTab &t, uint64_t h, int32_t missing)
{
    if (!t.hash) return missing;
    uint32_t s = (uint32_t)(h ^ (h >> 32)) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; probe++) {
        const int32_t v = t.val[s];
        if (v == INT32_MIN) return missing;  // empty slot
        if (t.hash[s] == h) return v;
        s = (s + 1) & t.mask;
    }
    return missing;
}

// at '{' of an id -> long map (instanceIds / failedIn): count the entries, and when out_pod != nullptr
// write them in document order (a TreeMap serialises in key order, which is what the paths expect)
__device__ __forceinline__ int32_t j_id_map(JCur &c, const HashTab &ids, int32_t *out_pod, int64_t *out_time)
{
    int32_t n = 0;
    j_ws(c);
    if (c.e - c.p >= 4 && c.p[0] == 'n' && c.p[1] == 'u' && c.p[2] == 'l' && c.p[3] == 'l') {
        c.p += 4;
        return 0;
    }
    if (!j_eat(c, '{')) {
        c.bad = true;
        return 0;
    }
    bool first = true;
    while (!c.bad) {
        j_ws(c);
        if (c.p < c.e && *c.p == '}') {
            c.p++;
            break;
        }
        if (!first && !j_eat(c, ',')) {
            c.bad = true;
            break;
        }
        first = false;
        j_ws(c);
        const uint64_t h = j_string_hash(c);
        if (c.bad || !j_eat(c, ':')) {
            c.bad = true;
            break;
        }
        const int64_t t = j_int(c);
        if (out_pod) {
            out_pod[n] = tab_find(ids, h, -1);
            out_time[n] = t;
        }
        n++;
    }
    return n;
}

struct IngestModelsArgs {
    const char *buf;
    const int64_t *off;
    int32_t n;
    HashTab ids, types;
    int32_t unknown_type;  // index for a type name that is not in the table
    int32_t default_type;  // index of ModelRecord.DEFAULT_TYPE ("NLCLASSIFIER", ModelRecord.java:133)
    mmp_model_row *rows;   // pass 0 writes type / n_loaded / n_failed / last_used; pass 1 reads ent_off
    int64_t *last_unload;
    int32_t *status;
    int32_t *ent_pod;      // pass 1
    int64_t *ent_time;
};

template <int PASS>
__global__ void ingest_models_kernel(IngestModelsArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    JCur c{A.buf + A.off[i], A.buf + A.off[i + 1], false};
    mmp_model_row r = A.rows[i];
    if (PASS == 0) {
        r.type = A.default_type;
        r.n_loaded = r.n_failed = 0;
        r.last_used = 0;
        r.ent_off = 0;
    } else if (A.status[i]) {
        return;  // malformed in pass 0: contributes no entries
    }
    int64_t lul = 0;
    if (!j_eat(c, '{')) c.bad = true;
    bool first = true;
    while (!c.bad) {
        j_ws(c);
        if (c.p < c.e && *c.p == '}') break;
        if (!first && !j_eat(c, ',')) {
            c.bad = true;
            break;
        }
        first = false;
        j_ws(c);
        const char *k0 = c.p;
        const uint64_t h = j_string_hash(c);
        const int klen = (int)(c.p - k0) - 2;
        if (c.bad || !j_eat(c, ':')) {
            c.bad = true;
            break;
        }
        if (h == MMP_KEY("instanceIds") && klen == 11) {
            const int32_t k = j_id_map(c, A.ids, PASS ? A.ent_pod + r.ent_off : nullptr, PASS ? A.ent_time + r.ent_off : nullptr);
            if (PASS == 0) r.n_loaded = k;
        } else if (h == MMP_KEY("failedIn") && klen == 8) {
            const int32_t k = j_id_map(c, A.ids, PASS ? A.ent_pod + r.ent_off + r.n_loaded : nullptr,
                                       PASS ? A.ent_time + r.ent_off + r.n_loaded : nullptr);
            if (PASS == 0) r.n_failed = k;
        } else if (PASS == 0 && h == MMP_KEY("type") && klen == 4) {
            j_ws(c);
            if (c.p < c.e && *c.p == '"')
                r.type = tab_find(A.types, j_string_hash(c), A.unknown_type);
            else
                j_skip_value(c);  // null -> DEFAULT_TYPE (ModelRecord.java:121)
        } else if (PASS == 0 && h == MMP_KEY("lu") && klen == 2)
            r.last_used = j_int(c);
        else if (PASS == 0 && h == MMP_KEY("lul") && klen == 3)
            lul = j_int(c);
        else
            j_skip_value(c);
    }
    if (PASS == 0) {
        A.status[i] = c.bad ? 1 : 0;
        if (c.bad) r.n_loaded = r.n_failed = 0;
        A.rows[i] = r;
        A.last_unload[i] = lul;
    }
}

// ent_off = exclusive scan of (n_loaded + n_failed): single-block scan, the registry is <= a few million rows
__global__ __launch_bounds__(1024) void model_offsets_kernel(mmp_model_row *rows, int32_t n, int32_t *total)
{
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int32_t s = 0;
    for (int i = lo; i < hi; i++) s += rows[i].n_loaded + rows[i].n_failed;
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int32_t v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = t ? part[t - 1] : 0;
    for (int i = lo; i < hi; i++) {
        rows[i].ent_off = run;
        run += rows[i].n_loaded + rows[i].n_failed;
    }
    if (t == 1023) *total = part[1023];
}

}  // namespace mmp
