// ingest_kernels.hpp — the KV-store wire format straight into the packed tables.
//
// ModelMesh keeps InstanceRecord and ModelRecord values as Jackson JSON in etcd / ZooKeeper
// (MM.java:346 INST_REC_SERIALIZER, :628 registry view).  These kernels parse the raw values on the
// device: one WAVEFRONT per record (j_scan: the record is staged into LDS with coalesced dword loads and
// classified 64 bytes at a time with ballots — unescaped quotes, string interiors by prefix-xor, nesting
// depth by popcounts — after which every field / map entry of the record is parsed by its own lane);
// records longer than the LDS tile are walked by one lane (the serial parser below, which is also the
// readable statement of the grammar).  Either way the parser matches the @JsonProperty names
// (InstanceRecord.java:37-69: lruTime,count,cap,used,lThreads,lInProg,rpm,shutdown,startTime,vers,
// loc,zone,labels; ModelRecord.java:61-114: type,encKey,mPath,instanceIds,failedIn,fails,refs,
// autoDel,lu,lul — `instanceIds` has no @JsonProperty and serialises under its bean name) by
// length + FNV-1a hash, and writes the packed row.  Fields Jackson omits because they hold the
// default value come out as 0 / false, exactly like the bean's defaults.  Instance ids inside a
// ModelRecord (the keys of instanceIds / failedIn) are resolved to pod indices through an
// open-addressing table of id hashes built when the ids are loaded.
//
// Pure byte / integer work: no MFMA, bound by the bytes of JSON read once.
//
// Malformed values (status 1, row untouched): truncated or unbalanced nesting / strings, a value that
// does not start with '{', a known field or a map entry whose value has the wrong type, missing ',' or
// ':' separators.  The grammar INSIDE values that are skipped (unknown fields, loc / zone / labels,
// fails) is not validated beyond balanced nesting.
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
// A field name is recognised by hash + length and then CONFIRMED byte for byte (kp = its first byte): an unknown
// field whose name collides on both (FNV-1a is not collision resistant against a chosen name) is skipped like any
// other unknown field instead of being parsed as the known one.
template <class B>
__device__ __forceinline__ bool key_bytes_equal(const B *kp, const char *lit, int n)
{
    for (int i = 0; i < n; i++)
        if ((unsigned char)kp[i] != (unsigned char)lit[i]) return false;
    return true;
}
#define KEY_IS(h, klen, kp, lit) \
    ((h) == MMP_KEY(lit) && (klen) == (int)sizeof(lit) - 1 && key_bytes_equal((kp), lit, (int)sizeof(lit) - 1))

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

// One InstanceRecord value, walked by ONE lane (records longer than the LDS tile of the wave path).
// `r` arrives with id_order / replica_set / flags(LIVE) set by the host; every numeric field is
// (re)written from the JSON.  Returns true when the value is malformed.
__device__ __forceinline__ bool pod_record_serial(const char *b, const char *e, mmp_pod_row &r, int64_t &st)
{
    JCur c{b, e, false};
    r.lru_time = r.capacity = r.used = r.version = 0;
    r.count = r.loading_threads = r.loading_in_progress = r.rpm = 0;
    r.flags &= ~MMP_POD_SHUTTING_DOWN;
    st = 0;
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
        const char *kp = k0 + 1;
        if (c.bad || !j_eat(c, ':')) {
            c.bad = true;
            break;
        }
        if (KEY_IS(h, klen, kp, "lruTime"))
            r.lru_time = j_int(c);
        else if (KEY_IS(h, klen, kp, "count"))
            r.count = (int32_t)j_int(c);
        else if (KEY_IS(h, klen, kp, "cap"))
            r.capacity = j_int(c);
        else if (KEY_IS(h, klen, kp, "used"))
            r.used = j_int(c);
        else if (KEY_IS(h, klen, kp, "lThreads"))
            r.loading_threads = (int32_t)j_int(c);
        else if (KEY_IS(h, klen, kp, "lInProg"))
            r.loading_in_progress = (int32_t)j_int(c);
        else if (KEY_IS(h, klen, kp, "rpm"))
            r.rpm = (int32_t)j_int(c);
        else if (KEY_IS(h, klen, kp, "shutdown")) {
            if (j_bool(c)) r.flags |= MMP_POD_SHUTTING_DOWN;
        } else if (KEY_IS(h, klen, kp, "startTime"))
            st = j_int(c);
        else if (KEY_IS(h, klen, kp, "vers"))
            r.version = j_int(c);
        else
            j_skip_value(c);  // loc, zone, labels (interned on the host), anything newer
    }
    return c.bad;
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
    mmp_model_row *rows;   // type / n_loaded / n_failed / last_used (ent_off: compact_entries_kernel)
    int64_t *last_unload;
    int32_t *status;
    int32_t *cnt;          // n_loaded + n_failed per record (scanned into the CSR offsets)
    int32_t *ent_pod;      // entries of record i are parked at slot off[i] / 6 (an entry takes >= 6 bytes of JSON,
    int64_t *ent_time;     // so the slots of consecutive records never overlap) until the offsets are known
    int32_t grp;           // records per wavefront (1..kJGroup)
};

// One ModelRecord value walked by ONE lane.  PASS 0: type / counts / lu / lul; PASS 1: the entries.
template <int PASS>
__device__ __forceinline__ bool model_record_serial(const IngestModelsArgs &A, const char *b, const char *e,
                                                    mmp_model_row &r, int64_t &lul)
{
    JCur c{b, e, false};
    lul = 0;
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
        const char *kp = k0 + 1;
        if (c.bad || !j_eat(c, ':')) {
            c.bad = true;
            break;
        }
        if (KEY_IS(h, klen, kp, "instanceIds")) {
            const int32_t k = j_id_map(c, A.ids, PASS ? A.ent_pod + r.ent_off : nullptr, PASS ? A.ent_time + r.ent_off : nullptr);
            if (PASS == 0) r.n_loaded = k;
        } else if (KEY_IS(h, klen, kp, "failedIn")) {
            const int32_t k = j_id_map(c, A.ids, PASS ? A.ent_pod + r.ent_off + r.n_loaded : nullptr,
                                       PASS ? A.ent_time + r.ent_off + r.n_loaded : nullptr);
            if (PASS == 0) r.n_failed = k;
        } else if (PASS == 0 && KEY_IS(h, klen, kp, "type")) {
            j_ws(c);
            if (c.p < c.e && *c.p == '"')
                r.type = tab_find(A.types, j_string_hash(c), A.unknown_type);
            else
                j_skip_value(c);  // null -> DEFAULT_TYPE (ModelRecord.java:121)
        } else if (PASS == 0 && KEY_IS(h, klen, kp, "lu"))
            r.last_used = j_int(c);
        else if (PASS == 0 && KEY_IS(h, klen, kp, "lul"))
            lul = j_int(c);
        else
            j_skip_value(c);
    }
    return c.bad;
}

// ---- the wave path ----------------------------------------------------------------------------------------
//
// A wavefront takes kJGroup consecutive records.  Their bytes are contiguous in the value buffer, so the
// whole group is staged into an LDS tile with coalesced dword loads.  Then, per record, j_scan classifies
// the bytes 64 at a time, one byte per lane, with ballots (the masks are wave-uniform 64-bit scalars):
//   unescaped quotes   = '"' & ~escaped, `escaped` from the odd-length-backslash-run carry arithmetic
//   string interiors   = prefix-xor of the unescaped quotes (carried across chunks)
//   nesting depth      = running popcount(open) - popcount(close) over the structural characters
// and keeps, per chunk, the ':' and ',' masks of the two levels the beans use (fields of the record;
// entries of the id -> time maps) plus the closers of level-2 containers.  After that the FIELDS of all
// records of the group are spread over the lanes — the k-th set bit of a record's ':' mask is its k-th
// field, the key is the string that ends before it, the value starts after it — and, for ModelRecords,
// the map ENTRIES of all records are spread over the lanes the same way.  (One lane per field of ONE record
// was measured first: 0.29 ms per pass over 100k ModelRecords, VALU-issue bound with ~6 of 64 lanes busy.)
constexpr int kJWaves = 4;           // wavefronts per workgroup
constexpr int kJGroup = 8;           // records per wavefront
constexpr int kJTileBytes = 2048;    // LDS tile of one wavefront; a record longer than this: serial path
constexpr int kJTileChunks = kJTileBytes / 64 + kJGroup;
constexpr int kJBlock = kJWaves * 64;
constexpr int kJSlots = 10;          // known fields per record (InstanceRecord has 10, ModelRecord 5 + 6 map words)

struct JRecInfo {  // one record of the tile
    int32_t base;  // first byte inside the tile
    int32_t L, nch;
    int32_t mb;    // first mask word
    int32_t f, g;  // first / last non-whitespace byte: the record's '{' and '}'
    int32_t n1;    // level-1 ':' = fields
    int32_t bad;
};

struct __attribute__((aligned(16))) JWaveLds {
    uint32_t dw[kJTileBytes / 4 + 4];  // the group's bytes, dword-staged from the aligned address below them
    uint64_t rq[kJTileChunks];         // unescaped '"'
    uint64_t c1[kJTileChunks];         // ':' outside strings, directly inside the record object
    uint64_t c2[kJTileChunks];         // ':' one level down (entries of instanceIds / failedIn / fails)
    uint64_t m1[kJTileChunks];         // ',' at those two levels
    uint64_t m2[kJTileChunks];
    uint64_t e2[kJTileChunks];         // closers of containers opened directly inside the record object
    JRecInfo rec[kJGroup];
    int32_t pf[kJGroup + 1];           // prefix sums of the per-record item counts (fields, then entries)
    int64_t val[kJGroup][kJSlots];     // value of each known field ...
    int32_t win[kJGroup][kJSlots];     // ... and the index of the field it came from (a later duplicate wins)
    int64_t off[kJGroup + 1];          // byte offsets of the wavefront's records
};

struct JView {  // what a lane needs to work on one record
    const uint8_t *by;
    const uint64_t *rq, *c1, *c2, *m1, *m2, *e2;
    int L, nch, f, g, n1;
};

__device__ __forceinline__ JView j_view(const JWaveLds &S, int r)
{
    const JRecInfo I = S.rec[r];
    JView V;
    V.by = reinterpret_cast<const uint8_t *>(S.dw) + I.base;
    V.rq = S.rq + I.mb;
    V.c1 = S.c1 + I.mb;
    V.c2 = S.c2 + I.mb;
    V.m1 = S.m1 + I.mb;
    V.m2 = S.m2 + I.mb;
    V.e2 = S.e2 + I.mb;
    V.L = I.L;
    V.nch = I.nch;
    V.f = I.f;
    V.g = I.g;
    V.n1 = I.n1;
    return V;
}

__device__ __forceinline__ bool j_is_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

__device__ __forceinline__ uint64_t j_prefix_xor(uint64_t x)
{
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

// Characters escaped by a backslash.  Runs of backslashes: a run that starts on an odd bit is told apart
// from one that starts on an even bit by letting an addition carry through the run (the usual
// bit-parallel formulation); `carry` = the first byte of the next chunk is escaped.
__device__ __forceinline__ uint64_t j_escaped(uint64_t backslash, uint64_t &carry)
{
    backslash &= ~carry;
    const uint64_t follows = (backslash << 1) | carry;
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t odd_starts = backslash & ~even & ~follows;
    const uint64_t sum = odd_starts + backslash;
    carry = sum < odd_starts ? 1ull : 0ull;
    return (even ^ (sum << 1)) & follows;
}

// Classify the record at tile bytes [base, base + L): whole wavefront, one byte per lane per step.
__device__ __forceinline__ JRecInfo j_scan(JWaveLds &S, int base, int L, int mb)
{
    const int lane = lane_id();
    const uint8_t *by = reinterpret_cast<const uint8_t *>(S.dw) + base;
    JRecInfo R;
    R.base = base;
    R.L = L;
    R.nch = (L + 63) >> 6;
    R.mb = mb;
    uint64_t esc_carry = 0, in_str = 0;
    int depth = 0, zero_pos = kNoPos, nm1 = 0, n_nonws = 0;
    R.f = kNoPos;
    R.g = -1;
    R.n1 = 0;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int c = 0; c < R.nch; c++) {
        const int idx = c * 64 + lane;
        const uint32_t B = idx < L ? by[idx] : (uint32_t)' ';
        const uint64_t bs = __ballot(B == '\\');
        const uint64_t esc = (bs | esc_carry) ? j_escaped(bs, esc_carry) : 0ull;
        const uint64_t rq = __ballot(B == '"') & ~esc;
        const uint64_t ins = j_prefix_xor(rq) ^ in_str;  // opening quote .. byte before the closing quote
        in_str = (uint64_t)((int64_t)ins >> 63);
        const uint64_t op = __ballot(B == '{' || B == '[') & ~ins;
        const uint64_t cl = __ballot(B == '}' || B == ']') & ~ins;
        const uint64_t co = __ballot(B == ':') & ~ins;
        const uint64_t cm = __ballot(B == ',') & ~ins;
        const uint64_t nonws = ~__ballot(j_is_ws(B));
        const int d_before = depth + __popcll((unsigned long long)(op & lt)) - __popcll((unsigned long long)(cl & lt));
        const uint64_t at1 = __ballot(d_before == 1), at2 = __ballot(d_before == 2);
        const uint64_t z = cl & __ballot(d_before <= 1);  // closers that bring the depth to <= 0
        if (z && zero_pos == kNoPos) zero_pos = c * 64 + (__ffsll((unsigned long long)z) - 1);
        if (nonws) {
            if (R.f == kNoPos) R.f = c * 64 + (__ffsll((unsigned long long)nonws) - 1);
            R.g = c * 64 + 63 - __clzll((unsigned long long)nonws);
            n_nonws += __popcll((unsigned long long)nonws);
        }
        if (lane == 0) {
            S.rq[mb + c] = rq;
            S.c1[mb + c] = co & at1;
            S.c2[mb + c] = co & at2;
            S.m1[mb + c] = cm & at1;
            S.m2[mb + c] = cm & at2;
            S.e2[mb + c] = cl & at2;
        }
        R.n1 += __popcll((unsigned long long)(co & at1));
        nm1 += __popcll((unsigned long long)(cm & at1));
        depth += __popcll((unsigned long long)op) - __popcll((unsigned long long)cl);
    }
    // one object, closed exactly by the last non-blank byte, nothing open at the end, fields separated by
    // exactly one ',' each, and "{}" holds nothing but blanks
    const bool bad = R.f == kNoPos || by[R.f] != '{' || zero_pos != R.g || by[R.g] != '}' || depth != 0 || in_str != 0 ||
                     nm1 != (R.n1 > 0 ? R.n1 - 1 : 0) || (R.n1 == 0 && n_nonws != 2);
    R.bad = bad ? 1 : 0;
    return R;
}

__device__ __forceinline__ bool j_bit(const uint64_t *m, int pos) { return (m[pos >> 6] >> (pos & 63)) & 1ull; }

// position of the k-th (0-based) set bit at a position > after; -1 if there is none
__device__ __forceinline__ int j_nth_after(const uint64_t *m, int nch, int after, int k)
{
    int w = (after + 1) >> 6;
    if (w >= nch) return -1;
    uint64_t v = m[w] & (~0ull << ((after + 1) & 63));
    for (;;) {
        const int c = __popcll((unsigned long long)v);
        if (k < c) return w * 64 + select_kth_bit(v, k);
        k -= c;
        if (++w >= nch) return -1;
        v = m[w];
    }
}

// highest set bit at a position < before; -1 if there is none
__device__ __forceinline__ int j_prev(const uint64_t *m, int before)
{
    if (before <= 0) return -1;
    int w = (before - 1) >> 6;
    const int hb = (before - 1) & 63;
    uint64_t v = m[w];
    if (hb != 63) v &= (1ull << (hb + 1)) - 1ull;
    for (;;) {
        if (v) return w * 64 + 63 - __clzll((unsigned long long)v);
        if (--w < 0) return -1;
        v = m[w];
    }
}

// set bits at positions in (lo, hi)
__device__ __forceinline__ int j_count(const uint64_t *m, int lo, int hi)
{
    if (hi - lo < 2) return 0;
    const int a = lo + 1, b = hi - 1, wa = a >> 6, wb = b >> 6;
    int n = 0;
    for (int w = wa; w <= wb; w++) {
        uint64_t v = m[w];
        if (w == wa) v &= ~0ull << (a & 63);
        if (w == wb && (b & 63) != 63) v &= (1ull << ((b & 63) + 1)) - 1ull;
        n += __popcll((unsigned long long)v);
    }
    return n;
}

// The key of the ':' at p — FNV-1a of the raw bytes between its quotes — and the separator before it:
// the container's opener for the first member, a ',' of this level (`commas`) otherwise.
__device__ __forceinline__ bool j_key(const JView &R, int p, const uint64_t *commas, int open_pos, bool first,
                                      uint64_t &h, int &klen, const uint8_t *&kp)
{
    int q = p - 1;
    while (q >= 0 && j_is_ws(R.by[q])) q--;
    if (q < 0 || !j_bit(R.rq, q)) return false;  // not a string
    const int ks = j_prev(R.rq, q);
    if (ks < 0) return false;
    int sp = ks - 1;
    while (sp >= 0 && j_is_ws(R.by[sp])) sp--;
    if (sp < 0 || (first ? sp != open_pos : !j_bit(commas, sp))) return false;
    klen = q - ks - 1;
    kp = R.by + ks + 1;
    h = 0xcbf29ce484222325ull;
    for (int i = ks + 1; i < q; i++) h = (h ^ (uint64_t)R.by[i]) * 0x100000001b3ull;
    return true;
}

// after a value that ended before byte p: a ',' of this level, or the container's closer after the last member
__device__ __forceinline__ bool j_term(const JView &R, int p, const uint64_t *commas, int close_pos, bool last)
{
    while (p < R.L && j_is_ws(R.by[p])) p++;
    if (p >= R.L) return false;
    return last ? p == close_pos : j_bit(commas, p);
}

__device__ __forceinline__ bool j_int_at(const JView &R, int &p, int64_t &out)
{
    bool neg = false;
    if (p < R.L && R.by[p] == '-') {
        neg = true;
        p++;
    }
    if (p >= R.L || R.by[p] < '0' || R.by[p] > '9') return false;
    uint64_t v = 0;
    while (p < R.L && R.by[p] >= '0' && R.by[p] <= '9') {
        v = v * 10u + (uint64_t)(R.by[p] - '0');
        p++;
    }
    out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;  // j_term rejects a fraction / exponent / junk behind the digits
}

__device__ __forceinline__ bool j_lit_at(const JView &R, int p, const char *lit, int n)
{
    if (p + n > R.L) return false;
    for (int i = 0; i < n; i++)
        if (R.by[p + i] != (uint8_t)lit[i]) return false;
    return true;
}

// The byte offsets of the wavefront's records, fetched with one load (S.off[k] = off[i0 + k]).
__device__ __forceinline__ void j_load_offsets(const int64_t *off, int i0, int i1, JWaveLds &S)
{
    const int lane = lane_id();
    if (lane <= i1 - i0) S.off[lane] = off[i0 + lane];
    wave_sync();
}

// How many of the records [k, k1) of the wavefront fit one tile together (0: the first alone is too long).
__device__ __forceinline__ int j_group_len(const JWaveLds &S, int k, int k1)
{
    const int64_t b0 = S.off[k];
    int c = 0;
    while (k + c < k1 && S.off[k + c + 1] - b0 <= kJTileBytes) c++;
    return c;
}

// Stage records [k, k+cnt) of the wavefront into the tile and scan them; fills S.rec[0..cnt).
__device__ __forceinline__ void j_stage_and_scan(const char *buf, int k, int cnt, JWaveLds &S)
{
    const int lane = lane_id();
    const int64_t *off = S.off + k;
    const int64_t b0 = off[0], a0 = b0 & ~3ll;  // the buffer is device-allocated (256-byte aligned)
    const int shift = (int)(b0 - a0);
    const int ndw = (shift + (int)(off[cnt] - b0) + 3) >> 2;
    const uint32_t *g32 = reinterpret_cast<const uint32_t *>(buf + a0);
    for (int q = lane; q < ndw; q += 64) S.dw[q] = g32[q];
    wave_sync();
    int mb = 0;
    for (int r = 0; r < cnt; r++) {
        const JRecInfo R = j_scan(S, shift + (int)(off[r] - b0), (int)(off[r + 1] - off[r]), mb);
        if (lane == 0) S.rec[r] = R;
        mb += R.nch;
    }
    for (int q = lane; q < cnt * kJSlots; q += 64) {
        (&S.val[0][0])[q] = 0;
        (&S.win[0][0])[q] = -1;
    }
    wave_sync();
}

// pf[0..cnt] = exclusive prefix sums of the per-record item counts held by lanes 0..cnt-1; returns the total
__device__ __forceinline__ int j_prefix_items(JWaveLds &S, int cnt, int mine)
{
    const int lane = lane_id();
    const int incl = wave_incl_scan_i32(lane < cnt ? mine : 0);
    if (lane < cnt) S.pf[lane + 1] = incl;
    if (lane == 0) S.pf[0] = 0;
    wave_sync();
    return S.pf[cnt];
}

// item t -> (record, index inside the record)
__device__ __forceinline__ int j_item_record(const JWaveLds &S, int cnt, int t, int &k)
{
    int r = 0;
#pragma unroll
    for (int q = 1; q < kJGroup; q++)
        if (q < cnt && t >= S.pf[q]) r = q;
    k = t - S.pf[r];
    return r;
}

// a known field's value: the field with the highest index wins (Jackson keeps the last duplicate)
__device__ __forceinline__ void j_claim(JWaveLds &S, int r, int fid, int j) { atomicMax(&S.win[r][fid], j); }

// One InstanceRecord value per lane-group, kJGroup records per WAVEFRONT.  rows[] arrive with id_order /
// replica_set / flags(LIVE) set by the host; every numeric field is (re)written from the JSON.
__global__ __launch_bounds__(kJBlock) void ingest_pods_kernel(const char *__restrict__ buf, const int64_t *__restrict__ off,
                                                              int32_t n, int32_t grp, mmp_pod_row *__restrict__ rows,
                                                              int64_t *__restrict__ start_time, int32_t *__restrict__ status)
{
    __shared__ JWaveLds lds[kJWaves];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lane_id();
    JWaveLds &S = lds[wave];
    const int i0 = (blockIdx.x * kJWaves + wave) * grp;
    if (i0 >= n) return;
    const int i1 = i0 + grp < n ? i0 + grp : n;
    j_load_offsets(off, i0, i1, S);
    int i = i0;
    while (i < i1) {
        const int cnt = j_group_len(S, i - i0, i1 - i0);
        if (cnt == 0) {  // does not fit the LDS tile: one lane walks it
            if (lane == 0) {
                mmp_pod_row r = rows[i];
                int64_t st;
                const bool bad = pod_record_serial(buf + off[i], buf + off[i + 1], r, st);
                status[i] = bad ? 1 : 0;
                if (!bad) {
                    rows[i] = r;
                    start_time[i] = st;
                }
            }
            i++;
            continue;
        }
        j_stage_and_scan(buf, i - i0, cnt, S);
        const int total = j_prefix_items(S, cnt, lane < cnt ? (S.rec[lane].bad ? 0 : S.rec[lane].n1) : 0);
        // every field of every record of the group on its own lane
        for (int base = 0; base < total; base += 64) {
            const int t = base + lane;
            int fid = -1, r = 0, j = 0;
            int64_t val = 0;
            bool lbad = false;
            if (t < total) {
                r = j_item_record(S, cnt, t, j);
                const JView R = j_view(S, r);
                const int p = j_nth_after(R.c1, R.nch, -1, j);
                uint64_t h;
                int klen;
                const uint8_t *kp = nullptr;
                if (!j_key(R, p, R.m1, R.f, j == 0, h, klen, kp))
                    lbad = true;
                else {
                    if (KEY_IS(h, klen, kp, "lruTime")) fid = 0;
                    else if (KEY_IS(h, klen, kp, "count")) fid = 1;
                    else if (KEY_IS(h, klen, kp, "cap")) fid = 2;
                    else if (KEY_IS(h, klen, kp, "used")) fid = 3;
                    else if (KEY_IS(h, klen, kp, "lThreads")) fid = 4;
                    else if (KEY_IS(h, klen, kp, "lInProg")) fid = 5;
                    else if (KEY_IS(h, klen, kp, "rpm")) fid = 6;
                    else if (KEY_IS(h, klen, kp, "shutdown")) fid = 7;
                    else if (KEY_IS(h, klen, kp, "startTime")) fid = 8;
                    else if (KEY_IS(h, klen, kp, "vers")) fid = 9;
                    if (fid >= 0) {
                        int v = p + 1;
                        while (v < R.L && j_is_ws(R.by[v])) v++;
                        if (fid == 7) {
                            if (j_lit_at(R, v, "true", 4)) {
                                val = 1;
                                v += 4;
                            } else if (j_lit_at(R, v, "false", 5))
                                v += 5;
                            else
                                lbad = true;
                        } else if (!j_int_at(R, v, val))
                            lbad = true;
                        if (!lbad && !j_term(R, v, R.m1, R.g, j == R.n1 - 1)) lbad = true;
                    }
                }
                if (lbad) S.rec[r].bad = 1;
                if (fid >= 0 && !lbad) j_claim(S, r, fid, j);
            }
            wave_sync();
            if (fid >= 0 && !lbad && S.win[r][fid] == j) S.val[r][fid] = val;
            wave_sync();
        }
        if (lane < cnt) {
            const bool bad = S.rec[lane].bad != 0;
            status[i + lane] = bad ? 1 : 0;
            if (!bad) {
                const int64_t *fv = S.val[lane];
                mmp_pod_row r = rows[i + lane];
                r.lru_time = fv[0];
                r.count = (int32_t)fv[1];
                r.capacity = fv[2];
                r.used = fv[3];
                r.loading_threads = (int32_t)fv[4];
                r.loading_in_progress = (int32_t)fv[5];
                r.rpm = (int32_t)fv[6];
                r.flags = fv[7] ? (r.flags | MMP_POD_SHUTTING_DOWN) : (r.flags & ~MMP_POD_SHUTTING_DOWN);
                r.version = fv[9];
                rows[i + lane] = r;
                start_time[i + lane] = fv[8];
            }
        }
        wave_sync();
        i += cnt;
    }
}

// ModelRecord values, A.grp per WAVEFRONT, one pass: type / n_loaded / n_failed / last_used, the status
// (the whole value is validated, entries included) and the entries themselves, parked at slot
// off[i] / 6 + e of ent_pod / ent_time until compact_entries_kernel moves them to their CSR position.
// Slots: 0 type, 1 lu, 2 lul, 3 instanceIds, 4 failedIn (value = entry count), 5..8 = the two maps'
// opener / closer positions.
__global__ __launch_bounds__(kJBlock) void ingest_models_kernel(IngestModelsArgs A)
{
    __shared__ JWaveLds lds[kJWaves];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lane_id();
    JWaveLds &S = lds[wave];
    const int i0 = (blockIdx.x * kJWaves + wave) * A.grp;
    if (i0 >= A.n) return;
    const int i1 = i0 + A.grp < A.n ? i0 + A.grp : A.n;
    j_load_offsets(A.off, i0, i1, S);
    int i = i0;
    while (i < i1) {
        const int cnt = j_group_len(S, i - i0, i1 - i0);
        if (cnt == 0) {  // longer than the tile: one lane walks it, a second time to write the entries
            if (lane == 0) {
                mmp_model_row r;
                r.type = A.default_type;
                r.n_loaded = r.n_failed = 0;
                r.last_used = 0;
                r.ent_off = 0;
                int64_t lul, lul2;
                const char *b = A.buf + S.off[i - i0], *e = A.buf + S.off[i - i0 + 1];
                bool bad = model_record_serial<0>(A, b, e, r, lul);
                if (!bad) {
                    mmp_model_row w = r;
                    w.ent_off = (int32_t)(S.off[i - i0] / 6);
                    (void)model_record_serial<1>(A, b, e, w, lul2);
                } else
                    r.n_loaded = r.n_failed = 0;
                A.status[i] = bad ? 1 : 0;
                A.rows[i] = r;
                A.cnt[i] = r.n_loaded + r.n_failed;
                A.last_unload[i] = bad ? 0 : lul;
            }
            i++;
            continue;
        }
        j_stage_and_scan(A.buf, i - i0, cnt, S);
        if (lane < cnt) S.val[lane][0] = A.default_type;
        wave_sync();
        int total = j_prefix_items(S, cnt, lane < cnt ? (S.rec[lane].bad ? 0 : S.rec[lane].n1) : 0);
        for (int base = 0; base < total; base += 64) {
            const int t = base + lane;
            int fid = -1, r = 0, j = 0;
            int64_t val = 0;
            int vopen = -1, vclose = -1;
            bool lbad = false;
            if (t < total) {
                r = j_item_record(S, cnt, t, j);
                const JView R = j_view(S, r);
                const int p = j_nth_after(R.c1, R.nch, -1, j);
                uint64_t h;
                int klen;
                const uint8_t *kp = nullptr;
                if (!j_key(R, p, R.m1, R.f, j == 0, h, klen, kp))
                    lbad = true;
                else {
                    if (KEY_IS(h, klen, kp, "type")) fid = 0;
                    else if (KEY_IS(h, klen, kp, "lu")) fid = 1;
                    else if (KEY_IS(h, klen, kp, "lul")) fid = 2;
                    else if (KEY_IS(h, klen, kp, "instanceIds")) fid = 3;
                    else if (KEY_IS(h, klen, kp, "failedIn")) fid = 4;
                    int v = p + 1;
                    while (v < R.L && j_is_ws(R.by[v])) v++;
                    const bool last = j == R.n1 - 1;
                    if (fid == 0) {
                        if (v < R.L && j_bit(R.rq, v)) {
                            const int ve = j_nth_after(R.rq, R.nch, v, 0);  // closing quote (strings are balanced)
                            uint64_t th = 0xcbf29ce484222325ull;
                            for (int k = v + 1; k < ve; k++) th = (th ^ (uint64_t)R.by[k]) * 0x100000001b3ull;
                            val = tab_find(A.types, th, A.unknown_type);
                            if (!j_term(R, ve + 1, R.m1, R.g, last)) lbad = true;
                        } else {
                            fid = -1;  // null (or a non-string) -> DEFAULT_TYPE (ModelRecord.java:121)
                        }
                    } else if (fid == 1 || fid == 2) {
                        if (!j_int_at(R, v, val) || !j_term(R, v, R.m1, R.g, last)) lbad = true;
                    } else if (fid >= 3) {
                        if (j_lit_at(R, v, "null", 4)) {
                            if (!j_term(R, v + 4, R.m1, R.g, last)) lbad = true;
                        } else if (v < R.L && R.by[v] == '{') {
                            const int ce = j_nth_after(R.e2, R.nch, v, 0);
                            if (ce < 0 || R.by[ce] != '}')
                                lbad = true;
                            else {
                                vopen = v;
                                vclose = ce;
                                val = j_count(R.c2, v, ce);
                                if (j_count(R.m2, v, ce) != (val > 0 ? val - 1 : 0)) lbad = true;
                                if (val == 0) {
                                    int q = v + 1;
                                    while (q < ce && j_is_ws(R.by[q])) q++;
                                    if (q != ce) lbad = true;
                                }
                                if (!j_term(R, ce + 1, R.m1, R.g, last)) lbad = true;
                            }
                        } else
                            lbad = true;
                    }
                }
                if (lbad) S.rec[r].bad = 1;
                if (fid >= 0 && !lbad) j_claim(S, r, fid, j);
            }
            wave_sync();
            if (fid >= 0 && !lbad && S.win[r][fid] == j) {
                S.val[r][fid] = val;
                if (fid >= 3) {
                    S.val[r][5 + 2 * (fid - 3)] = vopen;
                    S.val[r][6 + 2 * (fid - 3)] = vclose;
                }
            }
            wave_sync();
        }
        // the entries of both maps of every record, one lane each: instanceIds first, then failedIn (CSR layout)
        total = j_prefix_items(S, cnt, lane < cnt && !S.rec[lane].bad ? (int)(S.val[lane][3] + S.val[lane][4]) : 0);
        for (int base = 0; base < total; base += 64) {
            const int t = base + lane;
            if (t < total) {
                int e;
                const int r = j_item_record(S, cnt, t, e);
                const JView R = j_view(S, r);
                const int nl = (int)S.val[r][3], nfl = (int)S.val[r][4];
                const int which = e < nl ? 0 : 1;
                const int k = which ? e - nl : e, kcnt = which ? nfl : nl;
                const int open = (int)S.val[r][5 + 2 * which], close = (int)S.val[r][6 + 2 * which];
                const int p = j_nth_after(R.c2, R.nch, open, k);
                uint64_t h = 0;
                int klen;
                int64_t tm = 0;
                bool lbad = false;
                const uint8_t *kp = nullptr;
                if (p < 0 || p > close || !j_key(R, p, R.m2, open, k == 0, h, klen, kp))
                    lbad = true;
                else {
                    int v = p + 1;
                    while (v < R.L && j_is_ws(R.by[v])) v++;
                    if (!j_int_at(R, v, tm) || !j_term(R, v, R.m2, close, k == kcnt - 1)) lbad = true;
                }
                if (lbad)
                    S.rec[r].bad = 1;
                else {
                    const int64_t slot = S.off[i - i0 + r] / 6 + e;
                    A.ent_pod[slot] = tab_find(A.ids, h, -1);
                    A.ent_time[slot] = tm;
                }
            }
        }
        wave_sync();
        if (lane < cnt) {
            const bool bad = S.rec[lane].bad != 0;
            mmp_model_row r;
            r.type = bad ? A.default_type : (int32_t)S.val[lane][0];
            r.ent_off = 0;
            r.n_loaded = bad ? 0 : (int32_t)S.val[lane][3];
            r.n_failed = bad ? 0 : (int32_t)S.val[lane][4];
            r.last_used = bad ? 0 : S.val[lane][1];
            A.status[i + lane] = bad ? 1 : 0;
            A.rows[i + lane] = r;
            A.cnt[i + lane] = r.n_loaded + r.n_failed;
            A.last_unload[i + lane] = bad ? 0 : S.val[lane][2];
        }
        wave_sync();
        i += cnt;
    }
}

// offs = exclusive scan of cnt (rocPRIM): move every record's entries from their parking slots to
// [offs[i], offs[i] + cnt[i]) of the registry's entry arrays and publish ent_off.  One lane per record
// (a model has 1-3 copies).
__global__ void compact_entries_kernel(const int64_t *__restrict__ off, int32_t n, const int32_t *__restrict__ cnt,
                                       const int32_t *__restrict__ offs, const int32_t *__restrict__ tmp_pod,
                                       const int64_t *__restrict__ tmp_time, mmp_model_row *__restrict__ rows,
                                       int32_t *__restrict__ ent_pod, int64_t *__restrict__ ent_time)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t k = cnt[i], o = offs[i];
    const int64_t slot = off[i] / 6;
    rows[i].ent_off = o;
    for (int32_t e = 0; e < k; e++) {
        ent_pod[o + e] = tmp_pod[slot + e];
        ent_time[o + e] = tmp_time[slot + e];
    }
}

}  // namespace mmp
