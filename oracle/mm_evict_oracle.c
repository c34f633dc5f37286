/*
 * mm_evict_oracle.c — CPU restatement of the timestamp-ordered weighted LRU
 * (clhm/ConcurrentLinkedHashMap.java, clhm/LinkedDeque.java) and of the unload
 * buffer accounting (ModelCacheUnloadBufManager.java).
 *
 * TEST INFRASTRUCTURE ONLY (parity checker).  Models the drained ("strict",
 * single-threaded) order — SURVEY.md Appendix B#11.
 *
 * Pinned by the reference's own TEXT: the clhm / LinkedDeque / ModelCacheUnloadBufManager method bodies compiled as they
 * stand (oracle/ref_harness/clhm_harness.cc) and run over random operation streams -> tests/golden/ref_clhm.npz, which this
 * file reproduces operation by operation (tests/test_ref_clhm.py); and by the reference's own known-answer tests:
 * EvictionsModelMeshTest.java:36-125 and ModelMeshEvictionsTest.java:156-280 (tests/test_oracle_kat.py).
 *
 * Domain (what ModelMesh itself guarantees, MM.java:1766 "maybe ensure != 0"): entry weights stay >= 1, the unload reserve
 * is > 0 and the pinned unload-buffer entry is never evicted (capacity > reserve).
 */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

void orc_cache_init(orc_cache *c, int64_t capacity)
{
    memset(c, 0, sizeof *c);
    c->capacity = capacity;
    c->oldest_time = -1; /* EMPTY_OLDEST_TIME, clhm :1117 */
}

void orc_cache_free(orc_cache *c)
{
    free(c->nodes);
    memset(c, 0, sizeof *c);
}

static void reserve(orc_cache *c, int32_t n)
{
    if (n <= c->cap_nodes) return;
    int32_t nc = c->cap_nodes ? c->cap_nodes * 2 : 16;
    while (nc < n) nc *= 2;
    c->nodes = (orc_node *)realloc(c->nodes, (size_t)nc * sizeof(orc_node));
    c->cap_nodes = nc;
}

int32_t orc_cache_find(const orc_cache *c, int32_t key)
{
    for (int32_t i = 0; i < c->n; i++)
        if (c->nodes[i].key == key) return i;
    return -1;
}

/* updateOldestTime, clhm :1129-1133: runs at the end of afterWrite (:444) and of the read drain (:463) */
static void refresh_oldest(orc_cache *c) { c->oldest_time = c->n ? c->nodes[0].last_used : -1; }

/* Node.touch, clhm :1357-1360 */
static void touch(orc_node *nd, int64_t time, int64_t now)
{
    nd->last_used = time == 0 ? now : (nd->last_used > time ? nd->last_used : time);
}

/* LinkedDeque.insert, LinkedDeque.java:259-288: walk from the tail to the
 * first node whose lastUsed <= ts and link the new node after it. */
static int32_t deque_insert(orc_cache *c, orc_node e)
{
    reserve(c, c->n + 1);
    int32_t l = c->n - 1;
    while (l >= 0 && !(c->nodes[l].last_used <= e.last_used)) l--;
    int32_t pos = l + 1;
    memmove(&c->nodes[pos + 1], &c->nodes[pos], (size_t)(c->n - pos) * sizeof(orc_node));
    c->nodes[pos] = e;
    c->n++;
    return pos;
}

static orc_node deque_unlink(orc_cache *c, int32_t i)
{
    orc_node e = c->nodes[i];
    memmove(&c->nodes[i], &c->nodes[i + 1], (size_t)(c->n - i - 1) * sizeof(orc_node));
    c->n--;
    return e;
}

/* LinkedDeque.reposition, LinkedDeque.java:243-256 */
static void deque_reposition(orc_cache *c, int32_t i)
{
    const int64_t lu = c->nodes[i].last_used;
    if (i == 0 || c->nodes[i - 1].last_used <= lu) {
        if (i == c->n - 1 || c->nodes[i + 1].last_used >= lu) return; /* already in place */
    }
    orc_node e = deque_unlink(c, i);
    deque_insert(c, e);
}

/* evict(), clhm :329-352; makeDead subtracts |weight| (clhm :566-575) */
static int32_t evict(orc_cache *c, int32_t *victims, int32_t max_victims)
{
    int32_t nv = 0;
    while (c->weighted_size > c->capacity) {
        if (c->n == 0) return nv; /* poll() == null */
        orc_node e = deque_unlink(c, 0);
        int32_t w = e.weight < 0 ? -e.weight : e.weight;
        c->weighted_size -= w;
        if (victims && nv < max_victims) victims[nv] = e.key;
        nv++;
    }
    return nv;
}

/* putIfAbsent(key, value, lastUsed), clhm :804-834 with AddTask :590-611.
 * Returns the number of evicted keys (written to victims, oldest first), or
 * -1 if the key was already present (then only afterRead is applied). */
int32_t orc_cache_put_if_absent(orc_cache *c, int32_t key, int32_t weight, int64_t last_used,
                                int64_t now, int32_t *victims, int32_t max_victims,
                                int32_t *insert_pos)
{
    int32_t i = orc_cache_find(c, key);
    if (i >= 0) { /* afterRead(prior, lastUsed), clhm :829 */
        touch(&c->nodes[i], last_used, now);
        deque_reposition(c, i);
        refresh_oldest(c);
        return -1;
    }
    orc_node nd;
    nd.key = key;
    nd.weight = weight;
    nd.last_used = 0;
    touch(&nd, last_used, now);
    c->weighted_size += weight; /* AddTask.run :603 */
    int32_t pos = deque_insert(c, nd);
    if (insert_pos) *insert_pos = pos;
    int32_t nv = evict(c, victims, max_victims);
    refresh_oldest(c);
    return nv;
}

/* get(key, lastUsed), clhm :726-733 → afterRead → touch + applyRead :503-521 */
int orc_cache_get(orc_cache *c, int32_t key, int64_t last_used, int64_t now)
{
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return 0;
    touch(&c->nodes[i], last_used, now);
    deque_reposition(c, i);
    refresh_oldest(c);
    return 1;
}

/* replace / replaceQuietly with a new weight, clhm :902-985 + UpdateTask :629-652.
 * new_time: -1 quiet (no touch), 0 now, else a timestamp. */
int32_t orc_cache_update_weight(orc_cache *c, int32_t key, int32_t new_weight, int64_t new_time,
                                int64_t now, int32_t *victims, int32_t max_victims)
{
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return -1;
    int32_t diff = new_weight - c->nodes[i].weight;
    c->nodes[i].weight = new_weight;
    if (diff == 0) {
        if (new_time >= 0) { /* afterRead */
            touch(&c->nodes[i], new_time, now);
            deque_reposition(c, i);
            refresh_oldest(c);
        }
        return 0;
    }
    c->weighted_size += diff;
    if (new_time >= 0 && new_time != c->nodes[i].last_used) {
        touch(&c->nodes[i], new_time, now);
        deque_reposition(c, i);
    }
    int32_t nv = evict(c, victims, max_victims);
    refresh_oldest(c);
    return nv;
}

/* remove(key), clhm :861-870 + RemovalTask :614-627 */
int orc_cache_remove(orc_cache *c, int32_t key)
{
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return 0;
    orc_node e = deque_unlink(c, i);
    c->weighted_size -= e.weight < 0 ? -e.weight : e.weight;
    refresh_oldest(c);
    return 1;
}

/* oldestTime(), clhm :1125-1127: the field, not the head (equal except after setCapacity, see orc_ubm_unload_complete) */
int64_t orc_cache_oldest_time(const orc_cache *c) { return c->oldest_time; }

/* Convenience for the parity tests: rebuild a cache whose deque is exactly
 * (lu[i], wt[i]) oldest-first, then run one putIfAbsent of a new key. */
void orc_evict_eval(const int64_t *lu, const int32_t *wt, int32_t n, int64_t capacity, int32_t weight,
                    int64_t last_used, int64_t now, orc_evict_result *out)
{
    orc_cache c;
    orc_cache_init(&c, INT64_MAX);
    for (int32_t i = 0; i < n; i++) {
        /* append in deque order: equal timestamps keep FIFO order (LinkedDeque.java:267) */
        reserve(&c, c.n + 1);
        c.nodes[c.n].key = i;
        c.nodes[c.n].weight = wt[i];
        c.nodes[c.n].last_used = lu[i];
        c.n++;
        c.weighted_size += wt[i];
    }
    c.capacity = capacity;
    refresh_oldest(&c);
    int32_t pos = 0;
    int32_t *victims = (int32_t *)malloc((size_t)(n + 2) * sizeof(int32_t));
    int32_t nv = orc_cache_put_if_absent(&c, n /* new key */, weight, last_used, now, victims, n + 2, &pos);
    out->insert_pos = pos;
    out->n_victims = nv;
    out->self_evicted = 0;
    for (int32_t i = 0; i < nv; i++)
        if (victims[i] == n) out->self_evicted = 1;
    out->weighted_size = c.weighted_size;
    out->oldest_time = orc_cache_oldest_time(&c);
    free(victims);
    orc_cache_free(&c);
}

/* ======================================================================== */
/* ModelCacheUnloadBufManager.java — unload-buffer accounting on top of the  */
/* cache.  Evictions are fed back through entryRemoved exactly as            */
/* ModelMesh.onEviction does while still under the eviction lock             */
/* (MM.java:2876-2878).                                                     */
#define ORC_UNLOADBUF_KEY (-1000000)

static void ubm_adjust_agg(orc_ubm *u, int32_t delta, int64_t now);

static void ubm_on_evicted(orc_ubm *u, const int32_t *keys, const int32_t *weights, int32_t n, int64_t now)
{
    for (int32_t i = 0; i < n; i++) {
        if (u->n_evicted < ORC_UBM_MAX_EVICTED) u->evicted[u->n_evicted] = keys[i];
        u->n_evicted++;
        /* entryRemoved(weight), ModelCacheUnloadBufManager.java:311-316 */
        u->total_occupancy -= weights[i];
        ubm_adjust_agg(u, weights[i], now);
    }
}

/* evict() (clhm :329-352) with the victims handed to the listener while still under the lock */
static void ubm_evict(orc_ubm *u, int64_t now)
{
    orc_cache *c = u->cache;
    int32_t cap = 16, nv = 0;
    int32_t *vk = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *vw = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    while (c->weighted_size > c->capacity && c->n > 0) {
        orc_node e = c->nodes[0];
        memmove(&c->nodes[0], &c->nodes[1], (size_t)(c->n - 1) * sizeof(orc_node));
        c->n--;
        c->weighted_size -= e.weight < 0 ? -e.weight : e.weight;
        if (nv == cap) {
            cap *= 2;
            vk = (int32_t *)realloc(vk, (size_t)cap * sizeof(int32_t));
            vw = (int32_t *)realloc(vw, (size_t)cap * sizeof(int32_t));
        }
        vk[nv] = e.key;
        vw[nv] = e.weight;
        nv++;
    }
    ubm_on_evicted(u, vk, vw, nv, now);
    free(vk);
    free(vw);
}

/* CacheEntry.updateWeightLocked → replaceQuietly → UpdateTask(quiet) */
static void ubm_set_weight(orc_ubm *u, int32_t key, int32_t w, int64_t now)
{
    orc_cache *c = u->cache;
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return;
    int32_t diff = w - c->nodes[i].weight;
    if (diff == 0) return;
    c->nodes[i].weight = w;
    c->weighted_size += diff;
    ubm_evict(u, now);
    /* afterWrite(UpdateTask) stores oldestTime after the task's evict() and before the listener runs (:443-447); the listener's
       nested writes store it again.  Evictions only happen inside such tasks, so the LAST store always sees the final head: storing
       it here, after the nested notifications, leaves the same value */
    refresh_oldest(c);
}

/* adjustAggregateUnloadingWeight, :375-392 */
static void ubm_adjust_agg(orc_ubm *u, int32_t delta, int64_t now)
{
    if (delta == 0) return;
    u->total_unloading += delta;
    int32_t nw = u->total_unloading;
    if (nw <= u->reserved) {
        nw = u->reserved;
    } else {
        int32_t cap = u->cache->capacity > INT32_MAX ? INT32_MAX : (int32_t)u->cache->capacity;
        if (cap < nw) nw = cap;
    }
    ubm_set_weight(u, ORC_UNLOADBUF_KEY, nw, now);
}

void orc_ubm_init(orc_ubm *u, orc_cache *cache, int32_t reserved, int64_t now)
{
    memset(u, 0, sizeof *u);
    u->cache = cache;
    u->reserved = reserved;
    /* newInternalCacheEntry: pinned with Long.MAX_VALUE, MM.java:1617-1622 */
    orc_cache_put_if_absent(cache, ORC_UNLOADBUF_KEY, reserved, INT64_MAX, now, NULL, 0, NULL);
}

int32_t orc_ubm_buffer_weight(const orc_ubm *u)
{
    int32_t i = orc_cache_find(u->cache, ORC_UNLOADBUF_KEY);
    return i < 0 ? 0 : u->cache->nodes[i].weight;
}

/* insertNewEntry, :130-145 */
int orc_ubm_insert_new_entry(orc_ubm *u, int32_t key, int32_t weight, int64_t last_used, int64_t now)
{
    ubm_adjust_agg(u, -weight, now);
    orc_cache *c = u->cache;
    if (orc_cache_find(c, key) >= 0) {
        orc_cache_get(c, key, last_used, now);
        ubm_adjust_agg(u, weight, now);
        return 0;
    }
    /* putIfAbsent with the victims' weights captured */
    orc_node nd = {0, weight, key};
    nd.last_used = last_used == 0 ? now : last_used;
    c->weighted_size += weight;
    int32_t l = c->n - 1;
    while (l >= 0 && !(c->nodes[l].last_used <= nd.last_used)) l--;
    if (c->n + 1 > c->cap_nodes) {
        int32_t nc = c->cap_nodes ? c->cap_nodes * 2 : 16;
        c->nodes = (orc_node *)realloc(c->nodes, (size_t)nc * sizeof(orc_node));
        c->cap_nodes = nc;
    }
    memmove(&c->nodes[l + 2], &c->nodes[l + 1], (size_t)(c->n - l - 1) * sizeof(orc_node));
    c->nodes[l + 1] = nd;
    c->n++;
    u->total_occupancy += weight;
    ubm_evict(u, now);
    refresh_oldest(c);
    return 1;
}

/* adjustNewEntrySpaceRequest, :152-166 */
void orc_ubm_adjust_new_entry_space_request(orc_ubm *u, int32_t increase, int32_t key, int64_t now)
{
    int32_t i = orc_cache_find(u->cache, key);
    if (i < 0) return;
    int32_t nw = u->cache->nodes[i].weight + increase;
    u->total_occupancy += increase;
    ubm_adjust_agg(u, -increase, now);
    ubm_set_weight(u, key, nw, now);
}

/* cacheSpaceIsReady, :395-402 */
int orc_ubm_cache_space_is_ready(const orc_ubm *u, int32_t required)
{
    int32_t new_tuw = u->total_unloading + required;
    if (new_tuw <= u->reserved) return 1;
    return (int64_t)new_tuw + u->total_occupancy <= u->cache->capacity;
}

/* claimRequestedSpaceIfReady, :190-202 */
int orc_ubm_claim_requested_space_if_ready(orc_ubm *u, int32_t required, int64_t now)
{
    if (!orc_ubm_cache_space_is_ready(u, required)) return 0;
    ubm_adjust_agg(u, required, now);
    return 1;
}

/* payDownDeficitAndNotifyWaiters, :351-366 */
static void ubm_pay_down(orc_ubm *u, int32_t weight, int release, int64_t now)
{
    int32_t reduction = weight < u->cache_deficit ? weight : u->cache_deficit;
    if (reduction != 0) {
        u->cache_deficit -= reduction;
        weight -= reduction;
    }
    ubm_adjust_agg(u, release ? -weight : reduction, now);
}

/* adjustWeightAfterLoad, :224-246 */
void orc_ubm_adjust_weight_after_load(orc_ubm *u, int32_t delta, int32_t key, int64_t now)
{
    if (delta == 0) return;
    if (delta > 0) {
        int64_t rem = u->cache->capacity - u->cache->weighted_size;
        int32_t remaining = rem > INT32_MAX ? INT32_MAX : (int32_t)rem; /* cacheRemaining :340-342 */
        int32_t deficit = delta - remaining;
        if (deficit > 0) {
            ubm_adjust_agg(u, -deficit, now);
            u->cache_deficit += deficit;
        }
    }
    u->total_occupancy += delta;
    int32_t i = orc_cache_find(u->cache, key);
    if (i >= 0) ubm_set_weight(u, key, u->cache->nodes[i].weight + delta, now);
    if (delta < 0) ubm_pay_down(u, -delta, 0, now);
}

/* unloadComplete, :318-338 */
void orc_ubm_unload_complete(orc_ubm *u, int32_t weight, int success, int64_t now)
{
    if (success) {
        ubm_pay_down(u, weight, 1, now);
        return;
    }
    int64_t cap = u->cache->capacity;
    ubm_adjust_agg(u, -weight, now);
    u->cache->capacity = cap - weight > 1 ? cap - weight : 1;
    ubm_evict(u, now); /* setCapacity evicts and notifies under the lock, clhm :305-316 — without updateOldestTime(): the field
                          keeps the value of the last write unless a notification changes the buffer's weight */
}

/* removeEntry :281-298 with entryRemoved :311-316; returns the weight at removal or -1 */
int32_t orc_ubm_remove_entry(orc_ubm *u, int32_t key, int64_t now)
{
    orc_cache *c = u->cache;
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return -1;
    int32_t w = c->nodes[i].weight;
    orc_cache_remove(c, key);
    u->total_occupancy -= w;
    ubm_adjust_agg(u, w, now);
    return w;
}

/* discardFailedEntry, :343-349 */
void orc_ubm_discard_failed_entry(orc_ubm *u, int32_t weight, int64_t now)
{
    u->total_occupancy -= weight;
    ubm_pay_down(u, weight, 0, now);
}

/* insertFailedPlaceholderEntry, :250-274 */
int orc_ubm_insert_failed_placeholder_entry(orc_ubm *u, int32_t key, int32_t weight, int64_t last_used, int64_t now)
{
    orc_cache *c = u->cache;
    int64_t rem = c->capacity - c->weighted_size;
    int32_t remaining = rem > INT32_MAX ? INT32_MAX : (int32_t)rem;
    int32_t deficit = weight - remaining;
    if (deficit > 0) ubm_adjust_agg(u, -deficit, now);
    if (orc_cache_find(c, key) >= 0) {
        orc_cache_get(c, key, last_used, now);
        if (deficit > 0) ubm_adjust_agg(u, deficit, now);
        return 0;
    }
    orc_node nd = {0, weight, key};
    nd.last_used = last_used == 0 ? now : last_used;
    c->weighted_size += weight;
    int32_t l = c->n - 1;
    while (l >= 0 && !(c->nodes[l].last_used <= nd.last_used)) l--;
    reserve(c, c->n + 1);
    memmove(&c->nodes[l + 2], &c->nodes[l + 1], (size_t)(c->n - l - 1) * sizeof(orc_node));
    c->nodes[l + 1] = nd;
    c->n++;
    ubm_evict(u, now);
    refresh_oldest(c);
    u->total_occupancy += weight;
    if (deficit > 0) u->cache_deficit += deficit;
    return 1;
}
