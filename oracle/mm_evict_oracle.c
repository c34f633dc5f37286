/*
 * mm_evict_oracle.c — CPU restatement of the timestamp-ordered weighted LRU
 * (clhm/ConcurrentLinkedHashMap.java, clhm/LinkedDeque.java) and of the unload
 * buffer accounting (ModelCacheUnloadBufManager.java).
 *
 * TEST INFRASTRUCTURE ONLY (parity checker).  Models the drained ("strict",
 * single-threaded) order — SURVEY.md Appendix B#11.
 *
 * Pinned by the reference's own known-answer tests: EvictionsModelMeshTest.java
 * :36-125 and ModelMeshEvictionsTest.java:156-280 (tests/test_oracle_kat.py).
 */
#include <stdlib.h>
#include <string.h>

#include "mm_oracle.h"

void orc_cache_init(orc_cache *c, int64_t capacity)
{
    memset(c, 0, sizeof *c);
    c->capacity = capacity;
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
    return evict(c, victims, max_victims);
}

/* get(key, lastUsed), clhm :726-733 → afterRead → touch + applyRead :503-521 */
int orc_cache_get(orc_cache *c, int32_t key, int64_t last_used, int64_t now)
{
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return 0;
    touch(&c->nodes[i], last_used, now);
    deque_reposition(c, i);
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
        }
        return 0;
    }
    c->weighted_size += diff;
    if (new_time >= 0 && new_time != c->nodes[i].last_used) {
        touch(&c->nodes[i], new_time, now);
        deque_reposition(c, i);
    }
    return evict(c, victims, max_victims);
}

/* remove(key), clhm :861-870 + RemovalTask :614-627 */
int orc_cache_remove(orc_cache *c, int32_t key)
{
    int32_t i = orc_cache_find(c, key);
    if (i < 0) return 0;
    orc_node e = deque_unlink(c, i);
    c->weighted_size -= e.weight < 0 ? -e.weight : e.weight;
    return 1;
}

/* oldestTime(), clhm :1125-1133 */
int64_t orc_cache_oldest_time(const orc_cache *c) { return c->n ? c->nodes[0].last_used : -1; }

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
