/*
 * mm_oracle_batch.c — batch / multi-thread driver around orc_place so that the
 * parity tests and bench.py's cpu_baseline leg can run whole request tables
 * without per-call Python overhead.  TEST INFRASTRUCTURE ONLY.
 *
 * The flat structs mirror the wire layout of include/mmplace.h on purpose (the
 * same numpy arrays feed both sides), but nothing here is shared code.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mm_oracle.h"

typedef struct {
    const orc_snapshot *snap;
    const orc_flat_model *models;
    const int32_t *ent_pod;
    const orc_flat_req *reqs;
    const int32_t *extra;
    orc_flat_out *outs;
    int64_t now;
    int32_t begin, end;
    double *lat_ns; /* optional per-decision latency */
} job_t;

static double now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

static inline void make_req(orc_place_req *r, const orc_flat_req *q, const orc_flat_model *models, const int32_t *ent_pod,
                            const int32_t *extra, int64_t now)
{
    const orc_flat_model *m = &models[q->model];
    memset(r, 0, sizeof *r);
    r->type = m->type;
    r->self = q->self_pod;
    r->favour_self = (int32_t)(q->flags & 1u);
    r->pick = q->pick;
    r->last_used = q->last_used;
    r->now = now;
    r->loaded = ent_pod + m->ent_off;
    r->n_loaded = m->n_loaded;
    r->failed = ent_pod + m->ent_off + m->n_loaded;
    r->n_failed = m->n_failed;
    r->extra = extra + q->extra_off;
    r->n_extra = q->n_extra;
    r->fresh.lru_time = q->fresh_lru;
    r->fresh.capacity = q->fresh_capacity;
    r->fresh.used = q->fresh_used;
    r->fresh.count = q->fresh_count;
    r->fresh.rpm = q->fresh_rpm;
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (int32_t i = j->begin; i < j->end; i++) {
        orc_place_req r;
        make_req(&r, &j->reqs[i], j->models, j->ent_pod, j->extra, j->now);
        orc_place_out o;
        double t0 = j->lat_ns ? now_ns() : 0.0;
        orc_place(j->snap, &r, &o, NULL);
        if (j->lat_ns) j->lat_ns[i] = now_ns() - t0;
        j->outs[i].chosen = o.chosen;
        j->outs[i].best = o.best;
        j->outs[i].n_candidates = o.n_candidates;
        j->outs[i].hash = o.hash;
    }
    return NULL;
}

int orc_place_batch(const orc_snapshot *snap, const orc_flat_model *models, const int32_t *ent_pod,
                    const orc_flat_req *reqs, const int32_t *extra, int32_t n, int64_t now,
                    orc_flat_out *outs, int32_t n_threads, double *lat_ns)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    int32_t per = (n + n_threads - 1) / n_threads;
    int32_t used = 0;
    for (int32_t t = 0; t < n_threads; t++) {
        int32_t b = t * per, e = b + per > n ? n : b + per;
        if (b >= e) break;
        jobs[t] = (job_t){snap, models, ent_pod, reqs, extra, outs, now, b, e, lat_ns};
        if (n_threads == 1) {
            worker(&jobs[t]);
        } else if (pthread_create(&th[t], NULL, worker, &jobs[t]) != 0) {
            worker(&jobs[t]);
            th[t] = 0;
        }
        used = t + 1;
    }
    if (n_threads > 1)
        for (int32_t t = 0; t < used; t++)
            if (th[t]) pthread_join(th[t], NULL);
    return 0;
}

/* ---- cpu_baseline: persistent pool over orc_place_lean ------------------------------------------- */
#include <stdatomic.h>

#define POOL_CHUNK 128

struct orc_pool {
    int32_t n_threads, max_pods;
    pthread_t *th;
    int32_t **scratch; /* per worker: 4 * (max_pods + 1) ints */
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    uint64_t generation; /* bumped per batch */
    int32_t running, quit;
    /* the batch in flight */
    const orc_snapshot *snap;
    const orc_flat_model *models;
    const int32_t *ent_pod;
    const orc_flat_req *reqs;
    const int32_t *extra;
    orc_flat_out *outs;
    double *lat_ns;
    int64_t now;
    int32_t n;
    atomic_int next; /* next undealt decision */
};

typedef struct {
    orc_pool *p;
    int32_t id;
} pool_arg;

static void pool_work(orc_pool *p, int32_t id)
{
    int32_t *scratch = p->scratch[id];
    for (;;) {
        const int32_t b = atomic_fetch_add(&p->next, POOL_CHUNK);
        if (b >= p->n) break;
        const int32_t e = b + POOL_CHUNK > p->n ? p->n : b + POOL_CHUNK;
        for (int32_t i = b; i < e; i++) {
            orc_place_req r;
            make_req(&r, &p->reqs[i], p->models, p->ent_pod, p->extra, p->now);
            orc_place_out o;
            const double t0 = p->lat_ns ? now_ns() : 0.0;
            orc_place_lean(p->snap, &r, &o, scratch);
            if (p->lat_ns) p->lat_ns[i] = now_ns() - t0;
            p->outs[i].chosen = o.chosen;
            p->outs[i].best = o.best;
            p->outs[i].n_candidates = o.n_candidates;
            p->outs[i].hash = (uint32_t)o.n_remaining; /* no audit hash on this path: the slot carries n_remaining */
        }
    }
}

static void *pool_main(void *arg)
{
    pool_arg *a = (pool_arg *)arg;
    orc_pool *p = a->p;
    const int32_t id = a->id;
    free(a);
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (!p->quit && p->generation == seen) pthread_cond_wait(&p->go, &p->mu);
        if (p->quit) {
            pthread_mutex_unlock(&p->mu);
            return NULL;
        }
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        pool_work(p, id);
        pthread_mutex_lock(&p->mu);
        if (--p->running == 0) pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->mu);
    }
}

orc_pool *orc_pool_create(int32_t n_threads, int32_t max_pods)
{
    if (n_threads < 1) n_threads = 1;
    orc_pool *p = (orc_pool *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->n_threads = n_threads;
    p->max_pods = max_pods;
    p->scratch = (int32_t **)calloc((size_t)n_threads, sizeof(int32_t *));
    p->th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int32_t t = 0; t < n_threads; t++) p->scratch[t] = (int32_t *)malloc(4 * (size_t)(max_pods + 1) * sizeof(int32_t));
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->go, NULL);
    pthread_cond_init(&p->done, NULL);
    /* worker 0 is the calling thread; the others park on `go` */
    int32_t made = 1;
    for (int32_t t = 1; t < n_threads; t++) {
        pool_arg *a = (pool_arg *)malloc(sizeof *a);
        a->p = p;
        a->id = t;
        if (pthread_create(&p->th[t], NULL, pool_main, a) != 0) {
            free(a);
            break;
        }
        made++;
    }
    p->n_threads = made;
    return p;
}

void orc_pool_destroy(orc_pool *p)
{
    if (!p) return;
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    for (int32_t t = 1; t < p->n_threads; t++) pthread_join(p->th[t], NULL);
    for (int32_t t = 0; t < p->n_threads; t++) free(p->scratch[t]);
    free(p->scratch);
    free(p->th);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->go);
    pthread_cond_destroy(&p->done);
    free(p);
}

int orc_pool_place(orc_pool *p, const orc_snapshot *snap, const orc_flat_model *models, const int32_t *ent_pod,
                   const orc_flat_req *reqs, const int32_t *extra, int32_t n, int64_t now, orc_flat_out *outs,
                   double *lat_ns)
{
    if (!p || snap->n_pods > p->max_pods) return -1;
    pthread_mutex_lock(&p->mu);
    p->snap = snap;
    p->models = models;
    p->ent_pod = ent_pod;
    p->reqs = reqs;
    p->extra = extra;
    p->outs = outs;
    p->lat_ns = lat_ns;
    p->now = now;
    p->n = n;
    atomic_store(&p->next, 0);
    p->running = p->n_threads - 1;
    p->generation++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    pool_work(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->running > 0) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    return 0;
}
