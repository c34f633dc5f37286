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

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (int32_t i = j->begin; i < j->end; i++) {
        const orc_flat_req *q = &j->reqs[i];
        const orc_flat_model *m = &j->models[q->model];
        orc_place_req r;
        memset(&r, 0, sizeof r);
        r.type = m->type;
        r.self = q->self_pod;
        r.favour_self = (int32_t)(q->flags & 1u);
        r.pick = q->pick;
        r.last_used = q->last_used;
        r.now = j->now;
        r.loaded = j->ent_pod + m->ent_off;
        r.n_loaded = m->n_loaded;
        r.failed = j->ent_pod + m->ent_off + m->n_loaded;
        r.n_failed = m->n_failed;
        r.extra = j->extra + q->extra_off;
        r.n_extra = q->n_extra;
        r.fresh.lru_time = q->fresh_lru;
        r.fresh.capacity = q->fresh_capacity;
        r.fresh.used = q->fresh_used;
        r.fresh.count = q->fresh_count;
        r.fresh.rpm = q->fresh_rpm;
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
