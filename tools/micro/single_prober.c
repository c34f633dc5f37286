/* single_prober.c — a host thread that issues single load-target decisions (mmp_place_batch(n = 1), what the LB's getNext
 * does per request) back to back against a context while the caller does something else with it (bench.py: the C5 churn
 * leg), and records every call's wall time.  A C thread, not a Python one: a Python prober's samples would include its
 * waits for the interpreter lock.  Built by bench.py with gcc; takes the entry point as a function pointer. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*place_fn)(void *ctx, const void *reqs, int32_t n, const int32_t *extra, int32_t n_extra, int64_t now, void *outs);

static struct {
    pthread_t th;
    volatile int stop, running;
    place_fn fn;
    void *ctx;
    const char *reqs; /* n_reqs x 64 bytes */
    int32_t n_reqs;
    int64_t now;
    uint32_t *lat_ns;
    int64_t cap, n;
    int32_t errors;
} P;

static void *run(void *arg)
{
    (void)arg;
    char out[16];
    int64_t i = 0;
    while (!P.stop) {
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        const int rc = P.fn(P.ctx, P.reqs + 64 * (i % P.n_reqs), 1, NULL, 0, P.now, out);
        clock_gettime(CLOCK_MONOTONIC, &b);
        if (rc != 0) P.errors++;
        if (P.n < P.cap) P.lat_ns[P.n++] = (uint32_t)((b.tv_sec - a.tv_sec) * 1000000000LL + (b.tv_nsec - a.tv_nsec));
        i++;
    }
    return NULL;
}

int prober_start(void *fn, void *ctx, const void *reqs, int32_t n_reqs, int64_t now, int64_t max_samples)
{
    if (P.running) return -1;
    memset(&P, 0, sizeof P);
    P.fn = (place_fn)fn;
    P.ctx = ctx;
    P.reqs = (const char *)reqs;
    P.n_reqs = n_reqs;
    P.now = now;
    P.cap = max_samples;
    P.lat_ns = (uint32_t *)malloc((size_t)max_samples * sizeof(uint32_t));
    if (!P.lat_ns) return -2;
    P.running = 1;
    return pthread_create(&P.th, NULL, run, NULL);
}

/* stops the thread; copies up to `cap` samples (ns) into out, returns their number (negative: calls that failed) */
int64_t prober_stop(uint32_t *out, int64_t cap)
{
    if (!P.running) return 0;
    P.stop = 1;
    pthread_join(P.th, NULL);
    P.running = 0;
    const int64_t n = P.n < cap ? P.n : cap;
    memcpy(out, P.lat_ns, (size_t)n * sizeof(uint32_t));
    free(P.lat_ns);
    return P.errors ? -(int64_t)P.errors : n;
}
