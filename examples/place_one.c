/* The C ABI from plain C (C99, no C++/Python): build a 4-instance table and one model, commit, ask for one
 * load target — what a JNI / cgo / FFI host does through its binding.  Needs a GPU to run (mmp_create fails
 * with MMP_ENODEVICE otherwise; there is no CPU path).
 *   gcc -std=c99 -Iinclude examples/place_one.c -Lmodelmesh_amd/lib -lmmplace -Wl,-rpath,$PWD/modelmesh_amd/lib -o /tmp/place_one
 */
#include <stdio.h>
#include <string.h>

#include "mmplace.h"

int main(void)
{
    mmp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0;
    cfg.min_space_units = mmp_min_space_units(6400, 8, 8388608, 1); /* MM.java:765-771 */
    cfg.min_churn_age_ms = 600000;
    mmp_ctx *ctx = NULL;
    int rc = mmp_create(&cfg, &ctx);
    if (rc != MMP_OK) {
        fprintf(stderr, "mmp_create: %d (%s)\n", rc, mmp_last_error(NULL));
        return rc == MMP_ENODEVICE ? 77 : 1;
    }
    const int64_t now = 1760000000000LL;
    mmp_pod_row pods[4];
    memset(pods, 0, sizeof pods);
    for (int p = 0; p < 4; p++) {
        pods[p].lru_time = now - 3600000 - p;
        pods[p].capacity = 8388608;
        pods[p].used = 1000000 * (p + 1); /* pod 0 has the most free space */
        pods[p].version = 1;
        pods[p].count = 5;
        pods[p].loading_threads = 8;
        pods[p].id_order = (uint32_t)p;
        pods[p].replica_set = 0;
        pods[p].flags = MMP_POD_LIVE;
    }
    mmp_model_row model;
    memset(&model, 0, sizeof model);
    model.n_loaded = 1; /* already loaded on pod 0: excluded from the load targets */
    int32_t ent_pod[1] = {0};
    int64_t ent_time[1] = {now - 60000};
    mmp_place_req rq;
    memset(&rq, 0, sizeof rq);
    rq.self_pod = 3;
    rq.fresh_lru = pods[3].lru_time;
    rq.fresh_capacity = pods[3].capacity;
    rq.fresh_used = pods[3].used;
    rq.fresh_count = pods[3].count;
    mmp_place_out out;
    if ((rc = mmp_pods_load(ctx, pods, 4)) || (rc = mmp_models_load(ctx, &model, 1, ent_pod, ent_time, 1)) ||
        (rc = mmp_snapshot_commit(ctx)) || (rc = mmp_place_batch(ctx, &rq, 1, NULL, 0, now, &out))) {
        fprintf(stderr, "libmmplace: %d (%s)\n", rc, mmp_last_error(ctx));
        mmp_destroy(ctx);
        return 1;
    }
    printf("chosen=%d best=%d candidates=%d\n", out.chosen, out.best, out.n_candidates);
    mmp_destroy(ctx);
    /* pod 0 holds the model, so the most desirable eligible pod is pod 1 */
    return out.best == 1 ? 0 : 2;
}
