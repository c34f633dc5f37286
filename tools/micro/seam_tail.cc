// Where does the p99 of a single request at invokeModel's seams come from?  (serve / route n = 1 p99 27-30 us
// against p50 11-12 us and 12 us for the load target, on a quiet box.)  The four calls — mmp_place_batch, mmp_serve_batch,
// mmp_gate_batch, mmp_route_batch, n = 1 — are issued ROUND-ROBIN from one plain C++ thread (no interpreter), every call timed;
// printed per call: p50 / p90 / p99 / p99.9 / max, and for the slow calls (> 1.6 x the call's p50) WHEN they happen: their
// share per call kind, the distribution of the gaps between consecutive slow calls, and how many of them fall on a call whose
// predecessor (another kind) was slow too — a disturbance that does not care which call it hits (the host, the driver, the
// device's scheduler) shows as equal shares and as bursts; a slow call path shows as one kind's own tail.
//   g++ -O2 -std=c++17 -Iinclude tools/micro/seam_tail.cc -Lmodelmesh_amd/lib -lmmplace -Wl,-rpath,$PWD/modelmesh_amd/lib -lpthread -o /tmp/seam_tail
#include <sched.h>
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "mmplace.h"

static const int64_t NOW = 1760000000000LL;

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20000;
    // round 6: is the tail the host's?  `pin <cpu>` binds the calling thread to one CPU (sched_setaffinity) and asks for
    // SCHED_FIFO (granted or not, reported); either way the run reports what the kernel charged the thread while it measured: voluntary
    // and INVOLUNTARY context switches (getrusage RUSAGE_THREAD) — a slow call that coincides with neither is not the scheduler's.
    int pin_cpu = -1;
    for (int a = 2; a + 1 < argc; a++)
        if (!strcmp(argv[a], "pin")) pin_cpu = atoi(argv[a + 1]);
    bool pinned = false, fifo = false;
    if (pin_cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(pin_cpu, &set);
        pinned = sched_setaffinity(0, sizeof set, &set) == 0;
        sched_param sp;
        memset(&sp, 0, sizeof sp);
        sp.sched_priority = 10;
        fifo = sched_setscheduler(0, SCHED_FIFO, &sp) == 0;
    }
    const int P = 10000, M = 100000;
    std::vector<mmp_pod_row> pods(P);
    memset(pods.data(), 0, sizeof(mmp_pod_row) * P);
    std::mt19937 rng(3);
    for (int p = 0; p < P; p++) {
        pods[p].capacity = 8388608;
        pods[p].used = (int64_t)(8388608.0 * (0.4 + 0.5 * (rng() % 1000) / 1000.0));
        pods[p].count = std::poisson_distribution<int>(20)(rng);
        pods[p].lru_time = NOW - 3600000 - (int64_t)(rng() % 7200000);
        pods[p].rpm = (int32_t)(rng() % 2000);
        pods[p].loading_threads = 8;
        pods[p].version = 1;
        pods[p].id_order = (uint32_t)p;
        pods[p].flags = MMP_POD_LIVE;
    }
    mmp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.min_space_units = mmp_min_space_units(6400, 8, 8388608, 1);
    cfg.min_churn_age_ms = 600000;
    mmp_ctx *c = nullptr;
    if (mmp_create(&cfg, &c) != MMP_OK) { fprintf(stderr, "mmp_create: %s\n", mmp_last_error(nullptr)); return 77; }
    std::vector<mmp_model_row> models(M);
    memset(models.data(), 0, sizeof(mmp_model_row) * M);
    std::vector<int32_t> ep;
    std::vector<int64_t> et;
    for (int m = 0; m < M; m++) {
        models[m].ent_off = (int32_t)ep.size();
        models[m].n_loaded = 2;
        models[m].last_used = NOW - 1000 - (int64_t)(rng() % 3600000);
        for (int k = 0; k < 2; k++) { ep.push_back((int32_t)((m * 7919 + k * 104729) % P)); et.push_back(NOW - 60000); }
    }
    if (mmp_pods_load(c, pods.data(), P) || mmp_models_load(c, models.data(), M, ep.data(), et.data(), (int32_t)ep.size()) || mmp_snapshot_commit(c)) {
        fprintf(stderr, "load: %s\n", mmp_last_error(c));
        return 1;
    }
    const char *names[4] = {"place", "serve", "gates", "route"};
    rusage ru0;
    getrusage(RUSAGE_THREAD, &ru0);
    std::vector<double> us[4];
    std::vector<int> kind_of;     // call sequence
    std::vector<double> us_seq;
    for (int i = 0; i < 4 * (reps + 500); i++) {
        const int k = i & 3, j = i >> 2, m = (j * 31 + 7) % M, self = (j * 17) % P;
        mmp_place_req pr;
        memset(&pr, 0, sizeof pr);
        pr.model = m; pr.self_pod = self; pr.pick = (uint32_t)j * 2654435761u; pr.last_used = NOW - 5000;
        pr.fresh_lru = pods[self].lru_time; pr.fresh_capacity = pods[self].capacity; pr.fresh_used = pods[self].used; pr.fresh_count = pods[self].count;
        mmp_serve_req sr;
        memset(&sr, 0, sizeof sr);
        sr.model = m; sr.self_pod = self; sr.assume_completed_ms = 3000; sr.last_invoke_time = NOW - 10; sr.n_cnt = 2;
        mmp_serve_counter cnt[2];
        for (int q = 0; q < 2; q++) { cnt[q].pod = ep[models[m].ent_off + q]; cnt[q].in_use = q; cnt[q].last_used = NOW - 100 * q; }
        mmp_gate_req gr;
        memset(&gr, 0, sizeof gr);
        gr.model = m; gr.self_pod = self; gr.cache_capacity = 8388608; gr.loader_predicted = 6400;
        mmp_place_out po;
        mmp_serve_out so;
        mmp_gate_out go;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = 0;
        switch (k) {
        case 0: rc = mmp_place_batch(c, &pr, 1, nullptr, 0, NOW, &po); break;
        case 1: rc = mmp_serve_batch(c, &sr, 1, cnt, 2, nullptr, nullptr, 0, NOW, &so); break;
        case 2: rc = mmp_gate_batch(c, &gr, 1, nullptr, nullptr, 0, nullptr, 0, NOW, 450000, &go); break;
        case 3: rc = mmp_route_batch(c, &gr, &sr, 1, cnt, 2, nullptr, nullptr, 0, nullptr, 0, NOW, 450000, &go, &so); break;
        }
        const double d = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rc) { fprintf(stderr, "%s: %s\n", names[k], mmp_last_error(c)); return 1; }
        if (j >= 500) { us[k].push_back(d); kind_of.push_back(k); us_seq.push_back(d); }
    }
    rusage ru1;
    getrusage(RUSAGE_THREAD, &ru1);
    printf("calling thread: %s%s; while measuring: %ld voluntary and %ld involuntary context switches over %d calls (cpu now %d)\n",
           pin_cpu >= 0 ? (pinned ? "pinned to one CPU" : "pinning REFUSED") : "not pinned", pin_cpu >= 0 ? (fifo ? ", SCHED_FIFO" : ", SCHED_FIFO refused") : "",
           ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw, 4 * (reps + 500), sched_getcpu());
    double p50[4];
    for (int k = 0; k < 4; k++) {
        std::vector<double> s = us[k];
        std::sort(s.begin(), s.end());
        auto q = [&](double f) { return s[std::min(s.size() - 1, (size_t)(f * s.size()))]; };
        p50[k] = q(0.5);
        printf("%-6s n=1  p50 %6.2f  p90 %6.2f  p99 %6.2f  p99.9 %6.2f  max %7.2f us   (%zu calls)\n", names[k], q(0.5), q(0.9), q(0.99), q(0.999), s.back(),
               s.size());
    }
    // the slow calls in time
    std::vector<size_t> slow;
    int per_kind[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < us_seq.size(); i++)
        if (us_seq[i] > 1.6 * p50[kind_of[i]]) { slow.push_back(i); per_kind[kind_of[i]]++; }
    printf("slow calls (> 1.6 x the kind's p50): %zu of %zu = %.2f %%; by kind: place %d serve %d gates %d route %d\n", slow.size(), us_seq.size(),
           100.0 * slow.size() / us_seq.size(), per_kind[0], per_kind[1], per_kind[2], per_kind[3]);
    int adj = 0;
    std::vector<size_t> gaps;
    for (size_t i = 1; i < slow.size(); i++) {
        gaps.push_back(slow[i] - slow[i - 1]);
        if (slow[i] - slow[i - 1] == 1) adj++;
    }
    std::sort(gaps.begin(), gaps.end());
    if (!gaps.empty())
        printf("gaps between consecutive slow calls (in calls): min %zu  p10 %zu  p50 %zu  p90 %zu  max %zu; %d of %zu directly behind another slow call\n",
               gaps[0], gaps[gaps.size() / 10], gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10], gaps.back(), adj, slow.size());
    // wall-clock period: time between slow calls in microseconds (sum of the calls in between)
    std::vector<double> tgap;
    double acc = 0;
    size_t si = 0;
    for (size_t i = 0; i < us_seq.size(); i++) {
        acc += us_seq[i];
        if (si < slow.size() && slow[si] == i) { if (si) tgap.push_back(acc); acc = 0; si++; }
    }
    std::sort(tgap.begin(), tgap.end());
    if (!tgap.empty())
        printf("time between consecutive slow calls: p10 %.0f  p50 %.0f  p90 %.0f us\n", tgap[tgap.size() / 10], tgap[tgap.size() / 2], tgap[tgap.size() * 9 / 10]);
    mmp_destroy(c);
    return 0;
}
