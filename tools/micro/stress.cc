// Host-side stress of libmmplace for the sanitizer builds (tools/asan_lib.sh): everything the library's host side
// synchronises — the four latency slots with their completion flags, batch_mu / the state lock, the double-buffered snapshots, the
// submission threads, the resident kernel's doorbells, the delta-commit mirror, the registry arena, the per-stream buffers of split
// batches — exercised at once:
//   R request threads   single requests and small batches through the latency slots (mmp_place_batch n = 1 / 2 with exclusions / 300,
//                       mmp_gate_batch, mmp_serve_batch, mmp_miss_batch, mmp_route_batch), half the run with the resident kernel on
//   1 committer         a few InstanceRecords rewritten, then mmp_snapshot_commit (delta and full commits)
//   1 registry thread   mmp_models_upsert of a few ModelRecords (arena growth and squeezes)
//   1 batch thread      mmp_place_batch with 6000 host-pointer requests; device-pointer batches of 300 000 (split: two launches) on two
//                       streams of its own through mmp_issue_threads(4) + mmp_issue_flush
// Every return code must be MMP_OK and every result row plausible (a pod index, MMP_NONE or MMP_SELF).  No oracle here: the parity
// suites hold the results; this program is for AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer.
//   hipcc -O1 -g -std=c++17 -fsanitize=thread -fno-gpu-sanitize -Iinclude tools/micro/stress.cc \
//         -Lmodelmesh_amd/lib/variants -lmmplace_tsan -Wl,-rpath,$PWD/modelmesh_amd/lib/variants -lpthread -o /tmp/stress_tsan
// usage: stress [seconds = 10] [request threads = 8]
//        stress fuzz [calls = 20000]   hostile arguments from one thread (the C++ twin of tests/test_abi_fuzz_gpu.py, for the AddressSanitizer
//                                      build: Python does not start under a preloaded sanitizer runtime on the GPU box): every call returns
//                                      MMP_OK or an MMP_E* code and every row it wrote is well formed
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "mmplace.h"

static const int64_t NOW = 1760000000000LL;
static std::atomic<bool> g_stop{false};
static std::atomic<long> g_fail{0};
static std::atomic<long> g_calls[8];

#define CK(c, expr)                                                                           \
    do {                                                                                      \
        const int rc_ = (expr);                                                               \
        if (rc_ != MMP_OK) {                                                                  \
            if (g_fail.fetch_add(1) < 10) fprintf(stderr, "FAIL %s -> %d (%s)\n", #expr, rc_, mmp_last_error(c)); \
        }                                                                                     \
    } while (0)

static bool plausible(const mmp_place_out &o, int P)
{
    return (o.chosen >= 0 && o.chosen < P) || o.chosen == MMP_NONE || o.chosen == MMP_SELF;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 && strcmp(argv[1], "fuzz") ? atof(argv[1]) : 10.0;
    const int R = argc > 2 ? atoi(argv[2]) : 8;
    const int P = 3000, M = 30000;
    std::mt19937 rng0(11);
    std::vector<mmp_pod_row> pods(P);
    memset(pods.data(), 0, sizeof(mmp_pod_row) * P);
    for (int p = 0; p < P; p++) {
        pods[p].capacity = 8388608;
        pods[p].used = (int64_t)(8388608.0 * (0.3 + 0.6 * (rng0() % 1000) / 1000.0));
        pods[p].count = (int32_t)(rng0() % 40);
        pods[p].lru_time = NOW - 3600000 - (int64_t)(rng0() % 7200000);
        pods[p].rpm = (int32_t)(rng0() % 2000);
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
    if (mmp_create(&cfg, &c) != MMP_OK) {
        fprintf(stderr, "mmp_create: %s\n", mmp_last_error(nullptr));
        return 77;
    }
    std::vector<mmp_model_row> models(M);
    memset(models.data(), 0, sizeof(mmp_model_row) * M);
    std::vector<int32_t> ep;
    std::vector<int64_t> et;
    for (int m = 0; m < M; m++) {
        models[m].ent_off = (int32_t)ep.size();
        models[m].n_loaded = (int32_t)(rng0() % 3);
        models[m].last_used = NOW - 1000 - (int64_t)(rng0() % 3600000);
        for (int k = 0; k < models[m].n_loaded; k++) {
            ep.push_back((int32_t)((m * 7919 + k * 104729) % P));
            et.push_back(NOW - 60000);
        }
    }
    if (mmp_pods_load(c, pods.data(), P) || mmp_models_load(c, models.data(), M, ep.data(), et.data(), (int32_t)ep.size()) || mmp_snapshot_commit(c)) {
        fprintf(stderr, "load: %s\n", mmp_last_error(c));
        return 1;
    }
    auto make_req = [&](std::mt19937 &rng, mmp_place_req &r) {
        memset(&r, 0, sizeof r);
        r.model = (int32_t)(rng() % M);
        r.self_pod = (rng() % 16) ? (int32_t)(rng() % P) : -1;
        r.flags = (rng() % 8) ? 0u : MMP_REQ_FAVOUR_SELF;
        r.pick = rng();
        r.last_used = (rng() % 3) ? NOW - (int64_t)(rng() % 4000000) : 0;
        const mmp_pod_row &sp = pods[r.self_pod < 0 ? 0 : r.self_pod];  // (the committer rewrites rows of its OWN copy, see below)
        r.fresh_lru = sp.lru_time;
        r.fresh_capacity = sp.capacity;
        r.fresh_used = sp.used;
        r.fresh_count = sp.count;
    };
    auto request_thread = [&](int id) {
        std::mt19937 rng(100 + id);
        std::vector<mmp_place_req> rq(300);
        std::vector<mmp_place_out> out(300);
        std::vector<int32_t> extra(8);
        mmp_gate_req g;
        mmp_gate_out go;
        mmp_serve_req s;
        mmp_serve_out so;
        mmp_serve_counter cnt[3];
        while (!g_stop.load(std::memory_order_relaxed)) {
            const unsigned k = rng() % 8;
            if (k <= 2) {  // one request (the resident kernel or a latency slot)
                make_req(rng, rq[0]);
                CK(c, mmp_place_batch(c, rq.data(), 1, nullptr, 0, NOW, out.data()));
                if (!plausible(out[0], P)) g_fail++;
            } else if (k == 3) {  // two requests with exclusions of their own
                for (int i = 0; i < 2; i++) {
                    make_req(rng, rq[i]);
                    rq[i].extra_off = 3 * i;
                    rq[i].n_extra = (int32_t)(rng() % 4);
                }
                for (auto &e : extra) e = (int32_t)(rng() % P);
                CK(c, mmp_place_batch(c, rq.data(), 2, extra.data(), 8, NOW, out.data()));
                if (!plausible(out[0], P) || !plausible(out[1], P)) g_fail++;
            } else if (k == 4) {  // 300 requests: a latency slot, more than one workgroup, the completion counter
                for (int i = 0; i < 300; i++) make_req(rng, rq[i]);
                CK(c, mmp_place_batch(c, rq.data(), 300, nullptr, 0, NOW, out.data()));
                for (int i = 0; i < 300; i++)
                    if (!plausible(out[i], P)) g_fail++;
            } else {
                memset(&g, 0, sizeof g);
                g.model = (int32_t)(rng() % M);
                g.self_pod = (int32_t)(rng() % P);
                g.cache_capacity = 8388608;
                g.cache_weighted_size = 4000000;
                g.cache_oldest_time = NOW - 3600000;
                g.loaded_time = -1;
                g.fresh_lru = NOW - 3600000;
                g.fresh_capacity = 8388608;
                g.fresh_used = 4000000;
                g.fresh_loading_threads = 8;
                memset(&s, 0, sizeof s);
                s.model = g.model;
                s.self_pod = g.self_pod;
                s.n_cnt = 2;
                for (int i = 0; i < 2; i++) {
                    cnt[i].pod = (int32_t)((g.model * 7919 + i * 104729) % P);
                    cnt[i].in_use = (int32_t)(rng() % 4);
                    cnt[i].last_used = NOW - 1000;
                }
                if (k == 5)
                    CK(c, mmp_gate_batch(c, &g, 1, nullptr, nullptr, 0, nullptr, 0, NOW, 450000, &go));
                else if (k == 6)
                    CK(c, mmp_serve_batch(c, &s, 1, cnt, 2, nullptr, nullptr, 0, NOW, &so));
                else {
                    make_req(rng, rq[0]);
                    rq[0].model = g.model;
                    if (rng() & 1)
                        CK(c, mmp_miss_batch(c, &g, rq.data(), 1, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, NOW, 450000, &go, out.data()));
                    else
                        CK(c, mmp_route_batch(c, &g, &s, 1, cnt, 2, nullptr, nullptr, 0, nullptr, 0, NOW, 450000, &go, &so));
                }
            }
            g_calls[0]++;
        }
    };
    auto committer = [&]() {
        std::mt19937 rng(7);
        std::vector<mmp_pod_row> mine(pods);  // the committer's own copy of the table (request threads read `pods`, never written)
        while (!g_stop.load(std::memory_order_relaxed)) {
            const int k = 1 + (int)(rng() % ((rng() % 8) ? 6 : 40));  // mostly a few rows (the insertion commit), sometimes many (from scratch)
            std::vector<int32_t> idx(k);
            std::vector<mmp_pod_row> rows(k);
            for (int i = 0; i < k; i++) {
                idx[i] = (int32_t)(rng() % P);
                mmp_pod_row &r = mine[idx[i]];
                r.used = (int64_t)(8388608.0 * (0.2 + 0.79 * (rng() % 1000) / 1000.0));
                r.count = (int32_t)(rng() % 40);
                r.lru_time = NOW - 3600000 - (int64_t)(rng() % 7200000);
                rows[i] = r;
            }
            // (the same row twice in one call: the last one wins, as the header says)
            CK(c, mmp_pods_upsert(c, idx.data(), rows.data(), k));
            CK(c, mmp_snapshot_commit(c));
            g_calls[1]++;
            std::this_thread::sleep_for(std::chrono::microseconds(500 + rng() % 2000));
        }
    };
    auto registry = [&]() {
        std::mt19937 rng(9);
        while (!g_stop.load(std::memory_order_relaxed)) {
            const int k = 1 + (int)(rng() % 16);
            std::vector<int32_t> idx(k), e_pod;
            std::vector<int64_t> e_time;
            std::vector<mmp_model_row> rows(k);
            for (int i = 0; i < k; i++) {
                idx[i] = (int32_t)(rng() % M);
                memset(&rows[i], 0, sizeof rows[i]);
                rows[i].ent_off = (int32_t)e_pod.size();
                rows[i].n_loaded = (int32_t)(rng() % 4);
                rows[i].n_failed = (rng() % 10) ? 0 : 1;
                rows[i].last_used = NOW - (int64_t)(rng() % 3600000);
                const int32_t first = (int32_t)(rng() % P);
                for (int j = 0; j < rows[i].n_loaded + rows[i].n_failed; j++) {
                    e_pod.push_back((first + 37 * j) % P);
                    e_time.push_back(NOW - 1000);
                }
            }
            CK(c, mmp_models_upsert(c, idx.data(), rows.data(), k, e_pod.data(), e_time.data(), (int32_t)e_pod.size()));
            g_calls[2]++;
            std::this_thread::sleep_for(std::chrono::microseconds(300 + rng() % 1500));
        }
    };
    auto batches = [&]() {
        std::mt19937 rng(13);
        const int nb = 6000, nd = 300000;
        std::vector<mmp_place_req> rq(nd);
        std::vector<mmp_place_out> out(nd);
        for (auto &r : rq) make_req(rng, r);
        hipStream_t st[2];
        void *d_reqs = nullptr, *d_outs[2] = {nullptr, nullptr};
        if (hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&d_reqs, (size_t)nd * sizeof(mmp_place_req)) != hipSuccess || hipMalloc(&d_outs[0], (size_t)nd * 16) != hipSuccess ||
            hipMalloc(&d_outs[1], (size_t)nd * 16) != hipSuccess ||
            hipMemcpy(d_reqs, rq.data(), (size_t)nd * sizeof(mmp_place_req), hipMemcpyHostToDevice) != hipSuccess) {
            g_fail++;
            fprintf(stderr, "FAIL batch thread setup\n");
            return;
        }
        CK(c, mmp_issue_threads(c, 4));
        int round = 0;
        while (!g_stop.load(std::memory_order_relaxed)) {
            CK(c, mmp_place_batch(c, rq.data(), nb, nullptr, 0, NOW, out.data()));
            for (int i = 0; i < nb; i += 97)
                if (!plausible(out[i], P)) g_fail++;
            for (int i = 0; i < 6; i++) CK(c, mmp_place_batch_dev(c, d_reqs, nd, nullptr, NOW, d_outs[i & 1], st[i & 1]));
            CK(c, mmp_issue_flush(c));
            if (hipStreamSynchronize(st[0]) != hipSuccess || hipStreamSynchronize(st[1]) != hipSuccess) g_fail++;
            if (hipMemcpy(out.data(), d_outs[round & 1], (size_t)nd * 16, hipMemcpyDeviceToHost) != hipSuccess) g_fail++;
            for (int i = 0; i < nd; i += 997)
                if (!plausible(out[i], P)) g_fail++;
            if ((++round % 8) == 0) {  // the helpers stopped and started again under load
                CK(c, mmp_issue_threads(c, 0));
                CK(c, mmp_issue_threads(c, 4));
            }
            g_calls[3]++;
        }
        CK(c, mmp_issue_threads(c, 0));
        CK(c, mmp_stream_retire(c, st[0]));
        CK(c, mmp_stream_retire(c, st[1]));
        (void)hipStreamDestroy(st[0]);
        (void)hipStreamDestroy(st[1]);
        (void)hipFree(d_reqs);
        (void)hipFree(d_outs[0]);
        (void)hipFree(d_outs[1]);
    };
    if (argc > 1 && !strcmp(argv[1], "fuzz")) {
        const long calls = argc > 2 ? atol(argv[2]) : 20000;
        std::mt19937 rng(2024);
        static const int32_t W32[] = {0, 1, -1, -2, 5, P - 1, P, P + 1, M - 1, M, M + 1, 65535, 1 << 20, INT32_MAX, INT32_MAX - 1, INT32_MIN, INT32_MIN + 1};
        static const int64_t W64[] = {0, 1, -1, (int64_t)1 << 62, -((int64_t)1 << 62), INT64_MAX, INT64_MIN, NOW, NOW + 20000, 42};
        auto w32 = [&]() { return W32[rng() % (sizeof W32 / sizeof W32[0])]; };
        auto w64 = [&]() { return W64[rng() % (sizeof W64 / sizeof W64[0])]; };
        std::vector<mmp_place_req> rq(64);
        std::vector<mmp_place_out> out(64);
        std::vector<int32_t> pool(32);
        long refused = 0, decided = 0;
        for (long it = 0; it < calls; it++) {
            const int n = 1 + (int)(rng() % 40);
            for (int i = 0; i < n; i++) {
                make_req(rng, rq[i]);
                if (rng() & 1) rq[i].model = w32();
                if (rng() & 1) rq[i].self_pod = w32();
                if ((rng() & 3) == 0) {
                    rq[i].last_used = w64();
                    rq[i].fresh_lru = w64();
                    rq[i].fresh_capacity = w64();
                    rq[i].fresh_used = w64();
                    rq[i].fresh_count = w32();
                    rq[i].fresh_rpm = w32();
                    rq[i].flags = rng();
                }
                if ((rng() & 3) == 0) {
                    rq[i].extra_off = (rng() & 7) ? (int32_t)(rng() % 28) : w32();
                    rq[i].n_extra = (rng() & 7) ? (int32_t)(rng() % 5) : w32();
                }
            }
            for (auto &e : pool) e = (rng() & 3) ? (int32_t)(rng() % P) : w32();
            const int kind = (int)(rng() % 16);
            int rc;
            if (kind == 0)
                rc = mmp_place_batch(c, nullptr, n, pool.data(), 32, NOW, out.data());
            else if (kind == 1)
                rc = mmp_place_batch(c, rq.data(), -n, pool.data(), 32, NOW, out.data());
            else if (kind == 2)
                rc = mmp_place_batch(c, rq.data(), n, nullptr, 32, NOW, out.data());
            else if (kind == 3)
                rc = mmp_place_batch(c, rq.data(), n, pool.data(), -32, NOW, out.data());
            else if (kind == 4)
                rc = mmp_place_batch(c, rq.data(), n, pool.data(), 32, w64(), nullptr);
            else if (kind == 5) {  // instance rows that do not exist
                int32_t idx[3] = {w32(), (int32_t)(rng() % P), w32()};
                for (int32_t &k : idx)
                    if (k == P) k = P + 1;  // (index == the table's size APPENDS a row — legal, and a copy of row 0 would tie with it)
                mmp_pod_row rows[3] = {pods[0], pods[1], pods[2]};
                rc = (rng() & 1) ? mmp_pods_upsert(c, idx, rows, 3) : mmp_pods_remove(c, idx, 3);
                if (rc == MMP_OK) {  // (every index happened to exist: put the rows back as they were)
                    for (int k = 0; k < 3; k++) rows[k] = pods[idx[k] < P ? idx[k] : 0];
                    CK(c, mmp_pods_upsert(c, idx, rows, 3));
                }
            } else if (kind == 6) {  // registry rows whose entries the call does not bring / models beyond the table
                int32_t idx[2] = {(int32_t)(rng() % M), (rng() & 1) ? w32() : (int32_t)(rng() % M)};
                mmp_model_row rows[2];
                memset(rows, 0, sizeof rows);
                rows[0].n_loaded = w32();
                rows[1].ent_off = w32();
                rows[1].n_failed = (int32_t)(rng() % 3);
                int32_t e_pod[2] = {w32(), w32()};
                int64_t e_time[2] = {w64(), w64()};
                rc = mmp_models_upsert(c, idx, rows, 2, e_pod, e_time, 2);
            } else {
                rc = mmp_place_batch(c, rq.data(), n, pool.data(), 32, (rng() & 7) ? NOW : w64(), out.data());
                if (rc == MMP_OK) {
                    decided++;
                    for (int i = 0; i < n; i++)
                        if (!plausible(out[i], P)) {
                            if (g_fail.fetch_add(1) < 10) fprintf(stderr, "FAIL row %d: chosen %d best %d\n", i, out[i].chosen, out[i].best);
                        }
                }
            }
            if (rc != MMP_OK) {
                refused++;
                if (rc != MMP_EINVAL && rc != MMP_ESTATE) {
                    if (g_fail.fetch_add(1) < 10) fprintf(stderr, "FAIL kind %d -> %d (%s)\n", kind, rc, mmp_last_error(c));
                }
            }
            if ((it & 1023) == 1023) CK(c, mmp_snapshot_commit(c));  // (whatever the rejected calls left must still commit)
        }
        mmp_destroy(c);
        printf("fuzz: %ld calls, %ld refused with MMP_EINVAL, %ld batches decided; failures %ld\n", calls, refused, decided, g_fail.load());
        return g_fail.load() ? 1 : 0;
    }
    std::vector<std::thread> th;
    for (int i = 0; i < R; i++) th.emplace_back(request_thread, i);
    th.emplace_back(committer);
    th.emplace_back(registry);
    th.emplace_back(batches);
    const auto t0 = std::chrono::steady_clock::now();
    bool resident = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const bool want = el > seconds * 0.25 && el < seconds * 0.75;  // the resident kernel for the middle half of the run
        if (want != resident) {
            CK(c, mmp_resident(c, want ? 1 : 0));
            resident = want;
        }
    }
    g_stop.store(true);
    for (auto &t : th) t.join();
    uint64_t launches = 0, served = 0, punted = 0;
    (void)mmp_resident_stats(c, &launches, &served, &punted);
    int64_t n_split = 0, n_delta = 0;
    int32_t off = 0;
    (void)mmp_split_batches(c, &n_split, &off);
    (void)mmp_delta_commits(c, &n_delta);
    mmp_destroy(c);
    printf("stress: %.1f s, %d request threads: %ld request calls, %ld commits (%lld by insertion), %ld registry events, %ld batch rounds "
           "(%lld split batches); resident kernel: %llu launches, %llu served, %llu handed back; failures %ld\n",
           seconds, R, g_calls[0].load(), g_calls[1].load(), (long long)n_delta, g_calls[2].load(), g_calls[3].load(), (long long)n_split,
           (unsigned long long)launches, (unsigned long long)served, (unsigned long long)punted, g_fail.load());
    return g_fail.load() ? 1 : 0;
}
