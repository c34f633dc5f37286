// Round-trip floor of a persistent decision kernel: the host rings a doorbell, one wavefront that never exits sees it
// and answers into pinned host memory, the host spins on the answer.  Two doorbell placements:
//   A  pinned host memory (hipHostMalloc): the device polls across PCIe
//   B  fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) written by the host through
//      the BAR: the device polls its own memory
// and, for scale, C: the same echo as one ordinary kernel launch per request + a completion flag (what
// mmp_place_batch(n = 1) does today).
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/doorbell.hip -o /tmp/doorbell && /tmp/doorbell
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void echo_persistent(volatile uint32_t *bell, volatile uint32_t *answer, volatile uint32_t *stop, const int32_t *tab,
                                long long idle_ticks)
{
    if (threadIdx.x != 0) return;
    uint32_t seen = 0;
    long long last = wall_clock64();
    for (;;) {
        const uint32_t b = __hip_atomic_load(const_cast<uint32_t *>(bell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (b != seen) {
            seen = b;
            int x = tab[b & 1023];          // a little dependent work standing in for the decision
            x = tab[x & 1023];
            __hip_atomic_store(const_cast<uint32_t *>(answer), b + (uint32_t)(x & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            last = wall_clock64();
        } else {
            if (__hip_atomic_load(const_cast<uint32_t *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
            if (wall_clock64() - last > idle_ticks) return;  // nobody rang for a while: give the CU back
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

__global__ void echo_once(uint32_t b, volatile uint32_t *answer, const int32_t *tab)
{
    if (threadIdx.x != 0) return;
    int x = tab[b & 1023];
    x = tab[x & 1023];
    __hip_atomic_store(const_cast<uint32_t *>(answer), b + (uint32_t)(x & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// D: the resident kernel's pattern — 64 lanes, each polling its own 128-byte slot (8 x 8 B of request + an 8 B bell)
// with system-scope loads, answering into a line of its own
struct Slot { uint64_t req[8]; uint64_t bell; uint64_t pad[7]; };
struct Ans { uint64_t out[2]; uint32_t done; uint32_t pad[11]; };
__global__ void echo_slots(Slot *slots, Ans *ans, volatile uint32_t *stop, long long idle_ticks)
{
    Slot *s = &slots[threadIdx.x];
    Ans *a = &ans[threadIdx.x];
    uint32_t seen = 0;
    long long last = wall_clock64();
    for (;;) {
        uint64_t w[8];
        for (int k = 0; k < 8; k++) w[k] = __hip_atomic_load(&s->req[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t bell = __hip_atomic_load(&s->bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t tag = (uint32_t)(bell >> 44);
        const bool fresh = tag != seen;
        if (__ballot(fresh)) {
            if (fresh) {
                uint64_t x = 0;
                for (int k = 0; k < 8; k++) x += w[k];
                __hip_atomic_store(&a->out[0], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&a->out[1], bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&a->done, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                seen = tag;
            }
            last = wall_clock64();
        } else {
            if (__hip_atomic_load(const_cast<uint32_t *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
            if (wall_clock64() - last > idle_ticks) return;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// E: what makes a launch-per-request round trip slow?  The same echo with a ~400-byte by-value argument block (the size of
// place_single_kernel's Snap + PlaceArgs + request) and with 38 KB of dynamic LDS.
struct BigArgs { uint64_t w[50]; };
__global__ void echo_once_big(BigArgs a, uint32_t b, volatile uint32_t *answer, const int32_t *tab)
{
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x != 0) return;
    int x = tab[(b + (uint32_t)a.w[7]) & 1023];
    x = tab[x & 1023];
    if (a.w[3] == 12345) dyn[0] = 1;
    __hip_atomic_store(const_cast<uint32_t *>(answer), b + (uint32_t)(x & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static void report(const char *name, std::vector<double> &us)
{
    std::sort(us.begin(), us.end());
    printf("%-58s p50 %6.2f us  p99 %6.2f us  min %6.2f us\n", name, us[us.size() / 2], us[us.size() * 99 / 100], us[0]);
}

int main()
{
    uint32_t *answer, *stop, *bellA, *bellB = nullptr;
    int32_t *tab;
    CK(hipHostMalloc((void **)&answer, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&stop, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&bellA, 64, hipHostMallocDefault));
    CK(hipMalloc((void **)&tab, 4096));
    CK(hipMemset(tab, 0, 4096));
    hipError_t eb = hipExtMallocWithFlags((void **)&bellB, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(eb));
    bool b_ok = eb == hipSuccess;
    if (b_ok) {
        CK(hipMemset(bellB, 0, 4096));
        CK(hipDeviceSynchronize());
        signal(SIGSEGV, on_segv);
        signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            *(volatile uint32_t *)bellB = 0;  // can the host store through the BAR?
            printf("host store to fine-grained device memory: ok\n");
        } else {
            printf("host store to fine-grained device memory: FAULT (no host-visible BAR mapping)\n");
            b_ok = false;
        }
        signal(SIGSEGV, SIG_DFL);
        signal(SIGBUS, SIG_DFL);
    }
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int clk_khz = 100000;
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
    const long long idle = (long long)clk_khz * 200;  // 200 ms
    const int N = 20000;
    for (int which = 0; which < 2; which++) {
        volatile uint32_t *bell = which == 0 ? bellA : bellB;
        if (which == 1 && !b_ok) continue;
        *answer = 0;
        *stop = 0;
        *bell = 0;
        hipLaunchKernelGGL(echo_persistent, dim3(1), dim3(64), 0, st, bell, answer, stop, tab, idle);
        std::vector<double> us;
        for (uint32_t i = 1; i <= (uint32_t)N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            __atomic_store_n((uint32_t *)bell, i, __ATOMIC_RELEASE);
            while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != i) __builtin_ia32_pause();
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        report(which == 0 ? "A persistent kernel, doorbell in pinned host memory" : "B persistent kernel, doorbell in fine-grained device memory", us);
    }
    {
        std::vector<double> us;
        *answer = 0;
        for (uint32_t i = 1; i <= (uint32_t)N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(echo_once, dim3(1), dim3(64), 0, st, i, answer, tab);
            while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != i) __builtin_ia32_pause();
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        CK(hipStreamSynchronize(st));
        report("C one launch per request + completion flag in pinned memory", us);
    }
    for (int variant = 0; variant < 4; variant++) {
        // 0: big args, 64 threads, no LDS   1: big args + 38 KB dynamic LDS   2: big args, 256 threads + LDS   3: as 2 on a high-priority stream
        hipStream_t sx = st;
        if (variant == 3) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            CK(hipStreamCreateWithPriority(&sx, hipStreamNonBlocking, hi));
        }
        BigArgs ba;
        memset(&ba, 0, sizeof ba);
        std::vector<double> us;
        *answer = 0;
        for (uint32_t i = 1; i <= 5000u; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(echo_once_big, dim3(1), dim3(variant >= 2 ? 256 : 64), variant >= 1 ? 38912 : 0, sx, ba, i, answer, tab);
            while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != i) __builtin_ia32_pause();
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        CK(hipStreamSynchronize(sx));
        const char *names[] = {"E0 launch per request, 400-byte argument block", "E1 ... + 38 KB dynamic LDS", "E2 ... + 256 threads",
                               "E3 ... on a high-priority stream"};
        report(names[variant], us);
    }
    {
        Slot *slots;
        Ans *ans;
        CK(hipHostMalloc((void **)&slots, sizeof(Slot) * 64, hipHostMallocDefault));
        CK(hipHostMalloc((void **)&ans, sizeof(Ans) * 64, hipHostMallocDefault));
        memset(slots, 0, sizeof(Slot) * 64);
        memset(ans, 0, sizeof(Ans) * 64);
        *stop = 0;
        hipLaunchKernelGGL(echo_slots, dim3(1), dim3(64), 0, st, slots, ans, stop, idle);
        std::vector<double> us;
        uint32_t seq[64] = {0};
        bool ok = true;
        for (uint32_t i = 1; i <= 20000u && ok; i++) {
            const int si = i % 64;
            const uint32_t q = ++seq[si];
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 8; k++) slots[si].req[k] = i + k;
            __atomic_store_n(&slots[si].bell, ((uint64_t)q << 44) | 12345, __ATOMIC_RELEASE);
            while (__atomic_load_n(&ans[si].done, __ATOMIC_ACQUIRE) != q) {
                __builtin_ia32_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) { printf("D: request %u (slot %d) not answered in 20 ms\n", i, si); ok = false; break; }
            }
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        report("D 64 lanes polling 64 slots (8 x 8 B + bell), answers in own lines", us);
    }
    {   // idle exit: nobody rings; the kernel must leave by itself
        *stop = 0;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(echo_persistent, dim3(1), dim3(64), 0, st, bellA, answer, stop, tab, idle);
        CK(hipStreamSynchronize(st));
        printf("idle persistent kernel left by itself after %.1f ms\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}
