// Latency floors for a 100k-decision launch on this GPU: what an empty launch, a pure stream (64 B in, 16 B out
// per lane) and a stream with k dependent gathers cost, to read place_batch_kernel's 9 us against.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/floor.hip -o /tmp/floor && /tmp/floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Req { int32_t model, self; uint32_t flags, pick; int64_t a, b, c, d; int32_t e, f, g, h; };
static_assert(sizeof(Req) == 64, "");
struct Out { int32_t x, y, z, w; };
struct RM { int32_t v[8]; };

__global__ void k_empty(int n) {}
__global__ void k_stream(const Req *r, Out *o, int n)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    Req q = r[d];
    o[d] = Out{q.model, (int)(q.a + q.b + q.c + q.d), q.e + q.f + q.g + q.h, (int)q.pick};
}
template <int K>
__global__ void k_chain(const Req *r, const RM *rm, const int32_t *tab, int tabn, Out *o, int n)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    Req q = r[d];
    RM m = rm[q.model];
    int x = m.v[0] & (tabn - 1);
#pragma unroll
    for (int k = 0; k < K; k++) x = tab[x] & (tabn - 1);  // K dependent L2-resident gathers
    o[d] = Out{x, (int)(q.a + q.b + q.c + q.d), q.e + q.f + q.g + q.h + m.v[7], (int)q.pick};
}

template <class F>
static float timeit(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main()
{
    for (int n : {100000, 1000000}) {
        const int tabn = 1 << 16;
        std::vector<Req> hr(n);
        for (int i = 0; i < n; i++) { hr[i] = Req{}; hr[i].model = i; hr[i].pick = i * 2654435761u; }
        std::vector<RM> hm(n);
        for (int i = 0; i < n; i++) for (int k = 0; k < 8; k++) hm[i].v[k] = (i * 40503u + k * 977u);
        std::vector<int32_t> ht(tabn);
        for (int i = 0; i < tabn; i++) ht[i] = (int32_t)((i * 2654435761u) >> 7);
        Req *r; RM *rm; int32_t *tab; Out *o;
        hipMalloc(&r, n * sizeof(Req)); hipMalloc(&rm, n * sizeof(RM)); hipMalloc(&tab, tabn * 4); hipMalloc(&o, n * sizeof(Out));
        hipMemcpy(r, hr.data(), n * sizeof(Req), hipMemcpyHostToDevice);
        hipMemcpy(rm, hm.data(), n * sizeof(RM), hipMemcpyHostToDevice);
        hipMemcpy(tab, ht.data(), tabn * 4, hipMemcpyHostToDevice);
        const int B = 256, G = (n + B - 1) / B;
        printf("n=%d\n", n);
        printf("  empty            %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(G), dim3(B), 0, 0, n); }, 300));
        printf("  stream 64B->16B  %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream, dim3(G), dim3(B), 0, 0, r, o, n); }, 300));
        printf("  +gather 32B      %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<0>, dim3(G), dim3(B), 0, 0, r, rm, tab, tabn, o, n); }, 300));
        printf("  +2 L2 gathers    %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<2>, dim3(G), dim3(B), 0, 0, r, rm, tab, tabn, o, n); }, 300));
        printf("  +4 L2 gathers    %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<4>, dim3(G), dim3(B), 0, 0, r, rm, tab, tabn, o, n); }, 300));
        printf("  +6 L2 gathers    %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<6>, dim3(G), dim3(B), 0, 0, r, rm, tab, tabn, o, n); }, 300));
        hipFree(r); hipFree(rm); hipFree(tab); hipFree(o);
    }
    return 0;
}
