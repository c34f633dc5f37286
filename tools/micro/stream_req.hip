// What does it cost merely to STREAM the load-target batch — 64-byte request rows in, 16-byte result rows out, nothing decided?
// The floor under place_batch_kernel / place_batch_memo_kernel, by access pattern:
//   A  a lane reads its own 64-byte row as four 16-byte loads (64-byte stride between lanes: every load instruction touches all
//      32 lines of the wavefront's 4 KB) and writes its 16-byte row                       — what the kernels do
//   B  the wavefront's 4 KB read fully coalesced straight into LDS (global_load_lds_dwordx4, lane l moves bytes [16 l, +16) of each
//      1 KB chunk), each lane then reads its row from LDS
//   C  as B through registers with an XOR swizzle (no LDS bank conflict on either side)
//   D  plain copy of the same bytes (51.2 MB in as float4, 12.8 MB out), no rows at all
// each as 256-thread workgroups x 1 tile per wavefront, and as 64-thread workgroups x TILES tiles (loads of all tiles first).
// Rotates through 6 buffer sets (requests from HBM, not from the Infinity Cache).
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/stream_req.hip -o /tmp/stream_req && /tmp/stream_req
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct __attribute__((aligned(16))) Row { int4 a, b, c, d; };

__device__ __forceinline__ int4 fold(const int4 &a, const int4 &b, const int4 &c, const int4 &d)
{
    return make_int4(a.x ^ b.y ^ c.z ^ d.w, a.y + b.z + c.w + d.x, a.z ^ b.w ^ c.x ^ d.y, a.w + b.x + c.y + d.z);
}

template <int TILES, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_strided(const Row *__restrict__ in, int4 *__restrict__ out, int n)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int d0 = wave * (TILES * 64) + lane;
    Row r[TILES];
#pragma unroll
    for (int t = 0; t < TILES; t++) r[t] = in[d0 + t * 64 < n ? d0 + t * 64 : n - 1];
#pragma unroll
    for (int t = 0; t < TILES; t++)
        if (d0 + t * 64 < n) out[d0 + t * 64] = fold(r[t].a, r[t].b, r[t].c, r[t].d);
}

template <int TILES, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_lds(const Row *__restrict__ in, int4 *__restrict__ out, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[WPB][TILES][4096];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int base = wave * (TILES * 64);
    const char *src = reinterpret_cast<const char *>(in);
    const size_t lim = (size_t)n * 64 - 16;
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            size_t off = ((size_t)(base + t * 64)) * 64 + i * 1024 + lane * 16;
            if (off > lim) off = lim;
            __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(&tile[wib][t][i * 1024]), 16, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int4 *row = reinterpret_cast<const int4 *>(&tile[wib][t][lane * 64]);
        const int d = base + t * 64 + lane;
        if (d < n) out[d] = fold(row[0], row[1], row[2], row[3]);
    }
}

template <int TILES, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_swz(const Row *__restrict__ in, int4 *__restrict__ out, int n)
{
    __shared__ __attribute__((aligned(16))) int4 tile[WPB][TILES][256];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int base = wave * (TILES * 64);
    const int4 *src = reinterpret_cast<const int4 *>(in);
    const size_t lim = (size_t)n * 4 - 1;
    int4 v[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            size_t c = ((size_t)(base + t * 64)) * 4 + i * 64 + lane;
            v[t][i] = src[c > lim ? lim : c];
        }
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = i * 16 + (lane >> 2), j = lane & 3;
            tile[wib][t][r * 4 + (j ^ ((r >> 1) & 3))] = v[t][i];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int r = lane, f = (r >> 1) & 3;
        const int4 a = tile[wib][t][r * 4 + (0 ^ f)], b = tile[wib][t][r * 4 + (1 ^ f)], c = tile[wib][t][r * 4 + (2 ^ f)],
                   e = tile[wib][t][r * 4 + (3 ^ f)];
        const int d = base + t * 64 + lane;
        if (d < n) out[d] = fold(a, b, c, e);
    }
}

__global__ __launch_bounds__(256) void k_copy(const int4 *__restrict__ in, int4 *__restrict__ out, int n)
{
    // thread i: four coalesced 16-byte reads spread over the block's 16 KB, one 16-byte write
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const int4 a = in[b], c = in[b + 256], d = in[b + 512], e = in[b + 768];
    out[i] = fold(a, c, d, e);
}

// E / F: the shortlist kernel's first phase in miniature — the row, then two gathers that depend on it (a 400 KB and a 40 KB table,
// as the model's type word and the caller's position), a little arithmetic, the result row.  E: one tile per wavefront, as k_strided<1>;
// F: persistent wavefronts (grid = what is resident), each looping over tiles with the NEXT tile's row in flight while the current
// one's gathers are waited for.
__global__ __launch_bounds__(256) void k_gather(const Row *__restrict__ in, int4 *__restrict__ out, int n, const int *__restrict__ t1,
                                                const int *__restrict__ t2)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    const Row r = in[d < n ? d : n - 1];
    const int a = t1[(unsigned)(d + r.a.x) % 100000u], b = t2[(unsigned)(d + r.a.y) % 10000u];
    int4 o = fold(r.a, r.b, r.c, r.d);
    o.x += a;
    o.y ^= b;
    if (d < n) out[d] = o;
}
template <int WAVES_PER_EU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void k_gather_pipe(const Row *__restrict__ in,
                                                                                                          int4 *__restrict__ out, int n,
                                                                                                          const int *__restrict__ t1,
                                                                                                          const int *__restrict__ t2)
{
    const int tiles = (n + 255) / 256;
    int tile = blockIdx.x;
    if (tile >= tiles) return;
    int d = tile * 256 + threadIdx.x;
    Row r = in[d < n ? d : n - 1];
    while (true) {
        const int next = tile + gridDim.x;
        const int dn = next * 256 + threadIdx.x;
        Row rn;
        const bool more = next < tiles;  // (uniform)
        if (more) rn = in[dn < n ? dn : n - 1];
        const int a = t1[(unsigned)(d + r.a.x) % 100000u], b = t2[(unsigned)(d + r.a.y) % 10000u];
        int4 o = fold(r.a, r.b, r.c, r.d);
        o.x += a;
        o.y ^= b;
        if (d < n) out[d] = o;
        if (!more) break;
        r = rn;
        d = dn;
        tile = next;
    }
}

// G: E plus, switchable, what the real first phase has beyond it: STAGE — 2.5 KB of tables copied global -> LDS by the workgroup and a
// barrier before they are read (a dependent LDS read picks the result); WORK — ~150 dependent 64-bit integer instructions.
template <bool STAGE, int WORK, int WAVES_PER_EU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void k_gather2(const Row *__restrict__ in, int4 *__restrict__ out, int n,
                                                                                                      const int *__restrict__ t1, const int *__restrict__ t2)
{
    __shared__ int tab[640];
    const int d = blockIdx.x * 256 + threadIdx.x;
    const Row r = in[d < n ? d : n - 1];
    if (STAGE)
        for (int i = threadIdx.x; i < 640; i += 256) tab[i] = t1[i];
    const int a = t1[(unsigned)(d + r.a.x) % 100000u], b = t2[(unsigned)(d + r.a.y) % 10000u];
    if (STAGE) __syncthreads();
    int4 o = fold(r.a, r.b, r.c, r.d);
    long long acc = ((long long)r.b.x << 32) | (unsigned)r.b.y;
#pragma unroll 1
    for (int i = 0; i < WORK; i++) acc = acc * 0x9E3779B97F4A7C15ll + (acc >> 29) + a;
    o.x += a + (int)acc;
    o.y ^= b;
    if (STAGE) o.z += tab[(unsigned)(a + b + (int)(acc >> 40)) % 640u];
    if (d < n) out[d] = o;
}

int main()
{
    const int n = 800000, NB = 6, K = 200;
    std::vector<void *> in(NB), out(NB);
    for (int b = 0; b < NB; b++) {
        CK(hipMalloc(&in[b], (size_t)(n + 1024) * 64));
        CK(hipMalloc(&out[b], (size_t)(n + 1024) * 16));
        CK(hipMemset(in[b], b + 1, (size_t)(n + 1024) * 64));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 24; i++) launch(i % NB);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < K; i++) launch(i % NB);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / K;
        printf("%-44s %7.2f us per 800k  %5.2f TB/s of 64 MB\n", name, us, 64.0 / us);
        return 0;
    };
#define ROWK(kern, T, block)                                                                                                        \
    run(#kern "<" #T "> block " #block, [&](int b) {                                                                               \
        const int waves = (n + T * 64 - 1) / (T * 64);                                                                             \
        hipLaunchKernelGGL((kern<T, block / 64>), dim3((waves * 64 + block - 1) / block), dim3(block), 0, st, (const Row *)in[b], (int4 *)out[b], n); \
    })
    ROWK(k_strided, 1, 256);
    ROWK(k_strided, 2, 256);
    ROWK(k_strided, 4, 256);
    ROWK(k_strided, 4, 64);
    ROWK(k_lds, 1, 256);
    ROWK(k_lds, 2, 256);
    ROWK(k_lds, 4, 256);
    ROWK(k_lds, 4, 64);
    ROWK(k_swz, 1, 256);
    ROWK(k_swz, 2, 256);
    ROWK(k_swz, 4, 256);
    ROWK(k_swz, 4, 64);
    run("k_copy", [&](int b) { hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, (const int4 *)in[b], (int4 *)out[b], n); });
    int *t1, *t2;
    CK(hipMalloc(&t1, 100000 * 4));
    CK(hipMalloc(&t2, 10000 * 4));
    CK(hipMemset(t1, 1, 100000 * 4));
    CK(hipMemset(t2, 2, 10000 * 4));
    run("k_gather (1 tile per wavefront)", [&](int b) { hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, st, (const Row *)in[b], (int4 *)out[b], n, t1, t2); });
#define PIPE(W, BPC)                                                                                                                 \
    run("k_gather_pipe waves/SIMD " #W " blocks/CU " #BPC, [&](int b) {                                                              \
        hipLaunchKernelGGL(k_gather_pipe<W>, dim3(256 * BPC), dim3(256), 0, st, (const Row *)in[b], (int4 *)out[b], n, t1, t2);      \
    })
#define G2(STAGE, WORK, W)                                                                                                           \
    run("k_gather2 stage " #STAGE " work " #WORK " waves/SIMD " #W, [&](int b) {                                                   \
        hipLaunchKernelGGL((k_gather2<STAGE, WORK, W>), dim3((n + 255) / 256), dim3(256), 0, st, (const Row *)in[b], (int4 *)out[b], n, t1, t2); \
    })
    G2(false, 0, 8);
    G2(true, 0, 8);
    G2(false, 20, 8);
    G2(true, 20, 8);
    G2(true, 40, 8);
    G2(true, 20, 6);
    G2(true, 20, 4);
    PIPE(8, 8);
    PIPE(8, 6);
    PIPE(8, 4);
    PIPE(6, 6);
    PIPE(4, 4);
    PIPE(8, 12);
    run("empty-ish (n = 1, same grid)", [&](int b) { hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, (const int4 *)in[b], (int4 *)out[b], 1); });
    return 0;
}
