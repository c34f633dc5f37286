// wave.hpp — wave64 (gfx950) cross-lane helpers used by the solver kernels.
// A wavefront is 64 lanes; every helper assumes the whole wave is active.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmp {

constexpr int kWave = 64;
constexpr int kNoPos = 0x7fffffff;  // "no such position" for first_set_* searches

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS written by some lanes of a wave and read by other lanes of the same wave:
// DS operations of one wave execute in order, so only the compiler has to be
// stopped from moving accesses across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int32_t shfl_i32(int32_t v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl((int)lo, src, 64);
    hi = (uint32_t)__shfl((int)hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Read lane `src` when `src` is wave-uniform (a ballot-derived index): v_readlane, no LDS crossbar trip.
__device__ __forceinline__ int32_t readlane_i32(int32_t v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, s);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), s);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ int32_t wave_sum_i32(int32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int32_t t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) { return (int64_t)wave_sum_u64((uint64_t)v); }

__device__ __forceinline__ int64_t wave_min_i64(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int64_t t = (int64_t)shfl_u64((uint64_t)v, lane_id() ^ o);
        v = t < v ? t : v;
    }
    return v;
}

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ int32_t wave_incl_scan_i32(int32_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// position of the k-th (0-based) set bit of v; requires k < popcount(v)
__device__ __forceinline__ int select_kth_bit(uint64_t v, int k)
{
    int r = 0;
    uint32_t x = (uint32_t)v;
    int c = __popc(x);
    if (k >= c) { k -= c; r = 32; x = (uint32_t)(v >> 32); }
    c = __popc(x & 0xffffu);
    if (k >= c) { k -= c; r += 16; x >>= 16; }
    x &= 0xffffu;
    c = __popc(x & 0xffu);
    if (k >= c) { k -= c; r += 8; x >>= 8; }
    x &= 0xffu;
    c = __popc(x & 0xfu);
    if (k >= c) { k -= c; r += 4; x >>= 4; }
    x &= 0xfu;
    c = __popc(x & 0x3u);
    if (k >= c) { k -= c; r += 2; x >>= 2; }
    x &= 0x3u;
    c = (int)(x & 1u);
    if (k >= c) r += 1;
    return r;
}

// Latency path of the host-pointer entry points (results go to pinned host memory): once every result row is
// visible to the host, the kernel — its last workgroup to finish, counted in `blocks` (device memory, left at
// zero) — stores `seq` into the pinned word `flag`; the host, spinning on that word, returns without the
// completion-signal round trip of hipStreamSynchronize.  Called by EVERY thread of the grid at the very end
// (no thread may have returned early); flag == nullptr: nothing to do.
struct DoneFlag {
    uint32_t *flag;
    uint32_t *blocks;
    uint32_t seq;
};

__device__ __forceinline__ void announce_done(const DoneFlag &D)
{
    if (!D.flag) return;     // wave-uniform
    __threadfence_system();  // this thread's result rows are visible to the host ...
    __syncthreads();         // ... and so are the rest of the workgroup's
    if (threadIdx.x == 0) {
        bool last = gridDim.x == 1;
        if (!last && atomicAdd(D.blocks, 1u) == gridDim.x - 1) {
            *D.blocks = 0;  // the other workgroups have all passed their fence: this one announces
            last = true;
        }
        if (last) __hip_atomic_store(D.flag, D.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// The audit hash of a shortlist (machinery of this repository, not of the reference): the candidates as a bitmap
// over rank positions, H = sum over 64-position words of  bits(word) * audit_mul(global word index)  (mod 2^64), folded to 32
// bits.  Linear in the bits on purpose: the candidate at bit b of word w contributes audit_mul(w) << b, so taking ONE candidate
// out (an exclusion) is one subtraction, partial sums over words add up across prefix tables and across pod-axis shards, and a
// word's term is one multiply.  audit_mul is odd (a word's term determines its bits) and non-linear in w (sums of multipliers
// of different words do not coincide).
__host__ __device__ __forceinline__ uint64_t audit_mul(uint64_t global_word)
{
    uint64_t x = (global_word + 1ull) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return x | 1ull;
}
__host__ __device__ __forceinline__ uint64_t audit_term(uint64_t bits, uint64_t global_word) { return bits * audit_mul(global_word); }

}  // namespace mmp
