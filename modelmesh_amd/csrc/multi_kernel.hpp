// multi_kernel.hpp — several request arrays decided by ONE launch (mmp_place_multi_dev).
// A host that holds many batches of the size of one request set (100k decisions: 391 workgroups on 256 CUs, a launch that lasts an
// empty launch + one dependent chain + its tail, 0.18 of the HBM peak) pays a launch per batch; the same decisions in one launch
// run at the rate of a large batch (0.45).  The segments keep their own request / result / exclusion-pool arrays: a workgroup
// finds its segment from its index (a handful of scalar compares) and runs place_block on that segment with the array bases
// shifted so that the block's global decision index d = blockIdx.x * kPlaceBlock + lane addresses the segment's rows — the
// decision code itself is place_block's, unchanged.
#pragma once
#include "place_kernel.hpp"

namespace mmp {

constexpr int kMaxSegs = 16;
struct PlaceSeg {
    const mmp_place_req *reqs;
    mmp_place_out *outs;
    const int32_t *extra;
    int32_t first_block;  // workgroups [first_block, first_block + ceil(n / kPlaceBlock)) belong to this segment
    int32_t n;
};
struct PlaceSegs {
    int32_t n_segs;
    int32_t pad_;
    PlaceSeg seg[kMaxSegs];
};

__device__ __forceinline__ PlaceArgs segment_args(PlaceArgs A, const PlaceSegs &G)
{
    int k = 0;
    for (int i = 1; i < G.n_segs; i++)
        if ((int)blockIdx.x >= G.seg[i].first_block) k = i;  // (wave-uniform: scalar compares on the kernel arguments)
    const PlaceSeg sg = G.seg[k];
    const ptrdiff_t base = (ptrdiff_t)sg.first_block * kPlaceBlock;
    A.reqs = sg.reqs - base;   // only rows [base, base + n) are ever addressed
    A.outs = sg.outs - base;
    A.extra = sg.extra;
    A.n = (int32_t)base + sg.n;
    return A;
}

__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void place_multi_kernel(Snap S, PlaceArgs A, int32_t wpad,
                                                                                                          PlaceSegs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PlaceArgs As = segment_args(A, G);
    place_block<false>(S, As, wpad, smem);
}
// ... with the per-type shortlists checked first and no workgroup barrier (place_batch_m_kernel's workgroup; launches of >= kMemoFrom decisions)
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void place_multi_m_kernel(Snap S, PlaceArgs A, int32_t wpad, PlaceSegs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PlaceArgs As = segment_args(A, G);
    place_block<false, kReq64, true, true>(S, As, wpad, smem);
}
__global__ __launch_bounds__(kPlaceBlock) void place_multi_long_kernel(Snap S, PlaceArgs A, int32_t wpad, PlaceSegs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PlaceArgs As = segment_args(A, G);
    place_block<true>(S, As, wpad, smem);
}
__global__ __launch_bounds__(kPlaceBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void place_multi_long4_kernel(Snap S, PlaceArgs A, int32_t wpad,
                                                                                                                PlaceSegs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PlaceArgs As = segment_args(A, G);
    place_block<true>(S, As, wpad, smem);
}

// ---- one decision, one wavefront ------------------------------------------------------------------------------------------------
// mmp_place_batch(n = 1) through a launch is launch latency + this kernel + the completion flag's way to the host.  place_single_kernel
// is the batch kernel's workgroup (256 lanes, three workgroup barriers, a system-scope fence by every wavefront before the flag) run
// for ONE lane's work: 5.6-6.9 us in the kernel trace.  The same decision code on one wavefront: the request comes in the kernel
// arguments, the windows are staged by the one wavefront, lane 0 decides (lane_decide_win -> lane_decide_r -> the prefix-table
// phase), the wave path runs on the same wavefront if it is needed, ONE fence, the flag.
// the decision of ONE request by the calling wavefront (no workgroup barrier inside: other wavefronts of the workgroup may do
// something else meanwhile); writes A.outs[0]
__device__ __forceinline__ void single_place_wave(const Snap &S, PlaceArgs A, int32_t wpad, const mmp_place_req &rq, unsigned char *smem,
                                                  mmp_place_req *srq, int32_t *s_code)
{
    TypeWin *s_wins = reinterpret_cast<TypeWin *>(smem);
    uint64_t *s_scr = reinterpret_cast<uint64_t *>(smem + win_lds_bytes(S.T));
    const int lane = lane_id();
    const bool use_wins = A.wins != nullptr && !A.long_first;  // (a full cluster decides through the prefix tables: the windows are not read)
    if (use_wins) {
        const int chunks = ((S.T < kWinLds ? S.T : kWinLds) * (int)sizeof(TypeWin) + 1023) >> 10;
        const char *src = reinterpret_cast<const char *>(A.wins);
        char *dst = reinterpret_cast<char *>(s_wins);
        for (int c = 0; c < chunks; c++)
            __builtin_amdgcn_global_load_lds(src + (size_t)c * 1024 + lane * 16, (__attribute__((address_space(3))) void *)(dst + c * 1024), 16,
                                             0, 0);
    }
    if (lane == 0) {
        *srq = rq;  // (the prefix-table phase and the wave path read the request through A.reqs)
        *s_code = kLaneDone;
    }
    A.reqs = srq;
    A.n = 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lane == 0) {
        ResolvedReq r = resolve_req<false, true>(S, A, rq);
        mmp_place_out o;
        int code = kLaneHeadMiss;
        if (A.long_first) {
            merge_late_extras(r);
            // the type's recorded walk first (place_kernel.hpp: LongMemo): half the dependent levels of the walk itself
            code = kLaneDone;
            if (!(S.lmemo != nullptr && long_memo_try(S, A, r, o))) code = lane_decide_r<false, true>(S, A, r, o, BLds{});
        } else {
            if (use_wins) code = lane_decide_win(S, A, r, s_wins, s_scr, o);
            if (code == kLaneHeadMiss) {  // straight to the instantiation with the prefix-table phase: occupancy is nothing here, and a
                merge_late_extras(r);     // shortlist that spans the table would otherwise be walked twice (first pass: "long", second: decide)
                code = lane_decide_r<false, true>(S, A, r, o, BLds{});
            }
        }
        if (code == kLaneDone)
            A.outs[0] = o;
        else
            *s_code = code;
    }
    wave_sync();
    if (*s_code != kLaneDone) {  // (wave-uniform) the general path: the whole wavefront sweeps the table
        uint64_t *ew = reinterpret_cast<uint64_t *>(smem), *fw = ew + wpad;
        place_one(S, A, 0, ew, fw);
    }
}

__global__ __launch_bounds__(64) void place_single_lean_kernel(Snap S, PlaceArgs A, int32_t wpad, mmp_place_req rq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ mmp_place_req srq;
    __shared__ int32_t s_code;
    single_place_wave(S, A, wpad, rq, smem, &srq, &s_code);
    if (A.done_flag) {
        // the result row (whichever lane wrote it) before the flag.  (Relying on lane 0's release store alone for a row lane 0 wrote
        // itself was tried: the host then saw the flag before the row now and then — test_single_decisions_complete_while_a_commit_
        // is_running caught stale rows.)
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane_id() == 0) __hip_atomic_store(A.done_flag, A.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The cache-MISS route of ONE request (mmp_miss_batch, n = 1): the load target on wavefront 0, the request guards on lane 0 of
// wavefront 1 — side by side, both requests in the kernel arguments — one fence, the flag.
__global__ __launch_bounds__(128) void miss_single_kernel(Snap S, PlaceArgs A, int32_t wpad, mmp_place_req rq, GateArgs G, mmp_gate_req gr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ mmp_place_req srq;
    __shared__ int32_t s_code;
    if (threadIdx.x < 64)
        single_place_wave(S, A, wpad, rq, smem, &srq, &s_code);
    else if (threadIdx.x == 64)
        G.outs[0] = gate_eval(G, gr);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && A.done_flag) __hip_atomic_store(A.done_flag, A.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace mmp
