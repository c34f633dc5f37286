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

}  // namespace mmp
