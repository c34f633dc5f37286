// Phase-clock build of libmmplace (a profiling tool, not the product): the whole library compiled with
// -DMMP_PHASE_CLOCK (wave-level s_memtime deltas between the markers of lane_decide / place_block, see
// place_kernel.hpp) plus a reader for the accumulated counters.  Built and driven by tools/phase_clock.py.
#define MMP_PHASE_CLOCK 1
#define MMP_PLAN_CLOCK 1  // the one-launch reaper plan: the 100 MHz clock at its phase boundaries (tools/plan_clock.py)
#include "../../modelmesh_amd/csrc/mmplace.hip"

extern "C" int mmp_debug_phase_read(unsigned int *out, int reset)
{
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -1;
    int rc = 0;
    if (hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(mmp::g_phase), sizeof(unsigned int) * 4096 * 16, 0, hipMemcpyDeviceToHost, st) != hipSuccess) rc = -2;
    if (rc == 0 && reset) {
        static const unsigned int zero[4096 * 16] = {};
        if (hipMemcpyToSymbolAsync(HIP_SYMBOL(mmp::g_phase), zero, sizeof zero, 0, hipMemcpyHostToDevice, st) != hipSuccess) rc = -3;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = -4;
    (void)hipStreamDestroy(st);
    return rc;
}

// the last one-launch plan's phase boundaries (workgroup 0's clock, 10 ns ticks)
extern "C" int mmp_debug_plan_clock(mmp_ctx *c, long long *out16)
{
    if (!c || !c->r_ps.p) return -1;
    mmp::PlanScalars h{};
    if (hipMemcpy(&h, c->r_ps.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (int k = 0; k < 16; k++) out16[k] = h.t_phase[k];
    return 0;
}
