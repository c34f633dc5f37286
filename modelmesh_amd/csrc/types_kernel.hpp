// types_kernel.hpp — TypeConstraintManager's per-type instance sets rebuilt from labels on the
// device: ModelTypeConstraints.fromInstanceSet (TypeConstraintManager.java
// :416-447), instanceMatches (:478-486), refreshPerTypeInstanceSets (:680-725) and
// inferPreferredInstances (:727-747).  Labels are interned to bits of a 64-bit word per pod / type.
// Output bitmaps are over pod index — exactly the format mmp_types_load takes.  Row T (one past
// the configured types) is the row for model types that have no entry in the config:
// no constraint, preference = defaultPreferredInstances.
#pragma once
#include "snapshot.hpp"

namespace mmp {

// pass 1: one wave per 64 pods; per type the allowed / configured-preferred words, per pod the score
__global__ void type_sets_kernel(const mmp_pod_row *__restrict__ pods, const uint64_t *__restrict__ labels,
                                 int32_t P, int32_t W, int32_t T, const uint64_t *__restrict__ required,
                                 const uint64_t *__restrict__ preferred, uint64_t *__restrict__ allowed,
                                 uint64_t *__restrict__ cpref, int32_t *__restrict__ score)
{
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= W) return;
    const int p = w * 64 + lane;
    const bool present = p < P && !(pods[p].flags & (MMP_POD_SHUTTING_DOWN | MMP_POD_TOMBSTONE));
    const uint64_t lab = present ? labels[p] : 0;
    int32_t prohibited = 0, prefs = 0;
    for (int t = 0; t < T; t++) {
        const uint64_t R = required[t], F = preferred[t];
        // instanceMatches(labels, required, matchAll=true): false if either side is empty
        const bool in_req = present && R != 0 && lab != 0 && (R & ~lab) == 0;
        bool in_pref = false;
        if (!(R != 0 && in_req))  // the `else if` of :426-433: a required match is never also "preferred"
            in_pref = present && F != 0 && lab != 0 && (F & lab) != 0;
        if (present && R != 0 && !in_req) prohibited++;
        if (in_pref) prefs++;
        const uint64_t ba = __ballot(in_req), bp = __ballot(in_pref);
        if (lane == 0) {
            allowed[(size_t)t * W + w] = ba;
            cpref[(size_t)t * W + w] = bp;
        }
    }
    if (p < P) score[p] = present ? prohibited * 4 - prefs : INT32_MIN;  // :686-699 (INT32_MIN marks absent)
}

// pass 2: one workgroup per row r in [0, T]; resolves getPreferredInstances(type) for that row
__global__ __launch_bounds__(256) void type_prefer_kernel(int32_t P, int32_t W, int32_t T,
                                                          const uint64_t *__restrict__ required,
                                                          const uint64_t *__restrict__ allowed,
                                                          const uint64_t *__restrict__ cpref,
                                                          const int32_t *__restrict__ score,
                                                          uint64_t *__restrict__ prefer_out,
                                                          uint8_t *__restrict__ has_allowed,
                                                          uint8_t *__restrict__ has_prefer)
{
    __shared__ int32_t s_min[4], s_max[4], s_any[4], s_anya[4];
    const int r = blockIdx.x;  // r == T: the default row
    const bool has_req = r < T && required[r] != 0;
    // which score-based inference (if any) applies to this row, and over which include set
    //   no requirements  -> preferred = defaultPreferred = infer(scores, null)            (:714-715)
    //   requirements     -> configured preference if there is one, or if nothing is allowed;
    //                       otherwise infer(scores, allowedInstances)                       (:723-726)
    int32_t any_cp = 0, any_al = 0;
    if (has_req)
        for (int w = threadIdx.x; w < W; w += 256) {
            any_cp |= cpref[(size_t)r * W + w] != 0;
            any_al |= allowed[(size_t)r * W + w] != 0;
        }
    any_cp = __any(any_cp);
    any_al = __any(any_al);
    if (lane_id() == 0) {
        s_any[threadIdx.x >> 6] = any_cp;
        s_anya[threadIdx.x >> 6] = any_al;
    }
    __syncthreads();
    const bool have_cpref = s_any[0] | s_any[1] | s_any[2] | s_any[3];
    const bool have_allowed = s_anya[0] | s_anya[1] | s_anya[2] | s_anya[3];
    __syncthreads();
    const bool use_configured = has_req && (have_cpref || !have_allowed);
    const uint64_t *include = has_req ? allowed + (size_t)r * W : nullptr;

    int32_t mn = INT32_MAX, mx = 0;  // inferPreferredInstances: `max` starts at 0 (:729)
    if (!use_configured) {
        for (int p = threadIdx.x; p < P; p += 256) {
            const int32_t s = score[p];
            if (s == INT32_MIN) continue;
            if (include && !((include[p >> 6] >> (p & 63)) & 1ull)) continue;
            mn = s < mn ? s : mn;
            mx = s > mx ? s : mx;
        }
    }
    mn = wave_min_i32(mn);
    mx = -wave_min_i32(-mx);
    if (lane_id() == 0) {
        s_min[threadIdx.x >> 6] = mn;
        s_max[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
    mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const bool inferred_non_null = !use_configured && mn < mx;  // :746 `min < max ? set : null`
    for (int w = threadIdx.x >> 6; w < W; w += 4) {
        const int p = w * 64 + lane_id();
        uint64_t word;
        if (use_configured) {
            word = cpref[(size_t)r * W + w];
        } else {
            bool in = false;
            if (inferred_non_null && p < P) {
                const int32_t s = score[p];
                in = s != INT32_MIN && s == mx && (!include || ((include[w] >> lane_id()) & 1ull));
            }
            word = __ballot(in);
        }
        if (lane_id() == 0) prefer_out[(size_t)r * W + w] = word;
    }
    if (threadIdx.x == 0) {
        has_allowed[r] = has_req ? 1 : 0;
        has_prefer[r] = use_configured ? (have_cpref ? 1 : 0) : (inferred_non_null ? 1 : 0);
    }
}

}  // namespace mmp
