// Camera-grouped fused MSDA forward, many-camera rigs (9..16 levels): four lane groups of three waves share one staged
// window and take up to four cameras each (768 threads, one workgroup per CU) -- the instantiations of
// msda_group_kernel.h's template that msda_forward_group.hip hands over to, in their own translation unit so that the
// two sets compile in parallel.  Design notes: msda_forward_group.hip, DESIGN.md 4.1b.
#include "msda_group_kernel.h"

namespace mvdetr {

int msda_forward_group_many(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                            const float *off, const float *logit, const float *ref, int64_t ref_bstride, int fused,
                            SamplingLayout lay, int B, int S, int M, int D, int L, float *out, const int *local_hits,
                            int opts, float *stats)
{
#define GROUP_ARGS st, value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, out, local_hits, opts, stats
    switch ((D == 32 ? 100 : 0) + L) {
#ifndef MVDETR_QUAD_WAVES
#define MVDETR_QUAD_WAVES 3          // waves per SIMD the register budget is cut for (768-thread workgroups: 3)
#endif
#define QUAD_CASE(DD, LL, CFG)                                                                                       \
    case DD + LL: return fused == 2 ? launch_group<CFG, LL, MVDETR_QUAD_WAVES, 2, 4, true>(GROUP_ARGS) : fused ? launch_group<CFG, LL, MVDETR_QUAD_WAVES, 1, 4, true>(GROUP_ARGS) \
                                                                                    : launch_group<CFG, LL, MVDETR_QUAD_WAVES, 0, 4>(GROUP_ARGS);
    QUAD_CASE(0, 9, GQuad16) QUAD_CASE(0, 10, GQuad16) QUAD_CASE(0, 11, GQuad16) QUAD_CASE(0, 12, GQuad16)
    QUAD_CASE(0, 13, GQuad16) QUAD_CASE(0, 14, GQuad16) QUAD_CASE(0, 15, GQuad16) QUAD_CASE(0, 16, GQuad16)
    QUAD_CASE(100, 9, GQuad32) QUAD_CASE(100, 10, GQuad32) QUAD_CASE(100, 11, GQuad32) QUAD_CASE(100, 12, GQuad32)
    QUAD_CASE(100, 13, GQuad32) QUAD_CASE(100, 14, GQuad32) QUAD_CASE(100, 15, GQuad32) QUAD_CASE(100, 16, GQuad32)
#undef QUAD_CASE
    default: break;
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace mvdetr
