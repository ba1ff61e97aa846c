// Multi-scale deformable attention forward, fused + camera-grouped LDS-tiled kernel -- gfx950 (MI355X).
//
// What bounds msda_fwd_tile is the window staging: every (tile, slice, QUERY level) workgroup copies the
// same L source windows into LDS -- 1.35 GB of L2->LDS traffic per launch at Wildtrack size against 0.28 GB
// of algorithmic bytes -- and pays two barriers per level for 4 taps per lane.  In MVDeTr all levels
// (cameras) have the same shape and a cell's queries of all L cameras sample the same windows, so here one
// workgroup owns a (tile, slice) and walks ALL NG = L query levels per staged window: staging, window writes
// and barriers drop by L, the taps per barrier rise by L.  That needs the sampling data of (camera, level)
// when level's window is resident, which the reference layout [Lq, M, L, P] scatters over the tensor; the
// fused path's level-major raw layout [Lq, L, M, P] (a row permutation of the module's Linear) makes it one
// contiguous run per query -- that is where this kernel pays (170 -> 128 us); on the reference layout it is
// only marginally ahead of the tile kernel (188 vs 197 us).
//
// Lane = (cell, half slice) as in the tile kernel, but with NG accumulator sets (one per camera) and NG
// online-softmax states.  The window copy is synchronous (its latency is now amortised over NG x 4 taps),
// which is what frees the registers for the accumulators.
//
// Shapes live on the device, so the kernel itself checks that the levels are equal; if they are not, the
// same launch runs the tile kernel's body instead (msda_tile_body.h) -- no second launch, no host knowledge.
#include "msda_group2_kernel.h"

namespace mvdetr {

bool msda_group_supported(int D, int L)
{
    static const bool enabled = [] { const char *e = getenv("MVDETR_MSDA_GROUP"); return !(e && e[0] == '0'); }();
    return enabled && (D == 16 || D == 32) && (L == 6 || L == 7 || (L >= 9 && L <= 16));
}

// options of a launch, from the environment (read once): MVDETR_MSDA_WINDOW_SHIFT=0 keeps the fused kernels' windows centred
// on the tile (A/B knob); the jobs are dealt to the XCDs as 2-D blocks of the tile grid (GROUP_OPT_BLOCKS: 566 -> 314 MB of
// memory-side traffic per launch against bands of the job list, round 4)
static int group_opts(int fused)
{
    static const bool no_shift = [] { const char *e = getenv("MVDETR_MSDA_WINDOW_SHIFT"); return e && e[0] == '0'; }();
    return ((fused && no_shift) ? GROUP_OPT_NO_SHIFT : 0) | GROUP_OPT_BLOCKS;
}

#ifdef MVDETR_GROUP_TRACE
__device__ unsigned long long *g_group_trace = nullptr;
#endif

// reads in flight ahead of the FMAs in msda_fwd_group2's tap stream, in pairs (measured at Wildtrack size: 2 and 3 alike, 4
// slower -- registers)
constexpr int GROUP2_DEPTH = 2;

int msda_forward_group(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *off, const float *logit, const float *ref, int64_t ref_bstride, int fused,
                       SamplingLayout lay, int B, int S, int M, int D, int L, float *out, const int *local_hits, bool standdown,
                       float *stats)
{
    const int opts = group_opts(fused) | ((standdown && !fused) ? GROUP_OPT_STANDDOWN : 0);
#define GROUP_ARGS st, value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, out, local_hits, opts
    if (L >= 9 && L <= 16) {           // many cameras: 4 lane groups x up to 4 cameras (msda_forward_group_many.hip)
        return msda_forward_group_many(st, value, shapes, lsi, off, logit, ref, ref_bstride, fused, lay, B, S, M, D, L, out,
                                       local_hits, opts, fused ? stats : nullptr);
    }
    // 6 / 7 cameras: the software-pipelined kernel (msda_group2_kernel.h) for every entry -- fused (raw offsets / logits, one
    // or P reference points per (query, level)) and the public contract (final locations / weights)
#define G2(CFG, LL) (fused == 2 ? launch_group2<CFG, LL, 2, GROUP2_DEPTH>(GROUP_ARGS, stats) : fused ? launch_group2<CFG, LL, 1, GROUP2_DEPTH>(GROUP_ARGS, stats) \
                                                                                                  : launch_group2<CFG, LL, 0, GROUP2_DEPTH>(GROUP_ARGS))
    if (D == 16 && L == 7) return G2(GWide16, 7);
    if (D == 16 && L == 6) return G2(GWide16, 6);
    if (D == 32 && L == 7) return G2(GWide32, 7);
    if (D == 32 && L == 6) return G2(GWide32, 6);
#undef G2
    return (int)hipErrorInvalidValue;
}

}  // namespace mvdetr

#ifdef MVDETR_GROUP_TRACE
// trace builds only: (re)arm the stamp table (zeroed device memory of `workgroups` x 4 x GROUP_TRACE_SLOTS words, or NULL)
extern "C" int mvdetr_debug_group_trace_arm(unsigned long long *table)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mvdetr::g_group_trace), &table, sizeof(table));
}
#endif
