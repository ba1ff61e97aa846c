// Multi-scale deformable attention forward, LDS-tiled encoder kernel -- gfx950 (MI355X).
//
// Shape of the problem (MVDeTr's shadow transformer, and any deformable *encoder* self-attention):
// the queries ARE the value tokens (Lq == S), a query's reference point is its own cell, and the
// learned offsets are a few pixels.  So the 8x16 block of neighbouring queries of one level samples,
// in every source level, a small window around the same normalised position.  The gather kernel
// pulls every one of the 4 corners x L*P taps x 64-byte head segments through the texture-address
// path (4.3 GB of L2->L1 traffic at Wildtrack size, 67 M cache-line requests); here a workgroup
//
//   1. owns a TH x TW tile of query cells of one level and a 128-byte slice of the token row
//      (two 16-channel heads, or one 32-channel head),
//   2. for each source level copies the (TH+2R) x (TW+2R) window of that slice into LDS with
//      full-line coalesced loads (zero-filled outside the level, so zero padding costs nothing),
//   3. lets each lane -- one (query, head) pair, all D channels in registers -- take its P taps of
//      that level from LDS with ds_read_b128,
//   4. remembers (bit mask) the rare taps whose footprint falls outside the window and finishes
//      them afterwards straight from global memory.
//
// Correct for ANY sampling locations (step 4); fast when they are local.  Shapes and level
// offsets are read on the device, so the host never needs them: the grid is persistent and each
// workgroup strides over the tile list it derives from spatial_shapes.
//
// Replaces (together with msda_forward.hip) ms_deformable_im2col_gpu_kernel of the reference
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_tile_body.h"
#include "msda_gather_body.h"
#include <stdlib.h>
#include <string.h>

namespace mvdetr {

template <typename Cfg, int FUSED>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::WAVES_PER_SIMD) void msda_fwd_tile(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw,
    const float *__restrict__ ref, int64_t ref_bstride, SamplingLayout lay, QueryLevels qr, int B, int S, int M,
    int L, float *__restrict__ out, const int *__restrict__ local_hits)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    if constexpr (FUSED == 0) {
        // the locality probe found the taps far from their queries: windows would be wasted, gather instead
        if (local_hits && *local_hits * MSDA_PROBE_NEAR_DIV < MSDA_PROBE_SAMPLES) {
            msda_fwd_gather_body<float, 4>((int64_t)blockIdx.x * Cfg::THREADS + threadIdx.x, (int64_t)gridDim.x * Cfg::THREADS,
                                           value, shapes, lsi, loc, aw, B, S, M, Cfg::D, L, S, TILE_P, out);
            return;
        }
    }
    msda_fwd_tile_body<Cfg, FUSED>(win, value, shapes, lsi, loc, aw, ref, ref_bstride, lay, qr, B, S, M, L, out);
}

// (maximum, 1 / sum exp) per (query, head): what msda_fwd_group2 leaves in `stats` from its online softmax, for the fused
// training calls that kernel does not take.  The backward recomputes a = __expf(logit - max) * (1 / sum) with the same intrinsic.
__global__ __launch_bounds__(256) void msda_softmax_stats_kernel(const float *__restrict__ logits, SamplingLayout lay,
                                                                 int64_t queries, int M, int L, float *__restrict__ stats)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < queries * M; i += (int64_t)gridDim.x * 256) {
        const int head = (int)(i % M);
        const float *wp = logits + (i / M) * lay.q_w + lay.head_w(head);
        float m = -INFINITY;
        for (int l = 0; l < L; ++l) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + l * lay.l_w);
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
        float s = 0.f;
        for (int l = 0; l < L; ++l) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + l * lay.l_w);
            s += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
        }
        *reinterpret_cast<float2 *>(stats + i * 2) = make_float2(m, 1.f / s);
    }
}

int msda_softmax_stats(hipStream_t st, const float *logits, SamplingLayout lay, int64_t queries, int M, int L, float *stats)
{
    const int64_t n = queries * M;
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(msda_softmax_stats_kernel, dim3(blocks), dim3(256), 0, st, logits, lay, queries, M, L, stats);
    return (int)hipGetLastError();
}

bool msda_tile_supported(int B, int S, int M, int D, int L, int Lq, int P, bool aligned16, int ql0, int ql1)
{
    if (!aligned16 || P != TILE_P || L > TILE_MAX_LEVELS || B < 1) return false;
    // the LDS-DMA window copies (msda_forward_group.hip, msda_backward_sampling.hip) address one batch element's value
    // tokens through a 32-bit buffer descriptor whose out-of-range sentinel is offset 2^31: a per-batch value tensor of
    // 2 GiB or more would alias it.  Such calls take the gather / lane-group kernels (64-bit addressing).
    if ((int64_t)S * M * D * 4 >= 0x7fffffffLL) return false;
    const bool all_levels = ql0 == 0 && ql1 == L;
    if (ql0 < 0 || ql1 <= ql0 || ql1 > L || (all_levels ? Lq != S : Lq > S)) return false;
    return (D == 16 && M % 2 == 0) || D == 32;
}

template <typename Cfg, int FUSED>
static int launch_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *aw, const float *ref, int64_t ref_bstride, SamplingLayout lay,
                       QueryLevels qr, int B, int S, int M, int L, float *out, const int *local_hits)
{
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tile<Cfg, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_fwd_tile<Cfg, FUSED>, Cfg::THREADS,
                                                         Cfg::LDS_BYTES) != hipSuccess || per_cu < 1)
            per_cu = 2;
        int n = cus * per_cu;
        return (n + 7) / 8 * 8;                          // keep the XCD interleave whole
    });
    static const KernelResources res = kernel_resources(reinterpret_cast<const void *>(&msda_fwd_tile<Cfg, FUSED>));
    msda_note_forward_kernel("msda_fwd_tile", &res);
    hipLaunchKernelGGL((msda_fwd_tile<Cfg, FUSED>), dim3((unsigned)blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES,
                       st, value, shapes, lsi, loc, aw, ref, ref_bstride, lay, qr, B, S, M, L, out, local_hits);
    return (int)hipGetLastError();
}

#define TILE_ARGS st, value, shapes, lsi, loc, aw, ref, ref_bstride, lay, qr, B, S, M, L, out, local_hits

template <int FUSED>
static int dispatch_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *aw, const float *ref, int64_t ref_bstride, SamplingLayout lay,
                         QueryLevels qr, int B, int S, int M, int D, int L, float *out, const int *local_hits = nullptr)
{
    // (128-byte slices; 64-byte ones -- twice the workgroups per CU -- were measured slower: DESIGN 5.2)
    if (D == 16) return launch_tile<CfgWide16, FUSED>(TILE_ARGS);
    if (D == 32) return launch_tile<CfgWide32, FUSED>(TILE_ARGS);
    return (int)hipErrorInvalidValue;
}

static bool group2_takes(int B, int S, int M, int D, int L, const SamplingLayout &lay)
{
    return msda_group_supported(D, L) && L <= 7 && msda_group_fits(B, S, M * D, lay);
}

bool msda_forward_tile_wants_probe(int B, int S, int M, int D, int L)
{
    return !group2_takes(B, S, M, D, L, plain_layout(M * L * TILE_P * 2, L * TILE_P * 2, TILE_P * 2, M * L * TILE_P, L * TILE_P, TILE_P));
}

int msda_forward_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *aw, int B, int S, int M, int D, int L, int Lq,
                      int P, float *out, const int *local_hits)
{
    const SamplingLayout lay = plain_layout(M * L * P * 2, L * P * 2, P * 2, M * L * P, L * P, P);     // [.., Lq, M, L, P(, 2)]
    // camera-grouped kernel also for the public (unfused) contract: 188 vs 197 us at Wildtrack size -- the
    // reference layout re-touches every sampling_loc line in 4 level iterations, so the gain is small
    if (msda_group_supported(D, L) && msda_group_fits(B, S, M * D, lay))
        return msda_forward_group(st, value, shapes, lsi, loc, aw, nullptr, 0, 0, lay, B, S, M, D, L, out, local_hits,
                                  /*standdown=*/msda_fwd_impl_knob() == 0);
    return dispatch_tile<0>(st, value, shapes, lsi, loc, aw, nullptr, 0, lay, QueryLevels{0, L, S}, B, S, M, D, L, out, local_hits);
}

int msda_forward_tile_fused(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                            const float *ref, int64_t ref_bstride, const float *offsets, const float *logits,
                            int layout, int qstride_l, int qstride_w, int ql0, int ql1, int Lq, int B, int S,
                            int M, int D, int L, float *out, float *stats)
{
    const int level_major = layout & 1, shared_ref = layout & 2, slice_major = layout & 4, ref_level_major = layout & 8;
    const int P = TILE_P;
    // reference points: [.., Lq, L, P, 2], [.., Lq, L, 2] (shared_ref) or [.., L, Lq, 2] (shared_ref + ref_level_major)
    const int r_q = ref_level_major ? 2 : L * (shared_ref ? 2 : P * 2);
    const int r_l = ref_level_major ? Lq * 2 : (shared_ref ? 2 : P * 2);
    SamplingLayout lay;
    if (slice_major) {
        // one tensor [.., Lq, M / hps, L, (hps x P x 2 offsets | hps x P logits)]: what the workgroup of a 128-byte
        // slice reads for a (query, level) is one contiguous run of hps * 12 floats; `logits` = `offsets` + hps * 8
        // Bit 4: the level is the OUTER index, [.., Lq, L, M / hps, run] -- a (query, level)'s runs of all slices are then
        // M / hps * chunk * 4 = 384 contiguous bytes = three whole 128-byte lines (at D = 16, M = 8), shared by the four
        // slice jobs of a tile, which run at the same time on one XCD (one L2); with the slice outermost a run shares its
        // lines with the same slice's NEXT level, i.e. with a later phase of the same job, by when the line has left L2.
        const int hps = 32 / D, chunk = hps * P * 3;
        if (layout & 16)
            lay = SamplingLayout{qstride_l, P * 2, (M / hps) * chunk, qstride_w, P, (M / hps) * chunk, hps, chunk, chunk, r_q, r_l};
        else
            lay = SamplingLayout{qstride_l, P * 2, chunk, qstride_w, P, chunk, hps, L * chunk, L * chunk, r_q, r_l};
    } else if (level_major) {
        lay = plain_layout(qstride_l, P * 2, M * P * 2, qstride_w, P, M * P, r_q, r_l);              // [.., Lq, L, M, P(, 2)]
    } else {
        lay = plain_layout(qstride_l, L * P * 2, P * 2, qstride_w, L * P, P, r_q, r_l);              // [.., Lq, M, L, P(, 2)]
    }
    // MVDeTr's camera counts: the camera-grouped kernel (which falls back to this file's tile body, in the
    // same launch, when the levels turn out on the device to have unequal shapes) -- msda_forward_group.hip
    // A query-sharded call (a rank's own cameras as queries, mvdetr_amd/dist.py) has too few query levels per
    // window to amortise the grouped staging and runs the tile kernel.
    const bool all_levels = ql0 == 0 && ql1 == L;
    const bool grouped = all_levels && msda_group_supported(D, L) && msda_group_fits(B, S, M * D, lay);
    // the training entry's statistics: the camera-grouped kernels (6 / 7 levels: msda_fwd_group2; 9 - 16: msda_fwd_group) write them
    // themselves; the tile kernel runs as it does for inference and a small pass over the logits follows
    const bool own_stats = grouped;
    int rc;
    if (grouped)
        rc = msda_forward_group(st, value, shapes, lsi, offsets, logits, ref, ref_bstride, shared_ref ? 2 : 1, lay, B, S,
                                M, D, L, out, nullptr, false, own_stats ? stats : nullptr);
    else
        rc = shared_ref ? dispatch_tile<2>(st, value, shapes, lsi, offsets, logits, ref, ref_bstride, lay,
                                           QueryLevels{ql0, ql1, Lq}, B, S, M, D, L, out)
                        : dispatch_tile<1>(st, value, shapes, lsi, offsets, logits, ref, ref_bstride, lay,
                                           QueryLevels{ql0, ql1, Lq}, B, S, M, D, L, out);
    if (!rc && stats && !own_stats) rc = msda_softmax_stats(st, logits, lay, (int64_t)B * Lq, M, L, stats);
    return rc;
}

}  // namespace mvdetr
