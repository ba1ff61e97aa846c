// Shared configuration of the LDS-tiled MSDA kernels (forward: msda_forward_tile.hip, backward:
// msda_backward_tile.hip).  Internal, not part of the C ABI.
#pragma once
#include "common.h"

namespace mvdetr {

constexpr int TILE_MAX_LEVELS = 16;     // 64-bit miss mask = L * P bits with P == 4
constexpr int TILE_P = 4;

// R_: halo in x (and in y unless RY_ is given).  The LDS-DMA window copies start every window row at a wave-uniform LDS
// base; rows of WW * SLICE * 4 bytes that are not multiples of 512 bytes (WW = 26 at R = 5) fault on gfx950, so a narrower
// halo is only available in y.
template <int D_, int SLICE_, int TH_, int TW_, int R_, int THREADS_ = TH_ * TW_ * 2, int RY_ = R_> struct TileCfg {
    static constexpr int D = D_, TH = TH_, TW = TW_, R = R_, RY = RY_;
    static constexpr int SLICE = SLICE_;              // floats of a token row staged per workgroup (32 = 128 B, 16 = 64 B)
    static constexpr int SUBS = 2;                    // lanes per query, each owning half a slice
    static constexpr int NV = SLICE / SUBS / 4;       // 16-byte chunks (float4 accumulators) per lane
    static constexpr int PARTS = SLICE / 4;           // float4 per token in LDS
    static constexpr int WH = TH + 2 * RY, WW = TW + 2 * R;
    static constexpr int THREADS = THREADS_;          // >= TH*TW*SUBS compute lanes; the surplus only helps the window copy
    static_assert(THREADS_ >= TH_ * TW_ * 2 && THREADS_ % 64 == 0, "whole waves covering the tile");
    static constexpr int COLSLOTS = THREADS / PARTS;  // window columns a copy pass covers ...
    static constexpr int ROWS_PER_PASS = COLSLOTS / WW;   // ... i.e. this many whole rows
    static constexpr int NSTAGE = (WH + ROWS_PER_PASS - 1) / ROWS_PER_PASS;   // float4 per lane per window
    static constexpr int LDS_BYTES = WH * WW * SLICE * 4;
    static constexpr int TOK_PER_BANKROW = 256 / (SLICE * 4);                 // tokens per 256-byte LDS bank row
    // workgroups per CU the LDS admits; the register allocation is capped to match (waves per SIMD)
    static constexpr int WGS_PER_CU = (160 * 1024) / LDS_BYTES;
    static constexpr int WAVES_PER_SIMD = WGS_PER_CU * (THREADS / 64) / 4;
#ifndef MVDETR_TAP_FENCE
#define MVDETR_TAP_FENCE 1
#endif
    static constexpr bool TAP_FENCE = MVDETR_TAP_FENCE;
    static_assert(SLICE == 16 || SLICE == 32, "64- or 128-byte slices");
    static_assert(D % (SLICE / SUBS) == 0, "a lane's channels must lie inside one head");
    static_assert(ROWS_PER_PASS >= 1, "window copy: one pass must cover at least one row");
};

// 128-byte slices: 71.7 KB window, 2 workgroups / CU.  64-byte slices: 35.8 KB, 4 workgroups / CU (twice the
// waves to hide LDS and global latency behind, at the price of duplicating the per-tap address math).
using CfgWide16 = TileCfg<16, 32, 8, 16, 6>;
using CfgWide32 = TileCfg<32, 32, 8, 16, 6>;

// Where element (query, head, level) of the sampling locations / weights starts, in floats:
//     query * q + (head / hps) * s + (head % hps) * h + level * l
// hps > 1 groups the heads whose channels one workgroup stages together (a 128-byte slice of the token row): the
// fused path's slice-interleaved layout keeps everything ONE workgroup reads for a (query, level) -- offsets and
// logits of its hps heads -- in one contiguous run, so that a 128-byte line is not fetched for a quarter of its
// bytes.  Plain layouts have hps = 1, s = h.  Covers the reference layout [.., Lq, M, L, P(, 2)] and the fused path's
// level-major, column-block and slice-interleaved layouts; filled in by the host.
// Reference points: query * r_q + level * r_l (+ 2 * point when there is one per point).
struct SamplingLayout {
    int q_l, h_l, l_l;      // locations (or raw offsets)
    int q_w, h_w, l_w;      // weights (or raw logits)
    int hps, s_l, s_w;      // head grouping (see above)
    int r_q, r_l;           // reference points
    __host__ __device__ int head_l(int head) const { return (head / hps) * s_l + (head % hps) * h_l; }
    __host__ __device__ int head_w(int head) const { return (head / hps) * s_w + (head % hps) * h_w; }
};
inline SamplingLayout plain_layout(int q_l, int h_l, int l_l, int q_w, int h_w, int l_w, int r_q = 0, int r_l = 0)
{
    return SamplingLayout{q_l, h_l, l_l, q_w, h_w, l_w, 1, h_l, h_w, r_q, r_l};
}

// Which levels' tokens are the queries of a call: [begin, end) of the L levels, Lq tokens in all.
struct QueryLevels {
    int begin, end, Lq;
};

// the fused training forward's statistics -- (maximum, 1 / sum exp) of a (query, head)'s L x P logits -- for calls whose forward
// kernel does not write them itself (msda_fwd_group2 does): msda_forward_tile.hip
int msda_softmax_stats(hipStream_t st, const float *logits, SamplingLayout lay, int64_t queries, int M, int L, float *stats);

// camera-grouped fused forward (msda_forward_group.hip)
bool msda_group_supported(int D, int L);
// its per-lane addresses are 32-bit float offsets from per-batch, per-camera bases: one batch element's reference points and
// output must each stay below 2^32 bytes -- and the WHOLE sampling tensors (all B elements), because msda_fwd_group2 folds the
// batch index into the 32-bit scalar offset of one whole-tensor buffer descriptor (per-batch descriptors cost the headline
// instantiation its zero-scratch build: ADVICE r04).  Anything larger runs the tile kernel: 64-bit addresses.
inline bool msda_group_fits(int B, int S, int row, const SamplingLayout &lay)
{
    const int64_t lim = (int64_t)1 << 30;                   // floats
    return (int64_t)B * S * lay.q_l < lim && (int64_t)B * S * lay.q_w < lim && (int64_t)S * lay.r_q < lim && (int64_t)S * row < lim;
}
// fused: 0 = final locations / weights (ref unused), 1 = raw + reference points [.., Lq, L, P, 2], 2 = raw + one
// reference point per (query, level) [.., Lq, L, 2]
int msda_forward_group(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *off, const float *logit, const float *ref, int64_t ref_bstride, int fused,
                       SamplingLayout lay, int B, int S, int M, int D, int L, float *out,
                       const int *local_hits = nullptr, bool standdown = false, float *stats = nullptr);
// its many-camera instantiations (msda_forward_group_many.hip); `opts`: GROUP_OPT_* of msda_group_kernel.h
int msda_forward_group_many(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                            const float *off, const float *logit, const float *ref, int64_t ref_bstride, int fused,
                            SamplingLayout lay, int B, int S, int M, int D, int L, float *out, const int *local_hits,
                            int opts, float *stats = nullptr);

// Tile count of one level, recomputed by every workgroup from the device-side shapes (uniform ->
// scalar registers).
template <typename Cfg>
__device__ __forceinline__ int tiles_of_level(const int64_t *shapes, int l)
{
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    return ((H + Cfg::TH - 1) / Cfg::TH) * ((W + Cfg::TW - 1) / Cfg::TW);
}

}  // namespace mvdetr
