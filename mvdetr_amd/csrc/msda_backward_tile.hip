// Multi-scale deformable attention backward for deformable-ENCODER calls: the locality probe of rounds 2-4's two-kernel
// path and the entry points of its grad_value half -- gfx950 (MI355X).
//
// History of the grad_value kernels (the generic backward, msda_backward.hip, sends every bilinear corner of every tap to
// memory as fp32 atomics: 3.15 ms at Wildtrack size; LDS fp32 atomics are no way out -- ds_add_f32: 80 ns per wave
// instruction per CU -- but LDS INTEGER atomics run at 1.8 / 2.7 ns for 32 / 64 bits):
//   round 1-3  msda_bwd_value_win: channel-major fixed-point windows, lanes = cells (526 -> 390 us; removed in round 5)
//   round 4    msda_bwd_value_tok (msda_backward_value_tok.hip): token-major windows, lanes = (cell, corner, channel pair);
//              still what 32-channel heads and MVDETR_MSDA_BWD_IMPL=twopass run
//   round 5    msda_bwd_onepass<DOTS = 0> (msda_backward_onepass.hip): the same accumulation, jobs sequenced level by level
//              with a guessed fixed-point scale, no probe launch -- the default for 16-channel heads
// grad_sampling_loc / grad_attn_weight come from msda_backward_sampling.hip / msda_backward_fused.hip.
//
// Replaces (with the files above) ms_deformable_col2im_cuda's grad_value accumulation
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-152,301-920).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_backward_lanes.h"
#include <stdlib.h>
#include <string.h>

namespace mvdetr {

// Samples MSDA_PROBE_SAMPLES taps spread over the whole call -- block b samples head b % M -- and counts those within
// MSDA_PROBE_RADIUS pixels of their own query's cell (equal level shapes assumed; with unequal ones the count is ignored
// anyway); per head it also sums the displacement of the sampled taps from their cells (msda_dispatch.h).
__global__ __launch_bounds__(256) void msda_locality_probe(const float *__restrict__ loc, const int64_t *__restrict__ shapes,
                                                           int B, int S, int M, int L, int *__restrict__ probe)
{
    __shared__ int red[4][3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int H = (int)shapes[0], W = (int)shapes[1];
    const int head = blockIdx.x % M;
    const int64_t per_head = (int64_t)B * S * L * TILE_P;     // taps of one head
    // a fixed odd stride walks the (query, level, point) space of the block's head evenly
    const int64_t u = (int64_t)(((unsigned long long)i * 0x9E3779B97F4A7C15ull) % (unsigned long long)per_head);
    const int64_t bq = u / ((int64_t)L * TILE_P);
    const int64_t t = (bq * M + head) * L * TILE_P + (u - bq * L * TILE_P);
    const int cell = (int)((bq % S) % ((int64_t)H * W));
    const float x = loc[2 * t] * (float)W - 0.5f, y = loc[2 * t + 1] * (float)H - 0.5f;
    const float dx = x - (float)(cell % W), dy = y - (float)(cell / W);
    const bool in = i < MSDA_PROBE_SAMPLES;
    const bool hit = in && fabsf(dx) <= MSDA_PROBE_RADIUS && fabsf(dy) <= MSDA_PROBE_RADIUS;
    const bool near = in && fabsf(dx) <= 16.f && fabsf(dy) <= 16.f;      // (NaN: false)
    int sx = near ? (int)rintf(dx * 16.f) : 0, sy = near ? (int)rintf(dy * 16.f) : 0, sn = near ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
        sn += __shfl_xor(sn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; red[threadIdx.x >> 6][2] = sn; }
    const int n = __syncthreads_count(hit);
    if (threadIdx.x == 0) {
        if (n) atomicAdd(probe, n);
        if (head < MSDA_PROBE_MAXHEADS) {
            atomicAdd(probe + 1 + 3 * head, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
            atomicAdd(probe + 1 + 3 * head + 1, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
            atomicAdd(probe + 1 + 3 * head + 2, red[0][2] + red[1][2] + red[2][2] + red[3][2]);
        }
    }
}

int msda_launch_locality_probe(hipStream_t st, const float *loc, const int64_t *shapes, int B, int S, int M, int L, int *probe)
{
    hipError_t e = hipMemsetAsync(probe, 0, MSDA_PROBE_INTS * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_locality_probe, dim3(MSDA_PROBE_SAMPLES / 256), dim3(256), 0, st, loc, shapes, B, S, M, L, probe);
    return (int)hipGetLastError();
}

int msda_backward_value_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    return msda_backward_value_tok(st, go, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw, local_hits);
}

// msda_bwd_value_tok addresses one batch element's tensors with 32-bit byte offsets: larger calls keep the generic kernel
bool msda_backward_value_tile_fits(int S, int M, int D, int L)
{
    const int64_t lim = (int64_t)1 << 32;
    return (D == 16 || D == 32) && (int64_t)S * M * L * TILE_P * 2 * 4 < lim && (int64_t)S * M * D * 4 < lim;
}

// grad_value of the fused training backward: raw offsets / logits + the forward's statistics (see the kernel's header)
int msda_backward_value_tile_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                   const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                   const float *stats, int B, int S, int M, int D, int L, float *grad_value)
{
    const int64_t lim = (int64_t)1 << 32;
    if ((int64_t)S * raw_q * 4 < lim && (int64_t)S * M * D * 4 < lim)
        return msda_backward_value_tok_fused(st, go, value, shapes, lsi, raw, raw_q, ref, ref_bstride, stats, B, S, M, D, L, grad_value);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
