// Internal dispatch between the forward kernel variants (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvdetr {

enum class MsdaFwdImpl { Gather, Tile };

// name of the kernel the last forward on this thread launched (bench / tests only; set by the launchers)
// what the code object says about a kernel (hipFuncGetAttributes): registers per lane, scratch (spill) bytes per lane,
// static LDS bytes.  Queried once per instantiation by the launchers and reported through
// mvdetr_msda_last_forward_resources().
struct KernelResources {
    int num_regs, scratch_bytes, static_lds_bytes;
};
inline KernelResources kernel_resources(const void *func)
{
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, func) != hipSuccess) return KernelResources{-1, -1, -1};
    return KernelResources{at.numRegs, (int)at.localSizeBytes, (int)at.sharedSizeBytes};
}
void msda_note_forward_kernel(const char *name, const KernelResources *res = nullptr);

// fp32 LDS-tiled encoder kernel (msda_forward_tile.hip).  Only valid when
// msda_fwd_choose_impl() returned Tile.
// local_hits: device counter written by msda_launch_locality_probe (or NULL: no probe)
int msda_forward_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *aw, int B, int S, int M, int D, int L, int Lq,
                      int P, float *out, const int *local_hits);
// false when the kernel that takes the call decides per tile, inside the launch, whether to stage windows (msda_fwd_group2)
bool msda_forward_tile_wants_probe(int S, int M, int D, int L);
inline int msda_forward_tile(hipStream_t, const double *, const int64_t *, const int64_t *,
                             const double *, const double *, int, int, int, int, int, int, int,
                             double *, const int *)
{
    return 1;  // never chosen for fp64
}

// MVDETR_MSDA_FWD_IMPL = auto (default) | gather | tile; overridable at run time through
// mvdetr_msda_set_forward_impl().
int msda_fwd_impl_knob();

// queries = tokens of levels [ql0, ql1) (0, L: all, the plain encoder call, which needs Lq == S)
bool msda_tile_supported(int B, int S, int M, int D, int L, int Lq, int P, bool aligned16, int ql0, int ql1);

// fused variant: reference points + raw offsets + raw logits (see msda_forward_tile.hip)
int msda_forward_tile_fused(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                            const float *ref, int64_t ref_bstride, const float *offsets, const float *logits,
                            int layout, int qstride_l, int qstride_w, int ql0, int ql1, int Lq, int B, int S,
                            int M, int D, int L, float *out, float *stats = nullptr);

// grad_value of encoder-shaped fp32 calls through fixed-point LDS windows (msda_backward_tile.hip)
int msda_backward_value_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits);
// the same through token-major windows (msda_backward_value_tok.hip): what msda_backward_value_tile[_fused] dispatch to
int msda_backward_value_tok(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                            const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                            float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits);
int msda_backward_value_tok_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                  const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                  const float *stats, int B, int S, int M, int D, int L, float *grad_value);
// the two halves of the fused training backward (msda_backward_tile.hip, msda_backward_fused.hip)
int msda_backward_value_tile_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                   const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                   const float *stats, int B, int S, int M, int D, int L, float *grad_value);
int msda_backward_fused_sampling(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                 const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                 const float *stats, const float *out_fwd, int B, int S, int M, int D, int L, float *grad_raw);
// device-side locality probe shared by the kernels of a call (stream-ordered scratch of MSDA_PROBE_INTS ints):
//   probe[0]              how many of MSDA_PROBE_SAMPLES sampled taps lie within MSDA_PROBE_RADIUS pixels of their own query cell
//   probe[1 + 3 m + 0..2] for head m (< MSDA_PROBE_MAXHEADS): sum of the sampled taps' x / y displacement from their own cell in
//                         1/16 px, and how many were summed (those within 16 px) -- where that head's taps lie.  MSDeformAttn's
//                         offset bias is a ray per head (ms_deform_attn.py:64-69), so windows centred on the cell lose the
//                         far points of the ray; the LDS-tiled backward kernels shift their windows by msda_probe_shift().
#define MSDA_PROBE_SAMPLES 16384
#define MSDA_PROBE_RADIUS 5.5f
// a call stands down to the generic formulation when fewer than 1 / MSDA_PROBE_NEAR_DIV of the sampled taps are near (round 4's
// noise sweep: with a half, the backward left the windows at a spread of 6 px, 3.08 ms where they take 1.93; a quarter keeps them)
#ifndef MSDA_PROBE_NEAR_DIV
#define MSDA_PROBE_NEAR_DIV 4
#endif
#define MSDA_PROBE_MAXHEADS 64
#define MSDA_PROBE_INTS (1 + 3 * MSDA_PROBE_MAXHEADS)
#define MSDA_PROBE_MAXSHIFT 3
__device__ __forceinline__ void msda_probe_shift(const int *__restrict__ probe, int head, int &sx, int &sy)
{
    sx = sy = 0;
    if (!probe || head >= MSDA_PROBE_MAXHEADS) return;
    const int n = probe[1 + 3 * head + 2];
    if (n <= 0) return;
    const float inv = 1.f / (16.f * (float)n);
    sx = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf((float)probe[1 + 3 * head] * inv)));
    sy = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf((float)probe[1 + 3 * head + 1] * inv)));
}
int msda_launch_locality_probe(hipStream_t st, const float *loc, const int64_t *shapes, int B, int S, int M, int L, int *hits);

// grad_sampling_loc / grad_attn_weight of the same calls from LDS-staged value windows (msda_backward_sampling.hip)
int msda_backward_sampling_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                                float *grad_loc, float *grad_aw, const int *local_hits);

template <typename T>
inline MsdaFwdImpl msda_fwd_choose_impl(const T *value, const T *loc, const T *aw, const T *out, int B,
                                        int S, int M, int D, int L, int Lq, int P)
{
    if (sizeof(T) != 4) return MsdaFwdImpl::Gather;
    const int env = msda_fwd_impl_knob();
    if (env == 1) return MsdaFwdImpl::Gather;
    const bool a16 = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(loc) |
                       reinterpret_cast<uintptr_t>(aw) | reinterpret_cast<uintptr_t>(out)) % 16) == 0;
    if (!msda_tile_supported(B, S, M, D, L, Lq, P, a16, 0, L)) return MsdaFwdImpl::Gather;
    return MsdaFwdImpl::Tile;
}

}  // namespace mvdetr
