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
bool msda_forward_tile_wants_probe(int B, int S, int M, int D, int L);
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
bool msda_backward_value_tile_fits(int S, int M, int D, int L);
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
// the sampling half of the fused training backward for every other encoder shape (32-channel heads, other level counts): the
// level-groups kernel of msda_backward_sampling.hip on the raw tensor
int msda_backward_fused_sampling_groups(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                        const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                        const float *stats, const float *out_fwd, int B, int S, int M, int D, int L, float *grad_raw);
// the whole encoder-shaped fp32 backward in ONE kernel (msda_backward_onepass.hip): 16-channel heads; no probe, no scratch
bool msda_backward_onepass_supported(int B, int S, int M, int D, int L, int64_t q_floats);
int msda_backward_onepass(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                          float *grad_loc, float *grad_aw, bool standdown);
int msda_backward_onepass_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                const float *stats, const float *out_fwd, int B, int S, int M, int D, int L,
                                float *grad_value, float *grad_raw);
// the same kernel with grad_value summed in 64-bit fixed point (one binary point per call): bit-reproducible run to run.  Opt-in
// (mvdetr_msda_set_backward_deterministic / MVDETR_MSDA_BWD_DETERMINISTIC=1); scratch of 8 bytes per value element, cached per (device, stream)
bool msda_backward_deterministic_supported(int B, int S, int M, int D, int L, int64_t q_floats);
int msda_backward_onepass_det(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                              const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                              float *grad_loc, float *grad_aw);
int msda_release_det_scratch();
int msda_backward_onepass_fused_det(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                    const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                    const float *stats, const float *out_fwd, int B, int S, int M, int D, int L,
                                    float *grad_value, float *grad_raw);
// the grad_value half of the same kernel alone (no value window, no dot products): what the two-kernel backward launches for
// grad_value since round 5 (units of L level jobs, guessed fixed-point scale, in-kernel window shift and stand-down)
int msda_backward_scatter(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                          float *grad_loc, float *grad_aw, bool standdown);
int msda_backward_scatter_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                const float *stats, int B, int S, int M, int D, int L, float *grad_value);
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
// The same two answers WITHOUT a probe launch (round 5): every wave of a workgroup reduces ONE sample of the job -- 64 lanes'
// worth of (cell, 4 points) locations of camera 0, the job's head and one level, the same sample in every wave -- to the mean
// tap displacement (the window shift) and to whether fewer than 1 / MSDA_PROBE_NEAR_DIV of the sampled taps lie within
// MSDA_PROBE_RADIUS pixels of their own cell (`far`: the job computes its taps in the lane-group formulation instead of
// staging windows).  All waves see the same data and do the same arithmetic: the results are workgroup-uniform without LDS
// or a barrier.  la / lb: the lane's four normalised (x, y) locations; have: the lane holds a cell of the map.
// The tile whose sample decides: the grad_value kernel's (msda_backward_onepass.hip) -- the sampling kernels' 4 x 8 jobs are
// nested in it and repeat ITS sample (first 64 cells, camera 0, level 0), so that both kernels of a backward agree on which
// (tile, head) stand down: the grad_value kernel then computes all three gradients of such a tile in the lane-group
// formulation and the sampling kernel skips it.
#define MSDA_SAMPLE_TH 4
#define MSDA_SAMPLE_TW 16
__device__ __forceinline__ void msda_job_sample(const float4 &la, const float4 &lb, bool have, int qx, int qy, float fW, float fH,
                                                int &shx, int &shy, bool &far)
{
    float sx = 0.f, sy = 0.f, sn = 0.f, snear = 0.f, scnt = 0.f;
    if (have) {
        const float mx = 0.25f * ((la.x + la.z) + (lb.x + lb.z)) * fW - 0.5f - (float)qx;
        const float my = 0.25f * ((la.y + la.w) + (lb.y + lb.w)) * fH - 0.5f - (float)qy;
        if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
        const float ox_ = (float)qx + 0.5f, oy_ = (float)qy + 0.5f, rr = MSDA_PROBE_RADIUS;
        snear = (float)((fabsf(la.x * fW - ox_) < rr && fabsf(la.y * fH - oy_) < rr) + (fabsf(la.z * fW - ox_) < rr && fabsf(la.w * fH - oy_) < rr) +
                        (fabsf(lb.x * fW - ox_) < rr && fabsf(lb.y * fH - oy_) < rr) + (fabsf(lb.z * fW - ox_) < rr && fabsf(lb.w * fH - oy_) < rr));
        scnt = 4.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
        sn += __shfl_xor(sn, o, 64);
        snear += __shfl_xor(snear, o, 64);
        scnt += __shfl_xor(scnt, o, 64);
    }
    const float tx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
    const float ty = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
    const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
    const float tl = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, snear)));
    const float tc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scnt)));
    shx = shy = 0;
    if (tn > 0.f) {
        shx = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf(tx / tn)));
        shy = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf(ty / tn)));
    }
    far = tl * (float)MSDA_PROBE_NEAR_DIV < tc;
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
