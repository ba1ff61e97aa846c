// Multi-scale deformable attention, backward -- gfx950 (MI355X) kernels + C ABI.
//
// Replaces ms_deformable_col2im_cuda and its six kernel variants
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:956-1327, 301-920) and the
// device helpers ms_deform_attn_col2im_bilinear{,_gm} (cuh:87-234).
//
// The reference picks among shared-memory reductions by block size because its block IS the D
// channels of one (b,q,head) (16 threads at MVDeTr's D=16: a quarter of a wave64).  Here:
//   msda_bwd_lanes<T, VEC, G>  G = D/VEC lanes (power of two <= 64; VEC = 1 by default) own one (b,q,head); the
//                              per-tap partial sums for grad_sampling_loc / grad_attn_weight are
//                              reduced across those G lanes with DPP/xor shuffles inside the wave
//                              -- no LDS, no barrier -- and lane 0 of the group stores them.
//   msda_bwd_serial<T>         any D: one lane per (b,q,head) walks the channels itself.
// grad_value is accumulated with hardware fp atomics (global_atomic_add_f32/f64,
// -munsafe-fp-atomics), i.e. the summation order -- like the reference's atomicAdd
// (cuh:125-152) -- is not deterministic.
// Encoder-shaped fp32 calls do not end up here: backward_entry() sends them to msda_backward_tile.hip
// (grad_value, fixed-point LDS windows) + msda_backward_sampling.hip (the other two gradients).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_backward_lanes.h"
#include "../../include/mvdetr_ops.h"
#include <stdlib.h>
#include <string.h>

#include <atomic>

namespace mvdetr {

// deterministic backward requested (mvdetr_msda_set_backward_deterministic; MVDETR_MSDA_BWD_DETERMINISTIC=1 sets the initial state)
static std::atomic<int> &backward_deterministic()
{
    static std::atomic<int> on{[] { const char *e = getenv("MVDETR_MSDA_BWD_DETERMINISTIC"); return e && *e && strcmp(e, "0") ? 1 : 0; }()};
    return on;
}

template <typename T, int VEC, int G, bool VALUE_GRAD = true>
__global__ __launch_bounds__(256) void msda_bwd_lanes(
    const T *__restrict__ grad_col, const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw, int B, int S,
    int M, int D, int L, int Lq, int P, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_aw)
{
    msda_bwd_lanes_body<T, VEC, G, VALUE_GRAD>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, grad_col, value, shapes, lsi,
                                               loc, aw, B, S, M, D, L, Lq, P, grad_value, grad_loc, grad_aw);
}

template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_serial(
    const T *__restrict__ grad_col, const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw, int B, int S,
    int M, int D, int L, int Lq, int P, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_aw)
{
    const int64_t total = (int64_t)B * Lq * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t row = (int64_t)M * D;
    for (int64_t bqm = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; bqm < total; bqm += stride) {
        const int m = (int)(bqm % M);
        const int64_t bq = bqm / M;
        const int b = (int)(bq / Lq);
        const T *go = grad_col + bqm * D;
        const int64_t voff = (int64_t)b * S * row + (int64_t)m * D;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t poff = voff + lsi[l] * row;
            for (int p = 0; p < P; ++p) {
                const int64_t t = bqm * L * P + l * P + p;
                const T x = loc[t * 2 + 0] * T(W) - T(0.5);
                const T y = loc[t * 2 + 1] * T(H) - T(0.5);
                const T a = aw[t];
                T g_a = 0, g_x = 0, g_y = 0;
                if (y > T(-1) && x > T(-1) && y < T(H) && x < T(W)) {
                    const Footprint<T> f = footprint(y, x, H, W);
                    const int64_t o00 = poff + ((int64_t)f.y0 * W + f.x0) * row;
                    const int64_t o01 = o00 + row, o10 = o00 + (int64_t)W * row, o11 = o10 + row;
                    const bool v00 = f.vy0 && f.vx0, v01 = f.vy0 && f.vx1;
                    const bool v10 = f.vy1 && f.vx0, v11 = f.vy1 && f.vx1;
                    const T w00 = f.wy0 * f.wx0, w01 = f.wy0 * f.wx1, w10 = f.wy1 * f.wx0, w11 = f.wy1 * f.wx1;
                    for (int c = 0; c < D; ++c) {
                        const T g = go[c], ga = g * a;
                        const T c00 = v00 ? value[o00 + c] : T(0), c01 = v01 ? value[o01 + c] : T(0);
                        const T c10 = v10 ? value[o10 + c] : T(0), c11 = v11 ? value[o11 + c] : T(0);
                        g_a += g * (w00 * c00 + w01 * c01 + w10 * c10 + w11 * c11);
                        g_x += g * ((c01 - c00) * f.wy0 + (c11 - c10) * f.wy1);
                        g_y += g * ((c10 - c00) * f.wx0 + (c11 - c01) * f.wx1);
                        if (v00) atomic_add(grad_value + o00 + c, w00 * ga);
                        if (v01) atomic_add(grad_value + o01 + c, w01 * ga);
                        if (v10) atomic_add(grad_value + o10 + c, w10 * ga);
                        if (v11) atomic_add(grad_value + o11 + c, w11 * ga);
                    }
                }
                grad_aw[t] = g_a;
                grad_loc[t * 2 + 0] = T(W) * a * g_x;
                grad_loc[t * 2 + 1] = T(H) * a * g_y;
            }
        }
    }
}

#define MSDA_BWD_ARGS grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, grad_value, grad_loc, grad_aw

template <typename T, int VEC, int G, bool VALUE_GRAD = true>
static int launch_lanes(hipStream_t st, const T *grad_col, const T *value, const int64_t *shapes,
                        const int64_t *lsi, const T *loc, const T *aw, int B, int S, int M, int D, int L,
                        int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw)
{
    const int64_t total = (int64_t)B * Lq * M * G;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((msda_bwd_lanes<T, VEC, G, VALUE_GRAD>), dim3((unsigned)blocks), dim3(256), 0, st, MSDA_BWD_ARGS);
    return (int)hipGetLastError();
}

template <typename T>
static int backward_entry(void *stream, const T *grad_col, const T *value, const int64_t *shapes,
                          const int64_t *lsi, const T *loc, const T *aw, int B, int S, int M, int D, int L,
                          int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw)
{
    if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return (int)hipErrorInvalidValue;
    if ((int64_t)B * Lq == 0) return 0;
    if (!grad_col || !value || !shapes || !lsi || !loc || !aw || !grad_value || !grad_loc || !grad_aw)
        return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int WIDE = 16 / (int)sizeof(T);
    const bool a16 = aligned(value, 16) && aligned(grad_col, 16);
    if (sizeof(T) != 4 && backward_deterministic().load(std::memory_order_relaxed)) return (int)hipErrorNotSupported;
    if constexpr (sizeof(T) == 4) {
        // Encoder-shaped fp32 calls (the shapes the forward tile kernels take): grad_value through fixed-point LDS
        // windows (msda_backward_tile.hip), the other two gradients from LDS-staged value windows (msda_backward_sampling.hip).
        // MVDETR_MSDA_BWD_IMPL=atomic keeps everything on the direct-atomics kernel.
        static const bool tile_ok = [] { const char *e = getenv("MVDETR_MSDA_BWD_IMPL"); return !(e && !strcmp(e, "atomic")); }();
        const bool all16 = a16 && aligned(loc, 16) && aligned(aw, 16) && aligned(grad_value, 16) && aligned(grad_loc, 16) &&
                           aligned(grad_aw, 16);
        // MVDETR_MSDA_BWD_IMPL = twopass (default of this entry: msda_locality_probe + msda_bwd_value_tok + msda_bwd_sampling_*) |
        // split (grad_value from msda_bwd_onepass<DOTS = 0>, no probe launch; the default of the FUSED entry below) | onepass (all
        // three gradients from ONE kernel, msda_backward_onepass.hip) | atomic (the generic kernel).
        // The default is the measured one (profiles/r06_bwd_ab.txt, same box, realistic / uniform input): Wildtrack 625 / 3,383 us
        // twopass, 642 / 3,607 split, 671 / 3,960 onepass; MultiviewX 458 / 2,316, 458 / 2,360, 506 / 2,527 -- round 5 had made
        // `split` the default to save the 6-us probe launch; it moves 2.4 x the bytes and is not faster.
        static const int impl = [] {
            const char *e = getenv("MVDETR_MSDA_BWD_IMPL");
            return !e ? 2 : !strcmp(e, "onepass") ? 1 : !strcmp(e, "split") ? 0 : 2;
        }();
        const bool tile_shapes = tile_ok && msda_tile_supported(B, S, M, D, L, Lq, P, all16, 0, L);
        const bool op_ok = tile_shapes && msda_backward_onepass_supported(B, S, M, D, L, (int64_t)M * L * P * 2);
        if (backward_deterministic().load(std::memory_order_relaxed)) {
            // one kernel does it (msda_bwd_onepass<DET>); calls it does not take are refused, not served by a kernel that is not
            if (!tile_shapes || !msda_backward_deterministic_supported(B, S, M, D, L, (int64_t)M * L * P * 2)) return (int)hipErrorNotSupported;
            return msda_backward_onepass_det(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw);
        }
        if (op_ok && impl == 1)
            return msda_backward_onepass(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw, true);
        if (op_ok && impl == 0) {
            // no probe, no scratch: both kernels take the window shift and the stand-down decision from their jobs' own samples
            int rc = msda_backward_scatter(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw, true);
            if (!rc) rc = msda_backward_sampling_tile(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_loc, grad_aw, nullptr);
            return rc;
        }
        if (tile_shapes && msda_backward_value_tile_fits(S, M, D, L)) {
            // rounds 2-4's pair (and 32-channel heads): a stream-ordered scratch int carries the locality probe's verdict to both
            // kernels: calls whose taps are far from their queries (e.g. uniformly random locations) run the lane-group
            // backward inside the first launch instead -- no host synchronisation
            // (no probe means something else to the sampling kernels -- "take your own sample and skip the tiles the grad_value
            // kernel has taken along" -- and this path's grad_value kernel never takes a tile along: an allocation failure is
            // returned, not papered over)
            int *hits = nullptr;
            const hipError_t arc = hipMallocAsync(reinterpret_cast<void **>(&hits), MSDA_PROBE_INTS * sizeof(int), st);
            if (arc != hipSuccess || !hits) return (int)(arc != hipSuccess ? arc : hipErrorOutOfMemory);
            int rc = msda_launch_locality_probe(st, loc, shapes, B, S, M, L, hits);
            if (!rc) rc = msda_backward_value_tile(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw, hits);
            if (!rc) rc = msda_backward_sampling_tile(st, grad_col, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_loc, grad_aw, hits);
            (void)hipFreeAsync(hits, st);
            return rc;
        }
    }
    // One channel per lane (G = D lanes per head): a wave's atomic instruction then covers whole
    // 4*D-byte head segments, which the memory-side atomic units take as ONE request each, instead of four
    // partial ones with 16-byte-per-lane vectors (measured at Wildtrack size: 3.15 ms vs 12.7 ms -- the
    // kernel is bound by atomic requests, ~21 G/s, not by bytes).
    if (D <= 64 && (D & (D - 1)) == 0) {
        switch (D) {
        case 1: return launch_lanes<T, 1, 1>(st, MSDA_BWD_ARGS);
        case 2: return launch_lanes<T, 1, 2>(st, MSDA_BWD_ARGS);
        case 4: return launch_lanes<T, 1, 4>(st, MSDA_BWD_ARGS);
        case 8: return launch_lanes<T, 1, 8>(st, MSDA_BWD_ARGS);
        case 16: return launch_lanes<T, 1, 16>(st, MSDA_BWD_ARGS);
        case 32: return launch_lanes<T, 1, 32>(st, MSDA_BWD_ARGS);
        case 64: return launch_lanes<T, 1, 64>(st, MSDA_BWD_ARGS);
        default: break;
        }
    }
    if (a16 && D % WIDE == 0) {
        switch (D / WIDE) {
        case 1: return launch_lanes<T, WIDE, 1>(st, MSDA_BWD_ARGS);
        case 2: return launch_lanes<T, WIDE, 2>(st, MSDA_BWD_ARGS);
        case 4: return launch_lanes<T, WIDE, 4>(st, MSDA_BWD_ARGS);
        case 8: return launch_lanes<T, WIDE, 8>(st, MSDA_BWD_ARGS);
        case 16: return launch_lanes<T, WIDE, 16>(st, MSDA_BWD_ARGS);
        case 32: return launch_lanes<T, WIDE, 32>(st, MSDA_BWD_ARGS);
        case 64: return launch_lanes<T, WIDE, 64>(st, MSDA_BWD_ARGS);
        default: break;
        }
    }
    const int64_t total = (int64_t)B * Lq * M;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL((msda_bwd_serial<T>), dim3((unsigned)blocks), dim3(256), 0, st, MSDA_BWD_ARGS);
    return (int)hipGetLastError();
}

}  // namespace mvdetr

extern "C" {

int mvdetr_msda_backward_f32(void *stream, const float *grad_col, const float *value,
                             const int64_t *spatial_shapes, const int64_t *level_start_index,
                             const float *sampling_loc, const float *attn_weight, int batch,
                             int spatial_size, int num_heads, int channels, int num_levels,
                             int num_query, int num_point, float *grad_value,
                             float *grad_sampling_loc, float *grad_attn_weight)
{
    return mvdetr::backward_entry<float>(stream, grad_col, value, spatial_shapes, level_start_index,
                                         sampling_loc, attn_weight, batch, spatial_size, num_heads,
                                         channels, num_levels, num_query, num_point, grad_value,
                                         grad_sampling_loc, grad_attn_weight);
}

int mvdetr_msda_set_backward_deterministic(int on)
{
    return mvdetr::backward_deterministic().exchange(on ? 1 : 0);
}

int mvdetr_msda_get_backward_deterministic(void) { return mvdetr::backward_deterministic().load(std::memory_order_relaxed); }

int mvdetr_msda_release_scratch(void) { return mvdetr::msda_release_det_scratch(); }

int mvdetr_msda_backward_fused_f32(void *stream, const float *grad_output, const float *value,
                                   const int64_t *spatial_shapes, const int64_t *level_start_index,
                                   const float *reference_points, int64_t ref_batch_stride, const float *raw,
                                   int raw_query_stride, const float *stats, const float *out, int batch, int spatial_size,
                                   int num_heads, int channels, int num_levels, int num_point, float *grad_value,
                                   float *grad_raw)
{
    using namespace mvdetr;
    if (batch < 0 || spatial_size < 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_point <= 0)
        return (int)hipErrorInvalidValue;
    if ((int64_t)batch * spatial_size == 0) return 0;
    if (!grad_output || !value || !spatial_shapes || !level_start_index || !reference_points || !raw || !stats || !out ||
        !grad_value || !grad_raw)
        return (int)hipErrorInvalidValue;
    if (!mvdetr_msda_fused_train_supported(batch, spatial_size, num_heads, channels, num_levels, spatial_size, num_point))
        return (int)hipErrorNotSupported;
    if (raw_query_stride < num_heads * num_levels * num_point * 3 || raw_query_stride % 4) return (int)hipErrorInvalidValue;
    // the kernels address one batch element's raw tensor (and its gradient) with 32-bit offsets of the CALLER's query stride,
    // which may be wider than the dense width mvdetr_msda_fused_train_supported() bounds (a column block of a wider GEMM)
    if ((int64_t)spatial_size * raw_query_stride >= ((int64_t)1 << 29)) return (int)hipErrorNotSupported;
    const uintptr_t al = reinterpret_cast<uintptr_t>(grad_output) | reinterpret_cast<uintptr_t>(value) |
                         reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(out) |
                         reinterpret_cast<uintptr_t>(grad_value) | reinterpret_cast<uintptr_t>(grad_raw);
    if ((al & 15) || (reinterpret_cast<uintptr_t>(reference_points) & 7) || (reinterpret_cast<uintptr_t>(stats) & 7) ||
        (ref_batch_stride & 1))
        return (int)hipErrorNotSupported;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // MVDETR_MSDA_BWD_IMPL for this entry: split (default: msda_bwd_onepass<fused, DOTS = 0> + msda_bwd_fused_sampling) | twopass
    // (msda_bwd_value_tok<fused> for grad_value) | onepass.  Measured (profiles/r06_bwd_ab.txt): Wildtrack 549 us split, 564
    // twopass, 706 onepass; MultiviewX 386 / 404 / 493.
    static const int impl = [] {
        const char *e = getenv("MVDETR_MSDA_BWD_IMPL");
        return !e ? 0 : !strcmp(e, "onepass") ? 1 : !strcmp(e, "twopass") ? 2 : 0;
    }();
    const bool op_ok = msda_backward_onepass_supported(batch, spatial_size, num_heads, channels, num_levels, raw_query_stride);
    const bool det = backward_deterministic().load(std::memory_order_relaxed) != 0;
    if (!det && !(channels == 16 && (num_levels == 6 || num_levels == 7))) {
        // every other encoder shape (ABI 13): 16-channel heads -> the one-pass kernel (any level count); 32-channel heads ->
        // msda_bwd_value_tok<32, fused> + the level-groups sampling kernel on the raw tensor
        if (channels == 16 && op_ok)
            return msda_backward_onepass_fused(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                               reference_points, ref_batch_stride, stats, out, batch, spatial_size, num_heads,
                                               channels, num_levels, grad_value, grad_raw);
        int rc = msda_backward_value_tile_fused(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                                reference_points, ref_batch_stride, stats, batch, spatial_size, num_heads, channels,
                                                num_levels, grad_value);
        if (rc) return rc;
        return msda_backward_fused_sampling_groups(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                                   reference_points, ref_batch_stride, stats, out, batch, spatial_size, num_heads,
                                                   channels, num_levels, grad_raw);
    }
    if (det) {
        if (!msda_backward_deterministic_supported(batch, spatial_size, num_heads, channels, num_levels, raw_query_stride))
            return (int)hipErrorNotSupported;
        return msda_backward_onepass_fused_det(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                               reference_points, ref_batch_stride, stats, out, batch, spatial_size, num_heads,
                                               channels, num_levels, grad_value, grad_raw);
    }
    if (op_ok && impl == 1)
        return msda_backward_onepass_fused(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                           reference_points, ref_batch_stride, stats, out, batch, spatial_size, num_heads,
                                           channels, num_levels, grad_value, grad_raw);
    int rc = op_ok && impl == 0
                 ? msda_backward_scatter_fused(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                               reference_points, ref_batch_stride, stats, batch, spatial_size, num_heads, channels,
                                               num_levels, grad_value)
                 : msda_backward_value_tile_fused(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                                  reference_points, ref_batch_stride, stats, batch, spatial_size, num_heads,
                                                  channels, num_levels, grad_value);
    if (rc) return rc;
    return msda_backward_fused_sampling(st, grad_output, value, spatial_shapes, level_start_index, raw, raw_query_stride,
                                        reference_points, ref_batch_stride, stats, out, batch, spatial_size, num_heads,
                                        channels, num_levels, grad_raw);
}

int mvdetr_msda_backward_f64(void *stream, const double *grad_col, const double *value,
                             const int64_t *spatial_shapes, const int64_t *level_start_index,
                             const double *sampling_loc, const double *attn_weight, int batch,
                             int spatial_size, int num_heads, int channels, int num_levels,
                             int num_query, int num_point, double *grad_value,
                             double *grad_sampling_loc, double *grad_attn_weight)
{
    return mvdetr::backward_entry<double>(stream, grad_col, value, spatial_shapes, level_start_index,
                                          sampling_loc, attn_weight, batch, spatial_size, num_heads,
                                          channels, num_levels, num_query, num_point, grad_value,
                                          grad_sampling_loc, grad_attn_weight);
}

}  // extern "C"
