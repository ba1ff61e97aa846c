// Multi-scale deformable attention, forward -- gfx950 (MI355X) kernels + C ABI.
//
// Replaces ms_deformable_im2col_cuda / ms_deformable_im2col_gpu_kernel of the reference
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:923-954, 237-299, 33-84).
// Written for wave64 / CDNA4 from the operation's definition; not derived from the CUDA source.
//
// Kernels
//   msda_fwd_gather<T, VEC>   any shape/dtype.  One lane owns VEC consecutive channels of one
//                             (b, q, head); the D/VEC lanes of a head read one contiguous
//                             D*sizeof(T) segment per bilinear corner straight from L2/HBM.
//   (the LDS-tiled encoder kernel lives in msda_forward_tile.hip)
#include "common.h"
#include "../../include/mvdetr_ops.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_gather_body.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace mvdetr {

static int impl_from_env()
{
    const char *e = getenv("MVDETR_MSDA_FWD_IMPL");
    if (e && !strcmp(e, "gather")) return 1;
    if (e && !strcmp(e, "tile")) return 2;
    return 0;
}

static std::atomic<int> g_impl_knob{-1};

int msda_fwd_impl_knob()
{
    int v = g_impl_knob.load(std::memory_order_relaxed);
    if (v < 0) {
        v = impl_from_env();
        int expect = -1;
        g_impl_knob.compare_exchange_strong(expect, v);
        v = g_impl_knob.load(std::memory_order_relaxed);
    }
    return v;
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void msda_fwd_gather(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw,
    int B, int S, int M, int D, int L, int Lq, int P, T *__restrict__ out)
{
    msda_fwd_gather_body<T, VEC>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, value,
                                 shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
}

template <typename T, int VEC>
static int launch_gather(hipStream_t st, const T *value, const int64_t *shapes, const int64_t *lsi,
                         const T *loc, const T *aw, int B, int S, int M, int D, int L, int Lq, int P,
                         T *out)
{
    const int64_t total = (int64_t)B * Lq * M * (D / VEC);
    const int block = 256;
    int64_t blocks = (total + block - 1) / block;
    // Two workgroups per CU walking the items in order: the lanes in flight at any moment then belong to one compact
    // band of queries, whose taps share L2 lines.  Measured at Wildtrack size (realistic / uniform locations):
    // one item per lane 1,112 / 1,542 us; 1024 workgroups 837 / 1,437; 512 -> 537 / 950; 256 -> 819 / 1,017.
    static const int64_t resident = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return (int64_t)2 * cus;
    }();
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL((msda_fwd_gather<T, VEC>), dim3((unsigned)blocks), dim3(block), 0, st, value,
                       shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
    return (int)hipGetLastError();
}

template <typename T>
int msda_forward_gather(hipStream_t st, const T *value, const int64_t *shapes, const int64_t *lsi,
                        const T *loc, const T *aw, int B, int S, int M, int D, int L, int Lq, int P,
                        T *out)
{
    constexpr int WIDE = 16 / (int)sizeof(T);            // 4 floats / 2 doubles per 16-byte access
    const bool a16 = aligned(value, 16) && aligned(out, 16);
    if (a16 && D % WIDE == 0)
        return launch_gather<T, WIDE>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
    return launch_gather<T, 1>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
}

template int msda_forward_gather<float>(hipStream_t, const float *, const int64_t *, const int64_t *,
                                        const float *, const float *, int, int, int, int, int, int, int,
                                        float *);
template int msda_forward_gather<double>(hipStream_t, const double *, const int64_t *, const int64_t *,
                                         const double *, const double *, int, int, int, int, int, int,
                                         int, double *);

static thread_local const char *g_last_impl = "none";
static thread_local const char *g_last_kernel = "none";
static thread_local KernelResources g_last_resources = {-1, -1, -1};
void msda_note_forward_kernel(const char *name, const KernelResources *res)
{
    g_last_kernel = name;
    g_last_resources = res ? *res : KernelResources{-1, -1, -1};
}

static bool bad_dims(int B, int S, int M, int D, int L, int Lq, int P)
{
    return B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0;
}

template <typename T>
static int forward_entry(void *stream, const T *value, const int64_t *shapes, const int64_t *lsi,
                         const T *loc, const T *aw, int B, int S, int M, int D, int L, int Lq, int P,
                         T *out)
{
    if (bad_dims(B, S, M, D, L, Lq, P)) return (int)hipErrorInvalidValue;
    if ((int64_t)B * Lq == 0) { g_last_impl = "empty"; return 0; }     // nothing to write
    if (!value || !shapes || !lsi || !loc || !aw || !out) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const MsdaFwdImpl impl = msda_fwd_choose_impl<T>(value, loc, aw, out, B, S, M, D, L, Lq, P);
    if (impl == MsdaFwdImpl::Tile) {
        g_last_impl = "tile";
        if constexpr (sizeof(T) == 4) {
            // `auto`: a device-side probe (stream-ordered scratch int, no host sync) tells the tile kernel whether the
            // sampling locations are near the queries' cells; if not, the same launch runs the gather formulation.
            // A forced `tile` skips the probe (the kernel is then measured as it is).
            // (6 / 7 equal levels: msda_fwd_group2 looks at every tile's own taps and stands down job by job -- no probe
            // kernel, no scratch allocation in front of the forward)
            int *hits = nullptr;
            if (msda_fwd_impl_knob() == 0 && msda_forward_tile_wants_probe(B, S, M, D, L) &&
                hipMallocAsync(reinterpret_cast<void **>(&hits), MSDA_PROBE_INTS * sizeof(int), st) != hipSuccess)
                hits = nullptr;
            int rc = hits ? msda_launch_locality_probe(st, loc, shapes, B, S, M, L, hits) : 0;
            if (!rc) rc = msda_forward_tile(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out, hits);
            if (hits) (void)hipFreeAsync(hits, st);
            return rc;
        } else {
            return msda_forward_tile(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out, nullptr);
        }
    }
    g_last_impl = "gather";
    msda_note_forward_kernel("msda_fwd_gather");
    return msda_forward_gather<T>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
}

}  // namespace mvdetr

extern "C" {

int mvdetr_ops_abi_version(void) { return MVDETR_OPS_ABI_VERSION; }

const char *mvdetr_msda_last_forward_impl(void) { return mvdetr::g_last_impl; }
int mvdetr_msda_last_forward_resources(int *num_regs, int *scratch_bytes_per_lane, int *static_lds_bytes)
{
    const mvdetr::KernelResources r = mvdetr::g_last_resources;
    if (num_regs) *num_regs = r.num_regs;
    if (scratch_bytes_per_lane) *scratch_bytes_per_lane = r.scratch_bytes;
    if (static_lds_bytes) *static_lds_bytes = r.static_lds_bytes;
    return r.num_regs >= 0;
}
const char *mvdetr_msda_last_forward_kernel(void) { return mvdetr::g_last_kernel; }

int mvdetr_msda_set_forward_impl(int impl)
{
    if (impl < 0 || impl > 2) impl = 0;
    const int prev = mvdetr::msda_fwd_impl_knob();
    mvdetr::g_impl_knob.store(impl);
    return prev;
}

int mvdetr_msda_fused_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                int num_query, int num_point)
{
    return mvdetr_msda_fused_levels_supported(batch, spatial_size, num_heads, channels, num_levels, num_query,
                                              num_point, 0, num_levels);
}

int mvdetr_msda_fused_levels_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                       int num_query, int num_point, int query_level_begin, int query_level_end)
{
    return mvdetr::msda_tile_supported(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, true,
                                       query_level_begin, query_level_end) ? 1 : 0;
}

int mvdetr_msda_forward_fused_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const float *reference_points,
                                  int64_t ref_batch_stride, const float *sampling_offsets,
                                  const float *attn_logits, int level_major, int offsets_query_stride,
                                  int logits_query_stride, int batch, int spatial_size, int num_heads,
                                  int channels, int num_levels, int num_query, int num_point, float *out)
{
    return mvdetr_msda_forward_fused_levels_f32(stream, value, spatial_shapes, level_start_index, reference_points,
                                                ref_batch_stride, sampling_offsets, attn_logits, level_major,
                                                offsets_query_stride, logits_query_stride, 0, num_levels, batch,
                                                spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                                out);
}

static int fused_entry(void *stream, const float *value, const int64_t *spatial_shapes,
                       const int64_t *level_start_index, const float *reference_points,
                       int64_t ref_batch_stride, const float *sampling_offsets,
                       const float *attn_logits, int level_major, int offsets_query_stride,
                       int logits_query_stride, int query_level_begin, int query_level_end,
                       int batch, int spatial_size, int num_heads, int channels, int num_levels,
                       int num_query, int num_point, float *out, float *stats)
{
    using namespace mvdetr;
    const int dense_l = num_heads * num_levels * num_point * 2, dense_w = num_heads * num_levels * num_point;
    if (offsets_query_stride == 0) offsets_query_stride = dense_l;
    if (logits_query_stride == 0) logits_query_stride = dense_w;
    if (offsets_query_stride < dense_l || logits_query_stride < dense_w || offsets_query_stride % 4 || logits_query_stride % 4)
        return (int)hipErrorInvalidValue;
    if (bad_dims(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
        return (int)hipErrorInvalidValue;
    if (query_level_begin < 0 || query_level_end <= query_level_begin || query_level_end > num_levels)
        return (int)hipErrorInvalidValue;
    // bit 0 level-major raw tensors, bit 1 one reference point per (query, level), bit 2 slice-interleaved raw tensor
    // (offsets and logits in one run per (query, slice, level); excludes bit 0), bit 3 level-major reference points
    // [.., L, Lq, 2] (needs bit 1), bit 4 the slice-interleaved tensor with the LEVEL outermost, [.., Lq, L, M/g, run]
    // (needs bit 2)
    if ((level_major & ~31) || ((level_major & 4) && (level_major & 1)) || ((level_major & 8) && !(level_major & 2)) ||
        ((level_major & 16) && !(level_major & 4)))
        return (int)hipErrorInvalidValue;
    if (level_major & 4) {
        // the two pointers address one tensor: logits start behind the slice's offsets
        const int hps = channels == 16 ? 2 : 1;
        if ((channels != 16 && channels != 32) || attn_logits != sampling_offsets + hps * num_point * 2 ||
            offsets_query_stride != logits_query_stride || offsets_query_stride < dense_l + dense_w)
            return (int)hipErrorInvalidValue;
    }
    if (!value || !spatial_shapes || !level_start_index || !reference_points || !sampling_offsets || !attn_logits || !out)
        return (int)hipErrorInvalidValue;
    const bool a16 = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(reference_points) |
                       reinterpret_cast<uintptr_t>(sampling_offsets) | reinterpret_cast<uintptr_t>(attn_logits) |
                       reinterpret_cast<uintptr_t>(out)) % 16) == 0 && ref_batch_stride % ((level_major & 2) ? 2 : 4) == 0;
    if (!msda_tile_supported(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, a16,
                             query_level_begin, query_level_end))
        return (int)hipErrorNotSupported;
    g_last_impl = "tile_fused";
    return msda_forward_tile_fused(reinterpret_cast<hipStream_t>(stream), value, spatial_shapes, level_start_index,
                                   reference_points, ref_batch_stride, sampling_offsets, attn_logits,
                                   level_major, offsets_query_stride, logits_query_stride, query_level_begin,
                                   query_level_end, num_query, batch, spatial_size, num_heads, channels, num_levels,
                                   out, stats);
}

int mvdetr_msda_forward_fused_levels_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                         const int64_t *level_start_index, const float *reference_points,
                                         int64_t ref_batch_stride, const float *sampling_offsets,
                                         const float *attn_logits, int level_major, int offsets_query_stride,
                                         int logits_query_stride, int query_level_begin, int query_level_end,
                                         int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                         int num_query, int num_point, float *out)
{
    return fused_entry(stream, value, spatial_shapes, level_start_index, reference_points, ref_batch_stride, sampling_offsets,
                       attn_logits, level_major, offsets_query_stride, logits_query_stride, query_level_begin, query_level_end,
                       batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, out, nullptr);
}

int mvdetr_msda_fused_train_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                      int num_point)
{
    using namespace mvdetr;
    // every deformable-encoder shape the LDS-tiled kernels take: up to 16 levels (of equal shape: the caller's promise), 16- or
    // 32-channel heads, 4 points, queries = tokens.  6 / 7 levels of 16-channel heads (MVDeTr's own) run msda_fwd_group2 +
    // msda_bwd_onepass<grad_value only> + msda_bwd_fused_sampling; other level counts the one-pass backward (16 channels) or
    // msda_bwd_value_tok + the level-groups sampling kernel (32 channels) behind the inference forward + a statistics pass.
    if (!msda_tile_supported(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, true, 0, num_levels)) return 0;
    // (the whole raw tensor, all batch elements, in 32-bit float offsets: msda_group_fits; one element's in 2^29 for the backward)
    if ((int64_t)spatial_size * num_heads * num_levels * num_point * 3 >= ((int64_t)1 << 29)) return 0;
    return (int64_t)batch * spatial_size * num_heads * num_levels * num_point * 3 < ((int64_t)1 << 30) ? 1 : 0;
}

int mvdetr_msda_forward_fused_train_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                        const int64_t *level_start_index, const float *reference_points,
                                        int64_t ref_batch_stride, const float *raw, int raw_query_stride, int batch,
                                        int spatial_size, int num_heads, int channels, int num_levels, int num_point,
                                        float *out, float *stats)
{
    if (!stats || !raw) return (int)hipErrorInvalidValue;
    if (!mvdetr_msda_fused_train_supported(batch, spatial_size, num_heads, channels, num_levels, spatial_size, num_point))
        return (int)hipErrorNotSupported;
    const int hps = channels == 16 ? 2 : 1;
    // layout: the slice-interleaved raw tensor with the level outermost (bits 2 and 4), one reference point per
    // (query, level), level-major (bits 1 and 3)
    return fused_entry(stream, value, spatial_shapes, level_start_index, reference_points, ref_batch_stride, raw,
                       raw + hps * num_point * 2, 2 | 4 | 8 | 16, raw_query_stride, raw_query_stride, 0, num_levels, batch,
                       spatial_size, num_heads, channels, num_levels, spatial_size, num_point, out, stats);
}

int mvdetr_msda_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *sampling_loc,
                            const float *attn_weight, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, float *out)
{
    return mvdetr::forward_entry<float>(stream, value, spatial_shapes, level_start_index, sampling_loc,
                                        attn_weight, batch, spatial_size, num_heads, channels,
                                        num_levels, num_query, num_point, out);
}

int mvdetr_msda_forward_f64(void *stream, const double *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const double *sampling_loc,
                            const double *attn_weight, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, double *out)
{
    return mvdetr::forward_entry<double>(stream, value, spatial_shapes, level_start_index, sampling_loc,
                                         attn_weight, batch, spatial_size, num_heads, channels,
                                         num_levels, num_query, num_point, out);
}

}  // extern "C"
