/* mvdetr_ops.h -- C ABI of libmvdetr_ops.so: the MI355X (gfx950) kernels of MVDeTr's multiview
 * ground-plane fusion path.
 *
 * This is the drop-in boundary.  Each entry point replaces one host launcher of the reference's
 * CUDA extension (paths relative to the reference checkout), keeps that launcher's argument
 * order and meaning, and differs from it only in returning the HIP error code instead of
 * printf-ing it (ms_deform_im2col_cuda.cuh:948-952, 1321-1325 swallow launch failures).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (including spatial_shapes / level_start_index, which the
 *     reference also reads on the device: ms_deform_attn_cuda.cu:67-68); no host copies, no
 *     synchronisation, no allocation: work is enqueued on `stream` and the call returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - tensors are dense row-major with the shapes given below; fp32 (`_f32`) or fp64 (`_f64`),
 *     like AT_DISPATCH_FLOATING_TYPES in ms_deform_attn_cuda.cu:64,134.
 *   - return value: 0 (hipSuccess) or a hipError_t; 1 (hipErrorInvalidValue) for bad arguments.
 *   - thread-safe and re-entrant: the only global state is the forward-variant knob below
 *     (an atomic int, initialised from the environment variable MVDETR_MSDA_FWD_IMPL).
 */
#ifndef MVDETR_OPS_H
#define MVDETR_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVDETR_OPS_ABI_VERSION 14   /* 12: + mvdetr_warp_perspective_backward_tagged_*; 13: + mvdetr_msda_set_backward_deterministic; the fused training pair takes every encoder shape; 14: + mvdetr_msda_get_backward_deterministic */

/* ABI version of the loaded library (checked by the Python loader). */
int mvdetr_ops_abi_version(void);

/* ---- Multi-scale deformable attention, forward ------------------------------------------------
 * Replaces ms_deformable_im2col_cuda (ms_deform_im2col_cuda.cuh:923-954) and the kernel it
 * launches (cuh:237-299, device helper cuh:33-84).
 *   value            [batch, spatial_size, num_heads, channels]
 *   spatial_shapes   [num_levels, 2]  int64 (H_l, W_l)
 *   level_start_index[num_levels]     int64, token offset of level l inside spatial_size
 *   sampling_loc     [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *   attn_weight      [batch, num_query, num_heads, num_levels, num_point]
 *   out              [batch, num_query, num_heads*channels]   (every element is written)
 */
int mvdetr_msda_forward_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *sampling_loc,
                            const float *attn_weight, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, float *out);
int mvdetr_msda_forward_f64(void *stream, const double *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const double *sampling_loc,
                            const double *attn_weight, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, double *out);

/* 16-bit storage variants of the forward (an extension: the reference dispatches float and double only,
 * ms_deform_attn_cuda.cu:64, so under autocast its callers cast to fp32 around the op).  Tensors as above with
 * IEEE binary16 (`_f16`) or bfloat16 (`_bf16`) elements passed as raw 16-bit words; every arithmetic step is fp32 and
 * the result is rounded once (nearest-even) when stored.  Forward only; device pointers only. */
int mvdetr_msda_forward_f16(void *stream, const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const uint16_t *sampling_loc,
                            const uint16_t *attn_weight, int batch, int spatial_size, int num_heads, int channels,
                            int num_levels, int num_query, int num_point, uint16_t *out);
int mvdetr_msda_forward_bf16(void *stream, const uint16_t *value, const int64_t *spatial_shapes,
                             const int64_t *level_start_index, const uint16_t *sampling_loc,
                             const uint16_t *attn_weight, int batch, int spatial_size, int num_heads, int channels,
                             int num_levels, int num_query, int num_point, uint16_t *out);

/* Fused forward for deformable-encoder calls: the arithmetic MSDeformAttn.forward wraps around the core
 * (multiview_detector/models/ops/modules/ms_deform_attn.py:100-107) happens inside the kernel, so the
 * sampling_locations / attention_weights tensors are never written or re-read:
 *     loc = reference_points[:, :, None] + sampling_offsets / (W_l, H_l);  aw = softmax_{L*P}(attn_logits)
 *   reference_points [*, num_query, num_levels, num_point, 2]; consecutive batch elements are
 *                    `ref_batch_stride` floats apart (0 = one set shared by the whole batch)
 *   sampling_offsets [batch, num_query, num_heads, num_levels, num_point, 2]  (the Linear's raw output)
 *   attn_logits      [batch, num_query, num_heads, num_levels, num_point]     (the Linear's raw output)
 *   level_major      bit mask.  Bit 0 (1): the two raw tensors are [batch, num_query, num_levels, num_heads,
 *                    num_point(, 2)] instead -- the caller permutes the Linear's weight rows once; keeps what one
 *                    level iteration reads in the same cache lines.  Bit 1 (2): reference_points is
 *                    [*, num_query, num_levels, 2], ONE point per (query, level) shared by the num_point sampling
 *                    points -- what MVDeTr's reference map holds P copies of (mvdetr.py:49-58 with all heights 0);
 *                    a quarter of the reference bytes.  Bit 2 (4), not together with bit 0: ONE raw tensor
 *                    [batch, num_query, num_heads / g, num_levels, (g x num_point x 2 offsets | g x num_point logits)]
 *                    with g = 32 / channels heads per 128-byte slice of the token row, again a permutation of the
 *                    Linears' weight rows: everything one workgroup reads for a (query, level) is one contiguous run
 *                    of 12 g floats (the kernel is bound by the number of cache lines a CU can miss on, and the plain
 *                    layouts fetch a 128-byte line for 32 or 64 of its bytes); `sampling_offsets` points at the tensor,
 *                    `attn_logits` must equal sampling_offsets + 8 g and the two query strides must be equal.
 *                    Bit 3 (8), only with bit 1: reference_points is [*, num_levels, num_query, 2] (neighbouring
 *                    queries of a level in the same cache lines).  Other bits: hipErrorInvalidValue.
 *   offsets_query_stride / logits_query_stride: floats from one query's block to the next (0 = dense), so
 *                    both may be column blocks of one wider GEMM output; multiples of 4
 * Only the shapes the LDS-tiled kernel takes are supported (fp32, channels 16 or 32, num_point 4,
 * num_levels <= 16, num_query == spatial_size, 16-byte aligned pointers): mvdetr_msda_fused_supported()
 * returns 1 for them, and the forward returns hipErrorNotSupported (801) otherwise. */
int mvdetr_msda_fused_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                int num_query, int num_point);
int mvdetr_msda_forward_fused_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const float *reference_points,
                                  int64_t ref_batch_stride, const float *sampling_offsets,
                                  const float *attn_logits, int level_major, int offsets_query_stride,
                                  int logits_query_stride, int batch, int spatial_size, int num_heads,
                                  int channels, int num_levels, int num_query, int num_point, float *out);

/* The same with the queries restricted to the tokens of levels [query_level_begin, query_level_end): the
 * encoder call of ONE rank of a query-sharded run (cameras = levels partitioned over GPUs; SURVEY 8e option
 * B -- not in the reference, which is single-device).  `value` still holds all spatial_size tokens;
 * reference_points, sampling_offsets, attn_logits and out hold only the num_query =
 * sum_{l in range} H_l*W_l queries of the range, in token order.  (0, num_levels) is the call above. */
int mvdetr_msda_fused_levels_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                       int num_query, int num_point, int query_level_begin, int query_level_end);
int mvdetr_msda_forward_fused_levels_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                         const int64_t *level_start_index, const float *reference_points,
                                         int64_t ref_batch_stride, const float *sampling_offsets,
                                         const float *attn_logits, int level_major, int offsets_query_stride,
                                         int logits_query_stride, int query_level_begin, int query_level_end,
                                         int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                         int num_query, int num_point, float *out);

/* ---- Multi-scale deformable attention, backward -----------------------------------------------
 * Replaces ms_deformable_col2im_cuda (ms_deform_im2col_cuda.cuh:956-1327) and its six kernel
 * variants (cuh:301-920, device helpers cuh:87-234).
 *   grad_col          [batch, num_query, num_heads*channels]   upstream gradient
 *   grad_value        same shape as value; MUST BE ZERO on entry (accumulated with atomics,
 *                     like the reference: ms_deform_attn_cuda.cu:121)
 *   grad_sampling_loc same shape as sampling_loc; fully written
 *   grad_attn_weight  same shape as attn_weight; fully written
 */
int mvdetr_msda_backward_f32(void *stream, const float *grad_col, const float *value,
                             const int64_t *spatial_shapes, const int64_t *level_start_index,
                             const float *sampling_loc, const float *attn_weight, int batch,
                             int spatial_size, int num_heads, int channels, int num_levels,
                             int num_query, int num_point, float *grad_value,
                             float *grad_sampling_loc, float *grad_attn_weight);
int mvdetr_msda_backward_f64(void *stream, const double *grad_col, const double *value,
                             const int64_t *spatial_shapes, const int64_t *level_start_index,
                             const double *sampling_loc, const double *attn_weight, int batch,
                             int spatial_size, int num_heads, int channels, int num_levels,
                             int num_query, int num_point, double *grad_value,
                             double *grad_sampling_loc, double *grad_attn_weight);

/* ---- Fused TRAINING pair (ABI 10): MSDeformAttn.forward / backward with the module arithmetic inside the kernels --------
 * What the reference runs when gradients are needed -- softmax and location arithmetic in torch (ms_deform_attn.py:100-107),
 * the extension's forward / backward on the materialised sampling_locations (135 MB at Wildtrack size) and
 * attention_weights (68 MB) (func.py:21-38, cuh:237-299, 956-1327), and torch's backward of that arithmetic -- as two calls
 * on the module's RAW Linear outputs:
 *   raw   [batch, Lq, >= M*L*P*3]  one tensor, per query [L][M/g][g*P*2 offsets | g*P logits], g = 32 / channels heads per
 *         128-byte slice (the layout MultiScaleDeformableAttention.slice_major_rows(level_outer=True) gives the GEMM);
 *         raw_query_stride floats between queries
 *   reference_points [batch or 1, L, Lq, 2]  ONE point per (query, level), level-major (MVDeTr's map; ref_batch_stride 0 when
 *         shared by the batch)
 * Shapes: queries = tokens (Lq = spatial_size), up to 16 levels OF EQUAL SHAPE (the caller's promise: on other shapes the
 * outputs are NaN), 4 points, 16- or 32-channel heads (an even number of 16-channel heads); mvdetr_msda_fused_train_supported
 * says whether a call qualifies (1 / 0).  ABI 10 - 12 took 6 / 7 levels of 16-channel heads only (MVDeTr's own shapes: their
 * kernels are msda_fwd_group2, msda_bwd_onepass and msda_bwd_fused_sampling); since ABI 13 every other encoder shape runs the
 * inference forward of that shape + a statistics pass, and the one-pass backward (16 channels) or msda_bwd_value_tok + the
 * level-groups sampling kernel (32 channels; not in the deterministic mode).
 * forward: out [batch, Lq, M*D] as mvdetr_msda_forward_fused_f32, plus stats [batch, Lq, M, 2] = (maximum logit,
 *          1 / sum of exp(logit - maximum)) of every (query, head), which the backward needs to rebuild the weights.
 * backward: grad_value [batch, S, M, D] (ACCUMULATED: zero it first) and grad_raw [batch, Lq, raw_query_stride] (the
 *          M*L*P*3 leading columns of every query are written, in raw's layout) from grad_output, the same inputs, stats and
 *          the forward's out (softmax backward: d logit = a (d a - <grad_output, out>)).
 * Return 0, hipErrorNotSupported (801) for shapes / alignments outside the above, another hipError_t on failure. */
int mvdetr_msda_fused_train_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                      int num_point);
int mvdetr_msda_forward_fused_train_f32(void *stream, const float *value, const int64_t *spatial_shapes,
                                        const int64_t *level_start_index, const float *reference_points,
                                        int64_t ref_batch_stride, const float *raw, int raw_query_stride, int batch,
                                        int spatial_size, int num_heads, int channels, int num_levels, int num_point,
                                        float *out, float *stats);
int mvdetr_msda_backward_fused_f32(void *stream, const float *grad_output, const float *value,
                                   const int64_t *spatial_shapes, const int64_t *level_start_index,
                                   const float *reference_points, int64_t ref_batch_stride, const float *raw,
                                   int raw_query_stride, const float *stats, const float *out, int batch, int spatial_size,
                                   int num_heads, int channels, int num_levels, int num_point, float *grad_value,
                                   float *grad_raw);

/* ---- Feature -> ground-plane homography warp ---------------------------------------------------
 * Replaces the third-party call kornia.warp_perspective(src, M, dsize, mode='bilinear',
 * padding_mode='zeros', align_corners=False) at multiview_detector/models/mvdetr.py:194-195
 * (kornia: normalize_homography -> inverse -> transform_points(meshgrid) -> F.grid_sample),
 * fused into one kernel.
 *   src  [n, channels, src_h, src_w]        (NCHW, like the reference's imgs_feat)
 *   M    [n, 3, 3]  destination pixel <- source pixel homography, same dtype as src
 *   dst  [n, channels, dst_h, dst_w]        every element is written (zeros outside the view)
 * `layout_nhwc` is a bit mask (bit 2, value 4: mode='nearest' instead of bilinear -- frameDataset.py:80).  Bit 0 (value 1): dst is written as [n, dst_h, dst_w, channels] instead (the
 * token layout the shadow transformer consumes, trans_world_feat.py:92), saving the permute copy.  Bit 1
 * (value 2): src is [n, src_h, src_w, channels] -- what a channels_last trunk produces -- so every bilinear
 * corner is one contiguous channel vector; supported together with bit 0 (value 3), channels a multiple of
 * 16 bytes, 16-byte aligned pointers; hipErrorNotSupported (801) otherwise.  Other bits: hipErrorInvalidValue.
 */
int mvdetr_warp_perspective_forward_f32(void *stream, const float *src, const float *M, int n,
                                        int channels, int src_h, int src_w, int dst_h, int dst_w,
                                        int layout_nhwc, float *dst);
int mvdetr_warp_perspective_forward_f64(void *stream, const double *src, const double *M, int n,
                                        int channels, int src_h, int src_w, int dst_h, int dst_w,
                                        int layout_nhwc, double *dst);

/* Gradient of the warp w.r.t. src (what autograd reaches through grid_sample in the reference).
 *   grad_dst [n, channels, dst_h, dst_w] (or NHWC if layout_nhwc bit 0)
 *   grad_src [n, channels, src_h, src_w] (or NHWC if layout_nhwc bit 1, same restrictions as the forward);
 *            OVERWRITTEN: every element is stored, it need not be zeroed (ABI 9; up to ABI 8 it had to be zero on entry)
 * With both sides channel-last (layout_nhwc & 3 == 3) the gradient is computed as a GATHER over the destination pixels
 * whose bilinear footprint touches each source texel (the homography is invertible): no atomics, the order of every
 * sum is fixed, so the result is deterministic -- unlike grid_sample's atomicAdd backward.  It uses
 * stream-ordered scratch (80 bytes per 2x2 block of source texels + 4 bytes per destination pixel, hipMallocAsync on
 * `stream`), kept per (device, stream) between calls and grown on demand.
 * The other layouts scatter with fp atomics after a hipMemsetAsync (MVDETR_WARP_BWD_IMPL=scatter forces that for
 * channel-last tensors too).
 */
int mvdetr_warp_perspective_backward_f32(void *stream, const float *grad_dst, const float *M, int n,
                                         int channels, int src_h, int src_w, int dst_h, int dst_w,
                                         int layout_nhwc, float *grad_src);
int mvdetr_warp_perspective_backward_f64(void *stream, const double *grad_dst, const double *M,
                                         int n, int channels, int src_h, int src_w, int dst_h,
                                         int dst_w, int layout_nhwc, double *grad_src);

/* The one-call gradient with a caller-supplied VERSION TAG of the matrices (ABI 12): `matrices_tag` != 0 is the caller's
 * promise that calls with the same tag -- on this device and stream, with the same shapes -- pass the same M.  The library
 * keeps the gather's geometry (see below) in its per-(device, stream) scratch together with the tag, so the second and later
 * calls of a training loop without augmentation launch the gather alone: what mvdetr_warp_perspective_backward_* would do if
 * it could compare M on the host (it cannot: M is a device pointer, and comparing it on the device would cost every call a
 * dependent launch).  Tag 0 = unknown: identical to mvdetr_warp_perspective_backward_*.  A new tag (or an untagged call on
 * the stream in between) rebuilds the geometry.  Replaces kornia.warp_perspective's backward at mvdetr.py:194-195. */
int mvdetr_warp_perspective_backward_tagged_f32(void *stream, const float *grad_dst, const float *M, int n, int channels,
                                                int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                uint64_t matrices_tag, float *grad_src);
int mvdetr_warp_perspective_backward_tagged_f64(void *stream, const double *grad_dst, const double *M, int n, int channels,
                                                int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                uint64_t matrices_tag, double *grad_src);

/* The same gradient in two steps, for callers whose matrices stay the same from call to call (training without
 * augmentation: the projection matrices are constants, mvdetr.py:82-95,155-161).  The gather's geometry -- which destination
 * pixels can touch each 2x2 block of source texels -- depends on M and the shapes only:
 *   mvdetr_warp_backward_plan_bytes   size of the plan of such a call in bytes, 0 when the gather does not take the shapes
 *                                     (channels * elem_size not a multiple of 16, more than 2^31 elements, or
 *                                     MVDETR_WARP_BWD_IMPL=scatter): use mvdetr_warp_perspective_backward_* then
 *   mvdetr_warp_backward_plan_*       fills `plan` (caller-owned device memory of that size, 16-byte aligned) for M [n,3,3]
 *   mvdetr_warp_perspective_backward_planned_*   the gradient from a plan built for the same M and shapes; the plan is only
 *                                     read (any number of calls, any streams ordered after the plan's); layouts: channel-last
 *                                     on both sides (layout_nhwc bits 0 and 1 set; bit 2 = nearest), 16-byte aligned tensors
 * One kernel per call instead of two and a stream write, no library-side scratch.  Results are bit-identical to
 * mvdetr_warp_perspective_backward_* (the same kernels). */
int64_t mvdetr_warp_backward_plan_bytes(int n, int channels, int src_h, int src_w, int dst_h, int dst_w, int elem_size);
int mvdetr_warp_backward_plan_f32(void *stream, const float *M, int n, int channels, int src_h, int src_w, int dst_h, int dst_w,
                                  void *plan);
int mvdetr_warp_backward_plan_f64(void *stream, const double *M, int n, int channels, int src_h, int src_w, int dst_h, int dst_w,
                                  void *plan);
int mvdetr_warp_perspective_backward_planned_f32(void *stream, const float *grad_dst, const float *M, const void *plan, int n,
                                                 int channels, int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                 float *grad_src);
int mvdetr_warp_perspective_backward_planned_f64(void *stream, const double *grad_dst, const double *M, const void *plan, int n,
                                                 int channels, int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                 double *grad_src);

/* ---- Residual add + LayerNorm (the tail of both halves of the shadow transformer's encoder layer) -------
 * Replaces  self.norm1(src + self.dropout1(src2))  /  self.norm2(src + self.dropout3(ffn))  in eval mode
 * (multiview_detector/models/deformable_transformer.py:96-100; torch.nn.LayerNorm over the last dimension,
 * biased variance, eps inside the square root) by one pass:
 *     out[r, :] = LayerNorm(x[r, :] + residual[r, :]) * weight + bias
 *   x, residual, out [rows, cols] fp32, contiguous (residual may be NULL; out must not overlap the inputs)
 *   weight, bias     [cols]  (both NULL: no affine)
 * cols must be 64, 128 or 256 with (cols/64)*4-byte aligned pointers: hipErrorNotSupported (801) otherwise. */
int mvdetr_add_layernorm_f32(void *stream, const float *x, const float *residual, const float *weight,
                             const float *bias, int64_t rows, int cols, float eps, float *out);
/* The same with a second output  out2[r, :] = out[r, :] + add2[r % add2_rows, :]  -- the next encoder layer's query
 * `src + pos` (deformable_transformer.py:92, with_pos_embed), so that add is not a pass of its own.  add2
 * [add2_rows, cols] (the position embedding, shared by the batch); add2 and out2 are given together. */
int mvdetr_add_layernorm_add_f32(void *stream, const float *x, const float *residual, const float *weight,
                                 const float *bias, const float *add2, int64_t add2_rows, int64_t rows, int cols,
                                 float eps, float *out, float *out2);

/* ---- Introspection (used by bench.py / tests, not by the model code) ---------------------------
 * Name of the kernel variant the last forward call ON THIS THREAD dispatched to
 * ("gather", "tile16", ...).  Static storage; never NULL. */
const char *mvdetr_msda_last_forward_impl(void);
/* Name of the kernel that call launched ("msda_fwd_group[LDS-DMA windows]", "msda_fwd_tile", "msda_fwd_gather", ...). */
const char *mvdetr_msda_last_forward_kernel(void);

/* What the code object records for that kernel instantiation (hipFuncGetAttributes): registers per lane, scratch bytes per
 * lane (non-zero = the instantiation spills), static LDS bytes.  Returns 0 (and -1 in the outputs) when unknown. */
int mvdetr_msda_last_forward_resources(int *num_regs, int *scratch_bytes_per_lane, int *static_lds_bytes);

/* Name of the kernel the last warp call of this process launched (any thread: autograd runs backwards on its own) ("warp_fwd_cl", "warp_fwd<NCHW>", "warp_bwd_gather",
 * ...): one name per layout route, asserted by tests/test_warp_gpu.py.  Static storage; never NULL. */
const char *mvdetr_warp_last_kernel(void);

/* The warp gradient's one-call form keeps its geometry scratch per (device, stream) between calls (hipMallocAsync on the
 * call's stream, grown on demand).  This drops every cached buffer -- e.g. after destroying streams, or to hand the memory
 * back; call it when no warp backward is in flight.  Not usable under HIP graph capture: capture the two-step form
 * (mvdetr_warp_backward_plan_* + mvdetr_warp_perspective_backward_planned_*) with a caller-owned plan buffer instead. */
int mvdetr_warp_release_scratch(void);

/* Forward kernel variant selection: 0 = auto (default; tiled LDS kernel where it applies, else
 * the gather kernel), 1 = always gather, 2 = tile whenever the shape supports it.  Results are
 * identical up to fp32 summation order; this is a tuning/testing knob.  Returns the previous value.
 * Initial value comes from MVDETR_MSDA_FWD_IMPL = auto | gather | tile. */
int mvdetr_msda_set_forward_impl(int impl);

/* Deterministic backward (opt-in; ABI 13).  The reference's col2im adds grad_value with atomicAdd
 * (ms_deform_im2col_cuda.cuh:125-152) and is not reproducible run to run; neither are this library's default backward
 * kernels (fp32 atomics when LDS windows are flushed).  With on != 0, mvdetr_msda_backward_f32 and
 * mvdetr_msda_backward_fused_f32 sum grad_value in 64-bit fixed point with one binary point per call -- 38 bits below
 * max|grad_out| x max(1, max|attn_weight|) -- so the result is bit-identical run to run (grad_sampling_loc / grad_attn_weight have
 * one writer per element in every mode).  Costs a scratch of 8 bytes per value element, kept per (device, stream) between calls (mvdetr_msda_release_scratch), and three small launches.
 * Only deformable-encoder calls are served (fp32, 16-channel heads, equal level shapes, num_query == spatial_size,
 * spatial_size * num_levels * num_point < 2^24); every other call returns hipErrorNotSupported (801) while the mode is on, and
 * unequal level shapes (device data) fill grad_value with NaN.  Returns the previous state; the initial state comes from
 * MVDETR_MSDA_BWD_DETERMINISTIC=1. */
int mvdetr_msda_set_backward_deterministic(int on);
/* The mode's current state (1 = on), without changing it (ABI 14: a caller that only wants to KNOW -- MSDeformAttn.forward
 * refusing a training call the mode cannot serve -- must not toggle a process-wide switch other threads' backwards read). */
int mvdetr_msda_get_backward_deterministic(void);

/* The deterministic mode keeps its 64-bit accumulators (8 bytes per value element) per (device, stream) between calls
 * (hipMallocAsync on the call's stream, grown on demand).  This drops every cached buffer; call it when no backward is in flight.
 * Not usable under HIP graph capture. */
int mvdetr_msda_release_scratch(void);

/* dst[n][c][r] = src[n][r][c]: layout change between NCHW (rows = channels, cols = h*w) and the channel-last layout the
 * fast warp kernels read, and back.  Tiled through LDS, both sides move in 256-byte runs. */
int mvdetr_transpose_f32(void *stream, const float *src, int n, int rows, int cols, float *dst);
int mvdetr_transpose_f64(void *stream, const double *src, int n, int rows, int cols, double *dst);

/* ---- CPU path (host pointers, no stream, synchronous) ------------------------------------------------------------
 * The reference extension raises for CPU tensors (ms_deform_attn_cpu.cpp:17-41 are stubs; ms_deform_attn.h:38,60).
 * These entry points make the same contracts work on host memory: same argument meaning and layouts as the device
 * functions above; std::thread parallel (MVDETR_HOST_THREADS, else OMP_NUM_THREADS, else all cores up to 64);
 * deterministic (no atomics).  grad_value / grad_src are ACCUMULATED into: pass them zeroed.
 * warp: `layout_nhwc` bits 0/1 as above; `mode` 0 = bilinear, 1 = nearest.  Return 0 on success. */
int mvdetr_msda_forward_host_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const float *sampling_loc, const float *attn_weight, int batch, int spatial_size,
                                 int num_heads, int channels, int num_levels, int num_query, int num_point, float *out);
int mvdetr_msda_forward_host_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const double *sampling_loc, const double *attn_weight, int batch, int spatial_size,
                                 int num_heads, int channels, int num_levels, int num_query, int num_point, double *out);
int mvdetr_msda_backward_host_f32(const float *grad_output, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const float *sampling_loc, const float *attn_weight,
                                  int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                  int num_point, float *grad_value, float *grad_sampling_loc, float *grad_attn_weight);
int mvdetr_msda_backward_host_f64(const double *grad_output, const double *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const double *sampling_loc, const double *attn_weight,
                                  int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                  int num_point, double *grad_value, double *grad_sampling_loc, double *grad_attn_weight);
int mvdetr_warp_perspective_forward_host_f32(const float *src, const float *mats, int n, int channels, int src_h, int src_w,
                                             int dst_h, int dst_w, int layout_nhwc, int mode, float *dst);
int mvdetr_warp_perspective_forward_host_f64(const double *src, const double *mats, int n, int channels, int src_h, int src_w,
                                             int dst_h, int dst_w, int layout_nhwc, int mode, double *dst);
int mvdetr_warp_perspective_backward_host_f32(const float *grad_dst, const float *mats, int n, int channels, int src_h,
                                              int src_w, int dst_h, int dst_w, int layout_nhwc, int mode, float *grad_src);
int mvdetr_warp_perspective_backward_host_f64(const double *grad_dst, const double *mats, int n, int channels, int src_h,
                                              int src_w, int dst_h, int dst_w, int layout_nhwc, int mode, double *grad_src);

#ifdef __cplusplus
}
#endif
#endif /* MVDETR_OPS_H */
