// Multi-scale deformable attention, forward, 16-bit storage (float16 / bfloat16) -- gfx950 (MI355X).
//
// The reference dispatches float and double only (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64), so a model
// under autocast has to cast to fp32 around the op.  These two entry points take the tensors as they are: value,
// sampling locations and attention weights are READ in 16 bits (half the HBM bytes of the fp32 op), every arithmetic
// step is fp32 (the same formula as msda_gather_body.h: loc*size - 0.5, floor, four corners, zero padding), and the
// result is rounded once, to nearest-even, on the way out.  Forward only (inference): the backward stays fp32/fp64.
//
// One lane owns VEC consecutive channels of one (b, q, head): with D = 32 and VEC = 4 the eight lanes of a head read
// one contiguous 64-byte token segment per bilinear corner.
#include "common.h"
#include "../../include/mvdetr_ops.h"

namespace mvdetr {

struct KernelResources;
void msda_note_forward_kernel(const char *name, const KernelResources *res = nullptr);

struct F16 {
    static __device__ __forceinline__ float up(uint16_t h)
    {
        _Float16 x;
        __builtin_memcpy(&x, &h, 2);
        return (float)x;
    }
    static __device__ __forceinline__ uint16_t down(float f)
    {
        const _Float16 x = (_Float16)f;                   // v_cvt_f16_f32: round to nearest even
        uint16_t h;
        __builtin_memcpy(&h, &x, 2);
        return h;
    }
};

struct BF16 {
    static __device__ __forceinline__ float up(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
    static __device__ __forceinline__ uint16_t down(float f)
    {
        const uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);      // quiet NaN
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                      // round to nearest even
    }
};

template <int VEC> struct Raw;                            // VEC 16-bit values as one access
template <> struct Raw<1> { uint16_t v[1]; };
template <> struct alignas(8) Raw<4> { uint16_t v[4]; };
template <> struct alignas(16) Raw<8> { uint16_t v[8]; };

template <typename C, int VEC>
__global__ __launch_bounds__(256) void msda_fwd_gather_half(
    const uint16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const uint16_t *__restrict__ loc, const uint16_t *__restrict__ aw, int B, int S, int M, int D, int L, int Lq,
    int P, uint16_t *__restrict__ out)
{
    const int groups = D / VEC;
    const int64_t total = (int64_t)B * Lq * M * groups;
    const int64_t row = (int64_t)M * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int cg = (int)(idx % groups);
        const int64_t bqm = idx / groups;
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / M / Lq);
        const uint16_t *lp = loc + bqm * L * P * 2;
        const uint16_t *wp = aw + bqm * L * P;
        const uint16_t *vb = value + (int64_t)b * S * row + (int64_t)m * D + cg * VEC;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const uint16_t *plane = vb + lsi[l] * row;
            for (int p = 0; p < P; ++p) {
                const float x = C::up(lp[(l * P + p) * 2 + 0]) * (float)W - 0.5f;
                const float y = C::up(lp[(l * P + p) * 2 + 1]) * (float)H - 0.5f;
                const float a = C::up(wp[l * P + p]);
                if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
                const Footprint<float> f = footprint(y, x, H, W);
                const uint16_t *r0 = plane + ((int64_t)f.y0 * W + f.x0) * row;
                const uint16_t *r1 = r0 + (int64_t)W * row;
                Raw<VEC> c00{}, c01{}, c10{}, c11{};
                if (f.vy0 && f.vx0) c00 = *reinterpret_cast<const Raw<VEC> *>(r0);
                if (f.vy0 && f.vx1) c01 = *reinterpret_cast<const Raw<VEC> *>(r0 + row);
                if (f.vy1 && f.vx0) c10 = *reinterpret_cast<const Raw<VEC> *>(r1);
                if (f.vy1 && f.vx1) c11 = *reinterpret_cast<const Raw<VEC> *>(r1 + row);
                const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a;
                const float w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    acc[i] += w00 * C::up(c00.v[i]) + w01 * C::up(c01.v[i]) + w10 * C::up(c10.v[i]) +
                              w11 * C::up(c11.v[i]);
            }
        }
        Raw<VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = C::down(acc[i]);
        *reinterpret_cast<Raw<VEC> *>(out + bqm * D + cg * VEC) = o;
    }
}

template <typename C, int VEC>
static int launch_half(hipStream_t st, const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                       const uint16_t *loc, const uint16_t *aw, int B, int S, int M, int D, int L, int Lq, int P,
                       uint16_t *out)
{
    const int64_t total = (int64_t)B * Lq * M * (D / VEC);
    const int block = 256;
    int64_t blocks = (total + block - 1) / block;
    // as msda_forward.hip's gather launch: two workgroups per CU walking the items in order keep the lanes in flight in
    // one compact band of queries
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    if (blocks > 2 * (int64_t)cus) blocks = 2 * (int64_t)cus;
    hipLaunchKernelGGL((msda_fwd_gather_half<C, VEC>), dim3((unsigned)blocks), dim3(block), 0, st, value, shapes, lsi,
                       loc, aw, B, S, M, D, L, Lq, P, out);
    return (int)hipGetLastError();
}

template <typename C>
static int forward_half(void *stream, const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const uint16_t *loc, const uint16_t *aw, int B, int S, int M, int D, int L, int Lq, int P,
                        uint16_t *out)
{
    if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return (int)hipErrorInvalidValue;
    if ((int64_t)B * Lq == 0) return 0;
    if (!value || !shapes || !lsi || !loc || !aw || !out) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    msda_note_forward_kernel("msda_fwd_gather_half");
    if (D % 8 == 0 && aligned(value, 16) && aligned(out, 16))
        return launch_half<C, 8>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
    if (D % 4 == 0 && aligned(value, 8) && aligned(out, 8))
        return launch_half<C, 4>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
    return launch_half<C, 1>(st, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, out);
}

}  // namespace mvdetr

extern "C" {

int mvdetr_msda_forward_f16(void *stream, const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const uint16_t *sampling_loc,
                            const uint16_t *attn_weight, int batch, int spatial_size, int num_heads, int channels,
                            int num_levels, int num_query, int num_point, uint16_t *out)
{
    return mvdetr::forward_half<mvdetr::F16>(stream, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                             batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                                             out);
}

int mvdetr_msda_forward_bf16(void *stream, const uint16_t *value, const int64_t *spatial_shapes,
                             const int64_t *level_start_index, const uint16_t *sampling_loc,
                             const uint16_t *attn_weight, int batch, int spatial_size, int num_heads, int channels,
                             int num_levels, int num_query, int num_point, uint16_t *out)
{
    return mvdetr::forward_half<mvdetr::BF16>(stream, value, spatial_shapes, level_start_index, sampling_loc,
                                              attn_weight, batch, spatial_size, num_heads, channels, num_levels,
                                              num_query, num_point, out);
}

}  // extern "C"
