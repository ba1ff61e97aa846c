// Residual add + LayerNorm over the channel dimension, one pass -- gfx950 (MI355X).
//
// The shadow transformer's encoder layer ends each half with  norm(src + sublayer(src))
// (multiview_detector/models/deformable_transformer.py:96-100; torch: an elementwise add kernel, then
// LayerNorm: 30 + 73 us per call at 75,600 tokens x 128 channels).  Here one wave owns a token row: lanes hold
// cols/64 consecutive channels each (float2 for 128 channels: a row is one 512-byte access), the sum and the
// centred second moment are reduced across the wave with xor-shuffles (two-pass in registers, no LDS), and the
// normalised row is written once: 116 MB of traffic instead of 232 MB.  128 channels (the encoder's width) take
// add_layernorm_rows128x2 below: two rows per wave, 24.7 -> 18.6 us at 75,600 rows (0.78 of the HBM roofline).
#include "common.h"
#include "../../include/mvdetr_ops.h"

namespace mvdetr {

template <int VEC>
__global__ __launch_bounds__(256) void add_layernorm_rows(const float *__restrict__ x, const float *__restrict__ res,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ add2, int64_t add2_rows,
                                                          int64_t rows, float eps, float *__restrict__ out,
                                                          float *__restrict__ out2)
{
    constexpr int COLS = VEC * 64;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const Pack<float, VEC> g = gamma ? Pack<float, VEC>::load(gamma + lane * VEC) : Pack<float, VEC>::zero();
    const Pack<float, VEC> bt = beta ? Pack<float, VEC>::load(beta + lane * VEC) : Pack<float, VEC>::zero();
    for (int64_t r = wave; r < rows; r += nwaves) {
        Pack<float, VEC> v = Pack<float, VEC>::load(x + r * COLS + lane * VEC);
        if (res) {
            const Pack<float, VEC> q = Pack<float, VEC>::load(res + r * COLS + lane * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v.v[i] += q.v[i];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += v.v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.f / COLS);
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) m2 += (v.v[i] - mean) * (v.v[i] - mean);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        const float rstd = rsqrtf(m2 * (1.f / COLS) + eps);
        Pack<float, VEC> y;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float n = (v.v[i] - mean) * rstd;
            y.v[i] = gamma ? n * g.v[i] + bt.v[i] : n;
        }
        y.store(out + r * COLS + lane * VEC);
        if (out2) {                                           // second output: y + add2 (rows of add2 repeat)
            const Pack<float, VEC> p = Pack<float, VEC>::load(add2 + (r % add2_rows) * COLS + lane * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) y.v[i] += p.v[i];
            y.store(out2 + r * COLS + lane * VEC);
        }
    }
}

// 128 channels, TWO rows per wave: a half wave owns a token row (float4 per lane: a row is one 512-byte access), so a wave has
// two rows' loads in flight, and the 32-lane sums are four DPP adds (quad swaps, row_half_mirror, row_mirror) + ONE ds_bpermute
// (lane ^ 16) instead of six ds_bpermute per reduction through the LDS queue.
template <int CTRL> __device__ __forceinline__ float ln_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ln_sum32(float v)
{
    v += ln_dpp<0xB1>(v);                                     // lane ^ 1
    v += ln_dpp<0x4E>(v);                                     // lane ^ 2
    v += ln_dpp<0x141>(v);                                    // row_half_mirror: the other quad of the eight
    v += ln_dpp<0x140>(v);                                    // row_mirror: the other eight of the sixteen
    return v + __shfl_xor(v, 16, 64);                         // the other sixteen of the half wave
}

__global__ __launch_bounds__(256) void add_layernorm_rows128x2(const float *__restrict__ x, const float *__restrict__ res,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               const float *__restrict__ add2, int64_t add2_rows,
                                                               int64_t rows, float eps, float *__restrict__ out,
                                                               float *__restrict__ out2)
{
    constexpr int COLS = 128;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + hl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bt = beta ? *reinterpret_cast<const float4 *>(beta + hl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r0 = wave * 2; r0 < rows; r0 += nwaves * 2) {
        const int64_t r = r0 + (lane >> 5);
        const bool live = r < rows;                           // (an odd row count: the last wave's upper half idles, converged)
        const int64_t rr = live ? r : rows - 1;
        float4 v = *reinterpret_cast<const float4 *>(x + rr * COLS + hl * 4);
        if (res) {
            const float4 q = *reinterpret_cast<const float4 *>(res + rr * COLS + hl * 4);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const float mean = ln_sum32((v.x + v.y) + (v.z + v.w)) * (1.f / COLS);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float rstd = rsqrtf(ln_sum32((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / COLS) + eps);
        float4 y = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
        if (gamma) y = make_float4(y.x * g.x + bt.x, y.y * g.y + bt.y, y.z * g.z + bt.z, y.w * g.w + bt.w);
        if (live) *reinterpret_cast<float4 *>(out + r * COLS + hl * 4) = y;
        if (out2) {                                           // second output: y + add2 (rows of add2 repeat)
            const float4 p = *reinterpret_cast<const float4 *>(add2 + (rr % add2_rows) * COLS + hl * 4);
            if (live) *reinterpret_cast<float4 *>(out2 + r * COLS + hl * 4) = make_float4(y.x + p.x, y.y + p.y, y.z + p.z, y.w + p.w);
        }
    }
}

}  // namespace mvdetr

extern "C" int mvdetr_add_layernorm_f32(void *stream, const float *x, const float *residual, const float *weight,
                                        const float *bias, int64_t rows, int cols, float eps, float *out)
{
    return mvdetr_add_layernorm_add_f32(stream, x, residual, weight, bias, nullptr, 0, rows, cols, eps, out, nullptr);
}

extern "C" int mvdetr_add_layernorm_add_f32(void *stream, const float *x, const float *residual, const float *weight,
                                            const float *bias, const float *add2, int64_t add2_rows, int64_t rows,
                                            int cols, float eps, float *out, float *out2)
{
    using namespace mvdetr;
    if (rows < 0 || cols <= 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!x || !out || (weight == nullptr) != (bias == nullptr)) return (int)hipErrorInvalidValue;
    if ((add2 == nullptr) != (out2 == nullptr) || (add2 && add2_rows <= 0)) return (int)hipErrorInvalidValue;
    if (cols != 64 && cols != 128 && cols != 256) return (int)hipErrorNotSupported;
    const size_t al = cols == 64 ? 4 : cols == 128 ? 8 : 16;
    if (!aligned(x, al) || !aligned(out, al) || (residual && !aligned(residual, al)) || (weight && !aligned(weight, al)) ||
        (bias && !aligned(bias, al)) || (add2 && (!aligned(add2, al) || !aligned(out2, al))))
        return (int)hipErrorNotSupported;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t want = (rows + 3) / 4;
    const unsigned blocks = (unsigned)(want < 256 * 16 ? want : 256 * 16);         // grid-stride above 16 blocks per CU
    const bool a16 = aligned(x, 16) && aligned(out, 16) && (!residual || aligned(residual, 16)) && (!weight || (aligned(weight, 16) && aligned(bias, 16))) &&
                     (!add2 || (aligned(add2, 16) && aligned(out2, 16)));
    if (cols == 128 && a16) {
        const int64_t want2 = (rows + 7) / 8;                                          // two rows per wave, four waves per block
        const unsigned blocks2 = (unsigned)(want2 < 256 * 16 ? want2 : 256 * 16);
        hipLaunchKernelGGL(add_layernorm_rows128x2, dim3(blocks2), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
        return (int)hipGetLastError();
    }
    if (cols == 64) hipLaunchKernelGGL(add_layernorm_rows<1>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    else if (cols == 128) hipLaunchKernelGGL(add_layernorm_rows<2>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    else hipLaunchKernelGGL(add_layernorm_rows<4>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    return (int)hipGetLastError();
}
