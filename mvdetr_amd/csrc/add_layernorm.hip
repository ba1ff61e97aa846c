// Residual add + LayerNorm over the channel dimension, one pass -- gfx950 (MI355X).
//
// The shadow transformer's encoder layer ends each half with  norm(src + sublayer(src))
// (multiview_detector/models/deformable_transformer.py:96-100; torch: an elementwise add kernel, then
// LayerNorm: 30 + 73 us per call at 75,600 tokens x 128 channels).  Here one wave owns a token row: lanes hold
// cols/64 consecutive channels each (float2 for 128 channels: a row is one 512-byte access), the sum and the
// centred second moment are reduced across the wave with xor-shuffles (two-pass in registers, no LDS), and the
// normalised row is written once: 116 MB of traffic instead of 232 MB.
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
    if (cols == 64) hipLaunchKernelGGL(add_layernorm_rows<1>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    else if (cols == 128) hipLaunchKernelGGL(add_layernorm_rows<2>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    else hipLaunchKernelGGL(add_layernorm_rows<4>, dim3(blocks), dim3(256), 0, st, x, residual, weight, bias, add2, add2_rows, rows, eps, out, out2);
    return (int)hipGetLastError();
}
