// Lane-group formulation of the MSDA backward (any shapes / dtypes), shared by msda_backward.hip (its kernel) and
// msda_backward_tile.hip (in-launch fallback for levels of unequal shape).  Internal, not part of the C ABI.
#pragma once
#include "common.h"

namespace mvdetr {

template <typename T> __device__ __forceinline__ void atomic_add(T *p, T v) { atomicAdd(p, v); }

template <typename T, int G> __device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

// One lane of the lane-group backward: lane `idx` of B*Lq*M*G (G = D/VEC lanes per (b, q, head)).  Every lane of
// a wave must call it (idx past the end is fine): all G lanes of a group stay converged (same (b,q,m) => same
// branch decisions), which is what makes the shuffles legal.
// VALUE_GRAD = false: grad_sampling_loc / grad_attn_weight only.
// SAMPLING_GRAD = false: grad_value only (the one-pass kernel's grad_value-only variant, whose sampling gradients come from
// another kernel).
template <typename T, int VEC, int G, bool VALUE_GRAD = true, bool SAMPLING_GRAD = true>
__device__ __forceinline__ void msda_bwd_lanes_body(
    int64_t idx, const T *__restrict__ grad_col, const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw, int B, int S,
    int M, int D, int L, int Lq, int P, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_aw, int l_begin = 0, int l_end = -1)
{
    // [l_begin, l_end): the levels this call covers (default: all; the one-pass kernel's stand-down jobs own one level)
    if (l_end < 0) l_end = L;
    const int64_t total = (int64_t)B * Lq * M * G;
    const bool live = idx < total;
    const int64_t cidx = live ? idx : total - 1;
    const int cg = (int)(cidx % G);
    const int64_t bqm = cidx / G;
    const int m = (int)(bqm % M);
    const int64_t bq = bqm / M;
    const int b = (int)(bq / Lq);
    const int64_t row = (int64_t)M * D;
    const T *lp = loc + bqm * L * P * 2;
    const T *wp = aw + bqm * L * P;
    const int64_t voff = (int64_t)b * S * row + (int64_t)m * D + cg * VEC;
    const Pack<T, VEC> go = Pack<T, VEC>::load(grad_col + bqm * D + cg * VEC);
    for (int l = l_begin; l < l_end; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int64_t poff = voff + lsi[l] * row;
        for (int p = 0; p < P; ++p) {
            const int t = l * P + p;
            const T x = lp[t * 2 + 0] * T(W) - T(0.5);
            const T y = lp[t * 2 + 1] * T(H) - T(0.5);
            const T a = wp[t];
            T g_a = 0, g_x = 0, g_y = 0;
            if (y > T(-1) && x > T(-1) && y < T(H) && x < T(W)) {
                const Footprint<T> f = footprint(y, x, H, W);
                const int64_t o00 = poff + ((int64_t)f.y0 * W + f.x0) * row;
                const int64_t o01 = o00 + row, o10 = o00 + (int64_t)W * row, o11 = o10 + row;
                const bool v00 = f.vy0 && f.vx0, v01 = f.vy0 && f.vx1;
                const bool v10 = f.vy1 && f.vx0, v11 = f.vy1 && f.vx1;
                Pack<T, VEC> c00 = Pack<T, VEC>::zero(), c01 = c00, c10 = c00, c11 = c00;
                if (v00) c00 = Pack<T, VEC>::load(value + o00);
                if (v01) c01 = Pack<T, VEC>::load(value + o01);
                if (v10) c10 = Pack<T, VEC>::load(value + o10);
                if (v11) c11 = Pack<T, VEC>::load(value + o11);
                const T w00 = f.wy0 * f.wx0, w01 = f.wy0 * f.wx1, w10 = f.wy1 * f.wx0, w11 = f.wy1 * f.wx1;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const T g = go.v[i];
                    g_a += g * (w00 * c00.v[i] + w01 * c01.v[i] + w10 * c10.v[i] + w11 * c11.v[i]);
                    g_x += g * ((c01.v[i] - c00.v[i]) * f.wy0 + (c11.v[i] - c10.v[i]) * f.wy1);
                    g_y += g * ((c10.v[i] - c00.v[i]) * f.wx0 + (c11.v[i] - c01.v[i]) * f.wx1);
                    if (VALUE_GRAD && live) {
                        const T ga = g * a;
                        if (v00) atomic_add(grad_value + o00 + i, w00 * ga);
                        if (v01) atomic_add(grad_value + o01 + i, w01 * ga);
                        if (v10) atomic_add(grad_value + o10 + i, w10 * ga);
                        if (v11) atomic_add(grad_value + o11 + i, w11 * ga);
                    }
                }
            }
            g_a = group_sum<T, G>(g_a);
            g_x = group_sum<T, G>(g_x);
            g_y = group_sum<T, G>(g_y);
            if (SAMPLING_GRAD && live && cg == 0) {
                grad_aw[bqm * L * P + t] = g_a;
                grad_loc[(bqm * L * P + t) * 2 + 0] = T(W) * a * g_x;
                grad_loc[(bqm * L * P + t) * 2 + 1] = T(H) * a * g_y;
            }
        }
    }
}

}  // namespace mvdetr
