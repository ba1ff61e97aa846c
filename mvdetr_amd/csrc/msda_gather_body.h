// Gather formulation of the MSDA forward (any shape / dtype), shared by msda_forward.hip (its kernel) and the tile
// kernels' in-launch stand-down.  Internal, not part of the C ABI.
#pragma once
#include "common.h"

namespace mvdetr {

// Grid-stride gather formulation: work items first, first + stride, ... of B*Lq*M*(D/VEC); one item = VEC consecutive
// channels of one (b, q, head).  Body of msda_fwd_gather; also what the tile kernels run in their own launch when
// the locality probe says the sampling locations are far from the queries (msda_forward_tile.hip).
template <typename T, int VEC>
__device__ __forceinline__ void msda_fwd_gather_body(
    int64_t first, int64_t stride, const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw,
    int B, int S, int M, int D, int L, int Lq, int P, T *__restrict__ out)
{
    const int groups = D / VEC;                          // lanes per (b,q,m)
    const int64_t total = (int64_t)B * Lq * M * groups;
    const int64_t row = (int64_t)M * D;                  // elements per value token
    for (int64_t idx = first; idx < total; idx += stride) {
        const int cg = (int)(idx % groups);
        const int64_t bqm = idx / groups;
        const int m = (int)(bqm % M);
        const int64_t bq = bqm / M;
        const int b = (int)(bq / Lq);
        const T *lp = loc + bqm * L * P * 2;
        const T *wp = aw + bqm * L * P;
        const T *vb = value + (int64_t)b * S * row + (int64_t)m * D + cg * VEC;
        Pack<T, VEC> acc = Pack<T, VEC>::zero();
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const T *plane = vb + lsi[l] * row;
            for (int p = 0; p < P; ++p) {
                const T x = lp[(l * P + p) * 2 + 0] * T(W) - T(0.5);
                const T y = lp[(l * P + p) * 2 + 1] * T(H) - T(0.5);
                const T a = wp[l * P + p];
                if (!(y > T(-1) && x > T(-1) && y < T(H) && x < T(W))) continue;
                const Footprint<T> f = footprint(y, x, H, W);
                const T *r0 = plane + ((int64_t)f.y0 * W + f.x0) * row;
                const T *r1 = r0 + (int64_t)W * row;
                Pack<T, VEC> c00 = Pack<T, VEC>::zero(), c01 = c00, c10 = c00, c11 = c00;
                if (f.vy0 && f.vx0) c00 = Pack<T, VEC>::load(r0);
                if (f.vy0 && f.vx1) c01 = Pack<T, VEC>::load(r0 + row);
                if (f.vy1 && f.vx0) c10 = Pack<T, VEC>::load(r1);
                if (f.vy1 && f.vx1) c11 = Pack<T, VEC>::load(r1 + row);
                const T w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a;
                const T w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    acc.v[i] += w00 * c00.v[i] + w01 * c01.v[i] + w10 * c10.v[i] + w11 * c11.v[i];
            }
        }
        acc.store(out + bqm * D + cg * VEC);
    }
}

}  // namespace mvdetr
