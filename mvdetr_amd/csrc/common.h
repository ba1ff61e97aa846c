// Shared helpers for the gfx950 kernels of libmvdetr_ops.so (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MVDETR_WAVE 64

namespace mvdetr {

template <typename T, int N> struct VecOf;
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<double, 1> { using type = double; };
template <> struct VecOf<double, 2> { using type = double2; };

// N consecutive scalars, loaded/stored with one 4/8/16-byte access when N > 1.
template <typename T, int N> struct Pack {
    T v[N];
    __device__ __forceinline__ static Pack load(const T *p) {
        Pack r;
        if constexpr (N == 1) {
            r.v[0] = *p;
        } else {
            using V = typename VecOf<T, N>::type;
            V t = *reinterpret_cast<const V *>(p);
            const T *e = reinterpret_cast<const T *>(&t);
#pragma unroll
            for (int i = 0; i < N; ++i) r.v[i] = e[i];
        }
        return r;
    }
    __device__ __forceinline__ void store(T *p) const {
        if constexpr (N == 1) {
            *p = v[0];
        } else {
            using V = typename VecOf<T, N>::type;
            V t;
            T *e = reinterpret_cast<T *>(&t);
#pragma unroll
            for (int i = 0; i < N; ++i) e[i] = v[i];
            *reinterpret_cast<V *>(p) = t;
        }
    }
    __device__ __forceinline__ static Pack zero() {
        Pack r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = T(0);
        return r;
    }
};

__device__ __forceinline__ float ffloor(float x) { return floorf(x); }
__device__ __forceinline__ double ffloor(double x) { return floor(x); }

// Bilinear footprint of one sampling point in one level: integer corner, fractional weights and
// per-corner validity (zero padding).  (h, w) are already in pixel units: loc*size - 0.5, the
// align_corners=False convention of grid_sample and of ms_deform_im2col_cuda.cuh:285-286.
template <typename T> struct Footprint {
    int y0, x0;
    T wy0, wy1, wx0, wx1;
    bool vy0, vy1, vx0, vx1;
};

template <typename T>
__device__ __forceinline__ Footprint<T> footprint(T h, T w, int H, int W) {
    Footprint<T> f;
    T fy = ffloor(h), fx = ffloor(w);
    f.y0 = (int)fy;
    f.x0 = (int)fx;
    f.wy1 = h - fy;
    f.wx1 = w - fx;
    f.wy0 = T(1) - f.wy1;
    f.wx0 = T(1) - f.wx1;
    f.vy0 = f.y0 >= 0 && f.y0 < H;
    f.vy1 = f.y0 + 1 >= 0 && f.y0 + 1 < H;
    f.vx0 = f.x0 >= 0 && f.x0 < W;
    f.vx1 = f.x0 + 1 >= 0 && f.x0 + 1 < W;
    return f;
}

// The same footprint from a position already split into its integer and fractional part (see fused_px).
template <typename T>
__device__ __forceinline__ Footprint<T> footprint_split(T fy, T wy1, T fx, T wx1, int H, int W) {
    Footprint<T> f;
    f.y0 = (int)fy;
    f.x0 = (int)fx;
    f.wy1 = wy1;
    f.wx1 = wx1;
    f.wy0 = T(1) - wy1;
    f.wx0 = T(1) - wx1;
    f.vy0 = f.y0 >= 0 && f.y0 < H;
    f.vy1 = f.y0 + 1 >= 0 && f.y0 + 1 < H;
    f.vx0 = f.x0 >= 0 && f.x0 < W;
    f.vx1 = f.x0 + 1 >= 0 && f.x0 + 1 < W;
    return f;
}

// Pixel coordinate of a tap of the FUSED training backward, ref * size - 0.5 + off (reference point normalised, offset in pixels),
// in two parts: `fl` = its floor (an integer-valued float), `frac` = the bilinear weight of the far corner.  Formed as ONE fp32
// number -- (ref + off / size) * size - 0.5, the module's own arithmetic -- a position near 143.5 carries half an ulp = 7.6e-6 px
// of rounding, which a blend of four <grad_out, value> dots of +-19 turns into 3e-4 of grad_attn_weight (round 4's soak).  Here the
// reference point's own texel c = rint(ref * size - 0.5) is split off exactly (fma), so the roundings that remain happen at the
// magnitude of the offset (a few pixels: <= 5e-7 px).  `x` = fl + frac is the coarse position for window / image tests.
__device__ __forceinline__ void fused_px(float ref, float off, float size, float &x, float &fl, float &frac)
{
    const float c = rintf(__fmaf_rn(ref, size, -0.5f));
    const float r = __fmaf_rn(ref, size, -(c + 0.5f));
    const float t = r + off;
    const float ft = floorf(t);
    fl = c + ft;
    frac = t - ft;
    x = fl + frac;
}

// A 16-byte global load that yields zeros when `ok` is false, without a branch and without touching `p` then: the address
// is replaced by `safe` (any readable address) and the VALUE is selected afterwards.  (`ok ? *p : zero` makes hipcc select
// between the global pointer and a zero it puts on the stack -- a flat load and scratch in kernels that need none.)
__device__ __forceinline__ float4 load4_or_zero(const float *p, bool ok, const float *safe)
{
    const float4 v = *reinterpret_cast<const float4 *>(ok ? p : safe);
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of the
// wave (s_waitcnt vmcnt(0)): prefetched sampling data, fire-and-forget stores and atomics, a window copy that nothing behind
// the barrier depends on yet.  Use it where only LDS reads / writes of other waves must have completed.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A value computed once per DEVICE (the grid size of a persistent kernel: CU count x workgroups per CU, and the one-time
// hipFuncSetAttribute for its dynamic LDS, which is per device as well).  A function-local `static int` made the first caller's
// device decide for every later one -- a process that drives several GPUs (DataParallel, one thread per device) launched the wrong
// grid on the others, and the deterministic backward's job partition depends on the grid.  Lock-free: a race computes the value twice.
template <typename T> struct PerDevice {
    static constexpr int MAX_DEVICES = 64;
    T value[MAX_DEVICES];
    bool ready[MAX_DEVICES];
    template <typename F> T get(F compute)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return compute();
        if (!__atomic_load_n(&ready[dev], __ATOMIC_ACQUIRE)) {
            value[dev] = compute();
            __atomic_store_n(&ready[dev], true, __ATOMIC_RELEASE);
        }
        return value[dev];
    }
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace mvdetr
