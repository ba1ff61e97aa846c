// Multi-scale deformable attention backward, grad_value for deformable-ENCODER calls -- gfx950 (MI355X).
//
// The generic backward (msda_backward.hip) sends every bilinear corner of every tap to memory as fp32
// atomics: 1.08 G dword atomics per launch at Wildtrack size, and the L2 atomic units retire ~0.3 T of them
// per second -- 3.15 ms, 2 % of the HBM roofline, whatever the lane mapping.  LDS fp32 atomics are no way
// out (ds_add_f32: ~190 cycles per wave instruction on this part, tools/experiments/lds_atomic_rate.hip),
// but LDS *integer* atomics run at ~17 cycles per wave instruction.  So, for encoder-shaped calls (the
// queries are the value tokens, offsets are a few pixels; same shapes the forward tile kernels take):
//
//   * a workgroup owns a (tile of cells, 128-byte channel slice) and, per source level, a window of that
//     level's grad_value in LDS as 32-bit FIXED-POINT accumulators (channel-major, so the 64 lanes of an
//     atomic instruction -- neighbouring cells, hence neighbouring tokens -- fall into different banks);
//   * the taps of ALL cameras' queries of the tile (same window: equal level shapes) are added with
//     ds_add_u32; the scale is a power of two chosen per workgroup job from a bound on the largest possible
//     sum, so the accumulator can not overflow for ANY input, and the quantisation step is
//     <= 2^-20 of that bound -- below fp32 atomics' own order-dependent rounding for these sums;
//   * the window is then flushed once with fp32 atomics: whole 128-byte channel runs, zeros skipped --
//     ~20x fewer memory-side atomics than tap by tap;
//   * taps that leave the window, and jobs whose bound is not finite, take the direct fp32-atomic path, so
//     the result is right for any sampling locations.
//
// grad_sampling_loc / grad_attn_weight come from msda_backward_sampling.hip (LDS-staged value windows).
// Unequal level shapes are detected on the device; this launch then runs the lane-group backward for all three
// gradients itself (msda_backward_lanes.h) and the sampling kernel stands down.
//
// Replaces (with msda_backward.hip) ms_deformable_col2im_cuda's grad_value accumulation
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-152,301-920).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_backward_lanes.h"
#include <stdlib.h>
#include <string.h>

#ifdef MVDETR_BWD_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup's waves
__device__ unsigned long long g_bwd_trace[2048];
extern "C" int mvdetr_debug_bwd_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_trace), n * sizeof(unsigned long long));
}
#define BTRACE(i) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (i) < 2048) g_bwd_trace[(i)] = wall_clock64(); } while (0)
#else
#define BTRACE(i) do { } while (0)
#endif

namespace mvdetr {

template <typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void msda_bwd_value_tile(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const int *__restrict__ local_hits)
{
    extern __shared__ __attribute__((aligned(16))) int win[];        // [SLICE][NTOKP] fixed-point accumulators
    __shared__ float red[Cfg::THREADS / 64];
    constexpr int D = Cfg::D, TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW, SLICE = Cfg::SLICE, P = TILE_P;
    constexpr int LCH = SLICE / 2, NTOK = WH * WW, NTOKP = NTOK | 1, THREADS = Cfg::THREADS;
    static_assert(SLICE == 32 && LCH == 16, "lanes own 16 channels of a 32-channel slice");
    const int tid = threadIdx.x;
    const int HS = M * D / SLICE;
    const int64_t row = (int64_t)M * D;
    const int sub = tid & 1, qi = tid >> 1, qly = qi / TW, qlx = qi % TW, lane_off = sub * LCH;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    // local_hits: how many of msda_locality_probe's sampled taps stay near their own query cell; too few and
    // windows are pointless (every tap would take the far path) -- same stand-down as for unequal shapes
    if (local_hits && *local_hits * 2 < MSDA_PROBE_SAMPLES) equal = false;
    if (!equal) {
        // not this kernel's case: the lane-group backward (msda_backward_lanes.h) does all three gradients here,
        // and msda_bwd_sampling_tile, which sees the same shapes, stands down
        const int64_t total = (int64_t)B * S * M * D;
        for (int64_t base = (int64_t)blockIdx.x * THREADS; base < total; base += (int64_t)gridDim.x * THREADS)
            msda_bwd_lanes_body<float, 1, D, true>(base + tid, go, value, shapes, lsi, loc, aw, B, S, M, D, L, S, P,
                                                   grad_value, grad_loc, grad_aw);
        return;
    }

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * HS * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;

    int *const wsum = win + SLICE * NTOKP;                    // [2][NTOKP]: per (16-channel half, token) weight mass
    for (int i = tid; i < (SLICE + 2) * NTOKP; i += THREADS) win[i] = 0;
    __syncthreads();

    // block-wide maximum of a non-negative value (NaN-free); two barriers
    auto block_max = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        __syncthreads();                                      // previous readers of `red` are done
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        float m = red[0];
#pragma unroll
        for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, red[i]);
        return m;
    };

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        const int hs = job % HS, u2 = job / HS;
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int ch0 = hs * SLICE + lane_off, head = ch0 / D;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qi < TH * TW && qy < Hq && qx < Wq;
        const int64_t cell = active ? (int64_t)qy * Wq + qx : 0;
        auto query = [&](int c) { return (int64_t)b * S + lsi[c] + cell; };
        const int oy = Y0 + TH / 2 - WH / 2, ox = X0 + TW / 2 - WW / 2;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        // ---- largest |grad_out| of the job (inf if any is not finite) ----
        float gmax = 0.f;
        if (active) {
            for (int c = 0; c < L; ++c) {
                const float *gp = go + query(c) * row + ch0;
#pragma unroll
                for (int k = 0; k < LCH; k += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(gp + k);
                    gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) gmax = INFINITY;     // NaN
                }
            }
        }
        BTRACE((tid >> 6) * 128 + 120);
        const float Gmax = block_max(gmax);
        BTRACE((tid >> 6) * 128 + 121);
        if (Gmax == 0.f) continue;                            // all-zero upstream gradient: nothing to add

        for (int l = 0; l < L; ++l) {
            const int64_t level_base = ((int64_t)b * S + lsi[l]) * row;
            const int tr = (tid >> 6) * 128 + l * 16;
            BTRACE(tr + 0);
            // ---- largest sum_p |aw[l][p]| of the job: bounds the weight-mass pass below ----
            float al = 0.f;
            if (active) {
                for (int c = 0; c < L; ++c) {
                    const float4 v = *reinterpret_cast<const float4 *>(aw + ((query(c) * M + head) * L + l) * P);
                    const float s4 = (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
                    al = fmaxf(al, s4 == s4 ? s4 : INFINITY);
                }
            }
            const float Amax = block_max(al);
            BTRACE(tr + 1);
            if (Amax == 0.f) continue;                        // all weights of this level are zero
            // non-finite inputs: no fixed point; every tap goes to memory as fp32 atomics (same NaN/inf results)
            const bool direct_only = !(Gmax < INFINITY && Amax < INFINITY);
            // weight-mass fixed point: a lane adds at most Amax per token, TH*TW*L lanes -> < 2^30
            int ew = 0;
            (void)frexpf(Amax * (float)(TH * TW * L), &ew);
            ew = ew < -60 ? -60 : ew;
            const float wscale = ldexpf(1.f, 30 - ew);
            float scale = 0.f, inv_scale = 0.f;

            // pass 0: per (token, channel half) the mass  sum |aw| * bilinear weight  of the taps landing there --
            //         |grad_value contribution| <= Gmax * mass, a tight bound, so the accumulators of pass 1 can
            //         use (almost) all 31 bits: quantisation ~1e-9 of the largest |grad_out|.
            // pass 1: the accumulation itself, and the taps outside the window.
            for (int pass = direct_only ? 1 : 0; pass < 2; ++pass) {
                float4 g4[4] = {}, la, lb, wa;
                int64_t gofs = 0;
                auto load_cam = [&](int c) {
                    const int64_t q = query(c);
                    gofs = q * row + ch0;
                    if (pass == 1) {
                        const float *gp = go + gofs;
#pragma unroll
                        for (int k = 0; k < 4; ++k) g4[k] = *reinterpret_cast<const float4 *>(gp + 4 * k);
                    }
                    const float *lp = loc + ((q * M + head) * L + l) * P * 2;
                    la = *reinterpret_cast<const float4 *>(lp);
                    lb = *reinterpret_cast<const float4 *>(lp + 4);
                    wa = *reinterpret_cast<const float4 *>(aw + ((q * M + head) * L + l) * P);
                };
                load_cam(0);                  // inactive lanes read cell 0's data and add nothing
                BTRACE(tr + 2 + pass * 4);
                for (int c = 0; c < L; ++c) {
                    const float g[16] = {g4[0].x, g4[0].y, g4[0].z, g4[0].w, g4[1].x, g4[1].y, g4[1].z, g4[1].w,
                                         g4[2].x, g4[2].y, g4[2].z, g4[2].w, g4[3].x, g4[3].y, g4[3].z, g4[3].w};
                    const float xs[4] = {la.x * fW - 0.5f, la.z * fW - 0.5f, lb.x * fW - 0.5f, lb.z * fW - 0.5f};
                    const float ys[4] = {la.y * fH - 0.5f, la.w * fH - 0.5f, lb.y * fH - 0.5f, lb.w * fH - 0.5f};
                    const float as[4] = {wa.x, wa.y, wa.z, wa.w};
                    const int64_t my_gofs = gofs;
                    if (c + 1 < L) load_cam(c + 1);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = xs[p], y = ys[p], a = as[p];
                        const bool in_window = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                        if (active && !direct_only && in_window) {
                            const float fx = floorf(x), fy = floorf(y);
                            const int tok = ((int)fy - oy) * WW + ((int)fx - ox);
                            const float wx1 = x - fx, wy1 = y - fy;
                            if (pass == 0) {
                                const float s = fabsf(a) * wscale, ay1 = wy1 * s, ay0 = s - ay1;
                                const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                                int *w0 = wsum + sub * NTOKP + tok;       // rounded UP: the mass is an upper bound
                                __hip_atomic_fetch_add(w0, __float2int_ru(w00), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(w0 + 1, __float2int_ru(w01), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(w0 + WW, __float2int_ru(w10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(w0 + WW + 1, __float2int_ru(w11), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            } else {
                                const float s = a * scale, ay1 = wy1 * s, ay0 = s - ay1;
                                const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                                int *w0 = win + lane_off * NTOKP + tok;
#pragma unroll
                                for (int k = 0; k < LCH; ++k) {
                                    int *wk = w0 + k * NTOKP;
                                    __hip_atomic_fetch_add(wk, __float2int_rn(w00 * g[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(wk + 1, __float2int_rn(w01 * g[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(wk + WW, __float2int_rn(w10 * g[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(wk + WW + 1, __float2int_rn(w11 * g[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                        if (pass == 0) continue;
                        // taps outside the window (or every tap of a non-finite level): straight to memory, the whole
                        // wave working on four of them at a time with lanes = channels (whole 64-byte runs per atomic)
                        const bool miss = active && (direct_only || !in_window) && y > -1.f && x > -1.f && y < fH && x < fW;
                        unsigned long long pend = __ballot(miss);
                        while (pend) {
                            int src = -1;
#pragma unroll
                            for (int grp = 0; grp < 4; ++grp) {
                                const int s0 = pend ? __ffsll((long long)pend) - 1 : -1;
                                pend &= pend - 1;                      // (0 stays 0)
                                if ((tid & 63) >> 4 == grp) src = s0;
                            }
                            const int ss = src < 0 ? 0 : src;
                            const float sx = __shfl(x, ss, 64), sy = __shfl(y, ss, 64), sa = __shfl(a, ss, 64);
                            const int64_t sg = __shfl(my_gofs, ss, 64);
                            const int sch = __shfl(ch0, ss, 64);      // the source lane's first channel in the token row
                            if (src >= 0) {
                                const int j = tid & 15;
                                const float gk = go[sg + j];
                                const Footprint<float> f = footprint(sy, sx, Hq, Wq);
                                float *p00 = grad_value + level_base + sch + ((int64_t)f.y0 * Wq + f.x0) * row + j;
                                const float ga = gk * sa;
                                if (f.vy0 && f.vx0) atomicAdd(p00, f.wy0 * f.wx0 * ga);
                                if (f.vy0 && f.vx1) atomicAdd(p00 + row, f.wy0 * f.wx1 * ga);
                                if (f.vy1 && f.vx0) atomicAdd(p00 + (int64_t)Wq * row, f.wy1 * f.wx0 * ga);
                                if (f.vy1 && f.vx1) atomicAdd(p00 + (int64_t)Wq * row + row, f.wy1 * f.wx1 * ga);
                            }
                        }
                    }
                }
                BTRACE(tr + 3 + pass * 4);
                if (pass == 0) {
                    __syncthreads();
                    BTRACE(tr + 4);
                    // largest mass in the window (and leave the mass array zeroed for the next level)
                    int wm = 0;
                    for (int i = tid; i < 2 * NTOKP; i += THREADS) {
                        wm = max(wm, wsum[i]);
                        wsum[i] = 0;
                    }
                    const float Wmax = block_max((float)wm) * (1.f + 1e-6f) / wscale;       // (int -> float rounding)
                    // any accumulator's final |sum| <= Gmax * Wmax = m * 2^e, m < 1; nearest rounding adds < 2^14 steps
                    int e = 0;
                    const float bound = Gmax * Wmax;
                    (void)frexpf(bound, &e);
                    e = !(bound < INFINITY) ? 129 : e < -90 ? -90 : e;
                    scale = ldexpf(1.f, 30 - e);
                    inv_scale = ldexpf(1.f, e - 30);
                    BTRACE(tr + 5);
                }
            }
            __syncthreads();
            BTRACE(tr + 8);
            // ---- flush: whole channel runs of the touched tokens, fp32 atomics; leaves the window zeroed ----
            if (!direct_only) {
                for (int i = tid; i < NTOK * SLICE; i += THREADS) {
                    const int tok = i / SLICE, ch = i % SLICE;
                    const int v = win[ch * NTOKP + tok];
                    if (v != 0) {
                        win[ch * NTOKP + tok] = 0;
                        const int gy = oy + tok / WW, gx = ox + tok % WW;
                        // corners outside the level were accumulated like any other and are dropped here (zero padding)
                        if ((unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq)
                            atomicAdd(grad_value + ((int64_t)b * S + lsi[l] + (int64_t)gy * Wq + gx) * row + hs * SLICE + ch,
                                      (float)v * inv_scale);
                    }
                }
            }
            BTRACE(tr + 9);
        }
    }
}

// ---- second formulation: finer jobs, desynchronised workgroups -------------------------------------------------------
// The kernel above is one round of 480 long jobs at Wildtrack size, every workgroup walking the same phase sequence
// in step (per level: 5 us weight bound, 7 us mass pass, 36 us accumulation, 8 us flush -- tools/experiments/bwd_trace.py),
// a quarter of its lanes without a query (96-cell tiles in 128 query slots), and the LDS atomic unit idle outside the
// accumulation phase.  Here a job is (tile of 4x32 cells, 16-channel slice, SOURCE LEVEL): 5,040 jobs, three 2-wave
// workgroups per CU that drift apart, so one workgroup's loads / flush run under another's atomics.  A wave is two rows
// of 32 cells: undisturbed, the 64 lanes of an atomic fall on each bank exactly twice (the floor for 64 lanes).
// The mass pass reads its sampling data of all cameras in one go (one memory latency instead of one per camera).
template <int D>
__global__ __launch_bounds__(256, 3) void msda_bwd_value_win(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const int *__restrict__ local_hits)
{
    constexpr int TH = 4, TW = 32, R = 6, WH = TH + 2 * R, WW = TW + 2 * R, LCH = 16, P = TILE_P, THREADS = 256;
    constexpr int NTOK = WH * WW, NTOKP = NTOK + 4;           // channel stride = 4 (mod 32): the flush reads conflict-free
    static_assert(NTOK % 32 == 0 && D % LCH == 0, "window / slice geometry");
    // [LCH/2][NTOKP] 64-bit words, each TWO 32-bit fixed-point accumulators (channels k and k+8: one ds_add_u64 adds
    // hi * 2^32 + lo, the sum is exact mod 2^64, and |sum of lo| < 2^31 by the bound below, so both halves come apart
    // again at the flush) -- a conflicted LDS atomic costs the same for 8 bytes as for 4 (tools/experiments/
    // lds_atomic_rate.hip: 9.8 vs 8.8 ns per wave instruction with displaced taps).  Then [NTOKP] ints of weight mass.
    extern __shared__ __attribute__((aligned(16))) long long win64[];
    __shared__ float red[2][4];
    auto pack2 = [](float lo_f, float hi_f) {
        const int lo = __float2int_rn(lo_f), hi = __float2int_rn(hi_f) + (lo >> 31);
        return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    const int tid = threadIdx.x;
    const int HS = M * D / LCH;
    const int64_t row = (int64_t)M * D;
    // waves 0/1: the tile's rows 0-1 / 2-3, channel pairs 0-3 of the slice; waves 2/3: the same cells, channel pairs 4-7
    // (pair k = channels k and k + 8) -- twice the waves for the same LDS, and a lone wave gets half a SIMD's issue rate
    const int chalf = tid >> 7, qly = (tid & 127) / TW, qlx = tid % TW;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (local_hits && *local_hits * 2 < MSDA_PROBE_SAMPLES) equal = false;
    if (!equal) {                                             // see msda_bwd_value_tile
        const int64_t total = (int64_t)B * S * M * D;
        for (int64_t base = (int64_t)blockIdx.x * THREADS; base < total; base += (int64_t)gridDim.x * THREADS)
            msda_bwd_lanes_body<float, 1, D, true>(base + tid, go, value, shapes, lsi, loc, aw, B, S, M, D, L, S, P,
                                                   grad_value, grad_loc, grad_aw);
        return;
    }

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * HS * B * L, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;

    int *const wsum = reinterpret_cast<int *>(win64 + (LCH / 2) * NTOKP);
    for (int i = tid; i < (LCH + 1) * NTOKP; i += THREADS) reinterpret_cast<int *>(win64)[i] = 0;
    __syncthreads();

    // block-wide maxima of two non-negative values (NaN-free); two barriers
    auto block_max2 = [&](float &a, float &b) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = fmaxf(a, __shfl_xor(a, o, 64));
            b = fmaxf(b, __shfl_xor(b, o, 64));
        }
        __syncthreads();
        if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
        __syncthreads();
        a = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        b = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    };

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        const int l = job % L, u1 = job / L;                  // the levels of one (tile, slice) run back to back: same grad_out
        const int hs = u1 % HS, u2 = u1 / HS;
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int ch0 = hs * LCH, head = ch0 / D;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qy < Hq && qx < Wq;
        const int64_t cell = active ? (int64_t)qy * Wq + qx : 0;
        auto query = [&](int c) { return (int64_t)b * S + lsi[c] + cell; };
        const int oy = Y0 + TH / 2 - WH / 2, ox = X0 + TW / 2 - WW / 2;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        const int64_t level_base = ((int64_t)b * S + lsi[l]) * row;
        const int tr = ((t - (int)blockIdx.x) / (int)gridDim.x) * 64 + (tid >> 6) * 16;
        BTRACE(tr + 0);

        // ---- pass 0: bounds.  Gmax = largest |grad_out| of the job (inf if any is not finite), Amax = largest
        //      sum_p |aw[l][p]|; then, per window token, the mass  sum |aw| * bilinear weight  of the taps landing there:
        //      |grad_value contribution| <= Gmax * mass, a tight bound, so the accumulators of pass 1 can use (almost) all
        //      31 bits: quantisation ~1e-9 of the largest |grad_out|.  The sampling data of (up to) eight cameras is read
        //      in one go and stays in registers from the bounds to the mass atomics (more cameras: read again).
        //      The two wave pairs of a cell (channel halves) take alternate cameras, so each tap is counted once.
        float4 la8[4], lb8[4], wa8[4];
        auto load8 = [&](int c0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 2 * k + chalf < L ? c0 + 2 * k + chalf : L - 1;
                const int64_t q = query(c);
                const float *lp = loc + ((q * M + head) * L + l) * P * 2;
                la8[k] = *reinterpret_cast<const float4 *>(lp);
                lb8[k] = *reinterpret_cast<const float4 *>(lp + 4);
                wa8[k] = *reinterpret_cast<const float4 *>(aw + ((q * M + head) * L + l) * P);
            }
        };
        float gmax = 0.f, al = 0.f;
        for (int c0 = 0; c0 < L; c0 += 8) {
            load8(c0);
            float m = 0.f;
            for (int c = c0; c < L && c < c0 + 8; ++c) {
                const float *gp = go + query(c) * row + ch0 + 4 * chalf;
#pragma unroll
                for (int j = 0; j < LCH; j += 8) {
                    const float4 v = *reinterpret_cast<const float4 *>(gp + j);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) m = INFINITY;     // NaN
                }
            }
            float s8 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float s4 = (fabsf(wa8[k].x) + fabsf(wa8[k].y)) + (fabsf(wa8[k].z) + fabsf(wa8[k].w));
                s8 = fmaxf(s8, s4 == s4 ? s4 : INFINITY);
            }
            if (active) { gmax = fmaxf(gmax, m); al = fmaxf(al, s8); }
        }
        BTRACE(tr + 1);
        float Gmax = gmax, Amax = al;
        block_max2(Gmax, Amax);
        if (Gmax == 0.f || Amax == 0.f) continue;             // nothing to add (block-uniform)
        // non-finite inputs: no fixed point; every tap goes to memory as fp32 atomics (same NaN/inf results)
        const bool direct_only = !(Gmax < INFINITY && Amax < INFINITY);
        float scale = 0.f, inv_scale = 0.f;
        if (!direct_only) {
            // weight-mass fixed point: a lane adds at most Amax per token, TH*TW*L lanes -> < 2^30 (+ 2^15 for rounding up)
            int ew = 0;
            (void)frexpf(Amax * (float)(TH * TW * L), &ew);
            ew = ew < -60 ? -60 : ew;
            const float wscale = ldexpf(1.f, 30 - ew);
            for (int c0 = 0; c0 < L; c0 += 8) {
                if (L > 8) load8(c0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (c0 + 2 * k + chalf >= L) break;
                    const float xs[4] = {la8[k].x * fW - 0.5f, la8[k].z * fW - 0.5f, lb8[k].x * fW - 0.5f, lb8[k].z * fW - 0.5f};
                    const float ys[4] = {la8[k].y * fH - 0.5f, la8[k].w * fH - 0.5f, lb8[k].y * fH - 0.5f, lb8[k].w * fH - 0.5f};
                    const float as[4] = {wa8[k].x, wa8[k].y, wa8[k].z, wa8[k].w};
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = xs[p], y = ys[p];
                        const bool in_window = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                        if (active && in_window) {
                            const float fx = floorf(x), fy = floorf(y);
                            const int tok = ((int)fy - oy) * WW + ((int)fx - ox);
                            const float wx1 = x - fx, wy1 = y - fy;
                            const float s = fabsf(as[p]) * wscale, ay1 = wy1 * s, ay0 = s - ay1;
                            const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                            int *w0 = wsum + tok;                     // rounded UP: the mass is an upper bound
                            __hip_atomic_fetch_add(w0, __float2int_ru(w00), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(w0 + 1, __float2int_ru(w01), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(w0 + WW, __float2int_ru(w10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(w0 + WW + 1, __float2int_ru(w11), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            __syncthreads();
            // largest mass in the window (and leave the mass array zeroed for the next job)
            int wm = 0;
            for (int i = tid; i < NTOK; i += THREADS) {
                wm = max(wm, wsum[i]);
                wsum[i] = 0;
            }
            float Wmax = (float)wm, unused = 0.f;
            block_max2(Wmax, unused);
            Wmax = Wmax * (1.f + 1e-6f) / wscale;             // (int -> float rounding)
            // any accumulator's final |sum| <= Gmax * Wmax = m * 2^e, m < 1; nearest rounding adds < 2^14 steps
            const float bound = Gmax * Wmax;
            int e = 0;
            (void)frexpf(bound, &e);
            e = !(bound < INFINITY) ? 129 : e < -90 ? -90 : e;
            scale = ldexpf(1.f, 30 - e);
            inv_scale = ldexpf(1.f, e - 30);
        }
        BTRACE(tr + 2);

        // ---- pass 1: the accumulation, and the taps outside the window ----
        {
            float4 g4[2], la, lb, wa;
            int64_t gofs = 0;
            auto load_cam = [&](int c) {
                const int64_t q = query(c);
                gofs = q * row + ch0;
                const float *gp = go + gofs + 4 * chalf;
                g4[0] = *reinterpret_cast<const float4 *>(gp);        // channels 4*chalf .. +3
                g4[1] = *reinterpret_cast<const float4 *>(gp + 8);    // and their partners, + 8
                const float *lp = loc + ((q * M + head) * L + l) * P * 2;
                la = *reinterpret_cast<const float4 *>(lp);
                lb = *reinterpret_cast<const float4 *>(lp + 4);
                wa = *reinterpret_cast<const float4 *>(aw + ((q * M + head) * L + l) * P);
            };
            load_cam(0);                      // inactive lanes read cell 0's data and add nothing
            for (int c = 0; c < L; ++c) {
                const float g[8] = {g4[0].x, g4[0].y, g4[0].z, g4[0].w, g4[1].x, g4[1].y, g4[1].z, g4[1].w};
                const float xs[4] = {la.x * fW - 0.5f, la.z * fW - 0.5f, lb.x * fW - 0.5f, lb.z * fW - 0.5f};
                const float ys[4] = {la.y * fH - 0.5f, la.w * fH - 0.5f, lb.y * fH - 0.5f, lb.w * fH - 0.5f};
                const float as[4] = {wa.x, wa.y, wa.z, wa.w};
                const int64_t my_gofs = gofs;
                if (c + 1 < L) load_cam(c + 1);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float x = xs[p], y = ys[p], a = as[p];
                    const bool in_window = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                    if (active && !direct_only && in_window) {
                        const float fx = floorf(x), fy = floorf(y);
                        const int tok = ((int)fy - oy) * WW + ((int)fx - ox);
                        const float wx1 = x - fx, wy1 = y - fy;
                        const float s = a * scale, ay1 = wy1 * s, ay0 = s - ay1;
                        const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                        long long *w0 = win64 + chalf * 4 * NTOKP + tok;
#pragma unroll
                        for (int k = 0; k < LCH / 4; ++k) {
                            long long *wk = w0 + k * NTOKP;
                            __hip_atomic_fetch_add(wk, pack2(w00 * g[k], w00 * g[k + 4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(wk + 1, pack2(w01 * g[k], w01 * g[k + 4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(wk + WW, pack2(w10 * g[k], w10 * g[k + 4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(wk + WW + 1, pack2(w11 * g[k], w11 * g[k + 4]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    // taps outside the window (or every tap of a non-finite job): straight to memory, the whole wave
                    // working on eight of them at a time with lanes = this wave's eight channels
                    const bool miss = active && (direct_only || !in_window) && y > -1.f && x > -1.f && y < fH && x < fW;
                    unsigned long long pend = __ballot(miss);
                    while (pend) {
                        int src = -1;
#pragma unroll
                        for (int grp = 0; grp < 8; ++grp) {
                            const int s0 = pend ? __ffsll((long long)pend) - 1 : -1;
                            pend &= pend - 1;                      // (0 stays 0)
                            if ((tid & 63) >> 3 == grp) src = s0;
                        }
                        const int ss = src < 0 ? 0 : src;
                        const float sx = __shfl(x, ss, 64), sy = __shfl(y, ss, 64), sa = __shfl(a, ss, 64);
                        const int64_t sg = __shfl(my_gofs, ss, 64);
                        if (src >= 0) {
                            const int j = 4 * chalf + (tid & 3) + 2 * (tid & 4);     // channels 4*chalf..+3 and their partners
                            const float gk = go[sg + j];
                            const Footprint<float> f = footprint(sy, sx, Hq, Wq);
                            float *p00 = grad_value + level_base + ch0 + ((int64_t)f.y0 * Wq + f.x0) * row + j;
                            const float ga = gk * sa;
                            if (f.vy0 && f.vx0) atomicAdd(p00, f.wy0 * f.wx0 * ga);
                            if (f.vy0 && f.vx1) atomicAdd(p00 + row, f.wy0 * f.wx1 * ga);
                            if (f.vy1 && f.vx0) atomicAdd(p00 + (int64_t)Wq * row, f.wy1 * f.wx0 * ga);
                            if (f.vy1 && f.vx1) atomicAdd(p00 + (int64_t)Wq * row + row, f.wy1 * f.wx1 * ga);
                        }
                    }
                }
            }
        }
        BTRACE(tr + 3);
        __syncthreads();
        BTRACE(tr + 4);
        // ---- flush: 64-byte channel runs of the touched tokens, fp32 atomics; leaves the window zeroed ----
        if (!direct_only) {
            const int ch = tid % LCH, pair = ch & 7;
            const bool upper = ch >= 8;
            float *const gbase = grad_value + level_base + ch0 + ch;
            for (int i0 = tid / LCH; i0 < NTOK; i0 += 8 * (THREADS / LCH)) {
                long long v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int tok = i0 + k * (THREADS / LCH);
                    v[k] = tok < NTOK ? win64[pair * NTOKP + tok] : 0;          // both lanes of a pair read the word ...
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int tok = i0 + k * (THREADS / LCH);
                    if (v[k] != 0) {
                        if (!upper) win64[pair * NTOKP + tok] = 0;              // ... (same instruction) and one clears it
                        const int lo = (int)v[k], hi = (int)((v[k] - (long long)lo) >> 32);
                        const int mine = upper ? hi : lo;
                        const int gy = oy + tok / WW, gx = ox + tok % WW;
                        // corners outside the level were accumulated like any other and are dropped here (zero padding)
                        if (mine != 0 && (unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq)
                            atomicAdd(gbase + ((int64_t)gy * Wq + gx) * row, (float)mine * inv_scale);
                    }
                }
            }
        }
        BTRACE(tr + 5);
        __syncthreads();                                      // the window is zero again before the next job's atomics
    }
}

// Samples MSDA_PROBE_SAMPLES taps spread over the whole call and counts those within MSDA_PROBE_RADIUS pixels of
// their own query's cell (equal level shapes assumed; with unequal ones the count is ignored anyway).
__global__ __launch_bounds__(256) void msda_locality_probe(const float *__restrict__ loc, const int64_t *__restrict__ shapes,
                                                           int B, int S, int M, int L, int *__restrict__ hits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int H = (int)shapes[0], W = (int)shapes[1];
    const int64_t taps = (int64_t)B * S * M * L * TILE_P;
    // a fixed odd stride walks the tap index space evenly
    const int64_t t = (int64_t)(((unsigned long long)i * 0x9E3779B97F4A7C15ull) % (unsigned long long)taps);
    const int64_t bq = t / ((int64_t)M * L * TILE_P);
    const int cell = (int)((bq % S) % ((int64_t)H * W));
    const float x = loc[2 * t] * (float)W - 0.5f, y = loc[2 * t + 1] * (float)H - 0.5f;
    const bool hit = i < MSDA_PROBE_SAMPLES && fabsf(x - (float)(cell % W)) <= MSDA_PROBE_RADIUS &&
                     fabsf(y - (float)(cell / W)) <= MSDA_PROBE_RADIUS;
    const int n = __syncthreads_count(hit);
    if (threadIdx.x == 0 && n) atomicAdd(hits, n);
}

int msda_launch_locality_probe(hipStream_t st, const float *loc, const int64_t *shapes, int B, int S, int M, int L, int *hits)
{
    hipError_t e = hipMemsetAsync(hits, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_locality_probe, dim3(MSDA_PROBE_SAMPLES / 256), dim3(256), 0, st, loc, shapes, B, S, M, L, hits);
    return (int)hipGetLastError();
}

using BWide16 = TileCfg<16, 32, 6, 16, 6, 256>;
using BWide32 = TileCfg<32, 32, 6, 16, 6, 256>;

template <typename Cfg>
static int launch_value_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    constexpr int LDS = (Cfg::SLICE + 2) * ((Cfg::WH * Cfg::WW) | 1) * 4;      // accumulators + weight mass
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_value_tile<Cfg>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_value_tile<Cfg>, Cfg::THREADS, LDS) != hipSuccess ||
            per_cu < 1)
            per_cu = 2;
        return (cus * per_cu + 7) / 8 * 8;
    }();
    hipLaunchKernelGGL((msda_bwd_value_tile<Cfg>), dim3((unsigned)blocks), dim3(Cfg::THREADS), LDS, st, go, value, shapes,
                       lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    return (int)hipGetLastError();
}

template <int D>
static int launch_value_win(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                            const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                            float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    constexpr int LDS = (16 + 1) * (16 * 44 + 4) * 4;         // accumulators + weight mass
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_value_win<D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_value_win<D>, 256, LDS) != hipSuccess || per_cu < 1)
            per_cu = 3;
        return (cus * per_cu + 7) / 8 * 8;
    }();
    hipLaunchKernelGGL((msda_bwd_value_win<D>), dim3((unsigned)blocks), dim3(256), LDS, st, go, value, shapes, lsi, loc, aw,
                       B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    return (int)hipGetLastError();
}

int msda_backward_value_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    static const bool old_kernel = [] { const char *e = getenv("MVDETR_MSDA_BWD_VALUE"); return e && !strcmp(e, "tile"); }();
    if (!old_kernel) {
        if (D == 16) return launch_value_win<16>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
        if (D == 32) return launch_value_win<32>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    }
    if (D == 16) return launch_value_tile<BWide16>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    if (D == 32) return launch_value_tile<BWide32>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    return (int)hipErrorInvalidValue;
}

}  // namespace mvdetr
