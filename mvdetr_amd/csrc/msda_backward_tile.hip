// Multi-scale deformable attention backward, grad_value for deformable-ENCODER calls -- gfx950 (MI355X).
//
// The generic backward (msda_backward.hip) sends every bilinear corner of every tap to memory as fp32
// atomics: 1.08 G dword atomics per launch at Wildtrack size, and the L2 atomic units retire ~0.3 T of them
// per second -- 3.15 ms, 2 % of the HBM roofline, whatever the lane mapping.  LDS fp32 atomics are no way
// out (ds_add_f32: 80 ns per wave instruction per CU whatever the addresses, tools/experiments/lds_atomic_rate.hip),
// but LDS *integer* atomics run at 1.8 ns (2.7 ns for 64-bit ones) when the lanes fall on different banks.  So, for
// encoder-shaped calls (the queries are the value tokens, offsets are a few pixels; the shapes the forward tile
// kernels take), msda_bwd_value_win:
//
//   * a job is (tile of 4x32 cells, 16-channel slice, SOURCE level); the workgroup keeps that level's 16x44-token
//     window of grad_value in LDS as 32-bit FIXED-POINT accumulators, channel-major (the 64 lanes of an atomic --
//     two rows of 32 neighbouring cells, hence neighbouring tokens -- fall on each bank exactly twice when the taps
//     are undisturbed), two channels (k, k+8) to a 64-bit word so that one ds_add_u64 adds both;
//   * the taps of ALL cameras' queries of the tile (same window: equal level shapes) are added that way; the scale
//     is a power of two chosen per job from data (a first pass over the taps accumulates, per token, the weight mass
//     landing there), so no accumulator can overflow for ANY input and the quantisation step is 2^-30 of a tight
//     bound -- below fp32 atomics' own order-dependent rounding for these sums;
//   * the window is then flushed once with fp32 atomics: whole 64-byte channel runs, zeros skipped -- ~20x fewer
//     memory-side atomics than tap by tap;
//   * taps that leave the window, and jobs whose bound is not finite, take the direct fp32-atomic path, so
//     the result is right for any sampling locations.
//
// History (Wildtrack size, this kernel alone): 526 us as one round of 480 long (tile 6x16, 128-byte slice, all levels)
// jobs with 32-bit atomics -- every workgroup in the same phase at the same time, a quarter of the lanes without a
// cell, the LDS atomic unit idle outside the accumulation phase (tools/experiments/bwd_trace.py); 478 us with the
// 5,040 finer jobs below on three 2-wave workgroups per CU; 450 us with two channels per ds_add_u64; 390 us with the
// channel pairs of a cell split over two waves (a lone wave gets half a SIMD's issue rate).  Tried and dropped: levels
// inside the job with the flush / mass pass / loads software-pipelined (480 us: the workgroups fall back into step);
// v_cvt_rpi instead of v_rndne + v_cvt (no change: the kernel waits on the LDS atomic unit, whose cost with taps
// displaced by a pixel or two is ~4x the conflict-free one).
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

#ifdef MVDETR_BWD_TRACE_PLANES
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

// FUSED = 1 (the fused training backward, msda_backward_fused.hip): `loc` is the module's RAW tensor [.., Lq, L, M/g,
// (g*P*2 offsets | g*P logits)] (query stride raw_q floats), `aw` the forward's softmax statistics [.., Lq, M, (max, 1/sum)],
// `ref` one reference point per (query, level), level-major [B or 1, L, Lq, 2]: locations and weights are recomputed per tap
// with the forward's own expressions, and the windows follow the tile's taps as in the forward (no probe).
template <int D, int FUSED>
__global__ __launch_bounds__(256, 3) void msda_bwd_value_win(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const int *__restrict__ local_hits, const float *__restrict__ ref, int64_t ref_bstride, int raw_q)
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
    if (local_hits && *local_hits * MSDA_PROBE_NEAR_DIV < MSDA_PROBE_SAMPLES) equal = false;
    if (FUSED && !equal) {
        // (the fused entry's callers promise equal level shapes: make the misuse loud)
        for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < (int64_t)B * S * M * D; i += (int64_t)gridDim.x * THREADS)
            grad_value[i] = __builtin_nanf("");
        return;
    }
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
    const int jobs = per_level * HS * B * L, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;
    const float iw = 1.f / fW, ih = 1.f / fH;
    // sampling data of (query q, this job's head, level l): locations (x, y) x 4 points in la / lb, weights in wa
    constexpr int HPS = 32 / D;
    auto fetch = [&](int64_t q, int b, int head, int l, float4 &la, float4 &lb, float4 &wa) {
        if constexpr (FUSED) {
            const float *rp = loc + q * raw_q + (l * (M / HPS) + head / HPS) * (HPS * P * 3);
            const float4 oa = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2);
            const float4 ob = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2 + 4);
            const float4 lg = *reinterpret_cast<const float4 *>(rp + HPS * P * 2 + (head % HPS) * P);
            const float2 r = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)l * S + (q - (int64_t)b * S)) * 2);
            const float2 st = *reinterpret_cast<const float2 *>(aw + (q * M + head) * 2);
            la = make_float4(r.x + oa.x * iw, r.y + oa.y * ih, r.x + oa.z * iw, r.y + oa.w * ih);
            lb = make_float4(r.x + ob.x * iw, r.y + ob.y * ih, r.x + ob.z * iw, r.y + ob.w * ih);
            wa = make_float4(__expf(lg.x - st.x) * st.y, __expf(lg.y - st.x) * st.y, __expf(lg.z - st.x) * st.y, __expf(lg.w - st.x) * st.y);
        } else {
            const float *lp = loc + ((q * M + head) * L + l) * P * 2;
            la = *reinterpret_cast<const float4 *>(lp);
            lb = *reinterpret_cast<const float4 *>(lp + 4);
            wa = *reinterpret_cast<const float4 *>(aw + ((q * M + head) * L + l) * P);
        }
    };

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
        int shx, shy;                                         // where this head's taps lie (locality probe)
        msda_probe_shift(local_hits, head, shx, shy);
        if constexpr (FUSED) {
            // no probe in front of the fused backward: every wave reduces the same sample (the tile's first two rows, camera
            // 0, this level) to the head's mean tap displacement, as the forward does
            const int sl = tid & 63;
            const int s_qy = Y0 + sl / TW, s_qx = X0 + sl % TW;
            float sx = 0.f, sy = 0.f, sn = 0.f;
            if (s_qy < Hq && s_qx < Wq) {
                float4 a0, b0, w0;
                fetch((int64_t)b * S + lsi[0] + (int64_t)s_qy * Wq + s_qx, b, head, l, a0, b0, w0);
                const float mx = 0.25f * ((a0.x + a0.z) + (b0.x + b0.z)) * fW - 0.5f - (float)s_qx;
                const float my = 0.25f * ((a0.y + a0.w) + (b0.y + b0.w)) * fH - 0.5f - (float)s_qy;
                if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sx += __shfl_xor(sx, o, 64);
                sy += __shfl_xor(sy, o, 64);
                sn += __shfl_xor(sn, o, 64);
            }
            const float tx_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
            const float ty_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            if (tn > 0.f) {
                shx = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf(tx_ / tn)));
                shy = max(-MSDA_PROBE_MAXSHIFT, min(MSDA_PROBE_MAXSHIFT, (int)rintf(ty_ / tn)));
            }
        }
        const int oy = Y0 + TH / 2 - WH / 2 + shy, ox = X0 + TW / 2 - WW / 2 + shx;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        const int64_t level_base = ((int64_t)b * S + lsi[l]) * row;
        [[maybe_unused]] const int tr = ((t - (int)blockIdx.x) / (int)gridDim.x) * 64 + (tid >> 6) * 16;
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
                fetch(query(c), b, head, l, la8[k], lb8[k], wa8[k]);
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
                fetch(q, b, head, l, la, lb, wa);
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

// Samples MSDA_PROBE_SAMPLES taps spread over the whole call -- block b samples head b % M -- and counts those within
// MSDA_PROBE_RADIUS pixels of their own query's cell (equal level shapes assumed; with unequal ones the count is ignored
// anyway); per head it also sums the displacement of the sampled taps from their cells (msda_dispatch.h).
__global__ __launch_bounds__(256) void msda_locality_probe(const float *__restrict__ loc, const int64_t *__restrict__ shapes,
                                                           int B, int S, int M, int L, int *__restrict__ probe)
{
    __shared__ int red[4][3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int H = (int)shapes[0], W = (int)shapes[1];
    const int head = blockIdx.x % M;
    const int64_t per_head = (int64_t)B * S * L * TILE_P;     // taps of one head
    // a fixed odd stride walks the (query, level, point) space of the block's head evenly
    const int64_t u = (int64_t)(((unsigned long long)i * 0x9E3779B97F4A7C15ull) % (unsigned long long)per_head);
    const int64_t bq = u / ((int64_t)L * TILE_P);
    const int64_t t = (bq * M + head) * L * TILE_P + (u - bq * L * TILE_P);
    const int cell = (int)((bq % S) % ((int64_t)H * W));
    const float x = loc[2 * t] * (float)W - 0.5f, y = loc[2 * t + 1] * (float)H - 0.5f;
    const float dx = x - (float)(cell % W), dy = y - (float)(cell / W);
    const bool in = i < MSDA_PROBE_SAMPLES;
    const bool hit = in && fabsf(dx) <= MSDA_PROBE_RADIUS && fabsf(dy) <= MSDA_PROBE_RADIUS;
    const bool near = in && fabsf(dx) <= 16.f && fabsf(dy) <= 16.f;      // (NaN: false)
    int sx = near ? (int)rintf(dx * 16.f) : 0, sy = near ? (int)rintf(dy * 16.f) : 0, sn = near ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
        sn += __shfl_xor(sn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; red[threadIdx.x >> 6][2] = sn; }
    const int n = __syncthreads_count(hit);
    if (threadIdx.x == 0) {
        if (n) atomicAdd(probe, n);
        if (head < MSDA_PROBE_MAXHEADS) {
            atomicAdd(probe + 1 + 3 * head, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
            atomicAdd(probe + 1 + 3 * head + 1, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
            atomicAdd(probe + 1 + 3 * head + 2, red[0][2] + red[1][2] + red[2][2] + red[3][2]);
        }
    }
}

int msda_launch_locality_probe(hipStream_t st, const float *loc, const int64_t *shapes, int B, int S, int M, int L, int *probe)
{
    hipError_t e = hipMemsetAsync(probe, 0, MSDA_PROBE_INTS * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_locality_probe, dim3(MSDA_PROBE_SAMPLES / 256), dim3(256), 0, st, loc, shapes, B, S, M, L, probe);
    return (int)hipGetLastError();
}

template <int D, int FUSED>
static int launch_value_win(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                            const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                            float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits, const float *ref,
                            int64_t ref_bstride, int raw_q)
{
    constexpr int LDS = (16 + 1) * (16 * 44 + 4) * 4;         // accumulators + weight mass
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_value_win<D, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_value_win<D, FUSED>, 256, LDS) != hipSuccess || per_cu < 1)
            per_cu = 3;
        return (cus * per_cu + 7) / 8 * 8;
    }();
    hipLaunchKernelGGL((msda_bwd_value_win<D, FUSED>), dim3((unsigned)blocks), dim3(256), LDS, st, go, value, shapes, lsi, loc, aw,
                       B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, ref, ref_bstride, raw_q);
    return (int)hipGetLastError();
}

// MVDETR_MSDA_BWD_VALUE = tokens (default: msda_bwd_value_tok, token-major windows) | planes (msda_bwd_value_win, for A/B)
static bool value_planes()
{
    static const bool planes = [] { const char *e = getenv("MVDETR_MSDA_BWD_VALUE"); return e && !strcmp(e, "planes"); }();
    return planes;
}

int msda_backward_value_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    // msda_bwd_value_tok addresses one batch element's tensors with 32-bit byte offsets: larger ones keep the 64-bit kernel
    const int64_t lim = (int64_t)1 << 32;
    if (!value_planes() && (int64_t)S * M * L * TILE_P * 2 * 4 < lim && (int64_t)S * M * D * 4 < lim)
        return msda_backward_value_tok(st, go, value, shapes, lsi, loc, aw, B, S, M, D, L, grad_value, grad_loc, grad_aw, local_hits);
    if (D == 16) return launch_value_win<16, 0>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, nullptr, 0, 0);
    if (D == 32) return launch_value_win<32, 0>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, nullptr, 0, 0);
    return (int)hipErrorInvalidValue;
}

// grad_value of the fused training backward: raw offsets / logits + the forward's statistics (see the kernel's header)
int msda_backward_value_tile_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                   const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                   const float *stats, int B, int S, int M, int D, int L, float *grad_value)
{
    const int64_t lim = (int64_t)1 << 32;
    if (!value_planes() && (int64_t)S * raw_q * 4 < lim && (int64_t)S * M * D * 4 < lim)
        return msda_backward_value_tok_fused(st, go, value, shapes, lsi, raw, raw_q, ref, ref_bstride, stats, B, S, M, D, L, grad_value);
    if (D == 16) return launch_value_win<16, 1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, nullptr, nullptr, nullptr, ref, ref_bstride, raw_q);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
