// Multi-scale deformable attention backward for deformable-ENCODER calls in ONE pass -- gfx950 (MI355X).
//
// Rounds 2-4 ran two kernels per backward: grad_value through fixed-point LDS windows (msda_bwd_value_tok) and the two
// sampling gradients from LDS-staged value windows (msda_bwd_sampling_resident / msda_bwd_fused_sampling).  Both fetched the
// sampling data and grad_out, both rebuilt every tap's position, window test and bilinear weights.  Here a job --
// (4 x TW-cell tile, ONE 16-channel head, source level l), all cameras' queries of those cells -- keeps BOTH windows of that
// level in LDS, token-major, 64 bytes per token:
//     vwin  value[level l] of the head               (staged by LDS-DMA, zero outside the level)
//     gwin  grad_value[level l] of the head           (32-bit fixed point, two channels per qword: see msda_bwd_value_tok)
// and every tap is set up ONCE.  A wave owns one row of the tile and walks it two cells at a time:
//   * lanes as taps (2 cells x 8 cameras x 4 points): pixel position, window test, the four corner weights -> (weight x
//     attention weight x scale, record offset) entries in a wave-private LDS table;
//   * lanes as (cell, corner, channel pair): per tap of the lane's cell one ds_read_b64 of the entry's VALUE record, one
//     ds_add_u64 into the same record of gwin, and the lane's share of <grad_out, value[corner]> -- two FMAs, summed over the
//     8 pair lanes with three DPP adds (the 8 lanes of a corner cover one token's 64-byte record: conflict-free for both
//     the read and the atomic wherever the tokens lie).  Lane `pair` keeps the dots of taps k = pair (mod 8);
//   * lanes as taps again: the four dots of the lane's tap come back through LDS (4 ds_write_b32 + 1 ds_read_b128 per lane
//     and step), and the tap's three gradients are formed and stored:
//         public contract   grad_attn_weight = da, grad_sampling_loc = (W a gx, H a gy)            (cuh:155-158)
//         fused training    grad_raw offsets = (a gx, a gy), logits = a (da - <grad_out, out>)     (+ the module's softmax and
//                                                                                                   location arithmetic)
//     with da = bilinear(d00..d11), gx = (1-wy)(d01-d00) + wy (d11-d10), gy likewise.
// Taps outside the window (rare: the windows follow the head's mean displacement) take both halves from global memory, the
// whole wave on one tap.  Bounds, scale, flush, non-finite jobs and unequal level shapes: msda_bwd_value_tok's.  No probe
// launch and no scratch: the window shift comes from the job's own sample, and a job of the public contract whose sample is
// far from its cells (fewer than a quarter of the sampled taps within MSDA_PROBE_RADIUS) runs the lane-group body for its
// own taps.
//
// Replaces ms_deformable_col2im_cuda and its kernels (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:
// 956-1327, 301-920, 87-234) -- which is also ONE pass per (b, q, m) -- and, for the fused entry, torch's backward of
// ms_deform_attn.py:100-107 around it.
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_backward_lanes.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <utility>

#ifndef MVDETR_SCATTER_WGS
#define MVDETR_SCATTER_WGS 3     // workgroups per CU of the grad_value-only instantiation (register budget 512 / this per lane)
#endif
#ifndef MVDETR_OP_STAGGER
#define MVDETR_OP_STAGGER 16      // x 64 cycles between the waves of a workgroup at the start of pass 1 (0 = none)
#endif

#ifdef MVDETR_BWD_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup's waves
__device__ unsigned long long g_op_trace[4096];
extern "C" int mvdetr_debug_onepass_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_op_trace), n * sizeof(unsigned long long));
}
#define OTRACE(i) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (i) < 4096) g_op_trace[(i)] = wall_clock64(); } while (0)
#else
#define OTRACE(i) do { } while (0)
#endif

namespace mvdetr {

namespace {

// pixel coordinate of a normalised location; ONE expression for every pass (a tap must be inside the window in all or in none)
__device__ __forceinline__ float op_pix(float loc, float size) { return __fmaf_rn(loc, size, -0.5f); }

template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes of an aligned group, in every lane: lane ^ 1 (quad_perm [1,0,3,2]), lane ^ 2 (quad_perm [2,3,0,1]),
// the other quad (row_half_mirror: lane i <-> 7 - i, whose quad already holds its own sum)
__device__ __forceinline__ float sum8(float v)
{
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v;
}

// Deterministic mode (DET): grad_value is summed in 64-bit fixed point with ONE binary point for the whole call, 38 bits below
// the product of the call's largest finite |grad_out| and largest finite |attention weight| (>= 1): integer adds commute, so the
// result does not depend on the order in which workgroups flush.  hdr[0], hdr[1]: those two maxima as float bits (msda_det_absmax).
__device__ __forceinline__ int det_binary_point(const unsigned *hdr)
{
    int eg = 0, ea = 0;
    (void)frexpf(__uint_as_float(hdr[0]), &eg);
    (void)frexpf(fmaxf(__uint_as_float(hdr[1]), 1.f), &ea);
    return 38 - (eg + ea);
}
constexpr int DET_HDR_BYTES = 256;

}  // namespace

// DOTS_ = 0: the grad_value half alone (no value window, no dot products: the sampling gradients come from another kernel)
template <int TH_, int TW_, int R_, int DOTS_ = 1, int MAXWGS_ = 8> struct OnePassCfg {
    static constexpr int DOTS = DOTS_;
    // a wave per tile row: TH rows of TW cells, TH waves
    static constexpr int TH = TH_, TW = TW_, R = R_, WH = TH + 2 * R, WW = TW + 2 * R;
    // LDS row stride of the windows in tokens: = 2 (mod 4), so that the four corners of a tap -- tokens t, t + 1, t + WWP,
    // t + WWP + 1 -- fall into the four different quarters of the 256-byte bank row and the 32 lanes of a (cell) half wave
    // read them without a conflict (WW = 28 put corners 0 / 2 and 1 / 3 on the same banks: 18 % of all LDS cycles)
    static constexpr int WWP = (DOTS && WW % 4 == 0) ? WW + 2 : WW;
    static constexpr int NSLOT = (WH * WWP + 15) / 16 * 16;             // whole LDS-DMA instructions (16 slots each)
    static constexpr int LCH = 16, NPAIR = 8, NW = TH, THREADS = 64 * NW, CAMS = 8, CELLS = TH * TW;
    static constexpr int TSTRIDE = 34, TAB = 2 * 4 * TSTRIDE;           // entries of one wave's tap table
    static constexpr int WIN_BYTES = NSLOT * 64;
    static constexpr int TAB_BYTES = NW * TAB * 8;
    static constexpr int VWIN_BYTES = DOTS ? WIN_BYTES : 0;
    static constexpr int LDS = VWIN_BYTES + WIN_BYTES + TAB_BYTES + NSLOT * 4 + 2 * NW * 4;
    static constexpr int WGS_LDS = (160 * 1024) / LDS < 1 ? 1 : (160 * 1024) / LDS;  // workgroups per CU the LDS admits ...
    static constexpr int WGS = WGS_LDS > MAXWGS_ ? MAXWGS_ : WGS_LDS;                // ... and the register budget is cut for
    static constexpr int WPS = (WGS * NW + 3) / 4 > 8 ? 8 : (WGS * NW + 3) / 4;      // waves per SIMD (register budget)
    // bound pass: lanes = (cell, camera) items of a pass of 8 cameras
    static constexpr int IPT = (CELLS * CAMS + THREADS - 1) / THREADS;
    static_assert(TW % 2 == 0 && CELLS >= 64 && (!DOTS || WWP % 4 == 2), "cells in pairs; a full wave of sample cells; conflict-free corners");
    static_assert(NSLOT * 4 <= TAB_BYTES, "the exact weight-mass array lives in the tap tables' space");
    static_assert(WIN_BYTES < 65536 - 64, "value window offset must fit a ds instruction's immediate");
};

// NC: cameras (levels) per pass of the tap tables when the level count is a multiple of it (6, 7, 8: the pass is unrolled without
// tests); 0 = any level count, passes of 8 with tests.
// DET: deterministic grad_value -- flushes and far taps add to det_acc (int64 per element, binary point from det_hdr) instead of
// fp32 atomics on grad_value; msda_det_finish folds det_acc into grad_value afterwards.
template <int FUSED, typename Cfg, int NC, bool DET = false>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::WPS) void msda_bwd_onepass(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const float *__restrict__ ref, int64_t ref_bstride, int raw_q, const float *__restrict__ out_fwd, int opts,
    long long *__restrict__ det_acc, const unsigned *__restrict__ det_hdr)
{
    constexpr int D = 16, TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW, WWP = Cfg::WWP, LCH = Cfg::LCH, NPAIR = Cfg::NPAIR;
    constexpr int P = TILE_P, THREADS = Cfg::THREADS, NSLOT = Cfg::NSLOT, CAMS = Cfg::CAMS, NW = Cfg::NW, CELLS = Cfg::CELLS, IPT = Cfg::IPT;
    constexpr int TSTRIDE = Cfg::TSTRIDE, TAB = Cfg::TAB, VW = Cfg::VWIN_BYTES, GW = Cfg::WIN_BYTES;
    constexpr bool DOTS = Cfg::DOTS != 0;
    constexpr int CH = NC ? NC : CAMS;                        // cameras per pass of pass 1
    static_assert(NC == 0 || (NC >= 1 && NC <= CAMS), "at most 8 cameras per pass");
    constexpr unsigned MASS_ONE = 1u << 20, MASS_CLAMP = MASS_ONE + 1u;          // guessed-scale jobs: see `guess` below
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float *const vwin = reinterpret_cast<float *>(lds_raw);                               // [NSLOT][16]
    long long *const win64 = reinterpret_cast<long long *>(lds_raw + VW);                 // [NSLOT][8]
    int *const mass = reinterpret_cast<int *>(lds_raw + VW + GW);                         // [NSLOT], exact bound pass
    float2 *const table = reinterpret_cast<float2 *>(lds_raw + VW + GW);                  // [NW waves][TAB], pass 1
    unsigned *const mass2 = reinterpret_cast<unsigned *>(lds_raw + VW + GW + Cfg::TAB_BYTES); // [NSLOT], guessed-scale jobs
    float *const red0 = reinterpret_cast<float *>(mass2 + NSLOT), *const red1 = red0 + NW;   // the block reductions' slots
    auto rpi = [](float x) {
        int r;
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        return r;
    };
    auto pack2 = [&](float lo_f, float hi_f) {
        const int lo = rpi(lo_f), hi = rpi(hi_f) + (lo >> 31);
        return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if ((FUSED || DET) && !equal) {
        // (the fused entry's callers promise equal level shapes, and the deterministic mode has no other kernel: make the misuse loud)
        for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < (int64_t)B * S * M * D; i += (int64_t)gridDim.x * THREADS)
            grad_value[i] = __builtin_nanf("");
        // ... in every gradient this launch owes: the raw tensor's (fused, when this kernel forms it), or grad_sampling_loc /
        // grad_attn_weight (deterministic mode of the public contract)
        if (FUSED && DOTS && grad_loc)
            for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < (int64_t)B * S * raw_q; i += (int64_t)gridDim.x * THREADS)
                grad_loc[i] = __builtin_nanf("");
        if (!FUSED && DOTS && grad_loc && grad_aw)
            for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < (int64_t)B * S * M * L * P; i += (int64_t)gridDim.x * THREADS) {
                grad_loc[2 * i] = grad_loc[2 * i + 1] = __builtin_nanf("");
                grad_aw[i] = __builtin_nanf("");
            }
        return;
    }
    if (!equal) {
        // not this kernel's case: the lane-group backward (msda_backward_lanes.h) does all three gradients (also in the
        // grad_value-only variant: the sampling kernels stand down for such calls)
        const int64_t total = (int64_t)B * S * M * D;
        for (int64_t base = (int64_t)blockIdx.x * THREADS; base < total; base += (int64_t)gridDim.x * THREADS)
            msda_bwd_lanes_body<float, 1, D, true, true>(base + tid, go, value, shapes, lsi, loc, aw, B, S, M, D, L, S, P,
                                                         grad_value, grad_loc, grad_aw);
        return;
    }

    [[maybe_unused]] int det_s = 0;                           // DET: binary point of det_acc
    if constexpr (DET) det_s = __builtin_amdgcn_readfirstlane(det_binary_point(det_hdr));
    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int units = per_level * M * B;
    const float fW = (float)Wq, fH = (float)Hq;
    const float iw = 1.f / fW, ih = 1.f / fH;
    constexpr int HPS = 32 / D;
    // sampling data of (query q, head, level l): normalised locations (x, y) x 4 points in la / lb, weights in wa
    // (FUSED, rr given: la / lb stay the raw offsets in pixels and *rr is the reference point -- the bound pass forms the
    // position from them exactly as pass 1 does, fused_px: ONE expression for every pass)
    auto fetch = [&](int64_t q, int b, int head, int l, float4 &la, float4 &lb, float4 &wa, float2 *rr = nullptr) {
        if constexpr (FUSED) {
            const float *rp = loc + q * raw_q + (l * (M / HPS) + head / HPS) * (HPS * P * 3);
            const float4 oa = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2);
            const float4 ob = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2 + 4);
            const float4 lg = *reinterpret_cast<const float4 *>(rp + HPS * P * 2 + (head % HPS) * P);
            const float2 r = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)l * S + (q - (int64_t)b * S)) * 2);
            const float2 st = *reinterpret_cast<const float2 *>(aw + (q * M + head) * 2);
            if (rr) {
                la = oa;
                lb = ob;
                *rr = r;
            } else {
                la = make_float4(__fmaf_rn(oa.x, iw, r.x), __fmaf_rn(oa.y, ih, r.y), __fmaf_rn(oa.z, iw, r.x), __fmaf_rn(oa.w, ih, r.y));
                lb = make_float4(__fmaf_rn(ob.x, iw, r.x), __fmaf_rn(ob.y, ih, r.y), __fmaf_rn(ob.z, iw, r.x), __fmaf_rn(ob.w, ih, r.y));
            }
            wa = make_float4(__expf(lg.x - st.x) * st.y, __expf(lg.y - st.x) * st.y, __expf(lg.z - st.x) * st.y, __expf(lg.w - st.x) * st.y);
        } else {
            const float *lp = loc + ((q * M + head) * L + l) * P * 2;
            la = *reinterpret_cast<const float4 *>(lp);
            lb = *reinterpret_cast<const float4 *>(lp + 4);
            wa = *reinterpret_cast<const float4 *>(aw + ((q * M + head) * L + l) * P);
        }
    };
    // the same for ONE point (lanes = taps): raw pieces (loaded by the pipelined requests of pass 1), finished by tap_of
    struct TapRaw {
        float2 o;         // location (or raw offset)
        float w;          // weight (or raw logit)
        float2 r, st;     // fused: reference point, softmax statistics
        float4 gq, oq;    // fused: this lane's quarter of the (query, head)'s grad_out / forward output rows
    };
    // -> pixel position (x, y), its floor (fx, fy) and the far corner's weights (wx1, wy1), attention weight a
    auto tap_of = [&](const TapRaw &t, float &x, float &y, float &fx, float &fy, float &wx1, float &wy1, float &a) {
        if constexpr (FUSED) {
            // (the position in two parts, common.h: the corner weights keep 5e-7 px where ONE fp32 number has 8e-6 at x ~ 143)
            fused_px(t.r.x, t.o.x, fW, x, fx, wx1);
            fused_px(t.r.y, t.o.y, fH, y, fy, wy1);
            a = __expf(t.w - t.st.x) * t.st.y;
        } else {
            x = op_pix(t.o.x, fW);
            y = op_pix(t.o.y, fH);
            fx = floorf(x);
            fy = floorf(y);
            wx1 = x - fx;
            wy1 = y - fy;
            a = t.w;
        }
    };

    for (int i = tid; i < NSLOT * NPAIR; i += THREADS) win64[i] = 0;
    for (int i = tid; i < NSLOT; i += THREADS) mass2[i] = 0u;
    __syncthreads();

    // block-wide maxima of two non-negative values (NaN-free); two barriers
    auto block_max2 = [&](float &a, float &b) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = fmaxf(a, __shfl_xor(a, o, 64));
            b = fmaxf(b, __shfl_xor(b, o, 64));
        }
        lds_barrier();
        if ((tid & 63) == 0) { red0[tid >> 6] = a; red1[tid >> 6] = b; }
        lds_barrier();
        a = red0[0];
        b = red1[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            a = fmaxf(a, red0[w]);
            b = fmaxf(b, red1[w]);
        }
    };

    // A unit is (tile, head); its L source levels are jobs that run back to back in one workgroup: they share the grad_out
    // rows (one Gmax), the head's bias ray (one window shift) and -- what removes the bound pass from all but the first --
    // roughly the weight mass per token: a level GUESSES its fixed-point scale from the previous level's measured mass,
    // measures its own mass while it accumulates (one ds_add_u32 per tap) and repeats the job with the exact bound pass in the
    // rare case that the guess was too small.  Every workgroup takes one contiguous range of the unit-major job list (a range
    // may begin and end inside a unit: 10,080 jobs over 512 workgroups are 19 or 20 each, where whole units would be 2 or 3
    // of 7); workgroups of one XCD take neighbouring ranges.
    // opts bit 1: the other order, kept for A/B (MVDETR_MSDA_BWD_ORDER=spread) -- every level job on its own, dealt round-robin
    // inside an XCD's band of the job list, so that the L level jobs of a (tile, head) run AT THE SAME TIME on neighbouring
    // workgroups of one XCD.  With the public layout [q][head][level][point] a (query, head)'s seven levels share two or three
    // 128-byte lines, which the ranges above fetch once per level (5 % L2 hit rate, 1.9 GB fetched per launch against 0.33 GB
    // for rounds 2-4's kernel) and this order once -- and the launch is SLOWER for it (737 vs 667 us public, 619 vs 580 fused: every
    // job pays the exact bound pass again, and the kernel is bound by VALU issue, not by those bytes).
    const int64_t total_jobs = (int64_t)units * L;
    const int nwg = (int)gridDim.x, rank = ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3);      // (nwg is a multiple of 8)
    const bool spread = (opts & 2) != 0;
    const int64_t band = (total_jobs + 7) / 8;               // (spread) jobs of one XCD
    const int64_t j_begin = spread ? ((int64_t)blockIdx.x >> 3) : total_jobs * rank / nwg;
    const int64_t j_end = spread ? band : total_jobs * (rank + 1) / nwg;
    const int64_t j_step = spread ? (nwg >> 3) : 0;
    [[maybe_unused]] int job_no = 0;
    float Wprev = 0.f;                                        // the weight mass the workgroup's previous job measured (any unit)
    for (int64_t j = j_begin; j < j_end;) {
        const int64_t jj = spread ? ((int64_t)blockIdx.x & 7) * band + j : j;
        if (spread) {
            j += j_step;
            if (jj >= total_jobs) continue;
        }
        const int unit = (int)(jj / L), l_first = (int)(jj % L);
        const int l_last = spread ? l_first + 1
                                  : (int)((int64_t)L < l_first + (j_end - j) ? (int64_t)L : l_first + (j_end - j));    // (exclusive)
        if (!spread) j += l_last - l_first;
        const int head = unit % M, u2 = unit / M;             // the heads of a tile run back to back on one XCD
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int ch0 = head * D;
        [[maybe_unused]] const int tr0 = job_no * 64 + (tid >> 6) * 16;
        job_no += l_last - l_first;
        OTRACE(tr0 + 0);
        // ---- where this head's taps lie: every wave reduces the same sample (the tile's first 64 cells, camera 0, level 0)
        //      to the mean tap displacement and, for the public contract, to the share of taps near their cells
        int shx = 0, shy = 0;
        bool standdown = false;
        {
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));   // (opaque: what derives from it is computed here, not kept alive from kernel entry)
            const int s_qy = Y0 + lane_s / TW, s_qx = X0 + lane_s % TW;
            static_assert(TH == MSDA_SAMPLE_TH && TW == MSDA_SAMPLE_TW, "the sampling kernels repeat this tile's sample (msda_dispatch.h)");
            const bool have = s_qy < Hq && s_qx < Wq;
            float4 a0 = make_float4(0, 0, 0, 0), b0 = a0, w0;
            if (have) fetch((int64_t)b * S + lsi[0] + (int64_t)s_qy * Wq + s_qx, b, head, 0, a0, b0, w0);   // (level 0: msda_dispatch.h)
            bool far;
            msda_job_sample(a0, b0, have, s_qx, s_qy, fW, fH, shx, shy, far);
            standdown = !FUSED && (opts & 1) && far;
        }
        if (standdown) {
            // far-flung taps (e.g. uniformly random locations): this unit's (cell, camera) items through the lane-group body,
            // restricted to its head
            const int items = TH * TW * L * D;
            int tid_s = tid;
            asm volatile("" : "+v"(tid_s));
            for (int it = tid_s; it < (items + THREADS - 1) / THREADS * THREADS; it += THREADS) {
                const int cg = it % D, ci = (it / D) % (TH * TW), c = it / (D * TH * TW);
                const int y_ = Y0 + ci / TW, x_ = X0 + ci % TW;
                const bool ok = it < items && y_ < Hq && x_ < Wq;
                const int64_t q = ok ? (int64_t)b * S + lsi[c] + (int64_t)y_ * Wq + x_ : -1;
                // (lanes without an item pass an index past the end: they stay converged with their group and store nothing)
                const int64_t idx = ok ? (q * M + head) * D + cg : (int64_t)B * S * M * D + cg;
                // (all three gradients, also in the grad_value-only variant: the sampling kernel skips such tiles)
                msda_bwd_lanes_body<float, 1, D, true, true>(idx, go, value, shapes, lsi, loc, aw, B, S, M, D, L, S, P, grad_value,
                                                             grad_loc, grad_aw, l_first, l_last);
            }
            continue;
        }
        const int oy = Y0 + TH / 2 - WH / 2 + shy, ox = X0 + TW / 2 - WW / 2 + shx;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        auto in_window = [&](float x, float y) { return fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1); };

        // ---- a level's value window of this head: LDS-DMA, 16 consecutive window slots (1 KB) per wave instruction; positions
        //      outside the level (and the padding slots of a row) are out-of-range buffer offsets and store zeros
        auto issue_dma = [&](int l) {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(value + (int64_t)b * S * row + ch0), 0, (int)(((unsigned)S * (unsigned)row - (unsigned)ch0) * 4u), 0x00020000);
            const unsigned so = (unsigned)((int)lsi[l] * row) * 4u;
            int lane_d = lane;
            asm volatile("" : "+v"(lane_d));
            for (int k = wave; k < NSLOT / 16; k += NW) {
                const int wp = k * 16 + (lane_d >> 2), wy = wp / WWP, wx = wp % WWP, gy = oy + wy, gx = ox + wx;
                const unsigned vo = (wx < WW && (unsigned)gx < (unsigned)Wq && (unsigned)gy < (unsigned)Hq)
                                        ? (unsigned)((gy * Wq + gx) * row + (lane_d & 3) * 4) * 4u : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vwin + k * 16 * LCH),
                                                         16, (int)vo, (int)so, 0, 0);
            }
        };
        if constexpr (DOTS) issue_dma(l_first);

        // ---- the unit's largest |grad_out| (inf if any is not finite): one pass over the (cell, camera) items' 64-byte rows of
        //      this head, lanes = items.  With it every level job of the unit can GUESS its fixed-point scale (below) -- also the
        //      unit's first: the weight mass per token is a statistic of the call, not of the (tile, head), so the guess builds on
        //      whatever job the workgroup ran last.  Only a workgroup's very first job pays the exact bound pass.
        float Gmax_u = 0.f;
        if (Wprev > 0.f && Wprev < INFINITY) {
            int tid_u = tid;
            asm volatile("" : "+v"(tid_u));
            float gm = 0.f;
            for (int it = tid_u; it < CELLS * L; it += THREADS) {
                const int ci = it % CELLS, c = it / CELLS;
                const int qy = Y0 + ci / TW, qx = X0 + ci % TW;
                const bool ok = qy < Hq && qx < Wq;
                const float *gp = go + ((int64_t)b * S + lsi[c] + (ok ? (int64_t)qy * Wq + qx : 0)) * row + ch0;
                float m = 0.f;
#pragma unroll
                for (int j4 = 0; j4 < LCH; j4 += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(gp + j4);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) m = INFINITY;     // NaN
                }
                if (ok) gm = fmaxf(gm, m);
            }
            float unused_u = 0.f;
            block_max2(gm, unused_u);
            Gmax_u = gm;
        }
        for (int l = l_first; l < l_last; ++l) {
            [[maybe_unused]] const int tr = tr0 + (l - l_first) * 64;
            if (l > l_first) OTRACE(tr + 0);
            const int64_t level_base = ((int64_t)b * S + lsi[l]) * row;
            // `guess`: the fixed-point scale from the unit's Gmax and twice the previous job's measured weight mass, no bound
            // pass; the job measures its own mass (each add rounded up and clamped to MASS_CLAMP, so that neither a wrap nor a
            // single huge weight can hide an overflow) and is repeated exactly if that exceeds the guess
            bool guess = Wprev > 0.f && Wprev < INFINITY && Gmax_u > 0.f && Gmax_u < INFINITY;
            float Wmax = 0.f, scale = 0.f, inv_scale = 0.f, mscale = 0.f;
            [[maybe_unused]] int e_job = 0;                   // the job's fixed point: steps of 2^(e_job - 30)
            bool direct_only = false, no_scatter = false;
            bool repeat = false;                              // second run of the job: the far taps' scatter has been done
            for (;;) {
                if (guess) {
                    Wmax = 2.f * Wprev;
                    mscale = (float)MASS_ONE / Wmax;
                    if (!(mscale < INFINITY)) mscale = 3.0e38f;
                } else {
                    // ---- pass 0: bounds (lanes = cells).  Gmax = largest |grad_out| of the job (inf if any is not finite), Amax =
                    //      largest sum_p |aw[l][p]|; then a bound on the weight mass landing on any one window token
                    int tid_p = tid;
                    asm volatile("" : "+v"(tid_p));
                    for (int i = tid_p; i < NSLOT; i += THREADS) mass[i] = 0;     // (the tap tables lived there)
                    // lanes = (cell, camera) items of a pass of 8 cameras: the item's sampling data (kept for the mass pass) and
                    // its grad_out row of this head
                    float4 la8[IPT], lb8[IPT], wa8[IPT];
                    [[maybe_unused]] float2 rf8[IPT];        // (FUSED) the items' reference points
                    bool act8[IPT];
                    auto load8 = [&](int c0, float &gm) {
#pragma unroll
                        for (int k = 0; k < IPT; ++k) {
                            const int it = tid_p + k * THREADS, ci = it % CELLS, c = c0 + it / CELLS;
                            const int qy = Y0 + ci / TW, qx = X0 + ci % TW;
                            act8[k] = it < CELLS * CAMS && c < L && qy < Hq && qx < Wq;
                            const int64_t q = (int64_t)b * S + lsi[act8[k] ? c : 0] + (act8[k] ? (int64_t)qy * Wq + qx : 0);
                            fetch(q, b, head, l, la8[k], lb8[k], wa8[k], &rf8[k]);
                            const float *gp = go + q * row + ch0;
                            float m = 0.f;
#pragma unroll
                            for (int j = 0; j < LCH; j += 4) {
                                const float4 v = *reinterpret_cast<const float4 *>(gp + j);
                                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                                if (!(v.x == v.x && v.y == v.y && v.z == v.z && v.w == v.w)) m = INFINITY;     // NaN
                            }
                            if (act8[k]) gm = fmaxf(gm, m);
                        }
                    };
                    float gmax = 0.f, al = 0.f;
                    for (int c0 = 0; c0 < L; c0 += CAMS) {
                        load8(c0, gmax);
#pragma unroll
                        for (int k = 0; k < IPT; ++k) {
                            const float s4 = (fabsf(wa8[k].x) + fabsf(wa8[k].y)) + (fabsf(wa8[k].z) + fabsf(wa8[k].w));
                            if (act8[k]) al = fmaxf(al, s4 == s4 ? s4 : INFINITY);
                        }
                    }
                    OTRACE(tr + 1);
                    float Gmax = gmax, Amax = al;
                    block_max2(Gmax, Amax);
                    Gmax_u = Gmax;
                    // non-finite inputs: no fixed point; every tap goes to memory as fp32 atomics (same NaN/inf results)
                    direct_only = !(Gmax < INFINITY && Amax < INFINITY);
                    // nothing to add to grad_value (block-uniform); the sampling gradients are still due
                    no_scatter = Gmax == 0.f || Amax == 0.f;
                    Wmax = 0.f;
                    if (!direct_only && !no_scatter) {
                        // weight-mass fixed point: a lane adds at most Amax per (camera, level), TH*TW*L of them -> < 2^30 in all
                        int ew = 0;
                        (void)frexpf(Amax * (float)(CELLS * L), &ew);
                        ew = ew < -60 ? -60 : ew;
                        const float wscale = ldexpf(1.f, 30 - ew);
                        for (int c0 = 0; c0 < L; c0 += CAMS) {
                            float unused_g = 0.f;
                            if (L > CAMS) load8(c0, unused_g);
#pragma unroll
                            for (int k = 0; k < IPT; ++k) {
                                const float lxs[4] = {la8[k].x, la8[k].z, lb8[k].x, lb8[k].z}, lys[4] = {la8[k].y, la8[k].w, lb8[k].y, lb8[k].w};
                                const float as[4] = {wa8[k].x, wa8[k].y, wa8[k].z, wa8[k].w};
#pragma unroll
                                for (int p = 0; p < P; ++p) {
                                    // (pass 1's expressions, tap_of: a tap is inside the window, and on a token, in every pass or in none)
                                    float x, y, fx, fy;
                                    if constexpr (FUSED) {
                                        float wx_, wy_;
                                        fused_px(rf8[k].x, lxs[p], fW, x, fx, wx_);
                                        fused_px(rf8[k].y, lys[p], fH, y, fy, wy_);
                                    } else {
                                        x = op_pix(lxs[p], fW);
                                        y = op_pix(lys[p], fH);
                                        fx = floorf(x);
                                        fy = floorf(y);
                                    }
                                    if (act8[k] && in_window(x, y)) {
                                        const int tok = ((int)fy - oy) * WWP + ((int)fx - ox);
                                        __hip_atomic_fetch_add(mass + tok, __float2int_ru(fabsf(as[p]) * wscale), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                }
                            }
                        }
                        lds_barrier();
                        int wm = 0;
                        for (int i = tid_p; i < NSLOT; i += THREADS) {
                            const int wy = i / WWP, wx = i % WWP;
                            int s = mass[i];
                            if (wx > 0) s += mass[i - 1];
                            if (wy > 0) s += mass[i - WWP];
                            if (wx > 0 && wy > 0) s += mass[i - WWP - 1];
                            wm = max(wm, s);
                        }
                        float Wm = (float)wm, unused = 0.f;
                        block_max2(Wm, unused);               // (its first barrier: every lane has read the mass array)
                        Wmax = Wm * (1.f + 1e-6f) / wscale;   // (int -> float rounding)
                    }
                }
                {
                    // any accumulator's final |sum| <= Gmax * Wmax = m * 2^e, m < 1; nearest rounding adds < 2^14 steps
                    const float bound = Gmax_u * Wmax;
                    int e = 0;
                    (void)frexpf(bound, &e);
                    e = !(bound < INFINITY) ? 129 : e < -90 ? -90 : e;
                    scale = bound > 0.f && !direct_only ? ldexpf(1.f, 30 - e) : 0.f;
                    inv_scale = ldexpf(1.f, e - 30);
                    e_job = e;
                }
                // ---- what pass 1 needs (lanes = taps, then lanes = (cell, corner, channel pair)).  Derived HERE, from an opaque copy of
                //      the lane index, so that none of it is alive through the bound pass above
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                float2 *const tab = table + wave * TAB;               // this wave's tap table
                float *const dtab = reinterpret_cast<float *>(tab);   // ... whose first 1 KB carries the dots back to the tap lanes
                const int wy_row = Y0 + wave;                         // the tile row of this wave
                // lanes as taps
                const int pc = lane_o >> 5, pt = lane_o & 31, pcam = pt >> 2, pp = pt & 3;
                // lanes as (cell, corner, pair)
                const int ct = lane_o >> 5, corner = (lane_o >> 3) & 3, pair = lane_o & 7;
                const float2 *const my_entries = tab + (ct * 4 + corner) * TSTRIDE;
                const int pair_off = pair * 8;
                // Addresses: per-batch-element base pointers (uniform) + 32-bit byte offsets per lane (the launcher checks that one
                // batch element's tensors stay below 2 GB)
                const char *const go_b = reinterpret_cast<const char *>(go + ((int64_t)b * S * row + ch0));
                const unsigned row_b = (unsigned)row * 4u;
                const bool row_ok = wy_row < Hq;

                const char *loc_b, *aw_b, *ref_b = nullptr, *st_b = nullptr, *out_b = nullptr;
                char *gl_b, *ga_b;                                // where the tap lanes' gradients go
                unsigned loc_q, loc_c, aw_q, aw_c;                // bytes per query / constant part of this lane's tap
                if constexpr (FUSED) {
                    loc_b = reinterpret_cast<const char *>(loc + (int64_t)b * S * raw_q);
                    aw_b = loc_b;
                    gl_b = reinterpret_cast<char *>(grad_loc + (int64_t)b * S * raw_q);
                    ga_b = gl_b;
                    loc_q = aw_q = (unsigned)raw_q * 4u;
                    const unsigned run = (unsigned)((l * (M / HPS) + head / HPS) * (HPS * P * 3));
                    loc_c = (run + (unsigned)((head % HPS) * P * 2 + pp * 2)) * 4u;
                    aw_c = (run + (unsigned)(HPS * P * 2 + (head % HPS) * P + pp)) * 4u;
                    ref_b = reinterpret_cast<const char *>(ref + b * ref_bstride + (int64_t)l * S * 2);
                    st_b = reinterpret_cast<const char *>(aw + ((int64_t)b * S * M + head) * 2);
                    out_b = reinterpret_cast<const char *>(out_fwd + ((int64_t)b * S * row + ch0));
                } else {
                    loc_b = reinterpret_cast<const char *>(loc + (int64_t)b * S * M * L * P * 2);
                    aw_b = reinterpret_cast<const char *>(aw + (int64_t)b * S * M * L * P);
                    gl_b = reinterpret_cast<char *>(grad_loc + (int64_t)b * S * M * L * P * 2);
                    ga_b = reinterpret_cast<char *>(grad_aw + (int64_t)b * S * M * L * P);
                    loc_q = (unsigned)(M * L * P * 2) * 4u;
                    aw_q = (unsigned)(M * L * P) * 4u;
                    loc_c = (unsigned)((head * L + l) * P * 2 + pp * 2) * 4u;
                    aw_c = (unsigned)((head * L + l) * P + pp) * 4u;
                }

                unsigned cam_q[CAMS];                             // first token of the pass's cameras (uniform)
                bool tap_cam_ok = false;
                unsigned tap_q = 0;                               // of this lane's tap
                auto set_chunk = [&](int c0) {
    #pragma unroll
                    for (int k = 0; k < CAMS; ++k) cam_q[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)lsi[c0 + k < L ? c0 + k : L - 1]);
                    tap_cam_ok = pcam < CH && c0 + pcam < L;
                    tap_q = (unsigned)lsi[tap_cam_ok ? c0 + pcam : 0];
                };
                constexpr int steps = TW / 2;
                TapRaw nraw;
                float2 ng[CAMS];
                unsigned n_q = 0;                                 // of the tap lane's query, inside the batch element
                bool n_valid = false;
                auto request_taps = [&](int j, TapRaw &t, bool &ok, unsigned &q) {      // lanes as taps: this lane's tap
                    const int x_ = X0 + 2 * j + pc;
                    ok = row_ok && x_ < Wq && tap_cam_ok;
                    q = ok ? tap_q + (unsigned)(wy_row * Wq + x_) : 0u;                   // inside the batch element
                    t.o = *reinterpret_cast<const float2 *>(loc_b + (q * loc_q + loc_c));
                    t.w = *reinterpret_cast<const float *>(aw_b + (q * aw_q + aw_c));
                    if constexpr (FUSED) {
                        t.r = *reinterpret_cast<const float2 *>(ref_b + q * 8u);
                        t.st = *reinterpret_cast<const float2 *>(st_b + q * (unsigned)(M * 8));
                        if constexpr (DOTS) {
                            t.gq = *reinterpret_cast<const float4 *>(go_b + (q * row_b + (unsigned)pp * 16u));
                            t.oq = *reinterpret_cast<const float4 *>(out_b + (q * row_b + (unsigned)pp * 16u));
                        } else {
                            t.gq = t.oq = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    } else {
                        t.r = t.st = make_float2(0.f, 0.f);
                        t.gq = t.oq = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                };
                auto request_g = [&](int j) {    // lanes as (cell, corner, pair): grad_out pairs of the cell's cameras
                    const int x_ = X0 + 2 * j + ct;
                    const unsigned cellq = row_ok && x_ < Wq ? (unsigned)(wy_row * Wq + x_) : 0u;
                    const unsigned o = cellq * row_b + (unsigned)pair * 8u;
    #pragma unroll
                    for (int k = 0; k < CAMS; ++k) ng[k] = *reinterpret_cast<const float2 *>(go_b + (cam_q[k] * row_b + o));
                };

                // the first step's sampling data and grad_out pairs: requested before the wait below, not after it
                set_chunk(0);
                request_taps(0, nraw, n_valid, n_q);
                request_g(0);
                // the value window: every wave waits for its own LDS-DMA requests, then the barrier makes all of them visible
                if constexpr (DOTS) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) (expcnt / lgkmcnt untouched)
                    lds_barrier();
                }
#if MVDETR_OP_STAGGER
                // the four waves of a workgroup leave this barrier in step and would stay in step -- all in their LDS-bound
                // stream at the same time, then all outside it: start them a quarter of a step apart
                if (wave & 1) __builtin_amdgcn_s_sleep(MVDETR_OP_STAGGER);
                if (wave & 2) __builtin_amdgcn_s_sleep(2 * MVDETR_OP_STAGGER);
#endif

                OTRACE(tr + 2);
                // ---- pass 1: accumulation + dots (lanes = taps, then lanes = (cell, corner, channel pair), then lanes = taps) ----
                bool bad = false;                             // guessed scale: a weight that is not finite
                for (int c0 = 0; c0 < L; c0 += CH) {
                    if (c0) {
                        set_chunk(c0);
                        request_taps(0, nraw, n_valid, n_q);
                        request_g(0);
                    }
                    for (int s = 0; s < steps; ++s) {
                        const TapRaw raw_ = nraw;
                        float2 g[CAMS];
#pragma unroll
                        for (int k = 0; k < CAMS; ++k) g[k] = ng[k];
                        if (direct_only) {
                            // (uniform, rare) no fixed point for this job: the stream below still runs, on zeros -- the window stays clean
#pragma unroll
                            for (int k = 0; k < CAMS; ++k) g[k] = make_float2(0.f, 0.f);
                        }
                        const bool valid = n_valid;
                        const unsigned my_q = n_q;
                        if (s + 1 < steps) {
                            request_taps(s + 1, nraw, n_valid, n_q);
                            request_g(s + 1);
                        }

                        if (s < 4 && c0 == 0) OTRACE(tr + 6 + 2 * s);
                        // ---- lanes as taps: the four (weight, record) entries of this lane's tap
                        float x, y, fx, fy, wx1, wy1, a;
                        tap_of(raw_, x, y, fx, fy, wx1, wy1, a);
                        const bool inw = in_window(x, y);
                        const bool hit = valid && !direct_only && inw;
                        {
                            const int tok = hit ? ((int)fy - oy) * WWP + ((int)fx - ox) : 0;
                            const float sw = hit ? a * scale : 0.f, ay1 = hit ? wy1 * sw : 0.f, ay0 = sw - ay1;
                            const float w01 = hit ? ay0 * wx1 : 0.f, w00 = ay0 - w01, w11 = hit ? ay1 * wx1 : 0.f, w10 = ay1 - w11;
                            const int o00 = tok * (NPAIR * 8), o10 = hit ? o00 + WWP * (NPAIR * 8) : 0, o01 = hit ? o00 + NPAIR * 8 : 0;
                            float2 *e = tab + pc * 4 * TSTRIDE + pt;
                            e[0] = make_float2(w00, __int_as_float(o00));
                            e[TSTRIDE] = make_float2(w01, __int_as_float(o01));
                            e[2 * TSTRIDE] = make_float2(w10, __int_as_float(o10));
                            e[3 * TSTRIDE] = make_float2(w11, __int_as_float(hit ? o10 + NPAIR * 8 : 0));
                            if (guess) {
                                // (uniform) this job's own weight mass, as the exact pass would have measured it
                                bad = bad || (valid && !(fabsf(a) < INFINITY));
                                if (hit) {
                                    const float mf = fminf(fabsf(a) * mscale, (float)MASS_CLAMP);
                                    atomicAdd(mass2 + tok, min((unsigned)__float2int_ru(mf), MASS_CLAMP));
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                        if (s < 4 && c0 == 0) OTRACE(tr + 7 + 2 * s);
                        // ---- lanes as (cell, corner, pair): per tap of the lane's cell one value read, one ds_add_u64, the lane's
                        //      share of the dot.  A batch = one camera's four points = two 16-byte reads of the lane's run; the entries
                        //      are read a batch ahead (a third stage -- value records a batch ahead too -- costs 18 registers)
                        float dk[4] = {0.f, 0.f, 0.f, 0.f};   // dots of taps k = pair (mod 8) of this (cell, corner)
                        {
                            float4 E[CH][2];
                            float2 V[CH][4];
                            auto read_entries = [&](int bt) {
#pragma unroll
                                for (int k = 0; k < 2; ++k) E[bt][k] = *reinterpret_cast<const float4 *>(my_entries + (bt * 2 + k) * 2);
                            };
                            auto read_values = [&](int bt) {
                                const int of[4] = {__float_as_int(E[bt][0].y), __float_as_int(E[bt][0].w), __float_as_int(E[bt][1].y), __float_as_int(E[bt][1].w)};
#pragma unroll
                                for (int h = 0; h < 4; ++h) V[bt][h] = *reinterpret_cast<const float2 *>(lds_raw + of[h] + pair_off);
                            };
                            // (all of a step's entries requested up front -- no LDS round trip inside the stream -- was measured in
                            // round 6: the same time at three workgroups per CU, where it spills, and slower at two)
                            read_entries(0);
#pragma unroll
                            for (int bt = 0; bt < CH; ++bt) {
                                if (NC == 0 && c0 + bt >= L) break;             // (uniform)
                                if constexpr (DOTS) read_values(bt);
                                if (bt + 1 < CH) read_entries(bt + 1);
                                const float wg[4] = {E[bt][0].x, E[bt][0].z, E[bt][1].x, E[bt][1].z};
                                const int of[4] = {__float_as_int(E[bt][0].y), __float_as_int(E[bt][0].w), __float_as_int(E[bt][1].y), __float_as_int(E[bt][1].w)};
#pragma unroll
                                for (int h = 0; h < 4; ++h) {
                                    const long long v = pack2(wg[h] * g[bt].x, wg[h] * g[bt].y);
                                    long long *w = reinterpret_cast<long long *>(lds_raw + VW + of[h] + pair_off);    // (VW = 0 without the value window)
                                    __hip_atomic_fetch_add(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                                if constexpr (DOTS) {
#pragma unroll
                                    for (int h = 0; h < 4; ++h) {
                                        const int kk = bt * 4 + h;              // tap of the cell: camera bt, point h
                                        const float part = sum8(__fmaf_rn(V[bt][h].x, g[bt].x, V[bt][h].y * g[bt].y));
                                        dk[kk >> 3] = (kk & 7) == pair ? part : dk[kk >> 3];
                                    }
                                }
                            }
                        }
                        if (s == 0 && c0 == 0) OTRACE(tr + 14);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();      // every lane is done with the entries (the dots take their place)
                        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (DOTS) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) dtab[(ct * 32 + pair + 8 * i) * 4 + corner] = dk[i];
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            // ---- lanes as taps: the tap's four dots
                            dv = *reinterpret_cast<const float4 *>(dtab + lane_o * 4);     // (d00, d01, d10, d11)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();  // ... read by every lane before the next step's entries land
                        }

                        // ---- taps outside the window (or every tap of a non-finite job): both halves straight from memory, FOUR
                        //      taps per round with lanes = (tap of the round, channel) and the corners in turn -- an atomic
                        //      instruction is four 64-byte segments, as when the whole wave took one tap with lanes = (corner,
                        //      channel), but the shuffles, the footprint and the loop are paid once per four taps
                        const bool in_image = y > -1.f && x > -1.f && y < fH && x < fW;
                        const bool miss = valid && (direct_only || !inw) && in_image;
                        unsigned long long pend = __ballot(miss);
                        while (pend) {
                            int s_[4];
                            bool h_[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                h_[u] = pend != 0ull;
                                s_[u] = h_[u] ? __ffsll((long long)pend) - 1 : 0;
                                pend &= pend - 1ull;                              // (0 stays 0)
                            }
                            const int tq = lane_o >> 4, j = lane_o & 15;
                            const int src = tq == 0 ? s_[0] : tq == 1 ? s_[1] : tq == 2 ? s_[2] : s_[3];
                            const bool mine = tq == 0 ? h_[0] : tq == 1 ? h_[1] : tq == 2 ? h_[2] : h_[3];
                            const float sa = __shfl(a, src, 64);
                            const float sfx = __shfl(fx, src, 64), sfy = __shfl(fy, src, 64), swx = __shfl(wx1, src, 64), swy = __shfl(wy1, src, 64);
                            const unsigned sg = (unsigned)__shfl((int)my_q, src, 64) * row_b;
                            const float gk = *reinterpret_cast<const float *>(go_b + (sg + (unsigned)j * 4u));
                            const Footprint<float> f = footprint_split(sfy, swy, sfx, swx, Hq, Wq);
                            [[maybe_unused]] float prs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int cr = 0; cr < 4; ++cr) {
                                const int yy = f.y0 + (cr >> 1), xx = f.x0 + (cr & 1);
                                const float wgt = ((cr >> 1) ? f.wy1 : f.wy0) * ((cr & 1) ? f.wx1 : f.wx0);
                                const bool ok = mine && (unsigned)yy < (unsigned)Hq && (unsigned)xx < (unsigned)Wq;
                                const int64_t vo = level_base + ch0 + ((int64_t)(ok ? yy : 0) * Wq + (ok ? xx : 0)) * row + j;
                                if (ok && !repeat) {
                                    const float c = wgt * (gk * sa);
                                    if (DET && fabsf(c) < INFINITY)
                                        atomicAdd(reinterpret_cast<unsigned long long *>(det_acc + vo), (unsigned long long)__double2ll_rn(ldexp((double)c, det_s)));
                                    else        // (DET: NaN / inf sums do not depend on the order either)
                                        atomicAdd(grad_value + vo, c);
                                }
                                if constexpr (DOTS) {
                                    const float vk = ok ? value[vo] : 0.f;
                                    float pr = sum8(gk * vk);
                                    pr += dpp_f<0x140>(pr);                       // row_mirror: the other eight lanes of the sixteen
                                    prs[cr] = pr;
                                }
                            }
                            if constexpr (DOTS) {
                                const int myslot = (h_[0] && lane_o == s_[0]) ? 0 : (h_[1] && lane_o == s_[1]) ? 1 : (h_[2] && lane_o == s_[2]) ? 2
                                                 : (h_[3] && lane_o == s_[3]) ? 3 : -1;
                                const int from = (myslot < 0 ? 0 : myslot) * 16;
                                const float e00 = __shfl(prs[0], from, 64), e01 = __shfl(prs[1], from, 64), e10 = __shfl(prs[2], from, 64), e11 = __shfl(prs[3], from, 64);
                                if (myslot >= 0) dv = make_float4(e00, e01, e10, e11);
                            }
                        }

                        // ---- the tap's three gradients
                        if (DOTS && valid) {
                            const bool have = in_image && (hit || miss);       // (a NaN position has neither: zeros)
                            const float d00 = have ? dv.x : 0.f, d01 = have ? dv.y : 0.f, d10 = have ? dv.z : 0.f, d11 = have ? dv.w : 0.f;
                            const float top = d00 + wx1 * (d01 - d00), bot = d10 + wx1 * (d11 - d10);
                            const float da = have ? top + wy1 * (bot - top) : 0.f;
                            const float gx = have ? (d01 - d00) + wy1 * ((d11 - d10) - (d01 - d00)) : 0.f;
                            const float gy = have ? (d10 - d00) + wx1 * ((d11 - d01) - (d10 - d00)) : 0.f;
                            if constexpr (FUSED) {
                                // D = <grad_out, out> of the (query, head): this lane's quarter, summed over the 4 point lanes
                                float dq = (raw_.gq.x * raw_.oq.x + raw_.gq.y * raw_.oq.y) + (raw_.gq.z * raw_.oq.z + raw_.gq.w * raw_.oq.w);
                                dq += dpp_f<0xB1>(dq);
                                dq += dpp_f<0x4E>(dq);
                                *reinterpret_cast<float2 *>(gl_b + (my_q * loc_q + loc_c)) = make_float2(a * gx, a * gy);
                                *reinterpret_cast<float *>(ga_b + (my_q * aw_q + aw_c)) = a * (da - dq);
                            } else {
                                *reinterpret_cast<float2 *>(gl_b + (my_q * loc_q + loc_c)) = make_float2(fW * a * gx, fH * a * gy);
                                *reinterpret_cast<float *>(ga_b + (my_q * aw_q + aw_c)) = da;
                            }
                        }
                        if (s == 0 && c0 == 0) OTRACE(tr + 15);
                    }
                }
                OTRACE(tr + 3);
                lds_barrier();                                // (the last step's gradient stores stay in flight)
                OTRACE(tr + 4);
                if (!guess) break;
                // ---- the guessed scale's check: this job's own weight mass per token (2 x 2 neighbourhood sums, as in pass 0)
                int tid_v = tid;
                asm volatile("" : "+v"(tid_v));
                unsigned wm = 0;
                for (int i = tid_v; i < NSLOT; i += THREADS) {
                    const int wy = i / WWP, wx = i % WWP;
                    unsigned s = min(mass2[i], 2u * MASS_CLAMP);
                    if (wx > 0) s += min(mass2[i - 1], 2u * MASS_CLAMP);
                    if (wy > 0) s += min(mass2[i - WWP], 2u * MASS_CLAMP);
                    if (wx > 0 && wy > 0) s += min(mass2[i - WWP - 1], 2u * MASS_CLAMP);
                    wm = max(wm, s);
                }
                float Wm = (float)wm, fbad = bad ? 1.f : 0.f;
                block_max2(Wm, fbad);                         // (its first barrier: every lane has read the mass array)
                for (int i = tid_v; i < NSLOT; i += THREADS) mass2[i] = 0u;
                if (Wm <= (float)MASS_ONE && fbad == 0.f) {
                    Wmax = Wm * (1.f + 1e-6f) / mscale;       // what the next level builds its guess on
                    break;
                }
                // too small (or a weight that is not finite): once more, exactly.  The window starts from zero again; what pass 1
                // stored of the sampling gradients did not depend on the scale and is simply written a second time.
                for (int i = tid_v; i < NSLOT * NPAIR; i += THREADS) win64[i] = 0;
                guess = false;                                // (pass 0's barriers order the zeroing before the new adds)
                repeat = true;
            }
            Wprev = direct_only || no_scatter ? 0.f : Wmax;
            // the next level's value window: every wave is past pass 1 (the barrier above), the copy runs under the flush
            if constexpr (DOTS) {
                if (l + 1 < l_last) issue_dma(l + 1);
            }
            // ---- flush: the touched tokens' 64-byte records as fp32 atomics; leaves the window zeroed ----
            if (!direct_only && !no_scatter) {
                // lane = (token, channel): the two lanes of a channel pair read the same qword, one of them clears it
                int tid_f = tid;
                asm volatile("" : "+v"(tid_f));
                const int ch = tid_f % LCH, pair_f = ch >> 1;
                const bool upper = ch & 1;
                float *const gbase = grad_value + level_base + ch0 + ch;
                // DET: job steps -> call steps, a shift by k bits (rounded to nearest when it is to the right)
                [[maybe_unused]] long long *const dbase = DET ? det_acc + level_base + ch0 + ch : nullptr;
                [[maybe_unused]] const int k_up = e_job - 30 + det_s, k_dn = -k_up > 40 ? 40 : -k_up < 1 ? 1 : -k_up;
                for (int i0 = tid_f / LCH; i0 < NSLOT; i0 += 8 * (THREADS / LCH)) {
                    long long v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int tok = i0 + k * (THREADS / LCH);
                        v[k] = tok < NSLOT ? win64[tok * NPAIR + pair_f] : 0;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int tok = i0 + k * (THREADS / LCH);
                        if (v[k] != 0) {
                            if (!upper) win64[tok * NPAIR + pair_f] = 0;
                            const int lo = (int)v[k], hi = (int)((v[k] - (long long)lo) >> 32);
                            const int mine = upper ? hi : lo;
                            const int gy = oy + tok / WWP, gx = ox + tok % WWP;
                            // corners outside the level were accumulated like any other and are dropped here (zero padding)
                            if (mine != 0 && (unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq) {
                                if constexpr (DET) {
                                    const long long inc = k_up >= 0 ? (long long)mine << (k_up > 31 ? 31 : k_up)
                                                                    : ((long long)mine + (1ll << (k_dn - 1))) >> k_dn;
                                    if (inc) atomicAdd(reinterpret_cast<unsigned long long *>(dbase + ((int64_t)gy * Wq + gx) * row), (unsigned long long)inc);
                                } else {
                                    atomicAdd(gbase + ((int64_t)gy * Wq + gx) * row, (float)mine * inv_scale);
                                }
                            }
                        }
                    }
                }
            }
            OTRACE(tr + 5);
            lds_barrier();                                    // the window is clean before the next job adds to it (the flush's
                                                              // atomics and the next window's copy stay in flight)
        }
    }
}

template <int FUSED, typename Cfg, int NC, bool DET>
static int launch_onepass_nc(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int L, float *grad_value, float *grad_loc,
                          float *grad_aw, const float *ref, int64_t ref_bstride, int raw_q, const float *out_fwd, int opts,
                          long long *det_acc, const unsigned *det_hdr)
{
    constexpr int LDS = Cfg::LDS;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_onepass<FUSED, Cfg, NC, DET>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_onepass<FUSED, Cfg, NC, DET>, Cfg::THREADS, LDS) != hipSuccess || per_cu < 1)
            per_cu = Cfg::WGS;
        if (per_cu > Cfg::WGS) per_cu = Cfg::WGS;
        if (getenv("MVDETR_DEBUG_OCCUPANCY"))
            fprintf(stderr, "msda_bwd_onepass<%d, %dx%d, %d, %d>: %d workgroups per CU (LDS admits %d), %d B of LDS, %d threads\n", FUSED,
                    Cfg::TH, Cfg::TW, NC, (int)DET, per_cu, Cfg::WGS, LDS, Cfg::THREADS);
        return (cus * per_cu + 7) / 8 * 8;
    });
    hipLaunchKernelGGL((msda_bwd_onepass<FUSED, Cfg, NC, DET>), dim3((unsigned)blocks), dim3(Cfg::THREADS), LDS, st, go, value, shapes, lsi, loc, aw,
                       B, S, M, L, grad_value, grad_loc, grad_aw, ref, ref_bstride, raw_q, out_fwd, opts, det_acc, det_hdr);
    return (int)hipGetLastError();
}

template <int FUSED, typename Cfg, bool DET = false>
static int launch_onepass(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int L, float *grad_value, float *grad_loc,
                          float *grad_aw, const float *ref, int64_t ref_bstride, int raw_q, const float *out_fwd, int opts,
                          long long *det_acc = nullptr, const unsigned *det_hdr = nullptr)
{
#define NC_ARGS st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, ref, ref_bstride, raw_q, out_fwd, opts, det_acc, det_hdr
    if (L % 7 == 0) return launch_onepass_nc<FUSED, Cfg, 7, DET>(NC_ARGS);
    if (L % 8 == 0) return launch_onepass_nc<FUSED, Cfg, 8, DET>(NC_ARGS);
    if (L % 6 == 0) return launch_onepass_nc<FUSED, Cfg, 6, DET>(NC_ARGS);
    return launch_onepass_nc<FUSED, Cfg, 0, DET>(NC_ARGS);
#undef NC_ARGS
}

bool msda_backward_onepass_supported(int B, int S, int M, int D, int L, int64_t q_floats)
{
    // 16-channel heads; one batch element's tensors addressed with 32-bit byte offsets
    const int64_t lim = (int64_t)1 << 31;
    return D == 16 && L <= TILE_MAX_LEVELS && (int64_t)S * M * D * 4 < lim && (int64_t)S * q_floats * 4 < lim;
}

int msda_backward_onepass(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                          float *grad_loc, float *grad_aw, bool standdown)
{
    return launch_onepass<0, OnePassCfg<4, 16, 6, 1>>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw,
                                                      nullptr, 0, 0, nullptr, standdown ? 1 : 0);
}

int msda_backward_onepass_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                const float *stats, const float *out_fwd, int B, int S, int M, int D, int L,
                                float *grad_value, float *grad_raw)
{
    return launch_onepass<1, OnePassCfg<4, 16, 6, 1>>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, grad_raw, nullptr,
                                                      ref, ref_bstride, raw_q, out_fwd, 0);
}

// ---- deterministic mode: the one-pass kernel with 64-bit fixed-point sums for grad_value ------------------------------------
// (the reference's col2im adds with atomicAdd, cuh:125-152, and is not reproducible run to run; this is the opt-in that is.)
// largest finite |a[i]| -> hdr[0], largest finite |b[i]| (or 1 without b) -> hdr[1], as float bits (which order like the floats)
__global__ __launch_bounds__(256) void msda_det_absmax(const float *__restrict__ a, int64_t na4, const float *__restrict__ b,
                                                       int64_t nb4, unsigned *__restrict__ hdr)
{
    __shared__ unsigned red[2][4];
    auto scan = [&](const float *p, int64_t n4) {
        unsigned m = 0u;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 v = reinterpret_cast<const float4 *>(p)[i];
            const unsigned u[4] = {__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu,
                                   __float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu};
#pragma unroll
            for (int k = 0; k < 4; ++k) m = u[k] < 0x7f800000u && u[k] > m ? u[k] : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        return m;
    };
    const unsigned ma = scan(a, na4), mb = b ? scan(b, nb4) : __float_as_uint(1.f);
    // one atomic per workgroup and maximum (one per wave made 16 K atomics on two addresses: 200 us for 107 MB)
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ma; red[1][threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned m = max(max(red[threadIdx.x][0], red[threadIdx.x][1]), max(red[threadIdx.x][2], red[threadIdx.x][3]));
        if (m) atomicMax(hdr + threadIdx.x, m);
    }
}

// grad_value += det_acc * 2^-binary point (one writer per element)
__global__ __launch_bounds__(256) void msda_det_finish(const long long *__restrict__ acc, const unsigned *__restrict__ hdr,
                                                       float *__restrict__ grad_value, int64_t n)
{
    const int s = det_binary_point(hdr);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = acc[i];
        if (v) grad_value[i] += (float)ldexp((double)v, -s);
    }
}

bool msda_backward_deterministic_supported(int B, int S, int M, int D, int L, int64_t q_floats)
{
    // (an element's sum stays below 2^38 x the taps that can land on it: 2^24 of them leave a factor two to int64)
    return msda_backward_onepass_supported(B, S, M, D, L, q_floats) && (int64_t)S * L * TILE_P < ((int64_t)1 << 24);
}

// the deterministic mode's scratch (header + 8 bytes per value element), kept per (device, stream) between calls and grown on
// demand: a hipMallocAsync / hipFreeAsync pair per call made the pool grow while the previous call's free was still queued (avg
// 1.6 ms against a minimum of 0.91 in one bench run).  mvdetr_msda_release_scratch() hands it back.
namespace {
struct DetScratch { char *ptr; size_t size; };
std::mutex g_det_mu;
std::map<std::pair<int, hipStream_t>, DetScratch> g_det_scratch;
char *det_scratch(hipStream_t st, size_t bytes, int &rc)
{
    int dev = 0;
    rc = (int)hipGetDevice(&dev);
    if (rc) return nullptr;
    std::lock_guard<std::mutex> lock(g_det_mu);
    DetScratch &e = g_det_scratch[{dev, st}];
    if (e.size >= bytes) return e.ptr;
    if (e.ptr) (void)hipFreeAsync(e.ptr, st);
    e.ptr = nullptr;
    e.size = 0;
    rc = (int)hipMallocAsync(reinterpret_cast<void **>(&e.ptr), bytes, st);
    if (rc) { e.ptr = nullptr; return nullptr; }
    e.size = bytes;
    return e.ptr;
}
}  // namespace

int msda_release_det_scratch()
{
    std::lock_guard<std::mutex> lock(g_det_mu);
    int rc = 0;
    for (auto &kv : g_det_scratch)
        if (kv.second.ptr) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            (void)hipSetDevice(kv.first.first);
            const int r = (int)hipFree(kv.second.ptr);      // (synchronises: call it when no backward is in flight)
            (void)hipSetDevice(dev);
            rc = rc ? rc : r;
        }
    g_det_scratch.clear();
    return rc;
}

template <int FUSED>
static int onepass_det(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *aw, int B, int S, int M, int L, float *grad_value, float *grad_loc,
                       float *grad_aw, const float *ref, int64_t ref_bstride, int raw_q, const float *out_fwd)
{
    const int64_t n = (int64_t)B * S * M * 16;
    if (n == 0) return 0;
    const size_t bytes = DET_HDR_BYTES + (size_t)n * sizeof(long long);
    int rc = 0;
    char *scratch = det_scratch(st, bytes, rc);
    if (!scratch) return rc ? rc : (int)hipErrorOutOfMemory;
    unsigned *hdr = reinterpret_cast<unsigned *>(scratch);
    long long *acc = reinterpret_cast<long long *>(scratch + DET_HDR_BYTES);
    rc = (int)hipMemsetAsync(scratch, 0, bytes, st);
    if (!rc) {
        // (public contract: aw are the attention weights; fused: softmax weights, at most 1)
        hipLaunchKernelGGL(msda_det_absmax, dim3(2048), dim3(256), 0, st, go, n / 4, FUSED ? nullptr : aw,
                           FUSED ? (int64_t)0 : (int64_t)B * S * M * L * TILE_P / 4, hdr);
        rc = (int)hipGetLastError();
    }
    if (!rc)
        rc = launch_onepass<FUSED, OnePassCfg<4, 16, 6, 1>, true>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc,
                                                                  grad_aw, ref, ref_bstride, raw_q, out_fwd, 0, acc, hdr);
    if (!rc) {
        hipLaunchKernelGGL(msda_det_finish, dim3(4096), dim3(256), 0, st, acc, hdr, grad_value, n);
        rc = (int)hipGetLastError();
    }
    return rc;
}

int msda_backward_onepass_det(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                              const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                              float *grad_loc, float *grad_aw)
{
    return onepass_det<0>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, nullptr, 0, 0, nullptr);
}

int msda_backward_onepass_fused_det(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                    const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                    const float *stats, const float *out_fwd, int B, int S, int M, int D, int L,
                                    float *grad_value, float *grad_raw)
{
    return onepass_det<1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, grad_raw, nullptr, ref, ref_bstride, raw_q,
                          out_fwd);
}

// job order of the grad_value-only launches: ranges (default: levels of a (tile, head) one after the other in a workgroup, guessed
// scale) or spread (every level job on its own, the levels of a (tile, head) side by side on one XCD: see the kernel; measured
// slower).  MVDETR_MSDA_BWD_ORDER = ranges | spread for A/B.
static int onepass_order(int deflt)
{
    static const int forced = [] {
        const char *e = getenv("MVDETR_MSDA_BWD_ORDER");
        return !e ? -1 : !strcmp(e, "spread") ? 2 : !strcmp(e, "ranges") ? 0 : -1;
    }();
    return forced >= 0 ? forced : deflt;
}

// the grad_value half alone (DOTS = 0): the same jobs without the value window and the dot products
// (grad_loc / grad_aw: written only when the level shapes turn out unequal on the device -- the lane-group fallback)
int msda_backward_scatter(hipStream_t st, const float *go, const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, int B, int S, int M, int D, int L, float *grad_value,
                          float *grad_loc, float *grad_aw, bool standdown)
{
    return launch_onepass<0, OnePassCfg<4, 16, 6, 0, MVDETR_SCATTER_WGS>>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw,
                                                         nullptr, 0, 0, nullptr, (standdown ? 1 : 0) | onepass_order(0));
}

int msda_backward_scatter_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                const float *stats, int B, int S, int M, int D, int L, float *grad_value)
{
    return launch_onepass<1, OnePassCfg<4, 16, 6, 0, MVDETR_SCATTER_WGS>>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, nullptr, nullptr,
                                                         ref, ref_bstride, raw_q, nullptr, onepass_order(0));
}

}  // namespace mvdetr
