// Multi-scale deformable attention backward, grad_value for deformable-ENCODER calls, second generation -- gfx950 (MI355X).
//
// msda_bwd_value_win (msda_backward_tile.hip) keeps a source level's window of grad_value in LDS as fixed-point accumulators
// and adds every tap corner with ds_add_u64 -- lanes = cells, one channel plane per instruction.  With learned-like offsets
// (bias ray + ~1 px of noise) the 64 lanes of such an instruction fall on the 32 qword banks at random: 6.8 - 7.1 ns per wave
// instruction per CU against 2.7 ns for conflict-free addresses (tools/experiments/lds_atomic_rate2.hip), and the kernel is
// exactly that: 33 K atomics per CU, 390 us.  What the same experiment shows: when every 8 consecutive lanes cover ONE token's
// 64-byte record (8 qwords = 16 banks, aligned), the atomic unit runs at the conflict-free rate wherever the tokens lie.  So
// here
//
//   * the window is TOKEN-major: [16 x 44 tokens][8 qwords], a qword = two 32-bit fixed-point accumulators (channels 2k, 2k+1
//     of the job's 16-channel slice) -- same arithmetic, bounds and scale as msda_bwd_value_win;
//   * a wave instruction is 2 taps x 4 corners x 8 channel pairs: lane = (cell of a pair, corner, channel pair).  A wave owns
//     one row of the 4 x 32 tile and walks it two cells at a time.  Per pair of cells it first works lanes-as-taps (2 cells x
//     up to 8 cameras x 4 points = 64 lanes): pixel position, window test, the four corner weights x attention weight x
//     scale, written as (weight, record offset) entries into a wave-private 2 KB table in LDS -- then lanes-as-(corner, pair):
//     one ds_read_b64 of the entry, two multiplies with the lane's grad_out pair of that camera (registers, loaded once per
//     pair of cells), fixed-point packing, one ds_add_u64.  Taps outside the window (or cells outside the map) have weight
//     zero and record 0: the stream is branch-free.  The next pair's sampling data and grad_out rows are requested before
//     the current pair's taps are worked (software pipeline over the 16 pairs of the row);
//   * the weight-mass bound (scale selection, see msda_bwd_value_win) costs one ds_add_u32 per tap instead of four: the whole
//     |weight| goes to the tap's base token and a token's bound is the sum over its 2 x 2 up-left neighbourhood -- an upper
//     bound of the exact mass, at most 2 bits looser;
//   * flush, far taps, non-finite jobs, unequal level shapes and the locality stand-down are msda_bwd_value_win's.
//
// LDS: 45,056 B window + 8,704 B (weight mass during the bound pass, the four waves' tap tables afterwards) = 53,760 B:
// three workgroups per CU.
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

namespace {

// pixel coordinate of a normalised location; ONE expression for the bound pass and the accumulation pass, so that both
// see the same bits (a tap must be inside the window in both or in neither)
__device__ __forceinline__ float pix(float loc, float size) { return __fmaf_rn(loc, size, -0.5f); }

}  // namespace

#ifndef MVDETR_VT_WGS
#define MVDETR_VT_WGS 3      // workgroups per CU (register budget 512 / this per lane)
#endif
#ifndef MVDETR_VT_RB
#define MVDETR_VT_RB 2       // 16-byte entry reads (2 taps each) per batch of the accumulation stream
#endif
template <int D, int FUSED>
__global__ __launch_bounds__(256, MVDETR_VT_WGS) void msda_bwd_value_tok(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const int *__restrict__ local_hits, const float *__restrict__ ref, int64_t ref_bstride, int raw_q)
{
    constexpr int TH = 4, TW = 32, R = 6, WH = TH + 2 * R, WW = TW + 2 * R, LCH = 16, NPAIR = LCH / 2, P = TILE_P, THREADS = 256;
    constexpr int NTOK = WH * WW;                             // 704 tokens x 64 B
    constexpr int CAMS = 8;                                   // cameras per pass of a pair of cells: 2 x 8 x 4 taps = 64 lanes
    static_assert(D % LCH == 0 && THREADS / 64 == TH && TW % 2 == 0, "a wave per tile row, cells in pairs");
    extern __shared__ __attribute__((aligned(16))) long long win64[];
    int *const mass = reinterpret_cast<int *>(win64 + NTOK * NPAIR);                       // [NTOK], bound pass
    // tap tables: [4 waves][2 cells][4 corners][TSTRIDE taps] (weight, record offset); a (cell, corner)'s 32 taps are one run, so
    // a lane reads two taps per ds_read_b128, and the runs start 272 bytes apart: the 8 runs a wave instruction reads from fall
    // on different banks
    constexpr int TSTRIDE = 34, TAB = 2 * 4 * TSTRIDE;
    float2 *const table = reinterpret_cast<float2 *>(win64 + NTOK * NPAIR);
    // (no static LDS: 53,760 B is 42 allocation granules of 1,280 B, three workgroups per CU; 32 more bytes make it two.)  The
    // block reductions' eight slots sit in the two padding entries at the end of the last two runs of wave 3's table -- bytes
    // no tap entry is ever written to: a wave that leaves the second reduction early starts writing ITS table while a slower
    // wave still reads the slots (they may not share bytes with any table's entries: a first version had them behind the mass
    // array, i.e. inside wave 1's table, and one case of the seeded shape sweep failed once in a while)
    float *const red0 = reinterpret_cast<float *>(table + 3 * TAB + 6 * TSTRIDE + 32);
    float *const red1 = reinterpret_cast<float *>(table + 3 * TAB + 7 * TSTRIDE + 32);
    static_assert(NTOK * 4 <= 4 * TAB * 8 && TSTRIDE - 32 >= 2, "the mass array lives in the tap tables' space; two spare entries per run");
    // fixed point: floor(x + 1/2) in one instruction (v_rndne + v_cvt are two; ties are measure zero)
    auto rpi = [](float x) {
        int r;
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        return r;
    };
    auto pack2 = [&](float lo_f, float hi_f) {
        const int lo = rpi(lo_f), hi = rpi(hi_f) + (lo >> 31);
        return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HS = M * D / LCH;
    const int64_t row = (int64_t)M * D;
    // bound pass: lanes = cells (two half-blocks take alternate cameras)
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
        // and the sampling kernel, which sees the same shapes, stands down
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
    constexpr int HPS = 32 / D;
    // sampling data of (query q, head, level l): normalised locations (x, y) x 4 points in la / lb, weights in wa
    auto fetch = [&](int64_t q, int b, int head, int l, float4 &la, float4 &lb, float4 &wa) {
        if constexpr (FUSED) {
            const float *rp = loc + q * raw_q + (l * (M / HPS) + head / HPS) * (HPS * P * 3);
            const float4 oa = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2);
            const float4 ob = *reinterpret_cast<const float4 *>(rp + (head % HPS) * P * 2 + 4);
            const float4 lg = *reinterpret_cast<const float4 *>(rp + HPS * P * 2 + (head % HPS) * P);
            const float2 r = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)l * S + (q - (int64_t)b * S)) * 2);
            const float2 st = *reinterpret_cast<const float2 *>(aw + (q * M + head) * 2);
            la = make_float4(__fmaf_rn(oa.x, iw, r.x), __fmaf_rn(oa.y, ih, r.y), __fmaf_rn(oa.z, iw, r.x), __fmaf_rn(oa.w, ih, r.y));
            lb = make_float4(__fmaf_rn(ob.x, iw, r.x), __fmaf_rn(ob.y, ih, r.y), __fmaf_rn(ob.z, iw, r.x), __fmaf_rn(ob.w, ih, r.y));
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
    };
    auto tap_of = [&](const TapRaw &t, float &x, float &y, float &a) {
        if constexpr (FUSED) {
            x = pix(__fmaf_rn(t.o.x, iw, t.r.x), fW);
            y = pix(__fmaf_rn(t.o.y, ih, t.r.y), fH);
            a = __expf(t.w - t.st.x) * t.st.y;
        } else {
            x = pix(t.o.x, fW);
            y = pix(t.o.y, fH);
            a = t.w;
        }
    };

    for (int i = tid; i < NTOK * NPAIR; i += THREADS) win64[i] = 0;
    for (int i = tid; i < NTOK; i += THREADS) mass[i] = 0;
    __syncthreads();

    // block-wide maxima of two non-negative values (NaN-free); two barriers
    auto block_max2 = [&](float &a, float &b) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = fmaxf(a, __shfl_xor(a, o, 64));
            b = fmaxf(b, __shfl_xor(b, o, 64));
        }
        __syncthreads();
        if ((tid & 63) == 0) { red0[tid >> 6] = a; red1[tid >> 6] = b; }
        __syncthreads();
        a = fmaxf(fmaxf(red0[0], red0[1]), fmaxf(red0[2], red0[3]));
        b = fmaxf(fmaxf(red1[0], red1[1]), fmaxf(red1[2], red1[3]));
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
        int shx, shy;                                         // where this head's taps lie (locality probe)
        msda_probe_shift(local_hits, head, shx, shy);
        if constexpr (FUSED) {
            // no probe in front of the fused backward: every wave reduces the same sample (the tile's first two rows, camera
            // 0, this level) to the head's mean tap displacement, as the forward does
            const int s_qy = Y0 + lane / TW, s_qx = X0 + lane % TW;
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
        auto in_window = [&](float x, float y) { return fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1); };

        // ---- pass 0: bounds (lanes = cells).  Gmax = largest |grad_out| of the job (inf if any is not finite), Amax = largest
        //      sum_p |aw[l][p]|; then a bound on the weight mass sum |aw| * bilinear weight landing on any one window token:
        //      |grad_value contribution| <= Gmax * mass, so the accumulators can use (almost) all 31 bits.
        float4 la8[4], lb8[4], wa8[4];
        int64_t camq[8];                                      // first token of the pass's cameras (uniform: scalar loads, once)
        auto load_camq = [&](int c0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) camq[k] = lsi[c0 + k < L ? c0 + k : L - 1];
        };
        auto load8 = [&](int c0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t cq = chalf ? camq[2 * k + 1] : camq[2 * k];       // (a camera past L repeats camera L - 1)
                fetch((int64_t)b * S + cq + cell, b, head, l, la8[k], lb8[k], wa8[k]);
            }
        };
        float gmax = 0.f, al = 0.f;
        for (int c0 = 0; c0 < L; c0 += 8) {
            load_camq(c0);
            load8(c0);
            float m = 0.f;
            for (int c = c0; c < L && c < c0 + 8; ++c) {
                const float *gp = go + ((int64_t)b * S + lsi[c] + cell) * row + ch0 + 8 * chalf;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
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
        // ---- set-up of pass 1 (lanes = taps, then lanes = (cell, corner, channel pair)); its first loads go out here
            float2 *const tab = table + wave * TAB;                             // this wave's tap table
            const int wy_row = Y0 + wave;                                       // the tile row of this wave
            // lanes as taps
            const int pc = lane >> 5, pt = lane & 31, pcam = pt >> 2, pp = pt & 3;
            // lanes as (cell, corner, pair)
            const int ct = lane >> 5, corner = (lane >> 3) & 3, pair = lane & 7;
            const float2 *const my_entries = tab + (ct * 4 + corner) * TSTRIDE;
            const int pair_off = pair * 8;
            // Addresses: per-batch-element base pointers (uniform) + 32-bit byte offsets per lane (the launcher checks that one
            // batch element's tensors stay below 4 GB) -- no 64-bit multiplies and no dependent loads in the pipelined requests;
            // the cameras' first tokens (lsi) are read once per pass of 8 cameras.
            const char *const go_b = reinterpret_cast<const char *>(go + ((int64_t)b * S * row + ch0));
            const char *loc_b, *aw_b, *ref_b = nullptr, *st_b = nullptr;
            unsigned loc_q, loc_c, aw_q, aw_c;               // bytes per query / constant part of this lane's tap
            if constexpr (FUSED) {
                loc_b = reinterpret_cast<const char *>(loc + (int64_t)b * S * raw_q);
                aw_b = loc_b;
                loc_q = aw_q = (unsigned)raw_q * 4u;
                const unsigned run = (unsigned)((l * (M / HPS) + head / HPS) * (HPS * P * 3));
                loc_c = (run + (unsigned)((head % HPS) * P * 2 + pp * 2)) * 4u;
                aw_c = (run + (unsigned)(HPS * P * 2 + (head % HPS) * P + pp)) * 4u;
                ref_b = reinterpret_cast<const char *>(ref + b * ref_bstride + (int64_t)l * S * 2);
                st_b = reinterpret_cast<const char *>(aw + ((int64_t)b * S * M + head) * 2);
            } else {
                loc_b = reinterpret_cast<const char *>(loc + (int64_t)b * S * M * L * P * 2);
                aw_b = reinterpret_cast<const char *>(aw + (int64_t)b * S * M * L * P);
                loc_q = (unsigned)(M * L * P * 2) * 4u;
                aw_q = (unsigned)(M * L * P) * 4u;
                loc_c = (unsigned)((head * L + l) * P * 2 + pp * 2) * 4u;
                aw_c = (unsigned)((head * L + l) * P + pp) * 4u;
            }
            const unsigned row_b = (unsigned)row * 4u;
            const bool row_ok = wy_row < Hq;

            unsigned cam_q[CAMS];                             // first token of the pass's cameras (uniform)
            bool tap_cam_ok = false;
            unsigned tap_q = 0;                               // of this lane's tap
            auto set_chunk = [&](int c0) {
#pragma unroll
                for (int k = 0; k < CAMS; ++k) cam_q[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)lsi[c0 + k < L ? c0 + k : L - 1]);
                tap_cam_ok = c0 + pcam < L;
                tap_q = (unsigned)lsi[tap_cam_ok ? c0 + pcam : 0];
            };
            constexpr int steps = TW / 2;

            TapRaw nraw;
            float2 ng[CAMS];
            unsigned n_gofs = 0;                              // of the tap lane's (cell, camera) row, for the far path
            bool n_valid = false;
            auto request_taps = [&](int j, TapRaw &t, bool &ok, unsigned &gofs) {      // lanes as taps: this lane's tap
                const int x_ = X0 + 2 * j + pc;
                ok = row_ok && x_ < Wq && tap_cam_ok;
                const unsigned q = ok ? tap_q + (unsigned)(wy_row * Wq + x_) : 0u;            // inside the batch element
                gofs = q * row_b;
                t.o = *reinterpret_cast<const float2 *>(loc_b + (q * loc_q + loc_c));
                t.w = *reinterpret_cast<const float *>(aw_b + (q * aw_q + aw_c));
                if constexpr (FUSED) {
                    t.r = *reinterpret_cast<const float2 *>(ref_b + q * 8u);
                    t.st = *reinterpret_cast<const float2 *>(st_b + q * (unsigned)(M * 8));
                } else {
                    t.r = t.st = make_float2(0.f, 0.f);
                }
            };
            auto request_g = [&](int j) {        // lanes as (cell, corner, pair): grad_out pairs of the cell's cameras
                const int x_ = X0 + 2 * j + ct;
                const unsigned cellq = row_ok && x_ < Wq ? (unsigned)(wy_row * Wq + x_) : 0u;
                const unsigned o = cellq * row_b + (unsigned)pair * 8u;
#pragma unroll
                for (int k = 0; k < CAMS; ++k) ng[k] = *reinterpret_cast<const float2 *>(go_b + (cam_q[k] * row_b + o));
            };
        float scale = 0.f, inv_scale = 0.f;
        if (!direct_only) {
            // weight-mass fixed point: a lane adds at most Amax per (camera, level), TH*TW*L of them -> < 2^30 in all (+ rounding up)
            int ew = 0;
            (void)frexpf(Amax * (float)(TH * TW * L), &ew);
            ew = ew < -60 ? -60 : ew;
            const float wscale = ldexpf(1.f, 30 - ew);
            for (int c0 = 0; c0 < L; c0 += 8) {
                if (L > 8) { load_camq(c0); load8(c0); }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (c0 + 2 * k + chalf >= L) break;
                    const float xs[4] = {pix(la8[k].x, fW), pix(la8[k].z, fW), pix(lb8[k].x, fW), pix(lb8[k].z, fW)};
                    const float ys[4] = {pix(la8[k].y, fH), pix(la8[k].w, fH), pix(lb8[k].y, fH), pix(lb8[k].w, fH)};
                    const float as[4] = {wa8[k].x, wa8[k].y, wa8[k].z, wa8[k].w};
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = xs[p], y = ys[p];
                        if (active && in_window(x, y)) {
                            // the whole |weight| at the tap's base token (rounded UP); its four corners are that token and
                            // its right / lower / lower-right neighbours, so a token's mass is bounded by the sum over its
                            // up-left 2 x 2 neighbourhood (below)
                            const int tok = ((int)floorf(y) - oy) * WW + ((int)floorf(x) - ox);
                            __hip_atomic_fetch_add(mass + tok, __float2int_ru(fabsf(as[p]) * wscale), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            __syncthreads();
            int wm = 0;
            for (int i = tid; i < NTOK; i += THREADS) {
                const int wy = i / WW, wx = i % WW;
                int s = mass[i];
                if (wx > 0) s += mass[i - 1];
                if (wy > 0) s += mass[i - WW];
                if (wx > 0 && wy > 0) s += mass[i - WW - 1];
                wm = max(wm, s);
            }
            float Wmax = (float)wm, unused = 0.f;
            block_max2(Wmax, unused);                         // (its first barrier: every lane has read the mass array)
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
        // ---- pass 1: the accumulation (lanes = taps, then lanes = (cell, corner, channel pair)), and the taps outside the window
        {
            for (int c0 = 0; c0 < L; c0 += CAMS) {
            // (requesting the first pair before the mass bound, and batching the bound pass's grad_out loads, were both
            // measured SLOWER: 690 - 725 us against 671 for the whole backward -- DESIGN 4.3d)
            set_chunk(c0);
            request_taps(0, nraw, n_valid, n_gofs);
            request_g(0);
            for (int s = 0; s < steps; ++s) {
                const TapRaw raw_ = nraw;
                float2 g[CAMS];
#pragma unroll
                for (int k = 0; k < CAMS; ++k) g[k] = ng[k];
                const bool valid = n_valid;
                const unsigned my_gofs = n_gofs;
                // (requesting the tap data TWO steps ahead was measured too: 685 - 690 us against 662 - 677 for the whole backward)
                if (s + 1 < steps) {
                    request_taps(s + 1, nraw, n_valid, n_gofs);
                    request_g(s + 1);
                }

                if (s < 4 && c0 == 0) BTRACE(tr + 6 + 2 * s);
                // ---- lanes as taps: the four (weight, record) entries of this lane's tap
                float x, y, a;
                tap_of(raw_, x, y, a);
                const bool inw = in_window(x, y);
                const bool hit = valid && !direct_only && inw;
                {
                    const float fx = floorf(x), fy = floorf(y);
                    const int tok = hit ? ((int)fy - oy) * WW + ((int)fx - ox) : 0;
                    const float wx1 = x - fx, wy1 = y - fy;
                    const float sw = hit ? a * scale : 0.f, ay1 = hit ? wy1 * sw : 0.f, ay0 = sw - ay1;
                    const float w01 = hit ? ay0 * wx1 : 0.f, w00 = ay0 - w01, w11 = hit ? ay1 * wx1 : 0.f, w10 = ay1 - w11;
                    const int o00 = tok * (NPAIR * 8), o10 = hit ? o00 + WW * (NPAIR * 8) : 0, o01 = hit ? o00 + NPAIR * 8 : 0;
                    float2 *e = tab + pc * 4 * TSTRIDE + pt;
                    e[0] = make_float2(w00, __int_as_float(o00));
                    e[TSTRIDE] = make_float2(w01, __int_as_float(o01));
                    e[2 * TSTRIDE] = make_float2(w10, __int_as_float(o10));
                    e[3 * TSTRIDE] = make_float2(w11, __int_as_float(hit ? o10 + NPAIR * 8 : 0));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                if (s < 4 && c0 == 0) BTRACE(tr + 7 + 2 * s);
                // ---- lanes as (cell, corner, pair): one ds_add_u64 per tap of the lane's cell
                //      (entries read two taps at a time, a batch ahead of the adds that use them: an add must not wait for the LDS
                //      round trip of its own entry)
                if (!direct_only) {
                    // (2 x 16 bytes = 4 taps per batch: 4, 8 and 16 were measured too -- 8 and 16 spill inside the loop and lose 6 %,
                    // 4 leaves 16 - 44 bytes of scratch per lane and is 1 - 2 % slower than 2, which has none)
                    constexpr int RB = MVDETR_VT_RB, NB = CAMS * P / 2 / RB;             // float4 (= 2 taps) per batch, batches per step
                    float4 nx[RB];
                    auto read_batch = [&](int bt) {
#pragma unroll
                        for (int k = 0; k < RB; ++k) nx[k] = *reinterpret_cast<const float4 *>(my_entries + (bt * RB + k) * 2);
                    };
                    read_batch(0);
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        if (c0 + bt * RB / 2 >= L) break;                       // (uniform)
                        float4 cu[RB];
#pragma unroll
                        for (int k = 0; k < RB; ++k) cu[k] = nx[k];
                        if (bt + 1 < NB) read_batch(bt + 1);
#pragma unroll
                        for (int k = 0; k < RB; ++k) {
                            const int cam = (bt * RB + k) / 2;
                            if (c0 + cam >= L) break;                           // (uniform)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float wgt = h ? cu[k].z : cu[k].x;
                                const int off = __float_as_int(h ? cu[k].w : cu[k].y);
                                const long long v = pack2(wgt * g[cam].x, wgt * g[cam].y);
                                long long *w = reinterpret_cast<long long *>(reinterpret_cast<char *>(win64) + off + pair_off);
                                __hip_atomic_fetch_add(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();

                // ---- taps outside the window (or every tap of a non-finite job): straight to memory, the whole wave on one
                //      tap at a time with lanes = (corner, channel)
                const bool miss = valid && (direct_only || !inw) && y > -1.f && x > -1.f && y < fH && x < fW;
                unsigned long long pend = __ballot(miss);
                while (pend) {
                    const int src = __ffsll((long long)pend) - 1;
                    pend &= pend - 1;
                    const float sx = __shfl(x, src, 64), sy = __shfl(y, src, 64), sa = __shfl(a, src, 64);
                    const unsigned sg = (unsigned)__shfl((int)my_gofs, src, 64);
                    const int cr = lane >> 4, j = lane & 15;
                    const float gk = *reinterpret_cast<const float *>(go_b + (sg + (unsigned)j * 4u));
                    const Footprint<float> f = footprint(sy, sx, Hq, Wq);
                    const int yy = f.y0 + (cr >> 1), xx = f.x0 + (cr & 1);
                    const float wgt = ((cr >> 1) ? f.wy1 : f.wy0) * ((cr & 1) ? f.wx1 : f.wx0);
                    if ((unsigned)yy < (unsigned)Hq && (unsigned)xx < (unsigned)Wq)
                        atomicAdd(grad_value + level_base + ch0 + ((int64_t)yy * Wq + xx) * row + j, wgt * (gk * sa));
                }
            }
            }
        }
        BTRACE(tr + 3);
        __syncthreads();
        BTRACE(tr + 4);
        // ---- flush: the touched tokens' 64-byte records as fp32 atomics; leaves the window and the mass array zeroed ----
        if (!direct_only) {
            // lane = (token, channel): the two lanes of a channel pair read the same qword, one of them clears it
            const int ch = tid % LCH, pair = ch >> 1;
            const bool upper = ch & 1;
            float *const gbase = grad_value + level_base + ch0 + ch;
            for (int i0 = tid / LCH; i0 < NTOK; i0 += 8 * (THREADS / LCH)) {
                long long v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int tok = i0 + k * (THREADS / LCH);
                    v[k] = tok < NTOK ? win64[tok * NPAIR + pair] : 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int tok = i0 + k * (THREADS / LCH);
                    if (v[k] != 0) {
                        if (!upper) win64[tok * NPAIR + pair] = 0;
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
        for (int i = tid; i < NTOK; i += THREADS) mass[i] = 0;   // (the tap tables lived there)
        BTRACE(tr + 5);
        __syncthreads();                                      // window and mass array are zero again before the next job
    }
}

template <int D, int FUSED>
int launch_value_tok(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                     const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                     float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits, const float *ref,
                     int64_t ref_bstride, int raw_q)
{
    constexpr int LDS = 16 * 44 * 64 + 4 * 2 * 4 * 34 * 8;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_value_tok<D, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_value_tok<D, FUSED>, 256, LDS) != hipSuccess || per_cu < 1)
            per_cu = 3;
        if (per_cu > MVDETR_VT_WGS) per_cu = MVDETR_VT_WGS;
        return (cus * per_cu + 7) / 8 * 8;
    });
    hipLaunchKernelGGL((msda_bwd_value_tok<D, FUSED>), dim3((unsigned)blocks), dim3(256), LDS, st, go, value, shapes, lsi, loc, aw,
                       B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, ref, ref_bstride, raw_q);
    return (int)hipGetLastError();
}

int msda_backward_value_tok(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                            const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                            float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    if (D == 16) return launch_value_tok<16, 0>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, nullptr, 0, 0);
    if (D == 32) return launch_value_tok<32, 0>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits, nullptr, 0, 0);
    return (int)hipErrorInvalidValue;
}

int msda_backward_value_tok_fused(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                  const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                  const float *stats, int B, int S, int M, int D, int L, float *grad_value)
{
    if (D == 16) return launch_value_tok<16, 1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, nullptr, nullptr, nullptr, ref, ref_bstride, raw_q);
    if (D == 32) return launch_value_tok<32, 1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_value, nullptr, nullptr, nullptr, ref, ref_bstride, raw_q);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
