// EXPERIMENT, not part of libmvdetr_ops.so.  Round 2 shipped this file inside the library behind MVDETR_MSDA_QUAD=1 without a
// test that set the variable; it is parity-green but slower than msda_fwd_group (152 / 170 us against
// 128 / 139 us), so it was taken out of the product.  It builds against mvdetr_amd/csrc/{common.h,msda_tile.h}.
// Multi-scale deformable attention forward, camera-grouped "quad" kernel -- gfx950 (MI355X).
//
// Same job as msda_forward_group.hip -- one workgroup owns a (6 x 16 cell tile, 128-byte slice) and walks all
// NG = L query levels (cameras) per staged source window -- but with the work of one (cell, head) spread over a
// QUAD of lanes, 4 channels (one 16-byte chunk) each, which is what makes the rest possible:
//
//   * registers: a lane carries NG x 4 accumulators instead of NG x 16 (28 instead of 112 at 7 cameras; the group
//     kernel sits at 254 VGPRs with nothing left to pipeline with), so there is room for
//   * a software pipeline over the taps: the four ds_read_b128 of tap p+1 are in flight while tap p's sixteen
//     multiply-adds issue, across camera boundaries (the next camera's addresses and weights are computed while
//     the last two taps' reads fly).  A wave is an in-order machine; measured on the unpipelined version, a
//     (camera, level) step of ~125 VALU instructions took 1.3 us at 3 waves per SIMD -- read, wait, multiply, four
//     times over -- against 0.5 us of VALU time;
//   * sampling data requested three (camera, level) steps ahead, window chunks of the next level requested at the
//     start of this one (double-buffered window, one barrier per level): the copy costs the taps nothing;
//   * the per-tap arithmetic is not duplicated: lane j of the quad owns sampling point j of the (camera, level)
//     -- P == 4 -- computes that tap's LDS addresses and its four corner weights ONCE, and the quad's lanes pick
//     them up with DPP quad broadcasts;
//   * LDS bank conflicts are gone BY CONSTRUCTION, for any sampling locations: a ds_read_b128 is served in four
//     16-lane groups, i.e. four quads per group, each quad reading the 64 contiguous bytes of one (token, head).
//     With 128-byte tokens the slot of those 64 bytes in the 256-byte bank row is (token parity, head); the two
//     quads of a group that share a head are given opposite "roles" r, and every quad reads its two x-neighbour
//     corners in the order (token of parity r, token of parity 1-r) -- the two corners of a bilinear footprint
//     always differ in parity -- swapping the two x weights to match.  Every instruction then covers all four
//     slots exactly once (SQ_LDS_BANK_CONFLICT = 0; the group kernel: 46 % of its LDS cycles).
//
// Taps whose footprint leaves the window get zero weights in the main loop (no divergence) and a bit in a
// per-lane miss mask; they are finished from global memory after the last level -- correct for ANY locations.
// Shapes live on the device: if the levels turn out unequal the launch runs a plain per-(query, head) body.
//
// Replaces ms_deformable_im2col_gpu_kernel of the reference
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299) for encoder-shaped fp32 calls with
// equal level shapes (MVDeTr: levels = cameras), plus -- FUSED -- the module arithmetic around it
// (multiview_detector/models/ops/modules/ms_deform_attn.py:100-107).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_gather_body.h"
#include <stdlib.h>
#include <string.h>

#ifdef MVDETR_QUAD_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup's waves
__device__ unsigned long long g_quad_trace[2048];
extern "C" int mvdetr_debug_quad_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_quad_trace), n * sizeof(unsigned long long));
}
#define QTRACE(i) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (i) < 2048) g_quad_trace[(i)] = wall_clock64(); } while (0)
#else
#define QTRACE(i) do { } while (0)
#endif

namespace mvdetr {

namespace quad {

constexpr int TH = 6, TW = 16, R = 6;
constexpr int WH = TH + 2 * R, WW = TW + 2 * R;          // 18 x 28 tokens
constexpr int SLICE = 32;                                 // floats of a token row per workgroup (128 B)
constexpr int TOKB = SLICE * 4;                           // bytes per token in LDS
constexpr int WIN_FLOATS = WH * WW * SLICE;               // 16,128 floats = 64,512 B (a multiple of 256 B)
constexpr int THREADS = TH * TW * 8;                      // 8 lanes per cell: 2 sub-slices x 4 chunks = 768
constexpr int COPY_ITEMS = WH * WW * (SLICE / 4);         // float4 per window = 4,032
constexpr int NSTAGE = (COPY_ITEMS + THREADS - 1) / THREADS;   // 6
constexpr int LDS_BYTES = 2 * WIN_FLOATS * 4;             // double-buffered window
constexpr int AHEAD = 3;                                  // sampling data: (camera, level) steps requested ahead
static_assert(WW % 2 == 0, "token parity == column parity needs an even window width");
static_assert((WIN_FLOATS * 4) % 256 == 0, "both buffers start on a bank row");

// Raw buffer loads: the address is (SGPR descriptor) + (SGPR byte offset) + (one VGPR byte offset), so the per-camera /
// per-level part of every address lives in scalar registers and a lane keeps ONE offset register per tensor (flat
// loads made hipcc build a 64-bit VGPR address per camera: 40+ registers).  Reads beyond `bytes` return 0, which is
// also how window chunks outside the level get their zeros.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f1(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float2 buf_f2(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float4 buf_f4(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
constexpr unsigned OOB = 0x80000000u;                     // a byte offset no supported tensor reaches

template <int P> __device__ __forceinline__ int qb_i(int v)
{
    constexpr int ctrl = P | (P << 2) | (P << 4) | (P << 6);       // quad_perm:[P,P,P,P]
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
}
template <int P> __device__ __forceinline__ float qb_f(float v)
{
    return __builtin_bit_cast(float, qb_i<P>(__builtin_bit_cast(int, v)));
}
// (k is a constant after unrolling)
__device__ __forceinline__ float qb_sel(float v, int k)
{
    return k == 0 ? qb_f<0>(v) : k == 1 ? qb_f<1>(v) : k == 2 ? qb_f<2>(v) : qb_f<3>(v);
}
// butterfly over the quad: quad_perm:[1,0,3,2] then [2,3,0,1]
__device__ __forceinline__ float quad_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, quad_xor1(v));
    return fmaxf(v, quad_xor2(v));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += quad_xor1(v);
    return v + quad_xor2(v);
}

// A tap's descriptor, computed by its owner lane (lane P of the quad for point P): two LDS byte addresses (the
// x-neighbour corners in this quad's parity order, see the header) and the four corner weights, attention weight
// folded in.  Zero weights and address 0 for taps outside the window.
struct Desc {
    int addrA, addrB;
    float wAt, wAb, wBt, wBb;
};
// the four corners of one tap, as read from LDS by this lane (its 16-byte chunk of each)
struct Corners {
    float4 At, Ab, Bt, Bb;
};
template <int P> __device__ __forceinline__ Corners tap_read(const char *lane_base, const Desc &d)
{
    const char *pa = lane_base + qb_i<P>(d.addrA), *pb = lane_base + qb_i<P>(d.addrB);
    Corners c;
    c.At = *reinterpret_cast<const float4 *>(pa);
    c.Ab = *reinterpret_cast<const float4 *>(pa + WW * TOKB);
    c.Bt = *reinterpret_cast<const float4 *>(pb);
    c.Bb = *reinterpret_cast<const float4 *>(pb + WW * TOKB);
    return c;
}
// (measured, tools/experiments/valu_lds_rate.hip: v_fmac_f32_dpp issues at half the rate of v_fmac_f32 / v_pk_fma_f32,
// so each weight is broadcast once with v_mov_b32_dpp and the multiply-adds are plain)
template <int P> __device__ __forceinline__ void tap_fma(const Desc &d, const Corners &c, float4 &acc)
{
    const float a_t = qb_f<P>(d.wAt), a_b = qb_f<P>(d.wAb), b_t = qb_f<P>(d.wBt), b_b = qb_f<P>(d.wBb);
    acc.x = fmaf(a_t, c.At.x, acc.x); acc.y = fmaf(a_t, c.At.y, acc.y); acc.z = fmaf(a_t, c.At.z, acc.z); acc.w = fmaf(a_t, c.At.w, acc.w);
    acc.x = fmaf(a_b, c.Ab.x, acc.x); acc.y = fmaf(a_b, c.Ab.y, acc.y); acc.z = fmaf(a_b, c.Ab.z, acc.z); acc.w = fmaf(a_b, c.Ab.w, acc.w);
    acc.x = fmaf(b_t, c.Bt.x, acc.x); acc.y = fmaf(b_t, c.Bt.y, acc.y); acc.z = fmaf(b_t, c.Bt.z, acc.z); acc.w = fmaf(b_t, c.Bt.w, acc.w);
    acc.x = fmaf(b_b, c.Bb.x, acc.x); acc.y = fmaf(b_b, c.Bb.y, acc.y); acc.z = fmaf(b_b, c.Bb.z, acc.z); acc.w = fmaf(b_b, c.Bb.w, acc.w);
}

}  // namespace quad

// Plain per-(query, head) formulation for level shapes the windows cannot serve (unequal levels): lane = one
// (b, q, head), channels in chunks of 4.  Slow and only there for correctness; any SamplingLayout, FUSED as above.
template <int FUSED>
__device__ void msda_fwd_quad_generic(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                      const int64_t *__restrict__ lsi, const float *__restrict__ off,
                                      const float *__restrict__ logit, const float *__restrict__ ref, int64_t ref_bstride,
                                      SamplingLayout lay, int B, int S, int M, int D, int L, float *__restrict__ out)
{
    constexpr int P = TILE_P;
    const int64_t total = (int64_t)B * S * M;
    const int row = M * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx % M);
        const int64_t bq = idx / M;
        const int b = (int)(bq / S);
        const int64_t q = bq - (int64_t)b * S;
        const float *lp = off + bq * lay.q_l + lay.head_l(m);
        const float *wp = logit + bq * lay.q_w + lay.head_w(m);
        const float *rp = FUSED ? ref + b * ref_bstride + q * lay.r_q : nullptr;
        float mx = -INFINITY, sum = 1.f;
        if constexpr (FUSED != 0) {
            for (int l = 0; l < L; ++l)
                for (int p = 0; p < P; ++p) mx = fmaxf(mx, wp[l * lay.l_w + p]);
            sum = 0.f;
            for (int l = 0; l < L; ++l)
                for (int p = 0; p < P; ++p) sum += __expf(wp[l * lay.l_w + p] - mx);
        }
        const float inv = 1.f / sum;
        for (int c0 = 0; c0 < D; c0 += 4) {
            float4 acc = make_float4(0, 0, 0, 0);
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const float fW = (float)W, fH = (float)H;
                const float *plane = value + ((int64_t)b * S + lsi[l]) * row + m * D + c0;
                for (int p = 0; p < P; ++p) {
                    float lx = lp[l * lay.l_l + p * 2], ly = lp[l * lay.l_l + p * 2 + 1], a = wp[l * lay.l_w + p];
                    if constexpr (FUSED != 0) {
                        const int ri = l * lay.r_l + (FUSED == 2 ? 0 : p * 2);
                        lx = rp[ri] + lx * (1.f / fW);
                        ly = rp[ri + 1] + ly * (1.f / fH);
                        a = __expf(a - mx) * inv;
                    }
                    const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
                    if (!(y > -1.f && x > -1.f && y < fH && x < fW)) continue;
                    const Footprint<float> f = footprint(y, x, H, W);
                    const float *r0 = plane + ((int64_t)f.y0 * W + f.x0) * row, *r1 = r0 + (int64_t)W * row;
                    const float4 z = make_float4(0, 0, 0, 0);
                    const float4 c00 = (f.vy0 && f.vx0) ? *reinterpret_cast<const float4 *>(r0) : z;
                    const float4 c01 = (f.vy0 && f.vx1) ? *reinterpret_cast<const float4 *>(r0 + row) : z;
                    const float4 c10 = (f.vy1 && f.vx0) ? *reinterpret_cast<const float4 *>(r1) : z;
                    const float4 c11 = (f.vy1 && f.vx1) ? *reinterpret_cast<const float4 *>(r1 + row) : z;
                    const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a, w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
                    acc.x += w00 * c00.x + w01 * c01.x + w10 * c10.x + w11 * c11.x;
                    acc.y += w00 * c00.y + w01 * c01.y + w10 * c10.y + w11 * c11.y;
                    acc.z += w00 * c00.z + w01 * c01.z + w10 * c10.z + w11 * c11.z;
                    acc.w += w00 * c00.w + w01 * c01.w + w10 * c10.w + w11 * c11.w;
                }
            }
            *reinterpret_cast<float4 *>(out + bq * row + m * D + c0) = acc;
        }
    }
}

// FUSED: 0 = `off` / `logit` hold final sampling locations / attention weights (public contract), `ref` unused;
// 1 = raw offsets / logits + reference points, one per point; 2 = raw + one point per (query, level).
template <int D, int NG, int FUSED>
__global__ __launch_bounds__(quad::THREADS, 3) void msda_fwd_quad(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ off, const float *__restrict__ logit, const float *__restrict__ ref,
    int64_t ref_bstride, SamplingLayout lay, int B, int S, int M, float *__restrict__ out,
    const int *__restrict__ local_hits)
{
    using namespace quad;
    extern __shared__ __attribute__((aligned(256))) float win[];
    constexpr int P = TILE_P, L = NG;
    static_assert(NG <= 8, "8-bit level masks");
    if constexpr (FUSED == 0) {
        // the locality probe found the taps far from their queries: windows would be wasted, gather instead
        if (local_hits && *local_hits * 2 < MSDA_PROBE_SAMPLES) {
            msda_fwd_gather_body<float, 4>((int64_t)blockIdx.x * THREADS + threadIdx.x, (int64_t)gridDim.x * THREADS, value,
                                           shapes, lsi, off, logit, B, S, M, D, NG, S, TILE_P, out);
            return;
        }
    }
    for (int l = 1; l < L; ++l)
        if (shapes[2 * l] != shapes[0] || shapes[2 * l + 1] != shapes[1]) {
            msda_fwd_quad_generic<FUSED>(value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, D, L, out);
            return;
        }

    const int tid = threadIdx.x;
    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int HS = M * D / SLICE, row = M * D;
    // first token of every level, once: a scalar load inside the tap loop would share lgkmcnt with the LDS reads
    int lvl0[NG];
#pragma unroll
    for (int l = 0; l < NG; ++l) lvl0[l] = (int)lsi[l];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * HS * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq, iw = 1.f / fW, ih = 1.f / fH;

    // lane roles: wave = 8 consecutive cells of one tile row; lane = (cell, 64-byte sub-slice, 16-byte chunk)
    const int lane = tid & 63, wave = tid >> 6;
    const int c8 = lane >> 3, sub = (lane >> 2) & 1, j = lane & 3;
    const int qly = wave >> 1, qlx = (wave & 1) * 8 + c8;
    const int role = (c8 >> 1) & 1;                        // see the header: which token parity this quad reads first
    const int lane_byte = sub * 64 + j * 16;

    // t -> job: XCD k (workgroups t = k mod 8) takes a contiguous band of jobs
    auto job_of = [&](int t) { return (t >> 3) < jobs8 ? (t & 7) * jobs8 + (t >> 3) : jobs; };

    for (int t = blockIdx.x;; t += (int)gridDim.x) {
        const int job = job_of(t);
        if (job >= jobs) break;                            // (uniform)
        const int hs = job % HS, u2 = job / HS;
        const int tin = u2 % per_level, b = u2 / per_level;
        const int oy = (tin / tcols) * TH - R, ox = (tin % tcols) * TW - R;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        const int qy = oy + R + qly, qx = ox + R + qlx;
        const bool active = qy < Hq && qx < Wq;
        const int cell = active ? qy * Wq + qx : 0;
        const int head = (hs * SLICE + sub * 16) / D;
        const int64_t bS = (int64_t)b * S;
        // per-lane parts of the sampling-data addresses (bytes inside the tensor of batch element b): own point j
        const unsigned lane_l = (unsigned)(cell * lay.q_l + lay.head_l(head) + j * 2) * 4u;
        const unsigned lane_w = (unsigned)(cell * lay.q_w + lay.head_w(head) + j) * 4u;
        const unsigned lane_r = (unsigned)(cell * lay.r_q + (FUSED == 2 ? 0 : j * 2)) * 4u;
        const float *refb = FUSED ? ref + b * ref_bstride : nullptr;
        const rsrc_t r_val = make_rsrc(value + bS * row, (unsigned)S * row * 4u);
        const rsrc_t r_off = make_rsrc(off + bS * lay.q_l, 0x7ffffff0u);
        const rsrc_t r_log = make_rsrc(logit + bS * lay.q_w, 0x7ffffff0u);
        const rsrc_t r_ref = make_rsrc(FUSED ? (const void *)refb : (const void *)value, 0x7ffffff0u);

        float4 acc[NG];
        // running softmax state (max, sum) of camera c: the four lanes of a quad would hold identical copies, so lane
        // (c mod 4) keeps it and the others read it with a quad broadcast -- (NG + 3) / 4 registers each instead of NG
        constexpr int NS = (NG + 3) / 4;
        float smax_s[NS], ssum_s[NS];
        unsigned miss[NS];                                 // bit (c mod 4) * 8 + l of word c / 4: own point's tap left the window
#pragma unroll
        for (int c = 0; c < NG; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            smax_s[k] = -INFINITY;
            ssum_s[k] = 0.f;
            miss[k] = 0;
        }

        // ---- window staging: LDS-DMA (buffer_load ... lds), no registers: chunk i = tid + k * THREADS of the window goes
        // to byte 16 * i of the buffer, i.e. a wave's 64 chunks land contiguously behind a wave-uniform base; chunks
        // outside the level are out-of-range reads and store zeros
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        auto issue = [&](int l, float *buf) {
            const unsigned so = (unsigned)(lvl0[l] * row + hs * SLICE) * 4u;
#pragma unroll
            for (int k = 0; k < NSTAGE; ++k) {
                if (k * THREADS + wave_u * 64 >= COPY_ITEMS) continue;           // (wave-uniform: the last pass is 3 waves wide)
                const int i = tid + k * THREADS;
                const int tok = i >> 3, ch = i & 7;
                const int wy = tok / WW, wx = tok - wy * WW;
                const int gy = oy + wy, gx = ox + wx;
                const bool ok = (unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r_val, (__attribute__((address_space(3))) void *)(buf + (k * THREADS + wave_u * 64) * 4), 16,
                    ok ? (int)((unsigned)((gy * Wq + gx) * row + ch * 4) * 4u) : (int)OOB, (int)so, 0, 0);
            }
        };

        // ---- sampling data of this lane's point, one register set per camera, requested AHEAD steps ahead --------------
        float2 n_o[NG], n_r[NG];
        float n_w[NG];
        auto load_cam = [&](int c, int l) {
            n_o[c] = buf_f2(r_off, lane_l, (unsigned)(lvl0[c] * lay.q_l + l * lay.l_l) * 4u);
            n_w[c] = buf_f1(r_log, lane_w, (unsigned)(lvl0[c] * lay.q_w + l * lay.l_w) * 4u);
            n_r[c] = make_float2(0, 0);
            if constexpr (FUSED != 0) n_r[c] = buf_f2(r_ref, lane_r, (unsigned)(lvl0[c] * lay.r_q + l * lay.r_l) * 4u);
        };
        // descriptor of this lane's point for (camera c, level l) from the data in n_*[c]
        auto describe = [&](int c, int l) {
            const float2 o = n_o[c], r = n_r[c];
            const float lg = n_w[c];
            float x, y, a;
            if constexpr (FUSED != 0) {
                const float m_old = qb_sel(smax_s[c >> 2], c & 3), s_old = qb_sel(ssum_s[c >> 2], c & 3);
                const float m = fmaxf(m_old, quad_max(lg));
                const float sc = __expf(m_old - m);
                a = __expf(lg - m);
                const float s_new = s_old * sc + quad_sum(a);
                smax_s[c >> 2] = j == (c & 3) ? m : smax_s[c >> 2];
                ssum_s[c >> 2] = j == (c & 3) ? s_new : ssum_s[c >> 2];
                acc[c].x *= sc;
                acc[c].y *= sc;
                acc[c].z *= sc;
                acc[c].w *= sc;
                x = (r.x + o.x * iw) * fW - 0.5f;
                y = (r.y + o.y * ih) * fH - 0.5f;
            } else {
                a = lg;
                x = o.x * fW - 0.5f;
                y = o.y * fH - 0.5f;
            }
            // branch-free on purpose: with control flow inside the camera loop the compiler sinks every camera's FMAs
            // below the whole loop and spills the LDS data they wait for
            const bool in = (int)active & (int)(fabsf(x - cx) < 0.5f * (WW - 1)) & (int)(fabsf(y - cy) < 0.5f * (WH - 1));
            const float fx = floorf(x), fy = floorf(y);
            const int ix = in ? (int)fx - ox : 0, iy = in ? (int)fy - oy : 0;
            const float wx1 = in ? x - fx : 0.f, wy1 = in ? y - fy : 0.f;   // (NaN locations must not leak into the weights)
            a = in ? a : 0.f;
            miss[c >> 2] |= ((int)active & (int)!in) ? 1u << (l + 8 * (c & 3)) : 0u;
            const int s = (ix ^ role) & 1;                 // 1: the right-hand corner has this quad's parity
            const float wxA = s ? wx1 : 1.f - wx1;
            const float ay1 = wy1 * a, ay0 = a - ay1;
            Desc d;
            d.wAt = ay0 * wxA;
            d.wBt = ay0 - d.wAt;
            d.wAb = ay1 * wxA;
            d.wBb = ay1 - d.wAb;
            d.addrA = (iy * WW + ix + s) * TOKB;
            d.addrB = d.addrA + (s ? -TOKB : TOKB);
            return d;
        };

        // prologue: window of level 0, sampling data of the first AHEAD steps
        issue(0, win);
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) load_cam(c, 0);
        __syncthreads();

        for (int l = 0; l < L; ++l) {
            [[maybe_unused]] const int tr = wave * 128 + l * 16;
            QTRACE(tr + 0);
            const bool more = l + 1 < L;
            const char *lane_base = reinterpret_cast<const char *>(win + (l & 1) * WIN_FLOATS) + lane_byte;
            float *other = win + ((l & 1) ^ 1) * WIN_FLOATS;

            // Software pipeline over the 4 * NG taps of this level.  Entering step c, its descriptor `d` is known and
            // the reads of its tap 0 are in flight (slot A).  Program order below = issue order; the scheduling
            // fences keep the compiler from hoisting every read of the level to the top (and spilling what comes back).
            Desc d = describe(0, l);
            if (AHEAD < NG) load_cam(AHEAD, l); else if (more) load_cam(AHEAD - NG, l + 1);
            Corners A = tap_read<0>(lane_base, d), Bc;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                // the next level's window goes out once the loads this level still waits for across the loop's back
                // edge (steps 0 .. AHEAD-1; hipcc's vmcnt there is conservative) are behind us
                if (c == (AHEAD < NG ? AHEAD : NG - 1) && more) issue(l + 1, other);
                Bc = tap_read<1>(lane_base, d);
                tap_fma<0>(d, A, acc[c]);
                __builtin_amdgcn_sched_barrier(0);
                A = tap_read<2>(lane_base, d);
                tap_fma<1>(d, Bc, acc[c]);
                __builtin_amdgcn_sched_barrier(0);
                Bc = tap_read<3>(lane_base, d);
                const Desc dc = d;
                if (c + 1 < NG) {
                    // the next camera's descriptor while this camera's last two taps fly; its sampling data was
                    // requested AHEAD steps ago, and the request AHEAD steps on goes out now
                    d = describe(c + 1, l);
                    const int cn = c + 1 + AHEAD;
                    if (cn < NG) load_cam(cn, l); else if (more) load_cam(cn - NG, l + 1);
                }
                tap_fma<2>(dc, A, acc[c]);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NG) A = tap_read<0>(lane_base, d);
                tap_fma<3>(dc, Bc, acc[c]);
                asm volatile("" : "+v"(acc[c].x), "+v"(acc[c].y), "+v"(acc[c].z), "+v"(acc[c].w));   // this camera's FMAs stay here
                __builtin_amdgcn_sched_barrier(0);
                QTRACE(tr + 2 + c);
            }
            QTRACE(tr + 9);
            __syncthreads();
            QTRACE(tr + 10);
        }

        // ---- taps that left their window: straight from global memory (rare) ----------------------------------
        bool any_miss = false;
#pragma unroll
        for (int k = 0; k < NS; ++k) any_miss = any_miss || miss[k] != 0;
        if (__any(any_miss)) {
            const float *vb = value + bS * row + hs * SLICE + sub * 16 + j * 4;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const int mine = (int)((miss[c >> 2] >> (8 * (c & 3))) & 0xffu);
                const float smax_c = qb_sel(smax_s[c >> 2], c & 3);
                const int64_t cq = bS + lsi[c];
                const float *lp = off + cq * lay.q_l + (lane_l / 4 - j * 2);
                const float *wp = logit + cq * lay.q_w + (lane_w / 4 - j);
                const float *rp = FUSED ? refb + lsi[c] * lay.r_q + (lane_r / 4 - (FUSED == 2 ? 0 : j * 2)) : nullptr;
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    int mm = pp == 0 ? qb_i<0>(mine) : pp == 1 ? qb_i<1>(mine) : pp == 2 ? qb_i<2>(mine) : qb_i<3>(mine);
                    while (mm) {
                        const int l = __ffs(mm) - 1;
                        mm &= mm - 1;
                        float lx = lp[l * lay.l_l + pp * 2], ly = lp[l * lay.l_l + pp * 2 + 1], a = wp[l * lay.l_w + pp];
                        if constexpr (FUSED != 0) {
                            const int ri = l * lay.r_l + (FUSED == 2 ? 0 : pp * 2);
                            lx = rp[ri] + lx * iw;
                            ly = rp[ri + 1] + ly * ih;
                            a = __expf(a - smax_c);
                        }
                        const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
                        if (!(y > -1.f && x > -1.f && y < fH && x < fW)) continue;
                        const Footprint<float> f = footprint(y, x, Hq, Wq);
                        const float *r0 = vb + (lsi[l] + (int64_t)f.y0 * Wq + f.x0) * row, *r1 = r0 + (int64_t)Wq * row;
                        const float4 z = make_float4(0, 0, 0, 0);
                        const float4 c00 = (f.vy0 && f.vx0) ? *reinterpret_cast<const float4 *>(r0) : z;
                        const float4 c01 = (f.vy0 && f.vx1) ? *reinterpret_cast<const float4 *>(r0 + row) : z;
                        const float4 c10 = (f.vy1 && f.vx0) ? *reinterpret_cast<const float4 *>(r1) : z;
                        const float4 c11 = (f.vy1 && f.vx1) ? *reinterpret_cast<const float4 *>(r1 + row) : z;
                        const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a, w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
                        acc[c].x += w00 * c00.x + w01 * c01.x + w10 * c10.x + w11 * c11.x;
                        acc[c].y += w00 * c00.y + w01 * c01.y + w10 * c10.y + w11 * c11.y;
                        acc[c].z += w00 * c00.z + w01 * c01.z + w10 * c10.z + w11 * c11.z;
                        acc[c].w += w00 * c00.w + w01 * c01.w + w10 * c10.w + w11 * c11.w;
                    }
                }
            }
        }

        if (active) {
            float *ob = out + (bS + cell) * row + hs * SLICE + sub * 16 + j * 4;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const float inv = FUSED ? 1.f / qb_sel(ssum_s[c >> 2], c & 3) : 1.f;
                *reinterpret_cast<float4 *>(ob + lsi[c] * row) =
                    make_float4(acc[c].x * inv, acc[c].y * inv, acc[c].z * inv, acc[c].w * inv);
            }
        }

        __syncthreads();                                   // the next job's prologue overwrites the first window buffer
    }
}

template <int D, int NG, int FUSED>
static int launch_quad(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi, const float *off,
                       const float *logit, const float *ref, int64_t ref_bstride, SamplingLayout lay, int B, int S, int M,
                       float *out, const int *local_hits)
{
    auto kernel = &msda_fwd_quad<D, NG, FUSED>;
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_quad<D, NG, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, quad::LDS_BYTES);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return (cus + 7) / 8 * 8;                          // one workgroup per CU (2 x 64.5 KB of LDS each)
    }();
    msda_note_forward_kernel("msda_fwd_quad");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(quad::THREADS), quad::LDS_BYTES, st, value, shapes, lsi, off,
                       logit, ref, ref_bstride, lay, B, S, M, out, local_hits);
    return (int)hipGetLastError();
}

bool msda_quad_supported(int M, int D, int L)
{
    // opt-in (MVDETR_MSDA_QUAD=1): measured 3 % ahead of the group kernel on the public contract and 12 % behind it on the
    // fused one at Wildtrack size
    static const bool enabled = [] { const char *e = getenv("MVDETR_MSDA_QUAD"); return e && e[0] == '1'; }();
    return enabled && ((D == 16 && M % 2 == 0) || D == 32) && L >= 2 && L <= 8;
}

int msda_forward_quad(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi, const float *off,
                      const float *logit, const float *ref, int64_t ref_bstride, int fused, SamplingLayout lay, int B,
                      int S, int M, int D, int L, float *out, const int *local_hits)
{
#define QUAD_ARGS st, value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, out, local_hits
#define QUAD_CASE(DD, LL)                                                                                            \
    case DD * 100 + LL:                                                                                              \
        return fused == 2 ? launch_quad<DD, LL, 2>(QUAD_ARGS) : fused ? launch_quad<DD, LL, 1>(QUAD_ARGS)            \
                                                                      : launch_quad<DD, LL, 0>(QUAD_ARGS);
    switch (D * 100 + L) {
        QUAD_CASE(16, 2) QUAD_CASE(16, 3) QUAD_CASE(16, 4) QUAD_CASE(16, 5) QUAD_CASE(16, 6) QUAD_CASE(16, 7) QUAD_CASE(16, 8)
        QUAD_CASE(32, 2) QUAD_CASE(32, 3) QUAD_CASE(32, 4) QUAD_CASE(32, 5) QUAD_CASE(32, 6) QUAD_CASE(32, 7) QUAD_CASE(32, 8)
    default: break;
    }
#undef QUAD_CASE
#undef QUAD_ARGS
    return (int)hipErrorInvalidValue;
}

}  // namespace mvdetr
