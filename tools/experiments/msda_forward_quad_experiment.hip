// EXPERIMENT, NOT BUILT (round 2): a camera-grouped forward with the work of one (cell, head) spread over a quad of
// lanes.  Correct (1e-6 against the gather kernel) and free of LDS bank conflicts by construction, but not faster than
// msda_forward_group.hip: 180-190 us against 151 us at Wildtrack size.  What the measurements around it showed is the
// useful part: the forward kernels are bound by the rate at which a CU's L1 can miss --
// about one 128-byte line per 13 cycles -- not by VALU issue, LDS bandwidth or occupancy.  This version issues its
// loads from inline asm with hand-counted s_waitcnt; tools/experiments/check_quad_asm.py shows why that is unsafe
// (hipcc copies an asm load's destination register before the data has landed).
// Multi-scale deformable attention forward, camera-grouped "quad" kernel -- gfx950 (MI355X).
//
// Same job as msda_forward_group.hip -- one workgroup owns a (6 x 16 cell tile, 128-byte slice) and walks all
// NG = L query levels (cameras) per staged source window -- but with the work of one (cell, head) spread over a
// QUAD of lanes, 4 channels (one 16-byte chunk) each.  What that buys:
//
//   * registers: a lane carries NG x 4 accumulators instead of NG x 16 (28 instead of 112 at 7 cameras), so the
//     kernel runs 12 waves per CU with room for the next window's staging registers (the group kernel: 254 VGPRs,
//     6 compute waves per CU, 62 % of its wave-cycles parked in s_waitcnt);
//   * the per-tap arithmetic is not duplicated: lane j of the quad owns sampling point j of the (camera, level)
//     -- P == 4 -- computes that tap's LDS address and its four corner weights ONCE, and the quad's lanes pick
//     them up with DPP quad broadcasts (v_mov_b32_dpp quad_perm:[p,p,p,p]), one tap after the other;
//   * LDS bank conflicts are gone BY CONSTRUCTION, for any sampling locations: a ds_read_b128 is served in four
//     16-lane groups, i.e. four quads per group, each quad reading the 64 contiguous bytes of one (token, head).
//     With 128-byte tokens the slot of those 64 bytes in the 256-byte bank row is (token parity, head); the two
//     quads of a group that share a head are given opposite "roles" r, and every quad reads its two x-neighbour
//     corners in the order (token of parity r, token of parity 1-r) -- the two corners of a bilinear footprint
//     always differ in parity -- swapping the two x weights to match.  Every instruction then covers all four
//     slots exactly once;
//   * one workgroup per CU with the window DOUBLE-BUFFERED (2 x 64.5 KB): level l+1's window is loaded into
//     registers before level l's taps and written to the other buffer after them; one barrier per level.
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
// tuning aid (never in the shipped build): wall-clock stamps (100 MHz) of workgroup 0 / wave 0 at the phase boundaries
__device__ unsigned long long g_quad_trace[1024];
extern "C" int mvdetr_debug_quad_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_quad_trace), n * sizeof(unsigned long long));
}
#define QTRACE(i) do { if (blockIdx.x == 8 && threadIdx.x == 0 && (i) < 1024) g_quad_trace[(i)] = wall_clock64(); } while (0)
#else
#define QTRACE(i) do { } while (0)
#endif

namespace mvdetr {

typedef float qfloat2 __attribute__((ext_vector_type(2)));

namespace quad {

constexpr int TH = 6, TW = 16, R = 6;
constexpr int WH = TH + 2 * R, WW = TW + 2 * R;          // 18 x 28 tokens
constexpr int SLICE = 32;                                 // floats of a token row per workgroup (128 B)
constexpr int TOKB = SLICE * 4;                           // bytes per token in LDS
constexpr int WIN_FLOATS = WH * WW * SLICE;               // 16,128 floats = 64,512 B (a multiple of 256 B)
constexpr int THREADS = TH * TW * 8;                      // 8 lanes per cell: 2 sub-slices x 4 chunks = 768
constexpr int COPY_ITEMS = WH * WW * (SLICE / 4);         // float4 per window = 4,032
constexpr int NSTAGE = (COPY_ITEMS + THREADS - 1) / THREADS;   // 6
#ifndef MVDETR_QUAD_TAP
#define MVDETR_QUAD_TAP 1
#endif
constexpr int LDS_BYTES = 2 * WIN_FLOATS * 4;             // double-buffered window
static_assert(WW % 2 == 0, "token parity == column parity needs an even window width");
static_assert((WIN_FLOATS * 4) % 256 == 0, "both buffers start on a bank row");

// Raw buffer loads: the address is (SGPR descriptor) + (SGPR byte offset) + (one VGPR byte offset), so the per-camera /
// per-level part of every sampling-data and window address lives in scalar registers and a lane keeps ONE offset
// register per tensor (flat loads made hipcc build a 64-bit VGPR address per camera: 40+ registers).  Reads beyond
// `bytes` return 0, which is also how window chunks outside the level get their zeros.
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

// The loads of the main loop are issued from inline asm and waited for with hand-counted s_waitcnt vmcnt(N).  hipcc's
// own placement does not survive the loop's back edge: for data requested one level ahead it emitted vmcnt(0..9)
// where 24 loads may stay in flight, which parked every wave behind the loads it had just issued (measured: a
// (camera, level) step took 1.4 us with, 0.4 us without sampling loads in flight).  Rules that make this safe:
//   * every VMEM instruction of the steady-state loop is one of these (no compiler loads, no spills inside it), so
//     the number of younger loads at each wait is known; extra VMEM work elsewhere (the job epilogue's loads and
//     stores) is younger than anything waited for and only makes a wait stricter;
//   * a destination register is not touched between its load and the wait statement that names it ("+v")
//     (tools/check_quad_asm.py audits the generated code for that).
typedef int desc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ desc4 make_desc(const void *base, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    desc4 d = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    return d;
}
__device__ __forceinline__ float asm_load1(desc4 d, unsigned voff, unsigned soff)
{
    float r;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(d), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ float2 asm_load2(desc4 d, unsigned voff, unsigned soff)
{
    float2 r;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(d), "s"(soff) : "memory");
    return r;
}
// 16 bytes per lane straight into LDS (no VGPRs): lane i's chunk lands at lds_addr + 16 * i; out-of-range reads store 0
__device__ __forceinline__ void asm_dma16(desc4 d, unsigned voff, unsigned soff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(d), "s"(soff), "s"(lds_addr) : "memory");
}

template <int P> __device__ __forceinline__ int qb_i(int v)
{
    constexpr int ctrl = P | (P << 2) | (P << 4) | (P << 6);       // quad_perm:[P,P,P,P]
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
}
template <int P> __device__ __forceinline__ float qb_f(float v)
{
    return __builtin_bit_cast(float, qb_i<P>(__builtin_bit_cast(int, v)));
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

// One tap of the quad.  The owner lane P of each quad holds the tap's descriptor -- two LDS byte addresses and the
// four corner weights (attention weight folded in); the other lanes read them through DPP quad broadcasts: the
// addresses with v_add_u32_dpp, the weights INSIDE the multiply-add (v_fmac_f32_dpp: src0 = lane P's weight), so a
// tap costs 16 VALU instructions for its 16 FMAs and no broadcast registers.  Inline asm because hipcc does not fold
// a DPP move into v_fmac (it emits v_mov_b32_dpp + v_pk_fma_f32 with a wasted high half per weight); `s_nop 1` =
// the two wait states a DPP read needs after a VALU write of the same register, which hipcc cannot see inside asm.
#define MVDETR_FMAC_DPP(ACC, W, C) "v_fmac_f32_dpp " ACC ", " W ", " C " quad_perm:[%24,%24,%24,%24] row_mask:0xf bank_mask:0xf\n\t"
template <int P>
__device__ __forceinline__ void tap(const char *lane_base, int addrA, int addrB, float wAt, float wAb, float wBt,
                                    float wBb, float4 &acc)
{
    const char *pa = lane_base + qb_i<P>(addrA);
    const char *pb = lane_base + qb_i<P>(addrB);
    const float4 cAt = *reinterpret_cast<const float4 *>(pa);
    const float4 cAb = *reinterpret_cast<const float4 *>(pa + WW * TOKB);
    const float4 cBt = *reinterpret_cast<const float4 *>(pb);
    const float4 cBb = *reinterpret_cast<const float4 *>(pb + WW * TOKB);
#if MVDETR_QUAD_TAP == 1
    // measured (tools/experiments/valu_lds_rate.hip): v_fmac_f32_dpp issues at half the rate of v_fmac_f32 / v_pk_fma_f32,
    // so broadcast each weight once (v_mov_b32_dpp) and use plain FMAs
    const float a_t = qb_f<P>(wAt), a_b = qb_f<P>(wAb), b_t = qb_f<P>(wBt), b_b = qb_f<P>(wBb);
    acc.x = fmaf(a_t, cAt.x, acc.x); acc.y = fmaf(a_t, cAt.y, acc.y); acc.z = fmaf(a_t, cAt.z, acc.z); acc.w = fmaf(a_t, cAt.w, acc.w);
    acc.x = fmaf(a_b, cAb.x, acc.x); acc.y = fmaf(a_b, cAb.y, acc.y); acc.z = fmaf(a_b, cAb.z, acc.z); acc.w = fmaf(a_b, cAb.w, acc.w);
    acc.x = fmaf(b_t, cBt.x, acc.x); acc.y = fmaf(b_t, cBt.y, acc.y); acc.z = fmaf(b_t, cBt.z, acc.z); acc.w = fmaf(b_t, cBt.w, acc.w);
    acc.x = fmaf(b_b, cBb.x, acc.x); acc.y = fmaf(b_b, cBb.y, acc.y); acc.z = fmaf(b_b, cBb.z, acc.z); acc.w = fmaf(b_b, cBb.w, acc.w);
    return;
#endif
    asm("s_nop 1\n\t"
        MVDETR_FMAC_DPP("%0", "%4", "%8") MVDETR_FMAC_DPP("%1", "%4", "%9") MVDETR_FMAC_DPP("%2", "%4", "%10") MVDETR_FMAC_DPP("%3", "%4", "%11")
        MVDETR_FMAC_DPP("%0", "%5", "%12") MVDETR_FMAC_DPP("%1", "%5", "%13") MVDETR_FMAC_DPP("%2", "%5", "%14") MVDETR_FMAC_DPP("%3", "%5", "%15")
        MVDETR_FMAC_DPP("%0", "%6", "%16") MVDETR_FMAC_DPP("%1", "%6", "%17") MVDETR_FMAC_DPP("%2", "%6", "%18") MVDETR_FMAC_DPP("%3", "%6", "%19")
        MVDETR_FMAC_DPP("%0", "%7", "%20") MVDETR_FMAC_DPP("%1", "%7", "%21") MVDETR_FMAC_DPP("%2", "%7", "%22") MVDETR_FMAC_DPP("%3", "%7", "%23")
        : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w)
        : "v"(wAt), "v"(wAb), "v"(wBt), "v"(wBb),
          "v"(cAt.x), "v"(cAt.y), "v"(cAt.z), "v"(cAt.w), "v"(cAb.x), "v"(cAb.y), "v"(cAb.z), "v"(cAb.w),
          "v"(cBt.x), "v"(cBt.y), "v"(cBt.z), "v"(cBt.w), "v"(cBb.x), "v"(cBb.y), "v"(cBb.z), "v"(cBb.w), "i"(P));
}
#undef MVDETR_FMAC_DPP

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
        const float *lp = off + bq * lay.q_l + m * lay.h_l;
        const float *wp = logit + bq * lay.q_w + m * lay.h_w;
        const float *rp = FUSED ? ref + b * ref_bstride + q * L * (FUSED == 2 ? 2 : P * 2) : nullptr;
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
                        const int ri = FUSED == 2 ? l * 2 : (l * P + p) * 2;
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
// 1 = raw offsets / logits + reference points [.., Lq, L, P, 2]; 2 = raw + one point per (query, level) [.., Lq, L, 2].
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
    static_assert(NG <= 16, "16-bit level masks");
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

    // ---- what a lane needs to know about a job ----------------------------------------------------------------------
    struct JobCtx {
        int hs, b, oy, ox, cell;
        bool active;
        unsigned lane_l, lane_w, lane_r;                   // byte offsets of this lane's sampling point inside a query row
    };
    constexpr int RPL = FUSED == 2 ? 2 : P * 2;
    auto make_ctx = [&](int job) {
        JobCtx k;
        const int u2 = job / HS, tin = u2 % per_level;
        k.hs = job % HS;
        k.b = u2 / per_level;
        k.oy = (tin / tcols) * TH - R;
        k.ox = (tin % tcols) * TW - R;
        const int qy = k.oy + R + qly, qx = k.ox + R + qlx;
        k.active = qy < Hq && qx < Wq;
        k.cell = k.active ? qy * Wq + qx : 0;
        const int head = (k.hs * SLICE + sub * 16) / D;
        k.lane_l = (unsigned)(k.cell * lay.q_l + head * lay.h_l + j * 2) * 4u;
        k.lane_w = (unsigned)(k.cell * lay.q_w + head * lay.h_w + j) * 4u;
        k.lane_r = (unsigned)(k.cell * L * RPL + (FUSED == 2 ? 0 : j * 2)) * 4u;
        return k;
    };

    // ---- window staging: LDS-DMA, NSTAGE 16-byte chunks per lane and level, no registers ----------------------------
    // chunk i of the window (i = tid + k * THREADS) goes to byte 16 * i of the buffer: a wave's 64 chunks are contiguous
    unsigned stage_off[NSTAGE];                            // byte offset of this lane's k-th window chunk from the level base
    const unsigned lds_win = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)win;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    // where the window of a job's tile sits (per job; the level only moves the scalar offset)
    auto plan = [&](const JobCtx &k) {
#pragma unroll
        for (int i0 = 0; i0 < NSTAGE; ++i0) {
            // the last pass covers only 192 chunks (3 waves); the other waves repeat their previous chunk so that
            // every wave issues the same number of loads per level
            const int kk = (tid + i0 * THREADS < COPY_ITEMS) ? i0 : i0 - 1;
            const int i = tid + kk * THREADS;
            const int tok = i >> 3, ch = i & 7;
            const int wy = tok / WW, wx = tok - wy * WW;
            const int gy = k.oy + wy, gx = k.ox + wx;
            const bool ok = (unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq;
            stage_off[i0] = ok ? (unsigned)((gy * Wq + gx) * row + ch * 4) * 4u : OOB;     // outside the level: zeros
        }
    };
    auto issue = [&](const JobCtx &k, int l, int buf) {
        const desc4 dv = make_desc(value + (int64_t)k.b * S * row, (unsigned)S * row * 4u);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((lvl0[l] * row + k.hs * SLICE) * 4);
#pragma unroll
        for (int i0 = 0; i0 < NSTAGE; ++i0) {
            const unsigned kk = (wave_u * 64 + i0 * THREADS < COPY_ITEMS) ? i0 : i0 - 1;       // (wave-uniform)
            asm_dma16(dv, stage_off[i0], so, lds_win + (unsigned)buf * (WIN_FLOATS * 4) + (kk * THREADS + wave_u * 64) * 16u);
        }
    };

    // ---- sampling data of this lane's point, one register set per camera, refilled ONE LEVEL ahead -----------------
    // (A (camera, level) step issues in a few hundred cycles but a load takes a few thousand under load.)
    constexpr int LP = FUSED ? 3 : 2;                      // loads per (camera, level) step
    constexpr int AHEAD = (NG - 1) * LP + NSTAGE;          // loads younger than a step's data when it is consumed
    static_assert(AHEAD < 64 && NG * LP < 64, "vmcnt is a 6-bit counter");
    float2 n_o[NG], n_r[NG];
    float n_w[NG];
    auto load_cam = [&](int c, const JobCtx &k, int l) {
        const int64_t bS = (int64_t)k.b * S;
        const desc4 d_off = make_desc(off + bS * lay.q_l, (unsigned)S * lay.q_l * 4u);
        const desc4 d_log = make_desc(logit + bS * lay.q_w, (unsigned)S * lay.q_w * 4u);
        n_o[c] = asm_load2(d_off, k.lane_l, (unsigned)__builtin_amdgcn_readfirstlane((lvl0[c] * lay.q_l + l * lay.l_l) * 4));
        n_w[c] = asm_load1(d_log, k.lane_w, (unsigned)__builtin_amdgcn_readfirstlane((lvl0[c] * lay.q_w + l * lay.l_w) * 4));
        n_r[c] = make_float2(0, 0);
        if constexpr (FUSED != 0) {
            const desc4 d_ref = make_desc(ref + k.b * ref_bstride, (unsigned)S * L * RPL * 4u);
            n_r[c] = asm_load2(d_ref, k.lane_r, (unsigned)__builtin_amdgcn_readfirstlane((lvl0[c] * L * RPL + l * RPL) * 4));
        }
    };

    int t = blockIdx.x;
    int job = job_of(t);
    if (job >= jobs) return;                               // (uniform) nothing for this workgroup
    JobCtx cur = make_ctx(job);
    plan(cur);
    // prologue: the same load order as one level of the steady state (window, then camera by camera)
    issue(cur, 0, 0);
#pragma unroll
    for (int c = 0; c < NG; ++c) load_cam(c, cur, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * LP) : "memory");           // the window has landed
    __syncthreads();
    int pb = 0;

    for (;;) {
        const int hs = cur.hs, oy = cur.oy, ox = cur.ox, cell = cur.cell;
        const bool active = cur.active;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        const int64_t bS = (int64_t)cur.b * S;
        const unsigned lane_l = cur.lane_l, lane_w = cur.lane_w, lane_r = cur.lane_r;
        const float *refb = FUSED ? ref + cur.b * ref_bstride : nullptr;
        const int next_job = job_of(t + (int)gridDim.x);
        const JobCtx nxt = make_ctx(next_job < jobs ? next_job : job);     // (the last job prefetches itself: harmless)

        float4 acc[NG];
        float smax[NG], ssum[NG];
        unsigned miss[(NG + 1) / 2];
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            smax[c] = -INFINITY;
            ssum[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < (NG + 1) / 2; ++c) miss[c] = 0;

        for (int l = 0; l < L; ++l) {
            const int tr = ((t / (int)gridDim.x) * L + l) * 8;
            QTRACE(tr + 0);
            // the step after this one: next level of this job, or level 0 of the next job
            const bool last = l + 1 == L;
            JobCtx pf = cur;
            if (last) {
                pf = nxt;
                plan(nxt);
            }
            const int pl = last ? 0 : l + 1;
            issue(pf, pl, pb ^ 1);
            const char *lane_base = reinterpret_cast<const char *>(win + pb * WIN_FLOATS) + lane_byte;
            QTRACE(tr + 1);
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                if (c == 1) QTRACE(tr + 2);
                if (c == 4) QTRACE(tr + 3);
                // this step's sampling data was requested one level ago; AHEAD younger loads may stay in flight
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(n_o[c]), "+v"(n_w[c]), "+v"(n_r[c]) : "n"(AHEAD) : "memory");
                const float2 o = n_o[c], r = n_r[c];
                const float lg = n_w[c];
                load_cam(c, pf, pl);
                float x, y, a;
                if constexpr (FUSED != 0) {
                    const float m = fmaxf(smax[c], quad_max(lg));
                    const float sc = __expf(smax[c] - m);
                    a = __expf(lg - m);
                    ssum[c] = ssum[c] * sc + quad_sum(a);
                    smax[c] = m;
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
                const bool in = active & (fabsf(x - cx) < 0.5f * (WW - 1)) & (fabsf(y - cy) < 0.5f * (WH - 1));
                const float fx = floorf(x), fy = floorf(y);
                const int ix = in ? (int)fx - ox : 0, iy = in ? (int)fy - oy : 0;
                const float wx1 = in ? x - fx : 0.f, wy1 = in ? y - fy : 0.f;   // (NaN locations must not leak into the weights)
                a = in ? a : 0.f;
                miss[c >> 1] |= (active & !in) ? 1u << (l + 16 * (c & 1)) : 0u;
                const int s = (ix ^ role) & 1;             // 1: the right-hand corner has this quad's parity
                const float wxA = s ? wx1 : 1.f - wx1;
                const float ay1 = wy1 * a, ay0 = a - ay1;
                const float wAt = ay0 * wxA, wBt = ay0 - wAt, wAb = ay1 * wxA, wBb = ay1 - wAb;
                const int addrA = (iy * WW + ix + s) * TOKB;
                const int addrB = addrA + (s ? -TOKB : TOKB);
                tap<0>(lane_base, addrA, addrB, wAt, wAb, wBt, wBb, acc[c]);
                tap<1>(lane_base, addrA, addrB, wAt, wAb, wBt, wBb, acc[c]);
                __builtin_amdgcn_sched_barrier(0);         // two taps (8 ds_read_b128) in flight at a time
                tap<2>(lane_base, addrA, addrB, wAt, wAb, wBt, wBb, acc[c]);
                tap<3>(lane_base, addrA, addrB, wAt, wAb, wBt, wBb, acc[c]);
                asm volatile("" : "+v"(acc[c].x), "+v"(acc[c].y), "+v"(acc[c].z), "+v"(acc[c].w));   // this camera's FMAs stay here
                __builtin_amdgcn_sched_barrier(0);
            }
            QTRACE(tr + 4);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * LP) : "memory");       // the next window has landed
            QTRACE(tr + 5);
            __syncthreads();
            QTRACE(tr + 6);
            pb ^= 1;
        }

        // ---- taps that left their window: straight from global memory (rare) ----------------------------------
        bool any_miss = false;
#pragma unroll
        for (int c = 0; c < (NG + 1) / 2; ++c) any_miss = any_miss || miss[c] != 0;
        if (__any(any_miss)) {
            const float *vb = value + bS * row + hs * SLICE + sub * 16 + j * 4;
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const int mine = (int)((miss[c >> 1] >> (16 * (c & 1))) & 0xffffu);
                const int64_t cq = bS + lsi[c];
                const float *lp = off + cq * lay.q_l + (lane_l / 4 - j * 2);
                const float *wp = logit + cq * lay.q_w + (lane_w / 4 - j);
                const float *rp = FUSED ? refb + lsi[c] * L * RPL + (lane_r / 4 - (FUSED == 2 ? 0 : j * 2)) : nullptr;
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    int mm = pp == 0 ? qb_i<0>(mine) : pp == 1 ? qb_i<1>(mine) : pp == 2 ? qb_i<2>(mine) : qb_i<3>(mine);
                    while (mm) {
                        const int l = __ffs(mm) - 1;
                        mm &= mm - 1;
                        float lx = lp[l * lay.l_l + pp * 2], ly = lp[l * lay.l_l + pp * 2 + 1], a = wp[l * lay.l_w + pp];
                        if constexpr (FUSED != 0) {
                            const int ri = FUSED == 2 ? l * 2 : (l * P + pp) * 2;
                            lx = rp[ri] + lx * iw;
                            ly = rp[ri + 1] + ly * ih;
                            a = __expf(a - smax[c]);
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
                const float inv = FUSED ? 1.f / ssum[c] : 1.f;
                *reinterpret_cast<float4 *>(ob + lsi[c] * row) =
                    make_float4(acc[c].x * inv, acc[c].y * inv, acc[c].z * inv, acc[c].w * inv);
            }
        }

        if (next_job >= jobs) break;
        job = next_job;
        cur = nxt;
        t += (int)gridDim.x;
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
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(quad::THREADS), quad::LDS_BYTES, st, value, shapes, lsi, off,
                       logit, ref, ref_bstride, lay, B, S, M, out, local_hits);
    return (int)hipGetLastError();
}

bool msda_quad_supported(int M, int D, int L)
{
    static const bool enabled = [] { const char *e = getenv("MVDETR_MSDA_QUAD"); return !(e && e[0] == '0'); }();
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
