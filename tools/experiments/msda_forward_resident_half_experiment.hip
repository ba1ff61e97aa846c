// Multi-scale deformable attention forward for the PUBLIC contract of the reference's extension (final sampling locations and
// attention weights in the layout [.., Lq, M, L, P(, 2)]), all source windows resident in LDS -- gfx950 (MI355X).
// D = 16, P = 4, L <= 7 equal-shaped levels: MVDeTr's own shapes (levels = cameras).
//
// Why another kernel for this entry.  The camera-grouped kernels (msda_group2_kernel.h) stage ONE source level's window at a
// time and need the sampling data of (camera, level) while that level's window is resident.  In the reference layout a (query,
// head)'s 224 + 112 bytes hold all L levels back to back, so a level iteration uses 32 + 16 bytes of three 128-byte lines and the
// next iteration -- 14 us later, after every other workgroup of the XCD has streamed its own share through the 4 MB L2 -- fetches
// the same lines again: the entry was bound by those loads, 175 - 180 us at Wildtrack size for four rounds whatever was done to
// its tap stream.  The fused entries changed the layout; the public contract cannot.
//
// Here a job is (4 x 8 cells, ONE head, ONE half of its 16 channels) and the windows of ALL L source levels of that half head
// are resident at once: 16 x 20 tokens x 32 B = 10 KB per level, 71.7 KB at L = 7 -- TWO 4-wave workgroups per CU, so that one
// job's copy phase (LDS-DMA of its windows, its sampling data) runs under the other's taps.  (Round 2's experiment with this
// structure took whole 16-channel heads: 143 KB, ONE 8-wave workgroup per CU whose copy phase nothing overlapped -- 201 us.)
// A lane is (camera, cell): every camera's query at a cell samples the same windows, and a (query, head)'s locations and
// weights of all levels are ONE contiguous 336-byte run, read once per half-head job (the two halves of a head are neighbours
// in the job list: the second finds the lines in L2).  Eight accumulators per lane.
//
// The taps of a lane -- L levels x 4 points -- are a software pipeline: the eight 16-byte LDS reads (4 corners x 2 chunks) of the
// next DEPTH - 1 taps are in flight while a tap's FMAs run (two waves per SIMD cannot hide an LDS round trip per tap).  Taps
// outside their window are noted in a mask and finished from global memory by a list walk behind the stream (zero padding by
// test), so any locations give the right result.  The windows follow the head's mean tap displacement (the job's own sample,
// msda_dispatch.h); a job whose sampled taps are far from their cells (uniformly random locations) gathers all its taps from
// memory in the gather kernel's lane roles instead of staging windows.  Levels of unequal shape: the gather formulation for the
// whole call, in the same launch.
//
// Replaces ms_deformable_im2col_cuda / ms_deformable_im2col_gpu_kernel for these shapes
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:923-954, 237-299).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_gather_body.h"
#include <stdlib.h>
#include <string.h>

#ifndef MVDETR_RF_DEPTH
#define MVDETR_RF_DEPTH 3
#endif

#ifdef MVDETR_RF_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup's waves (tools/experiments/rf_trace.py)
__device__ unsigned long long g_rf_trace[2048];
extern "C" int mvdetr_debug_rf_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rf_trace), n * sizeof(unsigned long long));
}
#define RTRACE(i) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (i) < 2048) g_rf_trace[(i)] = wall_clock64(); } while (0)
#else
#define RTRACE(i) do { } while (0)
#endif

namespace mvdetr {

namespace {

constexpr int RF_TH = 4, RF_TW = 8, RF_R = 6, RF_WH = RF_TH + 2 * RF_R, RF_WW = RF_TW + 2 * RF_R, RF_NTOK = RF_WH * RF_WW;
constexpr int RF_D = 16, RF_CH = 8, RF_MAXL = 7, RF_THREADS = 256;
static_assert(RF_NTOK % 32 == 0, "a DMA instruction covers 32 window positions x 2 chunks of 16 bytes");
static_assert(RF_TH == MSDA_SAMPLE_TH && MSDA_SAMPLE_TW % RF_TW == 0, "the job's sample is the 4 x 16 tile it lies in (msda_dispatch.h)");

typedef float rf2 __attribute__((ext_vector_type(2)));

// acc (two float2) += w * c
__device__ __forceinline__ void rfma4(rf2 &a0, rf2 &a1, float w, const float4 &c)
{
    const rf2 wv = {w, w};
    a0 = __builtin_elementwise_fma(wv, (rf2){c.x, c.y}, a0);
    a1 = __builtin_elementwise_fma(wv, (rf2){c.z, c.w}, a1);
}

}  // namespace

// opts bit 0: jobs whose sampled taps are far from their cells gather instead of staging windows (the `auto` dispatch)
template <int DEPTH>
__global__ __launch_bounds__(RF_THREADS, 2) void msda_fwd_resident(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M, int L, float *__restrict__ out, int opts)
{
    extern __shared__ __attribute__((aligned(16))) float vwin[];      // [L][RF_NTOK][8]
    constexpr int D = RF_D, CH = RF_CH, TH = RF_TH, TW = RF_TW, WH = RF_WH, WW = RF_WW, NTOK = RF_NTOK, P = TILE_P;
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (!equal) {
        msda_fwd_gather_body<float, 4>((int64_t)blockIdx.x * RF_THREADS + tid, (int64_t)gridDim.x * RF_THREADS, value, shapes,
                                       lsi, loc, aw, B, S, M, D, L, S, P, out);
        return;
    }

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * M * 2 * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;

    // lane = (camera, cell): a wave is two cameras' 32 cells
    const int cam_raw = tid / (TH * TW), cam = cam_raw < L ? cam_raw : L - 1;
    const int qi = tid % (TH * TW), qly = qi / TW, qlx = qi % TW;
    // LDS bank spreading: a token is 32 bytes and a window row 640 = 2.5 x 256 bytes, so the first chunks of the cells of an
    // even and of an odd tile row fall on the same banks: odd rows read their second chunk first (accumulator k of a lane
    // holds chunk k ^ rot for the whole kernel)
    const int rot = qly & 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int my_pos = lane >> 1, my_chunk = lane & 1;       // window copy: 32 window positions x 2 chunks of 16 bytes

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        // the two halves of a head, then the heads of a tile, run back to back on one XCD: same sampling lines, same token rows
        const int half = job & 1, head = (job >> 1) % M, u2 = job / (2 * M);
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qy < Hq && qx < Wq && cam_raw < L;
        const int64_t q = (int64_t)b * S + lsi[cam] + (active ? (int64_t)qy * Wq + qx : 0);
        const int64_t e0 = (q * M + head) * L * P;            // this (query, head)'s first tap
        const float *vbatch = value + (int64_t)b * S * row + head * D + half * CH;
        // where this head's taps lie: every wave reduces the same sample (the 4 x 16 cells around the job, camera 0, level 0)
        int shx, shy;
        bool far;
        {
            const int sX0 = X0 / MSDA_SAMPLE_TW * MSDA_SAMPLE_TW;
            const int s_qy = Y0 + lane / MSDA_SAMPLE_TW, s_qx = sX0 + lane % MSDA_SAMPLE_TW;
            const bool have = s_qy < Hq && s_qx < Wq;
            const float *lp = loc + ((((int64_t)b * S + lsi[0] + (have ? (int64_t)s_qy * Wq + s_qx : 0)) * M + head) * L) * P * 2;
            const float4 a0 = *reinterpret_cast<const float4 *>(lp), b0 = *reinterpret_cast<const float4 *>(lp + 4);
            msda_job_sample(a0, b0, have, s_qx, s_qy, fW, fH, shx, shy, far);
        }
        if (far && (opts & 1)) {
            // far-flung taps: windows would be wasted.  The job's outputs in the gather formulation: items = (camera, cell,
            // 16-byte chunk), GU per lane and iteration, branch-free, so that their loads are in flight together
            constexpr int NCK = CH / 4, GU = 2;
            const int nit = L * TH * TW * NCK;
            for (int it0 = tid; it0 < nit; it0 += GU * RF_THREADS) {
                const float *lp[GU], *wp[GU], *vb[GU];
                float *op[GU];
                bool live[GU];
                float4 r[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int it = it0 + u * RF_THREADS;
                    const int ck = it % NCK, ci = (it / NCK) % (TH * TW), c = min(it / (NCK * TH * TW), L - 1);
                    const int gy_ = Y0 + ci / TW, gx_ = X0 + ci % TW;
                    live[u] = it < nit && gy_ < Hq && gx_ < Wq;
                    const int64_t qq = (int64_t)b * S + lsi[c] + (live[u] ? (int64_t)gy_ * Wq + gx_ : 0);
                    lp[u] = loc + (qq * M + head) * L * P * 2;
                    wp[u] = aw + (qq * M + head) * L * P;
                    vb[u] = vbatch + ck * 4;
                    op[u] = out + qq * row + head * D + half * CH + ck * 4;
                    r[u] = make_float4(0, 0, 0, 0);
                }
                for (int l = 0; l < L; ++l) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            float x = lp[u][(l * P + p) * 2] * fW - 0.5f, y = lp[u][(l * P + p) * 2 + 1] * fH - 0.5f;
                            float a = wp[u][l * P + p];
                            const bool ok = y > -1.f && x > -1.f && y < fH && x < fW;      // (false for NaN)
                            x = ok ? x : 0.f;
                            y = ok ? y : 0.f;
                            a = ok ? a : 0.f;
                            const Footprint<float> f = footprint(y, x, Hq, Wq);
                            const float *r0 = vb[u] + lsi[l] * row + ((int64_t)f.y0 * Wq + f.x0) * row, *r1 = r0 + (int64_t)Wq * row;
                            const float4 c00 = load4_or_zero(r0, f.vy0 && f.vx0, vb[u]), c01 = load4_or_zero(r0 + row, f.vy0 && f.vx1, vb[u]);
                            const float4 c10 = load4_or_zero(r1, f.vy1 && f.vx0, vb[u]), c11 = load4_or_zero(r1 + row, f.vy1 && f.vx1, vb[u]);
                            const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a, w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
                            r[u].x += w00 * c00.x + w01 * c01.x + w10 * c10.x + w11 * c11.x;
                            r[u].y += w00 * c00.y + w01 * c01.y + w10 * c10.y + w11 * c11.y;
                            r[u].z += w00 * c00.z + w01 * c01.z + w10 * c10.z + w11 * c11.z;
                            r[u].w += w00 * c00.w + w01 * c01.w + w10 * c10.w + w11 * c11.w;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
                    if (live[u]) *reinterpret_cast<float4 *>(op[u]) = r[u];
            }
            continue;
        }
        const int oy = Y0 + TH / 2 - WH / 2 + shy, ox = X0 + TW / 2 - WW / 2 + shx;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        [[maybe_unused]] const int tr = ((t - (int)blockIdx.x) / (int)gridDim.x) * 64 + (tid >> 6) * 16;
        RTRACE(tr + 0);
        __syncthreads();                                      // everyone is done reading the previous job's windows
        RTRACE(tr + 1);
        {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(vbatch), 0, (int)((unsigned)S * (unsigned)row * 4u - (unsigned)(head * D + half * CH) * 4u), 0x00020000);
            // a wave takes window positions [32 k, 32 k + 32) for k = wave, wave + 4, ... of every level: one address
            // computation per k, one instruction per (k, level); positions outside the level store zeros
            for (int k = wave_u; k < NTOK / 32; k += RF_THREADS / 64) {
                const int wp = k * 32 + my_pos, wy = wp / WW, wx = wp % WW, gy = oy + wy, gx = ox + wx;
                const unsigned vo = ((unsigned)gx < (unsigned)Wq && (unsigned)gy < (unsigned)Hq)
                                        ? (unsigned)((gy * Wq + gx) * (int)row + my_chunk * 4) * 4u : 0x80000000u;
                for (int l = 0; l < L; ++l) {
                    const unsigned so = (unsigned)((int)lsi[l] * (int)row) * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vwin + (l * NTOK + k * 32) * CH),
                                                             16, (int)vo, (int)so, 0, 0);
                }
            }
        }
        RTRACE(tr + 2);
        // the (query, head)'s sampling data of all levels: one contiguous run each
        float4 la[RF_MAXL], lb[RF_MAXL], wa[RF_MAXL];
#pragma unroll
        for (int l = 0; l < RF_MAXL; ++l) {
            const int ll = l < L ? l : L - 1;
            la[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2);
            lb[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2 + 4);
            wa[l] = *reinterpret_cast<const float4 *>(aw + e0 + ll * P);
        }
        RTRACE(tr + 3);
        __syncthreads();                                      // the windows have landed
        RTRACE(tr + 4);

        rf2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        unsigned far_taps = 0u;                               // bit 4 l + p: in the image, outside the window (finished below)
        float4 cbuf[DEPTH][8];
        float tx_[DEPTH], ty_[DEPTH];
        bool tin_[DEPTH];
        auto issue = [&](int l, int p, int s_) {
            const float lx = p == 0 ? la[l].x : p == 1 ? la[l].z : p == 2 ? lb[l].x : lb[l].z;
            const float ly = p == 0 ? la[l].y : p == 1 ? la[l].w : p == 2 ? lb[l].y : lb[l].w;
            const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
            const bool in = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
            const int ix = in ? (int)floorf(x) - ox : 0, iy = in ? (int)floorf(y) - oy : 0;
            const float *p00 = vwin + l * NTOK * CH + (iy * WW + ix) * CH;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float *pk = p00 + ((k ^ rot) << 2);
                cbuf[s_][4 * k + 0] = *reinterpret_cast<const float4 *>(pk);
                cbuf[s_][4 * k + 1] = *reinterpret_cast<const float4 *>(pk + CH);
                cbuf[s_][4 * k + 2] = *reinterpret_cast<const float4 *>(pk + WW * CH);
                cbuf[s_][4 * k + 3] = *reinterpret_cast<const float4 *>(pk + WW * CH + CH);
            }
            tx_[s_] = x;
            ty_[s_] = y;
            tin_[s_] = in;
        };
        auto finish = [&](int l, int p, int s_) {
            const float x = tx_[s_], y = ty_[s_];
            const bool in = tin_[s_];
            const float a = in ? (p == 0 ? wa[l].x : p == 1 ? wa[l].y : p == 2 ? wa[l].z : wa[l].w) : 0.f;
            const float wx1 = in ? x - floorf(x) : 0.f, wy1 = in ? y - floorf(y) : 0.f;
            const float ay1 = wy1 * a, ay0 = a - ay1;
            const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rfma4(acc[2 * k], acc[2 * k + 1], w00, cbuf[s_][4 * k + 0]);
                rfma4(acc[2 * k], acc[2 * k + 1], w01, cbuf[s_][4 * k + 1]);
                rfma4(acc[2 * k], acc[2 * k + 1], w10, cbuf[s_][4 * k + 2]);
                rfma4(acc[2 * k], acc[2 * k + 1], w11, cbuf[s_][4 * k + 3]);
            }
            // (lanes without a cell carry cell 0's taps; a NaN position is in neither set: no contribution, as the reference)
            if (!in && active && y > -1.f && x > -1.f && y < fH && x < fW) far_taps |= 1u << (l * P + p);
        };
#pragma unroll
        for (int t_ = 0; t_ < DEPTH - 1; ++t_)
            if (t_ / P < L) issue(t_ / P, t_ % P, t_ % DEPTH);
#pragma unroll
        for (int t_ = 0; t_ < RF_MAXL * P; ++t_) {
            const int l = t_ / P, p = t_ % P, tn = t_ + DEPTH - 1;
            if (l >= L) continue;                             // (uniform; `break` would keep the loop from unrolling)
            __builtin_amdgcn_sched_barrier(0);
            if (tn < RF_MAXL * P && tn / P < L) issue(tn / P, tn % P, tn % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            finish(l, p, t_ % DEPTH);
        }
        RTRACE(tr + 5);
        // ---- taps outside their window: one list per lane, walked with the NEXT entry's sampling data requested before the
        //      current entry's eight corner gathers (a round trip per far tap of the wave's worst lane)
        if (far_taps) {
            float2 nxy = make_float2(0.f, 0.f);
            float na_ = 0.f;
            int64_t nls = 0;
            auto request = [&]() {
                const int nt = __ffs((int)far_taps) - 1;
                far_taps &= far_taps - 1u;
                nxy = *reinterpret_cast<const float2 *>(loc + (e0 + nt) * 2);
                na_ = aw[e0 + nt];
                nls = lsi[nt >> 2];
            };
            request();
            for (;;) {
                const float2 xy = nxy;
                const float a = na_;
                const float *vlevel = vbatch + nls * row;
                const bool more = far_taps != 0u;
                if (more) request();
                const float x = xy.x * fW - 0.5f, y = xy.y * fH - 0.5f;       // (the tap stream's expressions)
                const Footprint<float> f = footprint(y, x, Hq, Wq);
                const float *r0 = vlevel + ((int64_t)f.y0 * Wq + f.x0) * row, *r1 = r0 + (int64_t)Wq * row;
                const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a, w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int ko = (k ^ rot) << 2;
                    const float4 c00 = load4_or_zero(r0 + ko, f.vy0 && f.vx0, vbatch), c01 = load4_or_zero(r0 + row + ko, f.vy0 && f.vx1, vbatch);
                    const float4 c10 = load4_or_zero(r1 + ko, f.vy1 && f.vx0, vbatch), c11 = load4_or_zero(r1 + row + ko, f.vy1 && f.vx1, vbatch);
                    rfma4(acc[2 * k], acc[2 * k + 1], w00, c00);
                    rfma4(acc[2 * k], acc[2 * k + 1], w01, c01);
                    rfma4(acc[2 * k], acc[2 * k + 1], w10, c10);
                    rfma4(acc[2 * k], acc[2 * k + 1], w11, c11);
                }
                if (!more) break;
            }
        }
        RTRACE(tr + 6);
        if (active) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                *reinterpret_cast<float4 *>(out + q * row + head * D + half * CH + ((k ^ rot) << 2)) =
                    make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y);
        }
    }
}

bool msda_forward_resident_supported(int B, int S, int M, int D, int L)
{
    static const bool on = [] { const char *e = getenv("MVDETR_MSDA_RESIDENT"); return !(e && e[0] == '0'); }();
    // 16-channel heads, at most seven equal levels (unequal ones fall to the gather formulation inside the launch, where the tile
    // kernel does better: callers with fewer than six levels keep the tile kernel); 32-bit buffer offsets in one batch element
    return on && D == RF_D && (L == 6 || L == 7) && M >= 1 && (int64_t)S * M * RF_D * 4 < 0x7fffffffLL;
}

int msda_forward_resident(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *aw, int B, int S, int M, int L, float *out, bool standdown)
{
    constexpr int DEPTH = MVDETR_RF_DEPTH;
    const int lds = L * RF_NTOK * RF_CH * 4;
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_resident<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  RF_MAXL * RF_NTOK * RF_CH * 4);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_fwd_resident<DEPTH>, RF_THREADS, RF_MAXL * RF_NTOK * RF_CH * 4) != hipSuccess || per_cu < 1)
            per_cu = 2;
        return (cus * per_cu + 7) / 8 * 8;
    }();
    static const KernelResources res = kernel_resources(reinterpret_cast<const void *>(&msda_fwd_resident<DEPTH>));
    msda_note_forward_kernel("msda_fwd_resident[all windows of a half head in LDS, pipelined taps]", &res);
    hipLaunchKernelGGL(msda_fwd_resident<DEPTH>, dim3((unsigned)blocks), dim3(RF_THREADS), lds, st, value, shapes, lsi, loc, aw, B, S, M,
                       L, out, standdown ? 1 : 0);
    return (int)hipGetLastError();
}

}  // namespace mvdetr
