// Multi-scale deformable attention forward for the PUBLIC (unfused) contract, all source windows resident in LDS --
// gfx950 (MI355X).  D = 16, P = 4, L <= 7 equal-shaped levels (MVDeTr's own shapes: levels = cameras).
//
// The camera-grouped kernel (msda_forward_group.hip) walks the source levels one staged window at a time and needs the
// sampling data of (camera, level) when that level's window is resident: in the reference layout
// [.., Lq, M, L, P(, 2)] that is a 32-byte + a 16-byte piece per (query, head, level) -- it is bound by those loads (172 us
// at Wildtrack size whatever the tap count).  Here a job is (4 x 8 cells, ONE head) and ALL L source
// windows of that head (16 x 20 tokens x 64 B = 20 KB each, 143 KB at L = 7) are staged by LDS-DMA before anything is
// read -- the structure of msda_bwd_sampling_resident (msda_backward_sampling.hip): one barrier per job, and a lane
// is (camera, cell, half of the head's 16 channels), so a (query, head)'s sampling locations and weights of all levels
// are ONE contiguous 336-byte run.  Eight accumulators per lane; one 8-wave workgroup per CU.
//
// Taps outside the window read global memory (zero padding by test), so any locations give the right result; a call
// whose levels turn out unequal, or whose taps the locality probe found far from their queries, runs the gather
// formulation inside the same launch.
//
// Replaces ms_deformable_im2col_gpu_kernel for these shapes (multiview_detector/models/ops/src/cuda/
// ms_deform_im2col_cuda.cuh:237-299).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_gather_body.h"
#include <stdlib.h>
#include <string.h>

namespace mvdetr {

void msda_note_forward_kernel(const char *name);

constexpr int RF_TH = 4, RF_TW = 8, RF_R = 6, RF_WH = RF_TH + 2 * RF_R, RF_WW = RF_TW + 2 * RF_R, RF_NTOK = RF_WH * RF_WW;
constexpr int RF_D = 16, RF_MAXL = 7, RF_THREADS = 512;
static_assert(RF_NTOK % 16 == 0, "a DMA instruction covers 16 window positions");

typedef float rf2 __attribute__((ext_vector_type(2)));

// acc (two float2) += w * c
__device__ __forceinline__ void rfma4(rf2 &a0, rf2 &a1, float w, const float4 &c)
{
    const rf2 wv = {w, w};
    a0 = __builtin_elementwise_fma(wv, (rf2){c.x, c.y}, a0);
    a1 = __builtin_elementwise_fma(wv, (rf2){c.z, c.w}, a1);
}

__global__ __launch_bounds__(RF_THREADS, 2) void msda_fwd_resident(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M, int L, float *__restrict__ out,
    const int *__restrict__ local_hits)
{
    extern __shared__ __attribute__((aligned(16))) float vwin[];      // [L][RF_NTOK][16]
    constexpr int D = RF_D, TH = RF_TH, TW = RF_TW, WH = RF_WH, WW = RF_WW, NTOK = RF_NTOK, P = TILE_P, NV = 2;
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (local_hits && *local_hits * 2 < MSDA_PROBE_SAMPLES) equal = false;      // far-flung taps: windows would be wasted
    if (!equal) {
        msda_fwd_gather_body<float, 4>((int64_t)blockIdx.x * RF_THREADS + tid, (int64_t)gridDim.x * RF_THREADS, value, shapes,
                                       lsi, loc, aw, B, S, M, D, L, S, P, out);
        return;
    }

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * M * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;

    // lane = (camera, cell, half of the head's 16 channels): a wave is one camera's 32 cells; two waves per SIMD
    const int sub = tid & 1, cam_raw = (tid >> 1) / (TH * TW), cam = cam_raw < L ? cam_raw : L - 1;
    const int qi = (tid >> 1) % (TH * TW), qly = qi / TW, qlx = qi % TW;
    // LDS bank spreading: lane reads chunks (2 * sub + k) ^ rot (see msda_bwd_sampling_resident)
    const int rot = (qlx >> 2) & 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int my_pos = lane >> 2, my_chunk = lane & 3;       // window copy: 16 window positions x 4 chunks of 16 bytes

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        const int head = job % M, u2 = job / M;               // the heads of a tile run back to back: same token rows
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qy < Hq && qx < Wq && cam_raw < L;
        const int64_t q = (int64_t)b * S + lsi[cam] + (active ? (int64_t)qy * Wq + qx : 0);
        const int64_t e0 = (q * M + head) * L * P;            // this (query, head)'s first tap
        const float *vbatch = value + (int64_t)b * S * row + head * D;
        const int oy = Y0 + TH / 2 - WH / 2, ox = X0 + TW / 2 - WW / 2;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        __syncthreads();                                      // everyone is done reading the previous job's windows
        {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(vbatch), 0, (int)((unsigned)S * (unsigned)row * 4u - (unsigned)(head * D) * 4u), 0x00020000);
            // a wave takes window positions [16 k, 16 k + 16) for k = wave, wave + 8, ... of every level: one address
            // computation per k, one instruction per (k, level); positions outside the level store zeros
            for (int k = wave_u; k < NTOK / 16; k += RF_THREADS / 64) {
                const int wp = k * 16 + my_pos, wy = wp / WW, wx = wp % WW, gy = oy + wy, gx = ox + wx;
                const unsigned vo = ((unsigned)gx < (unsigned)Wq && (unsigned)gy < (unsigned)Hq)
                                        ? (unsigned)((gy * Wq + gx) * (int)row + my_chunk * 4) * 4u : 0x80000000u;
                for (int l = 0; l < L; ++l) {
                    const unsigned so = (unsigned)((int)lsi[l] * (int)row) * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vwin + (l * NTOK + k * 16) * D),
                                                             16, (int)vo, (int)so, 0, 0);
                }
            }
        }
        // the (query, head)'s sampling data of all levels: one contiguous run each
        float4 la[RF_MAXL], lb[RF_MAXL], wa[RF_MAXL];
#pragma unroll
        for (int l = 0; l < RF_MAXL; ++l) {
            const int ll = l < L ? l : L - 1;
            la[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2);
            lb[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2 + 4);
            wa[l] = *reinterpret_cast<const float4 *>(aw + e0 + ll * P);
        }
        __syncthreads();                                      // the windows have landed

        rf2 acc[2 * NV] = {};
        if (active) {
#pragma unroll
            for (int l = 0; l < RF_MAXL; ++l) {
                if (l >= L) continue;                         // (uniform; `break` would keep the loop from unrolling)
                const float *wl = vwin + l * NTOK * D;
                const float xs[4] = {la[l].x * fW - 0.5f, la[l].z * fW - 0.5f, lb[l].x * fW - 0.5f, lb[l].z * fW - 0.5f};
                const float ys[4] = {la[l].y * fH - 0.5f, la[l].w * fH - 0.5f, lb[l].y * fH - 0.5f, lb[l].w * fH - 0.5f};
                const float as[4] = {wa[l].x, wa[l].y, wa[l].z, wa[l].w};
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float x = xs[p], y = ys[p], a = as[p];
                    if (fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1)) {
                        const float fx = floorf(x), fy = floorf(y);
                        const int ix = (int)fx - ox, iy = (int)fy - oy;
                        const float wx1 = x - fx, wy1 = y - fy;
                        const float ay1 = wy1 * a, ay0 = a - ay1;
                        const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                        const float *p00 = wl + (iy * WW + ix) * D;
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            const float *pk = p00 + (((2 * sub + k) ^ rot) << 2);
                            rfma4(acc[2 * k], acc[2 * k + 1], w00, *reinterpret_cast<const float4 *>(pk));
                            rfma4(acc[2 * k], acc[2 * k + 1], w01, *reinterpret_cast<const float4 *>(pk + D));
                            rfma4(acc[2 * k], acc[2 * k + 1], w10, *reinterpret_cast<const float4 *>(pk + WW * D));
                            rfma4(acc[2 * k], acc[2 * k + 1], w11, *reinterpret_cast<const float4 *>(pk + WW * D + D));
                        }
                    } else if (y > -1.f && x > -1.f && y < fH && x < fW) {
                        // outside the window: straight from global memory, zero padding by test
                        const Footprint<float> f = footprint(y, x, Hq, Wq);
                        const float *r0 = vbatch + lsi[l] * row + ((int64_t)f.y0 * Wq + f.x0) * row, *r1 = r0 + (int64_t)Wq * row;
                        const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a, w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            const int ko = ((2 * sub + k) ^ rot) << 2;
                            if (f.vy0 && f.vx0) rfma4(acc[2 * k], acc[2 * k + 1], w00, *reinterpret_cast<const float4 *>(r0 + ko));
                            if (f.vy0 && f.vx1) rfma4(acc[2 * k], acc[2 * k + 1], w01, *reinterpret_cast<const float4 *>(r0 + row + ko));
                            if (f.vy1 && f.vx0) rfma4(acc[2 * k], acc[2 * k + 1], w10, *reinterpret_cast<const float4 *>(r1 + ko));
                            if (f.vy1 && f.vx1) rfma4(acc[2 * k], acc[2 * k + 1], w11, *reinterpret_cast<const float4 *>(r1 + row + ko));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);        // one tap's 8 LDS reads in flight at a time
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k)
                *reinterpret_cast<float4 *>(out + q * row + head * D + (((2 * sub + k) ^ rot) << 2)) =
                    make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y);
        }
    }
}

bool msda_resident_supported(int M, int D, int L)
{
    static const bool on = [] { const char *e = getenv("MVDETR_MSDA_RESIDENT"); return !(e && !strcmp(e, "0")); }();
    // six or seven levels: MVDeTr's camera counts (equal shapes in practice; unequal ones would fall to the gather
    // formulation here, where the tile kernel does better -- other level counts keep the tile / group kernels)
    return on && D == RF_D && (L == 6 || L == 7) && M >= 1;
}

int msda_forward_resident(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *aw, int B, int S, int M, int L, float *out, const int *local_hits)
{
    if ((int64_t)S * M * RF_D * 4 >= 0x7fffffffLL) return (int)hipErrorNotSupported;      // 32-bit buffer offsets
    const int lds = L * RF_NTOK * RF_D * 4;
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_resident), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  RF_MAXL * RF_NTOK * RF_D * 4);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return (cus + 7) / 8 * 8;                             // one workgroup per CU (LDS)
    }();
    msda_note_forward_kernel("msda_fwd_resident");
    hipLaunchKernelGGL(msda_fwd_resident, dim3((unsigned)blocks), dim3(RF_THREADS), lds, st, value, shapes, lsi, loc, aw, B, S, M,
                       L, out, local_hits);
    return (int)hipGetLastError();
}

}  // namespace mvdetr
