// Multi-scale deformable attention backward, grad_sampling_loc and grad_attn_weight for deformable-ENCODER
// calls -- gfx950 (MI355X).
//
// Both gradients are dot products of grad_out with the four bilinear corners of each tap, i.e. the forward's
// reads with another reduction.  The generic backward (msda_backward.hip) gathers those corners from global
// memory like the forward gather kernel (1.1 ms at Wildtrack size with its atomics switched off); here the
// forward tile kernel's LDS staging is reused (msda_tile_body.h): a workgroup owns a (tile of query cells of one
// level, 128-byte channel slice), stages the window of each source level in turn and takes its taps from LDS.
// Lane = (cell, 16 channels), grad_out's 16 channels in registers; a 32-channel head is finished by adding the
// two half lanes (neighbours in the wave).  A lane keeps its (query, head)'s results of ALL levels in registers
// and stores them as one contiguous run: a camera-grouped variant (one staged window for all cameras' queries,
// results stored level by level) measured 590 us against 190 us for the forward of the same structure --
// 16/32-byte pieces 112/224 bytes apart are partial-line writes, which ECC memory turns into read-modify-writes.  Taps outside the
// window read global memory.  Levels of unequal shape are detected on the device: msda_bwd_value_win
// (msda_backward_tile.hip, always launched first) then computes all three gradients and this kernel returns.
//
// Replaces (with msda_backward.hip / msda_backward_tile.hip) the grad_sampling_loc / grad_attn_weight half of
// ms_deformable_col2im_cuda (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-234,301-920).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include <stdlib.h>
#include <string.h>

#ifndef MVDETR_RS_DEPTH
#define MVDETR_RS_DEPTH 2     // msda_bwd_sampling_resident: taps whose corner reads are in flight ahead of the tap being finished
#endif
#ifndef MVDETR_RS_LATE
#define MVDETR_RS_LATE 7      // ... and the first level whose sampling data is requested late (7 = none)
#endif

#ifdef MVDETR_BWD_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup's waves
__device__ unsigned long long g_bws_trace[2048];
extern "C" int mvdetr_debug_bws_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bws_trace), n * sizeof(unsigned long long));
}
#define STRACE(i) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (i) < 2048) g_bws_trace[(i)] = wall_clock64(); } while (0)
#else
#define STRACE(i) do { } while (0)
#endif

namespace mvdetr {

typedef float f2 __attribute__((ext_vector_type(2)));

// acc += a * b on the (x, y) and (z, w) halves: two v_pk_fma_f32, the halves are added once per tap (hsum)
__device__ __forceinline__ f2 dot4(const float4 &a, const float4 &b, f2 acc)
{
    acc = __builtin_elementwise_fma((f2){a.x, a.y}, (f2){b.x, b.y}, acc);
    return __builtin_elementwise_fma((f2){a.z, a.w}, (f2){b.z, b.w}, acc);
}
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }
// the value of lane ^ 1 (DPP quad_perm [1,0,3,2]; __shfl_xor goes through ds_bpermute_b32, i.e. the LDS queue)
__device__ __forceinline__ float neighbour(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// d = <g, corner> for the four corners of the tap at pixel position (x, y) of a level, corners read from
// global memory with zero padding; NV float4 chunks per lane, chunk k of `g` holds channels 4*(k^rot)..+3
template <int NV>
__device__ __forceinline__ void corners_of_footprint(const float *__restrict__ vlevel, int64_t row, int W, const Footprint<float> &f,
                                                     int rot, const float4 *g, f2 &d00, f2 &d01, f2 &d10, f2 &d11)
{
    const float *r0 = vlevel + ((int64_t)f.y0 * W + f.x0) * row, *r1 = r0 + (int64_t)W * row;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ko = (k ^ rot) << 2;
        if (f.vy0 && f.vx0) d00 = dot4(g[k], *reinterpret_cast<const float4 *>(r0 + ko), d00);
        if (f.vy0 && f.vx1) d01 = dot4(g[k], *reinterpret_cast<const float4 *>(r0 + row + ko), d01);
        if (f.vy1 && f.vx0) d10 = dot4(g[k], *reinterpret_cast<const float4 *>(r1 + ko), d10);
        if (f.vy1 && f.vx1) d11 = dot4(g[k], *reinterpret_cast<const float4 *>(r1 + row + ko), d11);
    }
}
template <int NV>
__device__ __forceinline__ void corners_from_memory(const float *__restrict__ vlevel, int64_t row, int H, int W, float x,
                                                    float y, int rot, const float4 *g, f2 &d00, f2 &d01, f2 &d10,
                                                    f2 &d11)
{
    corners_of_footprint<NV>(vlevel, row, W, footprint(y, x, H, W), rot, g, d00, d01, d10, d11);
}

// ---- all source windows resident: D = 16, L <= 7 (MVDeTr's own shapes) ------------------------------------------------
// msda_bwd_sampling_tile stages the L source windows once per QUERY level: 7 x 7 windows of 72 KB per 128 cells,
// 1.35 GB of L2->LDS copies per launch at Wildtrack size, most of them L2 misses (FETCH_SIZE 1.18 GB) -- and between two
// barriers per window it has 4 taps per lane to do.  Here a job is (4 x 8 cells, one head) and ALL L source windows of
// that head (16 x 20 tokens x 64 B = 20 KB each) are staged by LDS-DMA before anything is read: one barrier per job,
// 0.39 GB of copies per launch; a lane is one (camera, cell) -- every camera's query at that cell samples the same
// windows -- so a (query, head)'s sampling data of all levels is ONE contiguous 336-byte run on the way in (no
// 16/32-byte pieces per level) and its gradients one contiguous run on the way out.  One workgroup per CU (143 KB of
// LDS at L = 7), 4 waves with up to 512 VGPRs each: the level loop is fully unrolled.
constexpr int RS_TH = 4, RS_TW = 8, RS_R = 6, RS_WH = RS_TH + 2 * RS_R, RS_WW = RS_TW + 2 * RS_R, RS_NTOK = RS_WH * RS_WW;
constexpr int RS_D = 16, RS_MAXL = 7, RS_THREADS = 512;
static_assert(RS_NTOK % 16 == 0, "a DMA instruction covers 16 window positions");

__global__ __launch_bounds__(RS_THREADS, 2) void msda_bwd_sampling_resident(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_loc, float *__restrict__ grad_aw, const int *__restrict__ local_hits)
{
    extern __shared__ __attribute__((aligned(16))) float vwin[];      // [L][RS_NTOK][16]
    constexpr int D = RS_D, TH = RS_TH, TW = RS_TW, WH = RS_WH, WW = RS_WW, NTOK = RS_NTOK, P = TILE_P, NV = 2;
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (local_hits && *local_hits * MSDA_PROBE_NEAR_DIV < MSDA_PROBE_SAMPLES) equal = false;      // far-flung taps: same stand-down
    if (!equal) return;          // msda_bwd_value_win has done all three gradients for such calls

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * M * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;

    // lane = (camera, cell, half of the head's 16 channels): a wave is one camera's 32 cells; two waves per SIMD
    const int sub = tid & 1, cam_raw = (tid >> 1) / (TH * TW), cam = cam_raw < L ? cam_raw : L - 1;
    const int qi = (tid >> 1) % (TH * TW), qly = qi / TW, qlx = qi % TW;
    // LDS bank spreading: lane reads chunks (2 * sub + k) ^ rot.  A bank row is 256 B = four 64-byte tokens, and a quarter
    // wave (16 lanes x 16 B) is served in one pass if its lanes hit 16 different 16-byte slots: its lanes are 8 cells x
    // 2 halves, cells 0-3 cover the four tokens of a bank row with two chunks each, cells 4-7 take the other two
    const int rot = (qlx >> 2) & 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // window copy: an instruction moves 16 consecutive window positions x 4 chunks of 16 bytes
    const int my_pos = lane >> 2, my_chunk = lane & 3;

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
        int shx, shy;                                         // where this head's taps lie
        if (local_hits) {
            msda_probe_shift(local_hits, head, shx, shy);     // (the probe launch of rounds 2-4's callers)
        } else {
            // no probe (round 5): the sample of the grad_value kernel's tile this job lies in (msda_dispatch.h) gives the
            // window shift and says whether that kernel has taken the tile's sampling gradients along (far-flung taps)
            const int sY0 = Y0 / MSDA_SAMPLE_TH * MSDA_SAMPLE_TH, sX0 = X0 / MSDA_SAMPLE_TW * MSDA_SAMPLE_TW;
            const int s_qy = sY0 + lane / MSDA_SAMPLE_TW, s_qx = sX0 + lane % MSDA_SAMPLE_TW;
            const bool have = s_qy < Hq && s_qx < Wq;
            const float *lp = loc + ((((int64_t)b * S + lsi[0] + (have ? (int64_t)s_qy * Wq + s_qx : 0)) * M + head) * L) * P * 2;
            const float4 a0 = *reinterpret_cast<const float4 *>(lp), b0 = *reinterpret_cast<const float4 *>(lp + 4);
            bool far;
            msda_job_sample(a0, b0, have, s_qx, s_qy, fW, fH, shx, shy, far);
            if (far) continue;
        }
        const int oy = Y0 + TH / 2 - WH / 2 + shy, ox = X0 + TW / 2 - WW / 2 + shx;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        [[maybe_unused]] const int tr = ((t - (int)blockIdx.x) / (int)gridDim.x) * 128 + (tid >> 6) * 16;
        STRACE(tr + 0);
        __syncthreads();                                      // everyone is done reading the previous job's windows
        STRACE(tr + 1);
        {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(vbatch), 0, (int)((unsigned)S * (unsigned)row * 4u - (unsigned)(head * D) * 4u), 0x00020000);
            // a wave takes window positions [16 k, 16 k + 16) for k = wave, wave + 8, ... of every level: one address
            // computation per k, one instruction per (k, level)
            for (int k = wave_u; k < NTOK / 16; k += RS_THREADS / 64) {
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
        STRACE(tr + 2);
        float4 g[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) g[k] = *reinterpret_cast<const float4 *>(go + q * row + head * D + (((2 * sub + k) ^ rot) << 2));
        // the (query, head)'s sampling data of all levels: one contiguous run each
        float4 la[RS_MAXL], lb[RS_MAXL], wa[RS_MAXL];
        auto load_levels = [&](int l0, int l1) {
#pragma unroll
            for (int l = 0; l < RS_MAXL; ++l) {
                if (l < l0 || l >= l1) continue;
                const int ll = l < L ? l : L - 1;
                la[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2);
                lb[l] = *reinterpret_cast<const float4 *>(loc + (e0 + ll * P) * 2 + 4);
                wa[l] = *reinterpret_cast<const float4 *>(aw + e0 + ll * P);
            }
        };
        load_levels(0, MVDETR_RS_LATE);
        STRACE(tr + 3);
        __syncthreads();                                      // the windows have landed
        STRACE(tr + 4);

        unsigned far_taps = 0u;                               // bit 4 l + p: in the image, outside the window (finished below)
        // The taps as a software pipeline: the eight corner reads (4 corners x 2 chunks of 16 bytes) of the next DEPTH - 1 taps
        // are in flight while a tap's dots are taken, across level boundaries -- with two waves per SIMD the LDS round trip of
        // a tap (longer under bank conflicts: neighbouring cells' taps are displaced at random) was exposed 28 times per job
        // (round 5: 336 us; DEPTH 2: 284 us).  A tap outside its window reads window token 0 and its dots are discarded.  A
        // level's gradients leave as soon as its four taps are complete: the 336-byte run of a (query, head) is written within
        // one job, 48 bytes at a time, and merges in L2; holding all levels' results until the end cost 84 registers.
        constexpr int DEPTH = MVDETR_RS_DEPTH;
        float4 cbuf[DEPTH][8];
        float twx_[DEPTH], twy_[DEPTH];                       // the far corner's weights of the taps in flight
        bool tin_[DEPTH];
        auto issue = [&](int l, int p, int s_) {
            const float lx = p == 0 ? la[l].x : p == 1 ? la[l].z : p == 2 ? lb[l].x : lb[l].z;
            const float ly = p == 0 ? la[l].y : p == 1 ? la[l].w : p == 2 ? lb[l].y : lb[l].w;
            const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const bool in = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
            const int ix = in ? (int)fx - ox : 0, iy = in ? (int)fy - oy : 0;
            const float *p00 = vwin + l * NTOK * D + (iy * WW + ix) * D;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float *pk = p00 + (((2 * sub + k) ^ rot) << 2);
                cbuf[s_][4 * k + 0] = *reinterpret_cast<const float4 *>(pk);
                cbuf[s_][4 * k + 1] = *reinterpret_cast<const float4 *>(pk + D);
                cbuf[s_][4 * k + 2] = *reinterpret_cast<const float4 *>(pk + WW * D);
                cbuf[s_][4 * k + 3] = *reinterpret_cast<const float4 *>(pk + WW * D + D);
            }
            twx_[s_] = x - fx;
            twy_[s_] = y - fy;
            tin_[s_] = in;
            // a tap outside its window (rare: a wave-uniform branch skips the image test): no gathers between the taps' LDS reads
            // (round 5) -- it is noted, its gradients are zeros in the stream and are written again by the list walk behind the
            // job's stores.  (A tap INSIDE the window needs no image test: the window is zero-padded outside the level, so the
            // dots of a tap whose four corners are all outside are zero.)  Lanes without a cell carry cell 0's taps.
            if (__builtin_amdgcn_ballot_w64(!in) != 0ull) {
                if (!in && active && y > -1.f && x > -1.f && y < fH && x < fW) far_taps |= 1u << (l * P + p);
            }
        };
        float ga[4], gx[4], gy[4];
        auto finish = [&](int l, int p, int s_) {
            const float a = p == 0 ? wa[l].x : p == 1 ? wa[l].y : p == 2 ? wa[l].z : wa[l].w;
            f2 q00 = {0.f, 0.f}, q01 = q00, q10 = q00, q11 = q00;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                q00 = dot4(g[k], cbuf[s_][4 * k + 0], q00);
                q01 = dot4(g[k], cbuf[s_][4 * k + 1], q01);
                q10 = dot4(g[k], cbuf[s_][4 * k + 2], q10);
                q11 = dot4(g[k], cbuf[s_][4 * k + 3], q11);
            }
            // this lane's half of the four dots -> its half of the three gradients (they are linear in the dots), THEN the sum
            // with the other half of the head in the neighbouring lane: three cross-lane adds instead of four
            const float d00 = hsum(q00), d01 = hsum(q01), d10 = hsum(q10), d11 = hsum(q11);
            const float wx1 = twx_[s_], wy1 = twy_[s_];
            const float dx0 = d01 - d00, dx1 = d11 - d10, dy0 = d10 - d00, dy1 = d11 - d01;
            const float top = d00 + wx1 * dx0, bot = d10 + wx1 * dx1;
            float da = top + wy1 * (bot - top);               // = bilinear(d00 .. d11)
            float gxv = dx0 + wy1 * (dx1 - dx0);
            float gyv = dy0 + wx1 * (dy1 - dy0);
            da += neighbour(da);
            gxv += neighbour(gxv);
            gyv += neighbour(gyv);
            const bool in = tin_[s_];
            const float ae = in ? a : 0.f;
            ga[p] = in ? da : 0.f;
            gx[p] = (fW * ae) * gxv;
            gy[p] = (fH * ae) * gyv;
            if (p == P - 1 && active && sub == 0) {
                *reinterpret_cast<float4 *>(grad_aw + e0 + l * P) = make_float4(ga[0], ga[1], ga[2], ga[3]);
                *reinterpret_cast<float4 *>(grad_loc + (e0 + l * P) * 2) = make_float4(gx[0], gy[0], gx[1], gy[1]);
                *reinterpret_cast<float4 *>(grad_loc + (e0 + l * P) * 2 + 4) = make_float4(gx[2], gy[2], gx[3], gy[3]);
            }
        };
#pragma unroll
        for (int t_ = 0; t_ < DEPTH - 1; ++t_)
            if (t_ / P < L) issue(t_ / P, t_ % P, t_ % DEPTH);
#pragma unroll
        for (int t_ = 0; t_ < RS_MAXL * P; ++t_) {
            const int l = t_ / P, p = t_ % P, tn = t_ + DEPTH - 1;
            if (l >= L) continue;                             // (uniform; `break` would keep the loop from unrolling)
            // (the late levels' sampling data: requested once the first level is done, needed MVDETR_RS_LATE - 1 levels later)
            if (MVDETR_RS_LATE < RS_MAXL && t_ == P) load_levels(MVDETR_RS_LATE, RS_MAXL);
            __builtin_amdgcn_sched_barrier(0);
            if (tn < RS_MAXL * P && tn / P < L) issue(tn / P, tn % P, tn % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            finish(l, p, t_ % DEPTH);
        }
        STRACE(tr + 5);
        // ---- taps outside their window: one list per lane (both half-head lanes of a (camera, cell) hold the same list), walked
        //      with the NEXT entry's sampling data requested before the current entry's eight corner gathers -- a round trip per
        //      far tap of the wave's worst lane.  (Inside the tap loop they cost a divergent round trip per tap with any far lane
        //      in the wave: 326 -> 705 us for this kernel between offset spreads of 1 and 3 px.)  Same-lane program order puts
        //      these stores behind the zeros written above.
        if (far_taps) {
            float2 nxy = make_float2(0.f, 0.f);
            float na_ = 0.f;
            int64_t nls = 0;
            int nt = 0;
            auto request = [&]() {
                nt = __ffs((int)far_taps) - 1;
                far_taps &= far_taps - 1u;
                nxy = *reinterpret_cast<const float2 *>(loc + (e0 + nt) * 2);
                na_ = aw[e0 + nt];
                nls = lsi[nt >> 2];
            };
            request();
            for (;;) {
                const float2 xy = nxy;
                const float a = na_;
                const int t_ = nt;
                const float *vlevel = vbatch + nls * row;
                const bool more = far_taps != 0u;
                if (more) request();
                const float x = xy.x * fW - 0.5f, y = xy.y * fH - 0.5f;       // (the tap loop's expressions)
                const Footprint<float> f = footprint(y, x, Hq, Wq);
                const float *r0 = vlevel + ((int64_t)f.y0 * Wq + f.x0) * row, *r1 = r0 + (int64_t)Wq * row;
                f2 q00 = {0.f, 0.f}, q01 = q00, q10 = q00, q11 = q00;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int ko = ((2 * sub + k) ^ rot) << 2;
                    const float4 c00 = load4_or_zero(r0 + ko, f.vy0 && f.vx0, vbatch), c01 = load4_or_zero(r0 + row + ko, f.vy0 && f.vx1, vbatch);
                    const float4 c10 = load4_or_zero(r1 + ko, f.vy1 && f.vx0, vbatch), c11 = load4_or_zero(r1 + row + ko, f.vy1 && f.vx1, vbatch);
                    q00 = dot4(g[k], c00, q00);
                    q01 = dot4(g[k], c01, q01);
                    q10 = dot4(g[k], c10, q10);
                    q11 = dot4(g[k], c11, q11);
                }
                float d00 = hsum(q00), d01 = hsum(q01), d10 = hsum(q10), d11 = hsum(q11);
                d00 += neighbour(d00);
                d01 += neighbour(d01);
                d10 += neighbour(d10);
                d11 += neighbour(d11);
                const float wx1 = x - floorf(x), wy1 = y - floorf(y), wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                if (sub == 0) {
                    grad_aw[e0 + t_] = wy0 * (wx0 * d00 + wx1 * d01) + wy1 * (wx0 * d10 + wx1 * d11);
                    *reinterpret_cast<float2 *>(grad_loc + (e0 + t_) * 2) =
                        make_float2(fW * a * ((d01 - d00) * wy0 + (d11 - d10) * wy1), fH * a * ((d10 - d00) * wx0 + (d11 - d01) * wx1));
                }
                if (!more) break;
            }
        }
    }
}

// ---- level GROUPS resident: D = 32, or more than 7 levels (the 16-camera rig: D = 32, L = 16) ----------------------------
// msda_bwd_sampling_resident keeps ALL L windows of a head in LDS, which stops at 7 levels of 16-channel heads (143 KB); the
// tile kernel that took over beyond that stages every window once per QUERY level (5.0 ms at the stress configuration).
// Same job here -- (4 x 8 cells, one head), lane = (camera, cell, half head) -- but the source levels pass through LDS in
// groups of LG (3 windows of 16 x 20 tokens x 128 B at D = 32), and the cameras in passes of 8 (one wave per camera), so
// every window is still staged once per (tile, head).  A lane's sampling data / gradients of a level group are contiguous
// runs of LG x 32 / LG x 16 bytes.  (At Wildtrack size -- D = 16, L = 7 -- groups of 3 are no faster than all 7 resident:
// 764 vs 747 us for the whole backward with one workgroup per CU, 827 us with two at 128 VGPRs, which spill.)
// FUSED = 1: the fused TRAINING backward's half of the same job (mvdetr_msda_backward_fused_f32 for calls the 6 / 7-camera
// kernel msda_bwd_fused_sampling does not take: 32-channel heads, other level counts): `loc` is the module's raw tensor
// [B, Lq, L, M / hps, (hps x P x 2 offsets in pixels | hps x P logits)] with `raw_q` floats per query, `aw` the forward's softmax
// statistics [B, Lq, M, 2] (maximum, reciprocal sum), `ref` one reference point per (level, query) [.., L, Lq, 2], `out_fwd` the
// forward's output; `grad_loc` is the gradient of the raw tensor -- offsets a (gx, gy), logits a (da - <grad_out, out>) -- and
// `grad_aw` is unused.  The grad_value kernel of those calls never stands a tile down, so neither does this one.
template <int D, int LG, int WPE = 2, int FUSED = 0>
__global__ __launch_bounds__(RS_THREADS, WPE) void msda_bwd_sampling_groups(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_loc, float *__restrict__ grad_aw, const int *__restrict__ local_hits,
    const float *__restrict__ ref, int64_t ref_bstride, int raw_q, const float *__restrict__ out_fwd)
{
    extern __shared__ __attribute__((aligned(16))) float vwin[];      // [LG][RS_NTOK][D]
    constexpr int TH = RS_TH, TW = RS_TW, WH = RS_WH, WW = RS_WW, NTOK = RS_NTOK, P = TILE_P;
    constexpr int HALF = D / 2, NV = HALF / 4, CH = D / 4, POS = 64 / CH, CAMS = RS_THREADS / 64;
    constexpr int HPS = 32 / D, CHUNK = HPS * TILE_P * 3;     // (FUSED) heads per 128-byte slice; floats of a (query, level, slice) run
    static_assert(NTOK % POS == 0, "a DMA instruction covers POS window positions");
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (!FUSED && local_hits && *local_hits * MSDA_PROBE_NEAR_DIV < MSDA_PROBE_SAMPLES) equal = false;      // far-flung taps: same stand-down
    if (!equal) return;          // the grad_value kernel has done all three gradients for such calls (FUSED: it has made the misuse loud)

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * M * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;
    [[maybe_unused]] const float iw = 1.f / fW, ih = 1.f / fH;

    // lane = (camera of the pass, cell, half of the head's channels): a wave is one camera's 32 cells
    const int sub = tid & 1, qi = (tid >> 1) & (TH * TW - 1), qly = qi / TW, qlx = qi % TW;
    const int rot = (qlx / (64 / D)) & (NV - 1);              // LDS bank spreading, as in msda_tile_body.h
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int my_pos = lane / CH, my_chunk = lane % CH;       // window copy: POS positions x CH 16-byte chunks per instruction

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        const int head = job % M, u2 = job / M;               // the heads of a tile run back to back: same token rows
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool in_level = qy < Hq && qx < Wq;
        const float *vbatch = value + (int64_t)b * S * row + head * D;
        int shx, shy;                                         // where this head's taps lie
        if (local_hits) {
            msda_probe_shift(local_hits, head, shx, shy);     // (the probe launch of rounds 2-4's callers)
        } else {
            // no probe (round 5): the sample of the grad_value kernel's tile this job lies in (msda_dispatch.h) gives the
            // window shift and says whether that kernel has taken the tile's sampling gradients along (far-flung taps)
            const int sY0 = Y0 / MSDA_SAMPLE_TH * MSDA_SAMPLE_TH, sX0 = X0 / MSDA_SAMPLE_TW * MSDA_SAMPLE_TW;
            const int s_qy = sY0 + lane / MSDA_SAMPLE_TW, s_qx = sX0 + lane % MSDA_SAMPLE_TW;
            const bool have = s_qy < Hq && s_qx < Wq;
            const int64_t s_q = (int64_t)b * S + lsi[0] + (have ? (int64_t)s_qy * Wq + s_qx : 0);
            float4 a0, b0;
            if constexpr (FUSED) {
                const float *rp = loc + s_q * raw_q + (head / HPS) * CHUNK + (head % HPS) * P * 2;       // (level 0)
                const float2 r = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + (s_q - (int64_t)b * S) * 2);
                a0 = *reinterpret_cast<const float4 *>(rp);
                b0 = *reinterpret_cast<const float4 *>(rp + 4);
                a0 = make_float4(__fmaf_rn(a0.x, iw, r.x), __fmaf_rn(a0.y, ih, r.y), __fmaf_rn(a0.z, iw, r.x), __fmaf_rn(a0.w, ih, r.y));
                b0 = make_float4(__fmaf_rn(b0.x, iw, r.x), __fmaf_rn(b0.y, ih, r.y), __fmaf_rn(b0.z, iw, r.x), __fmaf_rn(b0.w, ih, r.y));
            } else {
                const float *lp = loc + ((s_q * M + head) * L) * P * 2;
                a0 = *reinterpret_cast<const float4 *>(lp);
                b0 = *reinterpret_cast<const float4 *>(lp + 4);
            }
            bool far;
            msda_job_sample(a0, b0, have, s_qx, s_qy, fW, fH, shx, shy, far);
            if (!FUSED && far) continue;
        }
        const int oy = Y0 + TH / 2 - WH / 2 + shy, ox = X0 + TW / 2 - WW / 2 + shx;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        for (int g0 = 0; g0 < L; g0 += LG) {
            const int ng = min(LG, L - g0);
            __syncthreads();                                  // everyone is done reading the previous group's windows
            {
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(vbatch), 0, (int)((unsigned)S * (unsigned)row * 4u - (unsigned)(head * D) * 4u), 0x00020000);
                for (int k = wave_u; k < NTOK / POS; k += RS_THREADS / 64) {
                    const int wp = k * POS + my_pos, wy = wp / WW, wx = wp % WW, gy = oy + wy, gx = ox + wx;
                    const unsigned vo = ((unsigned)gx < (unsigned)Wq && (unsigned)gy < (unsigned)Hq)
                                            ? (unsigned)((gy * Wq + gx) * (int)row + my_chunk * 4) * 4u : 0x80000000u;
                    for (int j = 0; j < ng; ++j) {
                        const unsigned so = (unsigned)((int)lsi[g0 + j] * (int)row) * 4u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(vwin + (j * NTOK + k * POS) * D),
                                                                 16, (int)vo, (int)so, 0, 0);
                    }
                }
            }
            for (int c0 = 0; c0 < L; c0 += CAMS) {
                const int cam = c0 + wave_u;                  // wave-uniform
                const bool active = in_level && cam < L;
                const int64_t q = (int64_t)b * S + lsi[cam < L ? cam : L - 1] + (active ? (int64_t)qy * Wq + qx : 0);
                const int64_t e0 = ((q * M + head) * L + g0) * P;     // this (query, head)'s first tap of the group
                float4 g[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) g[k] = *reinterpret_cast<const float4 *>(go + q * row + head * D + sub * HALF + ((k ^ rot) << 2));
                float4 la[LG], lb[LG], wa[LG];
                [[maybe_unused]] float2 rf[LG];               // (FUSED) reference point of (level, query)
                [[maybe_unused]] float2 st = make_float2(0.f, 1.f);
                [[maybe_unused]] float dq = 0.f;              // (FUSED) <grad_out, out> of the (query, head)
                // (FUSED) the raw tensor's run of (query, level, slice): this head's offsets, then its logits
                [[maybe_unused]] const int64_t r0_ = q * raw_q + (head / HPS) * CHUNK + (head % HPS) * P * 2;
                [[maybe_unused]] const int64_t w0_ = q * raw_q + (head / HPS) * CHUNK + HPS * P * 2 + (head % HPS) * P;
                [[maybe_unused]] const int l_stride = (M / HPS) * CHUNK;
#pragma unroll
                for (int j = 0; j < LG; ++j) {
                    const int jj = j < ng ? j : ng - 1;
                    if constexpr (FUSED) {
                        la[j] = *reinterpret_cast<const float4 *>(loc + r0_ + (g0 + jj) * l_stride);
                        lb[j] = *reinterpret_cast<const float4 *>(loc + r0_ + (g0 + jj) * l_stride + 4);
                        wa[j] = *reinterpret_cast<const float4 *>(loc + w0_ + (g0 + jj) * l_stride);
                        rf[j] = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)(g0 + jj) * S + (q - (int64_t)b * S)) * 2);
                    } else {
                        la[j] = *reinterpret_cast<const float4 *>(loc + (e0 + jj * P) * 2);
                        lb[j] = *reinterpret_cast<const float4 *>(loc + (e0 + jj * P) * 2 + 4);
                        wa[j] = *reinterpret_cast<const float4 *>(aw + e0 + jj * P);
                    }
                }
                if constexpr (FUSED) {
                    st = *reinterpret_cast<const float2 *>(aw + (q * M + head) * 2);
                    f2 dq2 = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        dq2 = dot4(g[k], *reinterpret_cast<const float4 *>(out_fwd + q * row + head * D + sub * HALF + ((k ^ rot) << 2)), dq2);
                    dq = hsum(dq2);
                    dq += neighbour(dq);
                }
                if (c0 == 0) __syncthreads();                 // the group's windows have landed
                // The group's taps -- ng levels x 4 points -- as a two-stage software pipeline, as in msda_bwd_sampling_resident:
                // tap t + 1's corner reads (4 corners x NV chunks of 16 bytes) are in flight while tap t's dots are taken.  A tap
                // outside its window reads window token 0, its dots are discarded and it is noted in `far_taps` (bit 4 j + p),
                // finished from global memory by the list walk behind the stream; a level's gradients leave as soon as its four
                // taps are complete.
                unsigned far_taps = 0u;
                float4 cbuf[2][4 * NV];
                // position of tap (level j of the group, point p_): pixel coordinates, their floors, the far corner's weights and
                // the window test -- evaluated when the tap's reads are issued AND when it is finished (the same expressions give
                // the same bits; carrying them through the pipeline cost registers the 32-channel instantiation does not have)
                auto tap_pos = [&](int j, int p_, float &x, float &y, float &fx, float &fy, float &wx1, float &wy1) {
                    const float lx = p_ == 0 ? la[j].x : p_ == 1 ? la[j].z : p_ == 2 ? lb[j].x : lb[j].z;
                    const float ly = p_ == 0 ? la[j].y : p_ == 1 ? la[j].w : p_ == 2 ? lb[j].y : lb[j].w;
                    if constexpr (FUSED) {
                        // (offsets in pixels; the position is formed in two parts, common.h fused_px)
                        fused_px(rf[j].x, lx, fW, x, fx, wx1);
                        fused_px(rf[j].y, ly, fH, y, fy, wy1);
                    } else {
                        x = lx * fW - 0.5f;
                        y = ly * fH - 0.5f;
                        fx = floorf(x);
                        fy = floorf(y);
                        wx1 = x - fx;
                        wy1 = y - fy;
                    }
                    return fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                };
                auto issue = [&](int j, int p_, int s_) {
                    float x, y, fx, fy, wx1, wy1;
                    const bool in = tap_pos(j, p_, x, y, fx, fy, wx1, wy1);
                    const int ix = in ? (int)fx - ox : 0, iy = in ? (int)fy - oy : 0;
                    const float *p00 = vwin + j * NTOK * D + (iy * WW + ix) * D + sub * HALF;
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        const float *pk = p00 + ((k ^ rot) << 2);
                        cbuf[s_][4 * k + 0] = *reinterpret_cast<const float4 *>(pk);
                        cbuf[s_][4 * k + 1] = *reinterpret_cast<const float4 *>(pk + D);
                        cbuf[s_][4 * k + 2] = *reinterpret_cast<const float4 *>(pk + WW * D);
                        cbuf[s_][4 * k + 3] = *reinterpret_cast<const float4 *>(pk + WW * D + D);
                    }
                };
                // the three gradients of one tap from its four dots (public contract: grad_attn_weight, grad_sampling_loc; FUSED:
                // logit, offsets -- a tap outside the image still has a logit: its weight takes part in the softmax)
                // (takes THIS lane's half of the four dots: the gradients are linear in them, so the lane forms its half of
                // (d a, d x, d y) and the two half-head lanes are summed afterwards -- three cross-lane adds instead of four)
                auto tap_grads = [&](float d00, float d01, float d10, float d11, float wx1, float wy1, float a, bool in_image, float &ga_,
                                     float &gx_, float &gy_) {
                    const float dx0 = d01 - d00, dx1 = d11 - d10, dy0 = d10 - d00, dy1 = d11 - d01;
                    const float top = d00 + wx1 * dx0, bot = d10 + wx1 * dx1;
                    float da = top + wy1 * (bot - top);       // = bilinear(d00 .. d11)
                    float gxv = dx0 + wy1 * (dx1 - dx0);
                    float gyv = dy0 + wx1 * (dy1 - dy0);
                    da += neighbour(da);                      // the other half of the head sits in the neighbouring lane
                    gxv += neighbour(gxv);
                    gyv += neighbour(gyv);
                    da = in_image ? da : 0.f;
                    const float ae = in_image ? a : 0.f;
                    if constexpr (FUSED) {
                        ga_ = a * (da - dq);
                        gx_ = ae * gxv;
                        gy_ = ae * gyv;
                    } else {
                        ga_ = da;
                        gx_ = (fW * ae) * gxv;
                        gy_ = (fH * ae) * gyv;
                    }
                };
                float ga[4], gx[4], gy[4];
                auto finish = [&](int j, int p_, int s_) {
                    float x, y, fx_, fy_, wx1, wy1;
                    const bool in = tap_pos(j, p_, x, y, fx_, fy_, wx1, wy1);
                    float a = p_ == 0 ? wa[j].x : p_ == 1 ? wa[j].y : p_ == 2 ? wa[j].z : wa[j].w;
                    if constexpr (FUSED) a = __expf(a - st.x) * st.y;
                    f2 q00 = {0.f, 0.f}, q01 = q00, q10 = q00, q11 = q00;
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        q00 = dot4(g[k], cbuf[s_][4 * k + 0], q00);
                        q01 = dot4(g[k], cbuf[s_][4 * k + 1], q01);
                        q10 = dot4(g[k], cbuf[s_][4 * k + 2], q10);
                        q11 = dot4(g[k], cbuf[s_][4 * k + 3], q11);
                    }
                    float d00 = hsum(q00), d01 = hsum(q01), d10 = hsum(q10), d11 = hsum(q11);
                    const bool in_image = y > -1.f && x > -1.f && y < fH && x < fW;
                    if (!in) {
                        d00 = d01 = d10 = d11 = 0.f;
                        if (active && in_image) far_taps |= 1u << (j * P + p_);      // (lanes without a cell carry cell 0's taps)
                    }
                    tap_grads(d00, d01, d10, d11, wx1, wy1, a, in_image, ga[p_], gx[p_], gy[p_]);
                    if (p_ == P - 1 && active && sub == 0) {
                        if constexpr (FUSED) {
                            *reinterpret_cast<float4 *>(grad_loc + w0_ + (g0 + j) * l_stride) = make_float4(ga[0], ga[1], ga[2], ga[3]);
                            *reinterpret_cast<float4 *>(grad_loc + r0_ + (g0 + j) * l_stride) = make_float4(gx[0], gy[0], gx[1], gy[1]);
                            *reinterpret_cast<float4 *>(grad_loc + r0_ + (g0 + j) * l_stride + 4) = make_float4(gx[2], gy[2], gx[3], gy[3]);
                        } else {
                            *reinterpret_cast<float4 *>(grad_aw + e0 + j * P) = make_float4(ga[0], ga[1], ga[2], ga[3]);
                            *reinterpret_cast<float4 *>(grad_loc + (e0 + j * P) * 2) = make_float4(gx[0], gy[0], gx[1], gy[1]);
                            *reinterpret_cast<float4 *>(grad_loc + (e0 + j * P) * 2 + 4) = make_float4(gx[2], gy[2], gx[3], gy[3]);
                        }
                    }
                };
                issue(0, 0, 0);
#pragma unroll
                for (int t_ = 0; t_ < LG * P; ++t_) {
                    const int j = t_ / P, p_ = t_ % P;
                    if (j >= ng) continue;                    // (uniform)
                    __builtin_amdgcn_sched_barrier(0);
                    if (t_ + 1 < LG * P && (t_ + 1) / P < ng) issue((t_ + 1) / P, (t_ + 1) % P, (t_ + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    finish(j, p_, t_ & 1);
                }
                // ---- taps outside their window: one list per lane (both half-head lanes of a (camera, cell) hold the same list),
                //      the tap's sampling data read again (the registers cannot be indexed by a run-time tap), its corners gathered
                //      from global memory with zero padding; same-lane program order puts these stores behind the zeros above
                while (far_taps) {
                    const int t_ = __ffs((int)far_taps) - 1, j = t_ >> 2, p_ = t_ & 3;
                    far_taps &= far_taps - 1u;
                    float x, y, fx, fy, wx1, wy1, a;
                    if constexpr (FUSED) {
                        const float2 o = *reinterpret_cast<const float2 *>(loc + r0_ + (g0 + j) * l_stride + p_ * 2);
                        const float2 r = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)(g0 + j) * S + (q - (int64_t)b * S)) * 2);
                        fused_px(r.x, o.x, fW, x, fx, wx1);
                        fused_px(r.y, o.y, fH, y, fy, wy1);
                        a = __expf(loc[w0_ + (g0 + j) * l_stride + p_] - st.x) * st.y;
                    } else {
                        const float2 o = *reinterpret_cast<const float2 *>(loc + (e0 + t_) * 2);
                        x = o.x * fW - 0.5f;                  // (the stream's expressions)
                        y = o.y * fH - 0.5f;
                        fx = floorf(x);
                        fy = floorf(y);
                        wx1 = x - fx;
                        wy1 = y - fy;
                        a = aw[e0 + t_];
                    }
                    f2 q00 = {0.f, 0.f}, q01 = q00, q10 = q00, q11 = q00;
                    corners_of_footprint<NV>(vbatch + lsi[g0 + j] * row + sub * HALF, row, Wq, footprint_split(fy, wy1, fx, wx1, Hq, Wq), rot, g,
                                             q00, q01, q10, q11);
                    const float d00 = hsum(q00), d01 = hsum(q01), d10 = hsum(q10), d11 = hsum(q11);
                    float ga1, gx1, gy1;
                    tap_grads(d00, d01, d10, d11, wx1, wy1, a, true, ga1, gx1, gy1);
                    if (sub == 0) {
                        if constexpr (FUSED) {
                            grad_loc[w0_ + (g0 + j) * l_stride + p_] = ga1;
                            *reinterpret_cast<float2 *>(grad_loc + r0_ + (g0 + j) * l_stride + p_ * 2) = make_float2(gx1, gy1);
                        } else {
                            grad_aw[e0 + t_] = ga1;
                            *reinterpret_cast<float2 *>(grad_loc + (e0 + t_) * 2) = make_float2(gx1, gy1);
                        }
                    }
                }
            }
        }
    }
}

template <int D, int LG, int WPE = 2, int FUSED = 0>
static int launch_sampling_groups(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                  const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                                  float *grad_loc, float *grad_aw, const int *local_hits, const float *ref = nullptr,
                                  int64_t ref_bstride = 0, int raw_q = 0, const float *out_fwd = nullptr)
{
    constexpr int LDS = LG * RS_NTOK * D * 4;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_sampling_groups<D, LG, WPE, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return (cus * (WPE / 2) + 7) / 8 * 8;                 // WPE / 2 workgroups of 8 waves per CU
    });
    hipLaunchKernelGGL((msda_bwd_sampling_groups<D, LG, WPE, FUSED>), dim3((unsigned)blocks), dim3(RS_THREADS), LDS, st, go, value, shapes,
                       lsi, loc, aw, B, S, M, L, grad_loc, grad_aw, local_hits, ref, ref_bstride, raw_q, out_fwd);
    return (int)hipGetLastError();
}

static int launch_sampling_resident(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                    const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                                    float *grad_loc, float *grad_aw, const int *local_hits)
{
    const int lds = L * RS_NTOK * RS_D * 4;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_sampling_resident),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RS_MAXL * RS_NTOK * RS_D * 4);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return (cus + 7) / 8 * 8;                             // one workgroup per CU (LDS)
    });
    hipLaunchKernelGGL(msda_bwd_sampling_resident, dim3((unsigned)blocks), dim3(RS_THREADS), lds, st, go, value, shapes, lsi,
                       loc, aw, B, S, M, L, grad_loc, grad_aw, local_hits);
    return (int)hipGetLastError();
}

int msda_backward_sampling_tile(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                                float *grad_loc, float *grad_aw, const int *local_hits)
{
#define SAMPLING_ARGS st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_loc, grad_aw, local_hits
    if (D == RS_D && L <= RS_MAXL && (int64_t)S * M * D * 4 < 0x7fffffffLL) return launch_sampling_resident(SAMPLING_ARGS);
    // more levels or 32-channel heads: the same job with the levels passing through LDS in groups
    if (D == 32) return launch_sampling_groups<32, 3>(SAMPLING_ARGS);
    if (D == 16) return launch_sampling_groups<16, 7>(SAMPLING_ARGS);
    return (int)hipErrorInvalidValue;
}

// the fused TRAINING backward's sampling half for calls msda_backward_fused_sampling (6 / 7 levels of 16-channel heads) does not
// take: the level-groups kernel on the raw tensor (see its header)
int msda_backward_fused_sampling_groups(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                        const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                        const float *stats, const float *out_fwd, int B, int S, int M, int D, int L, float *grad_raw)
{
    if ((int64_t)S * M * D * 4 >= 0x7fffffffLL) return (int)hipErrorNotSupported;
    if (D == 32) return launch_sampling_groups<32, 3, 2, 1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_raw, nullptr, nullptr, ref, ref_bstride, raw_q, out_fwd);
    if (D == 16) return launch_sampling_groups<16, 7, 2, 1>(st, go, value, shapes, lsi, raw, stats, B, S, M, L, grad_raw, nullptr, nullptr, ref, ref_bstride, raw_q, out_fwd);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
