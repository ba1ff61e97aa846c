// Camera-grouped fused MSDA forward: the kernel template and its launcher (see msda_forward_group.hip for the design notes).
// Shared by msda_forward_group.hip (6 / 7 cameras) and msda_forward_group_many.hip (9..16 cameras, camera-split lane groups)
// so that the two sets of instantiations compile in parallel.
#pragma once
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_tile_body.h"
#include "msda_gather_body.h"
#include <stdlib.h>
#include <string.h>

#ifndef MVDETR_SHIFT_MAX
#define MVDETR_SHIFT_MAX 3
#endif

// kernel options (the `opts` argument)
#define GROUP_OPT_NO_SHIFT 1      // keep the windows centred on the tile (A/B knob MVDETR_MSDA_WINDOW_SHIFT=0)
#define GROUP_OPT_BLOCKS 2        // XCD k takes a 2-D block of the tile grid instead of a band of the job list
#define GROUP_OPT_STANDDOWN 4     // (msda_fwd_group2, public contract) a job whose taps are far from its cells gathers instead
// -DMVDETR_GROUP_TRACE builds (libmvdetr_ops_trace.so, tools/experiments/group_trace.py): every wave's lane 0 stamps the
// 100 MHz wall clock at the phase boundaries of its workgroup's FIRST job into a global table
// [workgroup][wave 0..3][GROUP_TRACE_SLOTS]: 0 kernel entry, 1 shift known, 2 + 2l level l's window resident, 3 + 2l its
// taps done, 2 + 2 * TILE_MAX_LEVELS job stored; the last slot holds the hardware id (XCC_ID << 32 | HW_ID).
#ifdef MVDETR_GROUP_TRACE
#define GROUP_TRACE_SLOTS 40
namespace mvdetr { extern __device__ unsigned long long *g_group_trace; }
#define GROUP_STAMP(slot)                                                                                              \
    do {                                                                                                               \
        if (g_group_trace && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4) {                                       \
            unsigned long long *tr_ = g_group_trace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GROUP_TRACE_SLOTS; \
            if (tr_[slot] == 0) tr_[slot] = wall_clock64();                                                            \
            if ((slot) == 0)                                                                                           \
                tr_[GROUP_TRACE_SLOTS - 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) | \
                                             __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));                      \
        }                                                                                                              \
    } while (0)
#else
#define GROUP_STAMP(slot) do { } while (0)
#endif

namespace mvdetr {

typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gfma4(float2v &lo, float2v &hi, float w, const float4 &c)
{
    const float2v ww = {w, w};
    lo = __builtin_elementwise_fma(ww, (float2v){c.x, c.y}, lo);
    hi = __builtin_elementwise_fma(ww, (float2v){c.z, c.w}, hi);
}

// FUSED = false: `off` / `logit` hold final sampling locations / attention weights (the extension's public
// contract, any SamplingLayout) and `ref` is unused.
template <bool WIDE> struct MissMask { using type = unsigned; };
template <> struct MissMask<true> { using type = unsigned long long; };

// SPLIT > 1: the workgroup has SPLIT lane groups of TH*TW*2 lanes each; all use the same staged window, group g
// takes the cameras [g*NGA, (g+1)*NGA) with NGA = ceil(NG/SPLIT) -- NGA accumulator sets per lane instead of NG,
// which is what makes many-camera rigs (16 cameras: 4 groups x 4) fit the register file at all.
template <typename Cfg, int NG, int WAVES, int FUSED, int SPLIT = 1, bool DMA = false>
__global__ __launch_bounds__(Cfg::THREADS, WAVES) void msda_fwd_group(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ off, const float *__restrict__ logit,
    const float *__restrict__ ref, int64_t ref_bstride, SamplingLayout lay, int B, int S, int M,
    float *__restrict__ out, const int *__restrict__ local_hits, int opts, float *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    GROUP_STAMP(0);
    if constexpr (FUSED == 0) {
        // the locality probe found the taps far from their queries: windows would be wasted, gather instead
        if (local_hits && *local_hits * MSDA_PROBE_NEAR_DIV < MSDA_PROBE_SAMPLES) {
            msda_fwd_gather_body<float, 4>((int64_t)blockIdx.x * Cfg::THREADS + threadIdx.x, (int64_t)gridDim.x * Cfg::THREADS,
                                           value, shapes, lsi, off, logit, B, S, M, Cfg::D, NG, S, TILE_P, out);
            return;
        }
    }
    constexpr int D = Cfg::D, TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW;
    constexpr int SLICE = Cfg::SLICE, P = TILE_P, NV = Cfg::NV, NSTAGE = Cfg::NSTAGE, LCH = SLICE / 2, L = NG;
    constexpr int RPP = Cfg::ROWS_PER_PASS;
    const int tid = threadIdx.x;
    const int HS = M * D / SLICE;
    const int row = M * D;

    // levels of unequal shape are not ours (see the header of msda_forward_group.hip): the tile kernel's body takes the call
    auto levels_equal = [&]() {
        bool eq = true;
        for (int l = 1; l < L; ++l) eq = eq && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
        return eq;
    };
    auto fallback = [&]() {
#ifndef MVDETR_GROUP_NO_FALLBACK
        using Fallback = TileCfg<Cfg::D, 32, 8, 16, 6, Cfg::THREADS>;
        msda_fwd_tile_body<Fallback, FUSED>(win, value, shapes, lsi, off, logit, ref, ref_bstride, lay,
                                            QueryLevels{0, L, S}, B, S, M, L, out);
#endif
    };
    if (!levels_equal()) {
        fallback();
        // (the training entry needs equal level shapes -- its caller checks; statistics this path cannot give are NaN)
        if (stats)
            for (int64_t i = (int64_t)blockIdx.x * Cfg::THREADS + threadIdx.x; i < (int64_t)B * S * M * 2; i += (int64_t)gridDim.x * Cfg::THREADS)
                stats[i] = __builtin_nanf("");
        return;
    }
    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    // level size and its reciprocal once, in scalar registers (left inside the tap loop the two IEEE divisions were
    // re-done for every camera of every level)
    const float fW = (float)Wq, fH = (float)Hq;
    const float iw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / fW)));
    const float ih = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / fH)));
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * HS * B, jobs8 = (jobs + 7) / 8;

    constexpr int GROUP_LANES = SPLIT == 1 ? Cfg::THREADS : TH * TW * 2;       // lanes of one camera group
    static_assert(SPLIT == 1 || (GROUP_LANES % 64 == 0 && GROUP_LANES * SPLIT <= Cfg::THREADS), "whole waves per group");
    constexpr int NGA = (NG + SPLIT - 1) / SPLIT;             // cameras (accumulator sets) per lane
    using MissT = typename MissMask<(NG * TILE_P > 32)>::type;    // one bit per (level, point)
    const int grp = SPLIT == 1 ? 0 : tid / GROUP_LANES;       // wave-uniform
    const int ltid = tid - grp * GROUP_LANES;
    const int cam0 = grp * NGA;
    const int ncam = grp >= SPLIT ? 0 : (NG - cam0 < NGA ? (NG - cam0 < 0 ? 0 : NG - cam0) : NGA);
    const int sub = ltid & 1, qi = ltid >> 1;
    const int qly = qi / TW, qlx = qi % TW;
    const int rot = (qlx / Cfg::TOK_PER_BANKROW) & (NV - 1);
    const int lane_off = sub * LCH;
    // window copy: thread moves float4 `my_part` of window column `my_col`, rows my_row0 + i * RPP
    const int my_part = tid % Cfg::PARTS, my_slot = tid / Cfg::PARTS;
    const int my_row0 = my_slot / WW, my_col = my_slot % WW;
    const bool col_ok = my_row0 < RPP;
    float *const st_dst = win + (my_row0 * WW + my_col) * SLICE + my_part * 4;

    // Which jobs an XCD (workgroups t = k mod 8 share XCD k's L2) takes.  GROUP_OPT_BLOCKS: a 2-D block of the tile grid
    // -- the windows of neighbouring tiles overlap by 12 of their 18 rows / 28 columns, and what one XCD's jobs fetch is
    // the union of their windows: 15 tiles in a row (the band below) fetch 3x the tokens they own, a 5 x 3 block 1.75x.
    // Otherwise (batched calls: several rounds per workgroup) a contiguous band of the job list.
    const int trows = (Hq + TH - 1) / TH;
    int gy = 1, gx = 8;                                       // XCD grid over the tile grid, gy * gx = 8
    if (opts & GROUP_OPT_BLOCKS) {
        int best = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cy = 1 << e, cx = 8 >> e;
            const int bh = (trows + cy - 1) / cy, bw = (tcols + cx - 1) / cx;
            const int cost = bh * bw * 64 + bh + bw;          // fewest rounds first, then the shortest perimeter
            if (cost < best) { best = cost; gy = cy; gx = cx; }
        }
    }
    const int bh = (trows + gy - 1) / gy, bw = (tcols + gx - 1) / gx;
    const int blk_jobs = bh * bw * HS * B;                    // per XCD, incl. the block's cells outside the grid
    const int per_xcd = (opts & GROUP_OPT_BLOCKS) ? blk_jobs : jobs8;

    for (int t = blockIdx.x; t < per_xcd * 8; t += gridDim.x) {
        const int k = t & 7, idx = t >> 3;
        if (idx >= per_xcd) continue;
        int hs, b, ty, tx;
        if (opts & GROUP_OPT_BLOCKS) {
            hs = idx % HS;
            const int r = idx / HS, tib = r % (bh * bw);
            b = r / (bh * bw);
            ty = (k / gx) * bh + tib / bw;
            tx = (k % gx) * bw + tib % bw;
            if (ty >= trows || tx >= tcols) continue;
        } else {
            const int job = k * jobs8 + idx;                  // XCD k takes a contiguous band of jobs
            if (job >= jobs) continue;
            hs = job % HS;
            const int u2 = job / HS, tin = u2 % per_level;
            b = u2 / per_level;
            ty = tin / tcols;
            tx = tin % tcols;
        }
        const int Y0 = ty * TH, X0 = tx * TW;
        const int ch0 = hs * SLICE + lane_off;
        const int head = ch0 / D, ch_off = ch0 % D;
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = ncam > 0 && qi < TH * TW && qy < Hq && qx < Wq;
        // per-lane part of the sampling-data addresses (cell, head); camera c adds (b*S + lsi[c]) queries
        // Addresses = a wave-uniform 64-bit pointer (scalar registers: tensor base + camera / level part) + a 32-bit per-lane
        // offset in floats (one VGPR; the launcher keeps tensors whose per-batch extent exceeds 2^32 bytes away from this
        // kernel): 64-bit per-lane pointers cost two registers each plus the 64-bit adds of every load's address, and the
        // tap loop is short of exactly those registers (with them the compiler serialised the LDS reads two at a time).
        const unsigned cell = active ? (unsigned)(qy * Wq + qx) : 0u;
        const unsigned lane_l = cell * (unsigned)lay.q_l + (unsigned)lay.head_l(head);
        const unsigned lane_w = cell * (unsigned)lay.q_w + (unsigned)lay.head_w(head);
        const unsigned lane_r = cell * (unsigned)lay.r_q;
        const float *const refb = FUSED ? ref + b * ref_bstride : nullptr;
        auto cam_q = [&](int c) { return (int64_t)b * S + lsi[c]; };          // wave-uniform
        const float *vbatch = value + (int64_t)b * S * row + hs * SLICE;

        // sampling data of (camera, level): raw offsets (2 x float4), raw logits, reference points; camera c+1's loads are
        // in flight while camera c's taps run
        float4 na = make_float4(0, 0, 0, 0), nb = na, nw = na, nra = na, nrb = na;
        auto load_cam = [&](int c, int l) {
            const int64_t cq = cam_q(c);
            const float *ul = off + (cq * lay.q_l + l * lay.l_l);                       // uniform
            const float *uw = logit + (cq * lay.q_w + l * lay.l_w);
            na = *reinterpret_cast<const float4 *>(ul + lane_l);
            nb = *reinterpret_cast<const float4 *>(ul + lane_l + 4);
            nw = *reinterpret_cast<const float4 *>(uw + lane_w);
            if constexpr (FUSED == 1) {
                const float *ur = refb + ((cq - (int64_t)b * S) * lay.r_q + l * lay.r_l);
                nra = *reinterpret_cast<const float4 *>(ur + lane_r);
                nrb = *reinterpret_cast<const float4 *>(ur + lane_r + 4);
            } else if constexpr (FUSED == 2) {
                const float *ur = refb + ((cq - (int64_t)b * S) * lay.r_q + l * lay.r_l);
                const float2 r = *reinterpret_cast<const float2 *>(ur + lane_r);
                nra = nrb = make_float4(r.x, r.y, r.x, r.y);
            }
        };
        // camera-split variants with four lane groups (3 waves per SIMD, 168 registers): no look-ahead, the other waves
        // cover the load -- the prefetch registers spilled (91 VGPRs at 16 cameras)
        constexpr bool AHEAD = SPLIT <= 2;

        // Window centre (round 3): the taps of a slice's heads are not centred on the query cell -- MSDeformAttn's offset
        // bias is a ray per head (ms_deform_attn.py:64-69: 1..4 px along the head's direction), so with +-6 px windows around
        // the cell the far points of a ray leave the window as soon as the learned part adds a pixel or two, and every
        // such tap is a serialised global gather at the end of the job.  The tile measures where its taps lie -- the mean
        // displacement of the first camera's first-level taps from their own cells -- and shifts all of the job's windows
        // by that (rounded, at most +-3 px; the same for every level: the bias does not depend on the level).  Any shift
        // gives the same results -- taps outside the window are gathered from memory -- it only decides how many do.
        int shift_x = 0, shift_y = 0;
        {
            // every wave looks at the SAME sample -- the tile's first 32 cells x the slice's two halves, camera 0, level 0 --
            // so all waves arrive at the same shift without exchanging anything (no LDS, no barrier)
            const int sl = tid & 63, s_sub = sl & 1, s_qi = sl >> 1;
            const int s_qy = Y0 + s_qi / TW, s_qx = X0 + s_qi % TW;
            const int s_head = (hs * SLICE + s_sub * LCH) / D;
            const bool s_ok = s_qy < Hq && s_qx < Wq;
            float4 a0 = make_float4(0, 0, 0, 0), b0 = a0;
            float rx = 0.f, ry = 0.f;
            if (s_ok) {
                const int64_t s_cell = (int64_t)s_qy * Wq + s_qx, cq = cam_q(0);
                const float *lp = off + (cq + s_cell) * lay.q_l + lay.head_l(s_head);
                a0 = *reinterpret_cast<const float4 *>(lp);
                b0 = *reinterpret_cast<const float4 *>(lp + 4);
                if constexpr (FUSED) {
                    // raw offsets are already pixels relative to the reference point; that point relative to the cell:
                    const float *rp = ref + b * ref_bstride + (cq - (int64_t)b * S + s_cell) * lay.r_q;
                    rx = FUSED == 2 ? rp[0] : 0.25f * ((rp[0] + rp[2]) + (rp[4] + rp[6]));
                    ry = FUSED == 2 ? rp[1] : 0.25f * ((rp[1] + rp[3]) + (rp[5] + rp[7]));
                }
            }
            float sx = 0.f, sy = 0.f, sn = 0.f;
            if (s_ok) {
                float mx = 0.25f * ((a0.x + a0.z) + (b0.x + b0.z)), my = 0.25f * ((a0.y + a0.w) + (b0.y + b0.w));
                if constexpr (FUSED) {
                    mx += rx * fW - 0.5f - (float)s_qx;
                    my += ry * fH - 0.5f - (float)s_qy;
                } else {
                    mx = mx * fW - 0.5f - (float)s_qx;
                    my = my * fH - 0.5f - (float)s_qy;
                }
                if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sx += __shfl_xor(sx, o, 64);
                sy += __shfl_xor(sy, o, 64);
                sn += __shfl_xor(sn, o, 64);
            }
            // (the butterfly leaves every lane with the same sums only up to the order of the additions: take lane 0's)
            const float tx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
            const float ty = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            if (tn > 0.f && !(opts & GROUP_OPT_NO_SHIFT)) {
                shift_x = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(tx / tn)));
                shift_y = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(ty / tn)));
            }
        }
        GROUP_STAMP(1);

        float2v acc[NGA][2 * NV];
        float smax[NGA], ssum[NGA];
        MissT miss[NGA];
#pragma unroll
        for (int c = 0; c < NGA; ++c) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) acc[c][i] = (float2v){0.f, 0.f};
            smax[c] = -INFINITY;
            ssum[c] = 0.f;
            miss[c] = 0;
        }

        for (int l = 0; l < L; ++l) {
            // window of level l around the tile (all levels have the query level's shape), shifted to where the taps are
            const int oy = Y0 + TH / 2 - WH / 2 + shift_y, ox = X0 + TW / 2 - WW / 2 + shift_x;
            const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
            __syncthreads();                                  // everyone is done reading the old window
            if constexpr (DMA) {
                // LDS-DMA (buffer_load ... lds): a wave's 64 lanes are the 8 x 16-byte chunks of 8 consecutive window
                // positions (row-major), i.e. 1 KB contiguous in LDS behind a wave-uniform base -- no staging registers,
                // no ds_write; positions outside the level are out-of-range reads and store zeros.
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(vbatch), 0, (int)((unsigned)S * row * 4u - (unsigned)(hs * SLICE) * 4u), 0x00020000);
                const int gx = ox + my_col;
                const bool xok = (unsigned)gx < (unsigned)Wq;
                const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
                const unsigned so = (unsigned)((int)lsi[l] * row) * 4u;
                if (col_ok) {
#pragma unroll
                    for (int i = 0; i < NSTAGE; ++i) {
                        const int wy = my_row0 + i * RPP, gy = oy + wy;
                        if (wy < WH) {
                            const unsigned vo = (xok && (unsigned)gy < (unsigned)Hq) ? (unsigned)((gy * Wq + gx) * row + my_part * 4) * 4u : 0x80000000u;
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(win + (i * RPP * WW + wave_u * 8) * SLICE),
                                                                     16, (int)vo, (int)so, 0, 0);
                        }
                        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (a few offsets at a time: the accumulators fill the file)
                    }
                }
            } else {
                const int gx = ox + my_col;
                const bool xok = col_ok && (unsigned)gx < (unsigned)Wq;
                const float *colp = vbatch + lsi[l] * row + my_part * 4 + (xok ? gx : 0) * row;
                // SPLIT > 1 runs at 3 waves / SIMD (168 VGPRs): the column goes through the registers in chunks
                constexpr int CHUNK = SPLIT == 1 ? NSTAGE : 6;
#pragma unroll
                for (int i0 = 0; i0 < NSTAGE; i0 += CHUNK) {
                    float4 stage[CHUNK];
#pragma unroll
                    for (int j = 0; j < CHUNK; ++j) {
                        const int i = i0 + j;
                        const int wy = RPP == 1 ? i : my_row0 + i * RPP;       // RPP == 1: scalar row arithmetic
                        const int gy = oy + wy;
                        stage[j] = make_float4(0, 0, 0, 0);
                        if (i < NSTAGE && xok && wy < WH && (unsigned)gy < (unsigned)Hq)
                            stage[j] = *reinterpret_cast<const float4 *>(colp + (int64_t)gy * Wq * row);
                    }
                    if (col_ok) {
#pragma unroll
                        for (int j = 0; j < CHUNK; ++j)
                            if (i0 + j < NSTAGE && (RPP == 1 ? i0 + j : my_row0 + (i0 + j) * RPP) < WH)
                                *reinterpret_cast<float4 *>(st_dst + (i0 + j) * RPP * WW * SLICE) = stage[j];
                    }
                    if (CHUNK < NSTAGE) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            GROUP_STAMP(2 + 2 * l);

            if (active) {
                if (AHEAD) load_cam(cam0, l);
#pragma unroll
                for (int c = 0; c < NGA; ++c) {
                    if (c >= ncam) continue;                  // (wave-uniform)
                    if (!AHEAD) load_cam(cam0 + c, l);
                    float4 la = na, lb = nb, wa = nw;
                    const float4 ra = nra, rb = nrb;
                    if (AHEAD && c + 1 < ncam) load_cam(cam0 + c + 1, l);
                    float xs[4], ys[4];
                    if constexpr (FUSED) {
                        // fold this level's logits into camera c's running softmax
                        const float mx = fmaxf(fmaxf(wa.x, wa.y), fmaxf(wa.z, wa.w));
                        // lazy online softmax (round 4): the reference maximum moves -- and the accumulators are rescaled --
                        // only when some lane's logits exceed it by more than 8: weights stay below e^8 of the reference,
                        // exact up to rounding, and the 16 multiplications per (camera, level) are spent once per camera
                        if (__builtin_amdgcn_ballot_w64(mx > smax[c] + 8.f) != 0) {
                            const float m = fmaxf(smax[c], mx);
                            const float sc = __expf(smax[c] - m);
                            ssum[c] *= sc;
                            smax[c] = m;
                            const float2v scv = {sc, sc};
#pragma unroll
                            for (int i = 0; i < 2 * NV; ++i) acc[c][i] *= scv;
                        }
                        {
                            const float m = smax[c];
                            wa = make_float4(__expf(wa.x - m), __expf(wa.y - m), __expf(wa.z - m), __expf(wa.w - m));
                            ssum[c] += (wa.x + wa.y) + (wa.z + wa.w);
                        }
                        // pixel coordinates: (ref + off / size) * size - 0.5
                        xs[0] = (ra.x + la.x * iw) * fW - 0.5f; ys[0] = (ra.y + la.y * ih) * fH - 0.5f;
                        xs[1] = (ra.z + la.z * iw) * fW - 0.5f; ys[1] = (ra.w + la.w * ih) * fH - 0.5f;
                        xs[2] = (rb.x + lb.x * iw) * fW - 0.5f; ys[2] = (rb.y + lb.y * ih) * fH - 0.5f;
                        xs[3] = (rb.z + lb.z * iw) * fW - 0.5f; ys[3] = (rb.w + lb.w * ih) * fH - 0.5f;
                    } else {
                        xs[0] = la.x * fW - 0.5f; ys[0] = la.y * fH - 0.5f;
                        xs[1] = la.z * fW - 0.5f; ys[1] = la.w * fH - 0.5f;
                        xs[2] = lb.x * fW - 0.5f; ys[2] = lb.y * fH - 0.5f;
                        xs[3] = lb.z * fW - 0.5f; ys[3] = lb.w * fH - 0.5f;
                    }
                    const float aws[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = xs[p], y = ys[p];
                        if (fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1)) {
                            const float fx = floorf(x), fy = floorf(y);
                            const int ix = (int)fx - ox, iy = (int)fy - oy;
                            const float wx1 = x - fx, wy1 = y - fy, a = aws[p];
                            const float ay1 = wy1 * a, ay0 = a - ay1;
                            const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                            const float *p00 = win + __mul24(iy * WW + ix, SLICE) + lane_off;
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                const float *pk = p00 + ((k ^ rot) << 2);
                                const float4 c00 = *reinterpret_cast<const float4 *>(pk);
                                const float4 c01 = *reinterpret_cast<const float4 *>(pk + SLICE);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], w00, c00);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], w01, c01);
                            }
                            __builtin_amdgcn_sched_barrier(0);    // at most 8 LDS reads in flight
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                const float *pk = p00 + WW * SLICE + ((k ^ rot) << 2);
                                const float4 c10 = *reinterpret_cast<const float4 *>(pk);
                                const float4 c11 = *reinterpret_cast<const float4 *>(pk + SLICE);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], w10, c10);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], w11, c11);
                            }
                        } else {
                            miss[c] |= (MissT)1 << (l * P + p);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            GROUP_STAMP(3 + 2 * l);
        }

        if (active) {
#pragma unroll
            for (int c = 0; c < NGA; ++c) {
                if (c >= ncam) continue;
                const int64_t cq = cam_q(cam0 + c);
                const float *lp = off + cq * lay.q_l + lane_l, *wp = logit + cq * lay.q_w + lane_w;
                const float *rp = FUSED ? refb + lsi[cam0 + c] * lay.r_q + lane_r : nullptr;
                MissT mm = miss[c];
                // taps that left the window: straight from global memory (zero padding by test)
                while (mm) {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int l = bit / P, pp = bit - l * P;
                    const float fW = (float)Wq, fH = (float)Hq;
                    float lx = lp[l * lay.l_l + pp * 2 + 0], ly = lp[l * lay.l_l + pp * 2 + 1], a = wp[l * lay.l_w + pp];
                    if constexpr (FUSED) {
                        const int ri = l * lay.r_l + (FUSED == 2 ? 0 : pp * 2);
                        lx = rp[ri + 0] + lx * (1.f / fW);
                        ly = rp[ri + 1] + ly * (1.f / fH);
                        a = __expf(a - smax[c]);
                    }
                    const float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
                    if (!(y > -1.f && x > -1.f && y < fH && x < fW)) continue;
                    const Footprint<float> f = footprint(y, x, Hq, Wq);
                    const float *r0 = vbatch + lsi[l] * row + lane_off + ((int64_t)f.y0 * Wq + f.x0) * row;
                    const float *r1 = r0 + (int64_t)Wq * row;
                    const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a;
                    const float w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        const int ko = (k ^ rot) << 2;
                        const float4 c00 = load4_or_zero(r0 + ko, f.vy0 && f.vx0, vbatch);
                        const float4 c01 = load4_or_zero(r0 + row + ko, f.vy0 && f.vx1, vbatch);
                        const float4 c10 = load4_or_zero(r1 + ko, f.vy1 && f.vx0, vbatch);
                        const float4 c11 = load4_or_zero(r1 + row + ko, f.vy1 && f.vx1, vbatch);
                        gfma4(acc[c][2 * k], acc[c][2 * k + 1], w00, c00);
                        gfma4(acc[c][2 * k], acc[c][2 * k + 1], w01, c01);
                        gfma4(acc[c][2 * k], acc[c][2 * k + 1], w10, c10);
                        gfma4(acc[c][2 * k], acc[c][2 * k + 1], w11, c11);
                    }
                }
                const float inv = FUSED ? 1.f / ssum[c] : 1.f;
                // training (mvdetr_msda_forward_fused_train_f32): the softmax statistics of (query, head) -- the running maximum
                // the weights were formed against and the reciprocal of their sum -- for the fused backward (round 6: a
                // separate pass over the logits, msda_softmax_stats, took 0.35 of this kernel's 1.1 ms at 16 cameras)
                if (FUSED && stats && ch_off == 0)
                    *reinterpret_cast<float2 *>(stats + ((cq + cell) * M + head) * 2) = make_float2(smax[c], inv);
                float *o = out + cq * row + (cell * (unsigned)row + (unsigned)(head * D + ch_off));
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    *reinterpret_cast<float4 *>(o + ((k ^ rot) << 2)) =
                        make_float4(acc[c][2 * k].x * inv, acc[c][2 * k].y * inv, acc[c][2 * k + 1].x * inv,
                                    acc[c][2 * k + 1].y * inv);
            }
        }
        GROUP_STAMP(2 + 2 * TILE_MAX_LEVELS);
    }
}

// 6-row tiles: 60 = 10 x 6 and 80 rows = 13.3 -> 14; (6+12) x 28 tokens x 128 B = 64.5 KB -> 2 workgroups / CU;
// 480 jobs for 512 resident workgroups at Wildtrack size (8-row tiles would give 384)
// Wide: 128-byte slices, 6-row tiles: (6+12) x 28 tokens x 128 B = 64.5 KB -> 2 workgroups / CU; 480 jobs for
// 512 resident workgroups at Wildtrack size (8-row tiles would give 384).  192 compute lanes; all 256 move
// window columns.
using GWide16 = TileCfg<16, 32, 6, 16, 6, 256>;
using GWide32 = TileCfg<32, 32, 6, 16, 6, 256>;
// many cameras: 4 lane groups of 3 waves on one window, 4 cameras each -- 768 threads, one workgroup per CU
#ifndef MVDETR_QUAD_TH
#define MVDETR_QUAD_TH 6
#endif
using GQuad16 = TileCfg<16, 32, MVDETR_QUAD_TH, 16, 6, MVDETR_QUAD_TH * 16 * 2 * 4>;
using GQuad32 = TileCfg<32, 32, MVDETR_QUAD_TH, 16, 6, MVDETR_QUAD_TH * 16 * 2 * 4>;

template <typename Cfg, int NG, int WAVES, int FUSED, int SPLIT = 1, bool DMA = false>
static int launch_group(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                        const float *off, const float *logit, const float *ref, int64_t ref_bstride,
                        SamplingLayout lay, int B, int S, int M, float *out, const int *local_hits, int opts, float *stats = nullptr)
{
    // dynamic LDS: the larger of this kernel's window and the fallback body's
    constexpr int LDS = Cfg::LDS_BYTES > TileCfg<Cfg::D, 32, 8, 16, 6>::LDS_BYTES ? Cfg::LDS_BYTES
                                                                                   : TileCfg<Cfg::D, 32, 8, 16, 6>::LDS_BYTES;
    auto kernel = &msda_fwd_group<Cfg, NG, WAVES, FUSED, SPLIT, DMA>;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_group<Cfg, NG, WAVES, FUSED, SPLIT, DMA>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_fwd_group<Cfg, NG, WAVES, FUSED, SPLIT, DMA>, Cfg::THREADS,
                                                         LDS) != hipSuccess || per_cu < 1)
            per_cu = 2;
        return (cus * per_cu + 7) / 8 * 8;
    });
    static const KernelResources res = kernel_resources(reinterpret_cast<const void *>(&msda_fwd_group<Cfg, NG, WAVES, FUSED, SPLIT, DMA>));
    msda_note_forward_kernel(DMA ? (SPLIT > 1 ? "msda_fwd_group[camera-split, LDS-DMA windows]" : "msda_fwd_group[LDS-DMA windows]")
                                 : (SPLIT > 1 ? "msda_fwd_group[camera-split]" : "msda_fwd_group"), &res);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(Cfg::THREADS), LDS, st, value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, out, local_hits, opts, stats);
    return (int)hipGetLastError();
}

}  // namespace mvdetr
