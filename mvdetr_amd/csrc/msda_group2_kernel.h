// Camera-grouped fused MSDA forward, second generation: the tap stream as an explicit software pipeline.
//
// Same decomposition as msda_fwd_group (msda_group_kernel.h): a workgroup owns a (6 x 16 tile, 128-byte slice of the token
// row), stages one source level's window in LDS per iteration and walks all NG query levels (cameras) over it; a lane is
// (cell, half slice) with NG accumulator sets.  What round 4's phase stamps and the ISA showed about that kernel: 102 of
// its 128 us are the tap phases, and under its register pressure (256 VGPRs, spilling) hipcc issues the 16 ds_read_b128 of a
// tap two at a time, each pair behind its own wait -- eight LDS round trips in a row per tap, 1.5 compute waves per SIMD to
// overlap them.  This kernel
//   * keeps DEPTH pairs of LDS reads in flight at all times: the reads of a level -- NG cameras x 4 points x 8 (corner row,
//     chunk) pairs -- form one stream, pair i + DEPTH is issued before pair i's four FMAs, across tap and camera
//     boundaries.  The order is pinned in the source: empty asm statements take a pair's registers as operands (the pair
//     has landed before its first FMA) and scheduling barriers keep every issue where it was written;
//   * is branch-free inside the stream: a tap whose footprint leaves the window reads a zero pad in LDS with zero weights
//     (and is finished from global memory after the last level, as before), so there is no exec-mask code between taps;
//   * pays for the DEPTH x 8 extra registers by moving state out of the register file: the online-softmax state (running
//     maximum, running sum) of the NG cameras and the miss masks live in LDS (lane-private slots, read-modify-write once per
//     (camera, level)); sampling data is fetched with buffer loads -- wave-uniform base and offset in scalar registers, ONE
//     32-bit per-lane offset per tensor -- instead of 64-bit per-lane pointers and 64-bit address arithmetic per load; the
//     window copy's per-lane constants are re-derived from the thread index in every level instead of living through the
//     tap loop;
//   * rescales the accumulators lazily (only when a running maximum moves by more than 8) and requests the next level's
//     first camera before the barriers and the window copy.
//
// LDS per workgroup: window 64,512 B + two 256-byte zero pads one window row apart + softmax state NG x 192 x 8 B + miss
// masks NG x 192 x 4 B = 81,408 B at NG = 7: two workgroups per CU, as before.
//
// Levels of unequal shape: the same launch runs the tile kernel's body (msda_tile_body.h), as in msda_fwd_group.
#pragma once
#include "msda_group_kernel.h"

#ifndef MVDETR_GROUP2_TAIL_FLAT
#define MVDETR_GROUP2_TAIL_FLAT 1     // far taps per round of the tail (2 was measured: the 32 gathers in flight spill INTO the tap
                                      // stream, 111 vs 99 us at the headline); 0: round 4's far-tap tail (per (camera, level): reload the level's sampling data, then gather)
#endif

namespace mvdetr {

template <typename Cfg, int NG> struct Group2Lds {
    static constexpr int NCL = Cfg::TH * Cfg::TW * 2;                    // compute lanes
    static constexpr int WIN = Cfg::WH * Cfg::WW * Cfg::SLICE;           // floats
    static constexpr int ROWF = Cfg::WW * Cfg::SLICE;                    // one window row
    static constexpr int ZPAD = 2 * Cfg::SLICE;                          // two tokens of zeros
    static constexpr int ZA = WIN, ZB = WIN + ROWF;                      // a missed tap's "row 0" and "row 1"
    static constexpr int GAP = ROWF - ZPAD;                              // floats between the pads
    static constexpr int ST_ROW = NCL * 2;                               // (max, sum) per lane, one row per camera
    static constexpr int ST_IN_GAP = GAP / ST_ROW;                       // state rows that fit between the pads
    static constexpr int AFTER = ZB + ZPAD;
    static constexpr int st_row(int c) { return c < ST_IN_GAP ? ZA + ZPAD + c * ST_ROW : AFTER + (c - ST_IN_GAP) * ST_ROW; }
    static constexpr int MS = AFTER + (NG > ST_IN_GAP ? NG - ST_IN_GAP : 0) * ST_ROW;      // miss masks [level][lane]
    static constexpr int FLOATS = MS + NG * NCL;
    static constexpr int BYTES = FLOATS * 4;
};

// (the whole vector is bit-cast, then taken apart: __builtin_bit_cast(float, v.y) on an element of the builtin's result read
// element 0 for every component with this hipcc -- the load shrank to one dword)
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    const float4v f = __builtin_bit_cast(float4v, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    const float2v f = __builtin_bit_cast(float2v, v);
    return make_float2(f.x, f.y);
}

template <typename Cfg, int NG, int FUSED, int DEPTH>
__global__ __launch_bounds__(Cfg::THREADS, 2) void msda_fwd_group2(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ off, const float *__restrict__ logit,
    const float *__restrict__ ref, int64_t ref_bstride, SamplingLayout lay, int B, int S, int M,
    float *__restrict__ out, const int *__restrict__ local_hits, int opts, float *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    GROUP_STAMP(0);
    using Lds = Group2Lds<Cfg, NG>;
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
    constexpr int RPP = Cfg::ROWS_PER_PASS, NCL = Lds::NCL;
    static_assert(NV == 4 && P == 4, "the pair stream is written for 4 chunks per lane and 4 points");
    const int tid = threadIdx.x;
    const int HS = M * D / SLICE;
    const int row = M * D;

    bool eq = true;
    for (int l = 1; l < L; ++l) eq = eq && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (!eq) {                                                // not ours (see header)
        using Fallback = TileCfg<Cfg::D, 32, 8, 16, 6, Cfg::THREADS>;
        msda_fwd_tile_body<Fallback, FUSED>(win, value, shapes, lsi, off, logit, ref, ref_bstride, lay,
                                            QueryLevels{0, L, S}, B, S, M, L, out);
        // (the training entry needs equal level shapes -- its caller checks; statistics this path cannot give are NaN)
        if (stats)
            for (int64_t i = (int64_t)blockIdx.x * Cfg::THREADS + threadIdx.x; i < (int64_t)B * S * M * 2; i += (int64_t)gridDim.x * Cfg::THREADS)
                stats[i] = __builtin_nanf("");
        return;
    }
    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const float fW = (float)Wq, fH = (float)Hq;
    const float iw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / fW)));
    const float ih = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / fH)));
    const int tcols = (Wq + TW - 1) / TW, trows = (Hq + TH - 1) / TH, per_level = trows * tcols;
    const int jobs = per_level * HS * B, jobs8 = (jobs + 7) / 8;

    const int sub = tid & 1, qi = tid >> 1;
    const int qly = qi / TW, qlx = qi % TW;
    const int rot = (qlx / Cfg::TOK_PER_BANKROW) & (NV - 1);
    const int lane_off = sub * LCH;
    // per-lane LDS slots: softmax state (float2 per camera) and miss masks (one word per level)
    float2 *const st_lane = reinterpret_cast<float2 *>(win) + (tid < NCL ? tid : 0);
    unsigned *const ms_lane = reinterpret_cast<unsigned *>(win + Lds::MS) + (tid < NCL ? tid : 0);
    if (tid < 2 * Lds::ZPAD) win[(tid < Lds::ZPAD ? Lds::ZA : Lds::ZB - Lds::ZPAD) + tid] = 0.f;   // (visible after the first barrier)

    // XCD job map: see msda_fwd_group
    int gy = 1, gx = 8;
    if (opts & GROUP_OPT_BLOCKS) {
        int best = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cy = 1 << e, cx = 8 >> e;
            const int bh = (trows + cy - 1) / cy, bw = (tcols + cx - 1) / cx;
            const int cost = bh * bw * 64 + bh + bw;
            if (cost < best) { best = cost; gy = cy; gx = cx; }
        }
    }
    const int bh = (trows + gy - 1) / gy, bw = (tcols + gx - 1) / gx;
    const int per_xcd = (opts & GROUP_OPT_BLOCKS) ? bh * bw * HS * B : jobs8;

    for (int t = blockIdx.x; t < per_xcd * 8; t += gridDim.x) {
        const int k8 = t & 7, idx = t >> 3;
        if (idx >= per_xcd) continue;
        int hs, b, ty, tx;
        if (opts & GROUP_OPT_BLOCKS) {
            hs = idx % HS;
            const int r = idx / HS, tib = r % (bh * bw);
            b = r / (bh * bw);
            ty = (k8 / gx) * bh + tib / bw;
            tx = (k8 % gx) * bw + tib % bw;
            if (ty >= trows || tx >= tcols) continue;
        } else {
            const int job = k8 * jobs8 + idx;
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
        const bool active = qi < TH * TW && qy < Hq && qx < Wq;
        // sampling data: buffer loads, voffset = one 32-bit per-lane byte offset per tensor, soffset = the camera / level part
        const unsigned cell = active ? (unsigned)(qy * Wq + qx) : 0u;
        const unsigned vo_l = (cell * (unsigned)lay.q_l + (unsigned)lay.head_l(head)) * 4u;
        const unsigned vo_w = (cell * (unsigned)lay.q_w + (unsigned)lay.head_w(head)) * 4u;
        const unsigned vo_r = cell * (unsigned)lay.r_q * 4u;
        const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(off), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(logit), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(FUSED ? ref + b * ref_bstride : value), 0, 0x7fffffff, 0x00020000);
        auto cam_q = [&](int c) { return (int64_t)b * S + lsi[c]; };          // wave-uniform
        const float *vbatch = value + (int64_t)b * S * row + hs * SLICE;

        float4 na = make_float4(0, 0, 0, 0), nb = na, nw = na, nr0 = na, nr1 = na;
        auto load_cam = [&](int c, int l) {
            const int64_t cq = cam_q(c);
            const unsigned so_l = (unsigned)(cq * lay.q_l + l * lay.l_l) * 4u, so_w = (unsigned)(cq * lay.q_w + l * lay.l_w) * 4u;
            na = buf_load4(rs_l, vo_l, so_l);
            nb = buf_load4(rs_l, vo_l + 16u, so_l);
            nw = buf_load4(rs_w, vo_w, so_w);
            if constexpr (FUSED == 1) {
                const unsigned so_r = (unsigned)((cq - (int64_t)b * S) * lay.r_q + l * lay.r_l) * 4u;
                nr0 = buf_load4(rs_r, vo_r, so_r);
                nr1 = buf_load4(rs_r, vo_r + 16u, so_r);
            } else if constexpr (FUSED == 2) {
                const unsigned so_r = (unsigned)((cq - (int64_t)b * S) * lay.r_q + l * lay.r_l) * 4u;
                const float2 r = buf_load2(rs_r, vo_r, so_r);
                nr0 = make_float4(r.x, r.y, r.x, r.y);
            }
        };

        // window shift: see msda_fwd_group
        int shift_x = 0, shift_y = 0;
        {
            const int sl = tid & 63, s_sub = sl & 1, s_qi = sl >> 1;
            const int s_qy = Y0 + s_qi / TW, s_qx = X0 + s_qi % TW;
            const int s_head = (hs * SLICE + s_sub * LCH) / D;
            float sx = 0.f, sy = 0.f, sn = 0.f, sloc = 0.f;
            if (s_qy < Hq && s_qx < Wq) {
                const int64_t s_cell = (int64_t)s_qy * Wq + s_qx, cq = cam_q(0);
                const float *lp = off + (cq + s_cell) * lay.q_l + lay.head_l(s_head);
                const float4 a0 = *reinterpret_cast<const float4 *>(lp), b0 = *reinterpret_cast<const float4 *>(lp + 4);
                float mx = 0.25f * ((a0.x + a0.z) + (b0.x + b0.z)), my = 0.25f * ((a0.y + a0.w) + (b0.y + b0.w));
                if constexpr (FUSED) {
                    const float *rp = ref + b * ref_bstride + (cq - (int64_t)b * S + s_cell) * lay.r_q;
                    const float rx = FUSED == 2 ? rp[0] : 0.25f * ((rp[0] + rp[2]) + (rp[4] + rp[6]));
                    const float ry = FUSED == 2 ? rp[1] : 0.25f * ((rp[1] + rp[3]) + (rp[5] + rp[7]));
                    mx += rx * fW - 0.5f - (float)s_qx;
                    my += ry * fH - 0.5f - (float)s_qy;
                } else {
                    mx = mx * fW - 0.5f - (float)s_qx;
                    my = my * fH - 0.5f - (float)s_qy;
                    // how many of this lane's four taps lie within MSDA_PROBE_RADIUS pixels of its own cell (the stand-down
                    // test below; the comparison is false for NaN)
                    const float ox_ = (float)s_qx + 0.5f, oy_ = (float)s_qy + 0.5f, rr = MSDA_PROBE_RADIUS;
                    sloc = (float)((fabsf(a0.x * fW - ox_) < rr && fabsf(a0.y * fH - oy_) < rr) + (fabsf(a0.z * fW - ox_) < rr && fabsf(a0.w * fH - oy_) < rr) +
                                   (fabsf(b0.x * fW - ox_) < rr && fabsf(b0.y * fH - oy_) < rr) + (fabsf(b0.z * fW - ox_) < rr && fabsf(b0.w * fH - oy_) < rr));
                }
                if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
            }
            float scnt = (s_qy < Hq && s_qx < Wq) ? 4.f : 0.f;                 // taps sampled by this lane
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sx += __shfl_xor(sx, o, 64);
                sy += __shfl_xor(sy, o, 64);
                sn += __shfl_xor(sn, o, 64);
                if constexpr (FUSED == 0) {
                    sloc += __shfl_xor(sloc, o, 64);
                    scnt += __shfl_xor(scnt, o, 64);
                }
            }
            if constexpr (FUSED == 0) {
                // Public contract, `auto`: the sample also says whether this tile's taps are near their cells at all.  If
                // fewer than a QUARTER are, windows would be wasted on it (half, the first version's test, sent jobs to the gather
                // body at a spread of 4 px where the windows still win: 749 vs 533 us for the whole call; a third does the same at 6 px:
                // 833 vs 715; with a quarter the worst point of the sweep is 9 px, 908 us against the gather kernel's 722): the job computes its outputs in the gather formulation
                // instead (every wave sees the same sample and takes the same way; this replaces round 3's separate probe
                // kernel, its scratch allocation and two launch gaps in front of every forward).
                const float tl = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sloc)));
                const float tc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scnt)));
                if ((opts & GROUP_OPT_STANDDOWN) && tl * 4.f < tc) {
                    // items of the job: (camera, cell of the tile, 16-byte chunk of the slice); GU per lane and iteration,
                    // branch-free, so that their loads are in flight together
                    constexpr int CH = SLICE / 4, NIT = NG * TH * TW * CH, T = Cfg::THREADS, GU = 4;
                    for (int it0 = tid; it0 < NIT; it0 += GU * T) {
                        const float *lp[GU], *wp[GU], *vb[GU];
                        float *op[GU];
                        bool live[GU];
                        float4 r[GU];
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            const int it = it0 + u * T;
                            const int ck = it % CH, ci = (it / CH) % (TH * TW), c = min(it / (CH * TH * TW), NG - 1);
                            const int gy_ = Y0 + ci / TW, gx_ = X0 + ci % TW;
                            live[u] = it < NIT && gy_ < Hq && gx_ < Wq;
                            const int chn = hs * SLICE + ck * 4, hd = chn / D;
                            const int64_t q = cam_q(c) + (live[u] ? (int64_t)gy_ * Wq + gx_ : 0);
                            lp[u] = off + q * lay.q_l + lay.head_l(hd);
                            wp[u] = logit + q * lay.q_w + lay.head_w(hd);
                            vb[u] = value + (int64_t)b * S * row + chn;
                            op[u] = out + q * row + chn;
                            r[u] = make_float4(0, 0, 0, 0);
                        }
                        for (int l = 0; l < L; ++l) {
#pragma unroll
                            for (int p = 0; p < P; ++p) {
#pragma unroll
                                for (int u = 0; u < GU; ++u) {
                                    float x = lp[u][l * lay.l_l + p * 2] * fW - 0.5f, y = lp[u][l * lay.l_l + p * 2 + 1] * fH - 0.5f;
                                    float a = wp[u][l * lay.l_w + p];
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
            }
            if (active) load_cam(0, 0);
            const float tx_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
            const float ty_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            if (tn > 0.f && !(opts & GROUP_OPT_NO_SHIFT)) {
                shift_x = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(tx_ / tn)));
                shift_y = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(ty_ / tn)));
            }
        }
        GROUP_STAMP(1);

        float2v acc[NG][2 * NV];
#pragma unroll
        for (int c = 0; c < NG; ++c) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) acc[c][i] = (float2v){0.f, 0.f};
            if (FUSED && tid < NCL) st_lane[Lds::st_row(c) / 2] = make_float2(-INFINITY, 0.f);
        }

        const int oy = Y0 + TH / 2 - WH / 2 + shift_y, ox = X0 + TW / 2 - WW / 2 + shift_x;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        // LDS byte addresses of this lane's half slice in token 0 of the window, of the zero pad, and of the lane's 4 chunks
        const float *const wbase = win + lane_off;
        const float *const zbase = win + Lds::ZA + lane_off;
        int choff[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) choff[k] = (k ^ rot) << 2;

        for (int l = 0; l < L; ++l) {
            // everyone is done reading the old window (an LDS-only barrier: the sampling data requested above stays in flight
            // and overlaps the window copy; __syncthreads() waited for it here, then for the copy)
            lds_barrier();
            {
                // LDS-DMA window copy (see msda_fwd_group); its per-lane constants are re-derived here from an opaque copy
                // of the thread index so that they do not stay in registers through the tap stream
                int t2 = tid;
                asm volatile("" : "+v"(t2));
                const int my_part = t2 % Cfg::PARTS, my_slot = t2 / Cfg::PARTS;
                const int my_row0 = my_slot / WW, my_col = my_slot % WW;
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(vbatch), 0, (int)((unsigned)S * row * 4u - (unsigned)(hs * SLICE) * 4u), 0x00020000);
                const int gxx = ox + my_col;
                const bool xok = (unsigned)gxx < (unsigned)Wq;
                const int wave_u = __builtin_amdgcn_readfirstlane(t2 >> 6);
                const unsigned so = (unsigned)((int)lsi[l] * row) * 4u;
                if (my_row0 < RPP) {
#pragma unroll
                    for (int i = 0; i < NSTAGE; ++i) {
                        const int wy = my_row0 + i * RPP, gyy = oy + wy;
                        if (wy < WH) {
                            const unsigned vo = (xok && (unsigned)gyy < (unsigned)Hq) ? (unsigned)((gyy * Wq + gxx) * row + my_part * 4) * 4u : 0x80000000u;
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(win + (i * RPP * WW + wave_u * 8) * SLICE),
                                                                     16, (int)vo, (int)so, 0, 0);
                        }
                        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();
            GROUP_STAMP(2 + 2 * l);

            if (active) {
                constexpr int NP = NG * P * 8;                // pairs of this level's stream
                constexpr int R = DEPTH + 1;
                float4v ring[R][2];
                float w[2][4];                                // weights of the tap being consumed / the tap being issued
                const float *p0 = wbase;
                float4 la = na, lb = nb, wa = nw, ra = nr0, rb = nr1;
                float aws[4] = {0.f, 0.f, 0.f, 0.f};
                unsigned mlevel = 0;
                auto consume = [&](int m) {
                    const int c = m / (P * 8), r = (m / 4) % 2, k = m % 4;
                    float4v &cl = ring[m % R][0], &cr = ring[m % R][1];
                    asm volatile("" : "+v"(cl), "+v"(cr));
                    const float *ww = w[(m / 8) & 1];
                    const float wl = ww[2 * r], wr = ww[2 * r + 1];
                    gfma4(acc[c][2 * k], acc[c][2 * k + 1], wl, make_float4(cl.x, cl.y, cl.z, cl.w));
                    gfma4(acc[c][2 * k], acc[c][2 * k + 1], wr, make_float4(cr.x, cr.y, cr.z, cr.w));
                    // (without this pin the optimiser sinks the FMAs of the whole level below the stream -- there is control
                    // flow between the cameras -- and keeps every loaded pair alive until then)
                    asm volatile("" : "+v"(acc[c][2 * k]), "+v"(acc[c][2 * k + 1]));
                };
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    // ---- the issue side enters camera c: its sampling data becomes current, the next camera's (or the
                    // next level's first) is requested, its logits fold into its running softmax
                    la = na; lb = nb; wa = nw; ra = nr0; rb = nr1;
                    if (c + 1 < NG) load_cam(c + 1, l);
                    else if (l + 1 < L) load_cam(0, l + 1);
                    if constexpr (FUSED) {
                        float2 s = st_lane[Lds::st_row(c) / 2];
                        const float mx = fmaxf(fmaxf(wa.x, wa.y), fmaxf(wa.z, wa.w));
                        if (__builtin_amdgcn_ballot_w64(mx > s.x + 8.f) != 0) {
                            const float m = fmaxf(s.x, mx);
                            const float sc = __expf(s.x - m);
                            s.y *= sc;
                            s.x = m;
                            const float2v scv = {sc, sc};
#pragma unroll
                            for (int j = 0; j < 2 * NV; ++j) acc[c][j] *= scv;
                        }
                        aws[0] = __expf(wa.x - s.x); aws[1] = __expf(wa.y - s.x);
                        aws[2] = __expf(wa.z - s.x); aws[3] = __expf(wa.w - s.x);
                        s.y += (aws[0] + aws[1]) + (aws[2] + aws[3]);
                        st_lane[Lds::st_row(c) / 2] = s;
                    } else {
                        aws[0] = wa.x; aws[1] = wa.y; aws[2] = wa.z; aws[3] = wa.w;
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        // ---- tap (c, p): position, window test, bilinear weights, LDS address (branch-free: a tap outside
                        // the window reads the zero pad with zero weights and is noted for the far path)
                        {
                            const float lx = p == 0 ? la.x : p == 1 ? la.z : p == 2 ? lb.x : lb.z;
                            const float ly = p == 0 ? la.y : p == 1 ? la.w : p == 2 ? lb.y : lb.w;
                            float x, y;
                            if constexpr (FUSED == 1) {
                                const float rx = p == 0 ? ra.x : p == 1 ? ra.z : p == 2 ? rb.x : rb.z;
                                const float ry = p == 0 ? ra.y : p == 1 ? ra.w : p == 2 ? rb.y : rb.w;
                                x = (rx + lx * iw) * fW - 0.5f;
                                y = (ry + ly * ih) * fH - 0.5f;
                            } else if constexpr (FUSED == 2) {
                                x = (ra.x + lx * iw) * fW - 0.5f;
                                y = (ra.y + ly * ih) * fH - 0.5f;
                            } else {
                                x = lx * fW - 0.5f;
                                y = ly * fH - 0.5f;
                            }
                            const bool in = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                            x = in ? x : cx;                  // (finite stand-ins: a NaN position must not reach the weights)
                            y = in ? y : cy;
                            const float a = in ? aws[p] : 0.f;
                            mlevel |= in ? 0u : (1u << (c * P + p));
                            const float fx = floorf(x), fy = floorf(y);
                            const int ix = (int)fx - ox, iy = (int)fy - oy;
                            const float wx1 = x - fx, wy1 = y - fy;
                            const float ay1 = wy1 * a, ay0 = a - ay1;
                            float *ww = w[(c * P + p) & 1];
                            ww[1] = ay0 * wx1; ww[0] = ay0 - ww[1]; ww[3] = ay1 * wx1; ww[2] = ay1 - ww[3];
                            const float *pw = wbase + __mul24(iy * WW + ix, SLICE);
                            p0 = in ? pw : zbase;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = (c * P + p) * 8 + j, r = j / 4, k = j % 4;
                            const float *pk = p0 + r * Lds::ROWF + choff[k];
                            ring[i % R][0] = *reinterpret_cast<const float4v *>(pk);
                            ring[i % R][1] = *reinterpret_cast<const float4v *>(pk + SLICE);
                            __builtin_amdgcn_sched_barrier(0);
                            if (i >= DEPTH) consume(i - DEPTH);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int m = NP - DEPTH; m < NP; ++m) consume(m);
                ms_lane[l * NCL] = mlevel;
            }
            GROUP_STAMP(3 + 2 * l);
        }

        if (active) {
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const int64_t cq = cam_q(c);
                const float *lp = off + cq * lay.q_l + (vo_l >> 2), *wp = logit + cq * lay.q_w + (vo_w >> 2);
                const float *rp = FUSED ? ref + b * ref_bstride + lsi[c] * lay.r_q + (vo_r >> 2) : nullptr;
                float2 s = make_float2(0.f, 1.f);
                if constexpr (FUSED) s = st_lane[Lds::st_row(c) / 2];
                // taps that left the window: straight from global memory (zero padding by test)
#if MVDETR_GROUP2_TAIL_FLAT
                // ONE list per (lane, camera) -- bit 4 l + p of `mc` = point p of level l -- walked FU entries at a time, with the
                // next entries' sampling data requested before the current ones' gathers: a round trip per FU missed taps of
                // the wave's worst lane.
                // (Before: per (camera, level) with any miss in the wave, the level's sampling data and then the gathers, two
                // dependent round trips each -- 2 x 49 per job once the offsets spread to 2 px.)
                unsigned mc = 0u;
#pragma unroll
                for (int l = 0; l < L; ++l) mc |= ((ms_lane[l * NCL] >> (c * P)) & 15u) << (4 * l);
                if (mc) {
                    constexpr int FU = MVDETR_GROUP2_TAIL_FLAT;           // entries per round
                    float2 nxy[FU], nrf[FU];
                    float nlg[FU];
                    int nl[FU];
                    bool nok[FU];
                    auto request = [&]() {                    // the list's first FU entries: sampling data on its way
#pragma unroll
                        for (int u = 0; u < FU; ++u) {
                            nok[u] = mc != 0u;
                            const int t_ = nok[u] ? __ffs((int)mc) - 1 : 0, l_ = t_ >> 2, pp_ = t_ & 3;
                            mc &= mc - 1u;                    // (0 stays 0)
                            nl[u] = l_;
                            nxy[u] = *reinterpret_cast<const float2 *>(lp + l_ * lay.l_l + pp_ * 2);
                            nlg[u] = wp[l_ * lay.l_w + pp_];
                            nrf[u] = make_float2(0.f, 0.f);
                            if constexpr (FUSED) nrf[u] = *reinterpret_cast<const float2 *>(rp + l_ * lay.r_l + (FUSED == 2 ? 0 : pp_ * 2));
                        }
                    };
                    request();
                    for (;;) {
                        const float *r0[FU], *r1[FU];
                        float wgt[FU][4];
                        bool v00[FU], v01[FU], v10[FU], v11[FU];
#pragma unroll
                        for (int u = 0; u < FU; ++u) {
                            float lx = nxy[u].x, ly = nxy[u].y, a = nlg[u];
                            if constexpr (FUSED) {
                                lx = nrf[u].x + lx * (1.f / fW);
                                ly = nrf[u].y + ly * (1.f / fH);
                                a = __expf(a - s.x);
                            }
                            float x = lx * fW - 0.5f, y = ly * fH - 0.5f;
                            const bool ok = nok[u] && y > -1.f && x > -1.f && y < fH && x < fW;        // (false for NaN)
                            x = ok ? x : 0.f;
                            y = ok ? y : 0.f;
                            a = ok ? a : 0.f;
                            const Footprint<float> f = footprint(y, x, Hq, Wq);
                            r0[u] = vbatch + lsi[nl[u]] * row + lane_off + ((int64_t)f.y0 * Wq + f.x0) * row;
                            r1[u] = r0[u] + (int64_t)Wq * row;
                            wgt[u][0] = f.wy0 * f.wx0 * a; wgt[u][1] = f.wy0 * f.wx1 * a;
                            wgt[u][2] = f.wy1 * f.wx0 * a; wgt[u][3] = f.wy1 * f.wx1 * a;
                            v00[u] = ok && f.vy0 && f.vx0; v01[u] = ok && f.vy0 && f.vx1;
                            v10[u] = ok && f.vy1 && f.vx0; v11[u] = ok && f.vy1 && f.vx1;
                        }
                        const bool more = mc != 0u;
                        if (more) request();
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            const int ko = (k ^ rot) << 2;
                            float4 c00[FU], c01[FU], c10[FU], c11[FU];
#pragma unroll
                            for (int u = 0; u < FU; ++u) {
                                c00[u] = load4_or_zero(r0[u] + ko, v00[u], vbatch);
                                c01[u] = load4_or_zero(r0[u] + row + ko, v01[u], vbatch);
                                c10[u] = load4_or_zero(r1[u] + ko, v10[u], vbatch);
                                c11[u] = load4_or_zero(r1[u] + row + ko, v11[u], vbatch);
                            }
#pragma unroll
                            for (int u = 0; u < FU; ++u) {
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], wgt[u][0], c00[u]);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], wgt[u][1], c01[u]);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], wgt[u][2], c10[u]);
                                gfma4(acc[c][2 * k], acc[c][2 * k + 1], wgt[u][3], c11[u]);
                            }
                        }
                        if (!more) break;
                    }
                }
#else
                for (int l = 0; l < L; ++l) {
                    unsigned mm = (ms_lane[l * NCL] >> (c * P)) & 15u;
                    if (!mm) continue;
                    // the (query, level)'s four points in ONE round trip (16-byte loads, as the stream took them), not one per
                    // missed tap in front of its gathers
                    const float4 la4 = *reinterpret_cast<const float4 *>(lp + l * lay.l_l), lb4 = *reinterpret_cast<const float4 *>(lp + l * lay.l_l + 4);
                    const float4 wa4 = *reinterpret_cast<const float4 *>(wp + l * lay.l_w);
                    while (mm) {
                        const int pp = __ffs((int)mm) - 1;
                        mm &= mm - 1;
                        float lx = pp == 0 ? la4.x : pp == 1 ? la4.z : pp == 2 ? lb4.x : lb4.z;
                        float ly = pp == 0 ? la4.y : pp == 1 ? la4.w : pp == 2 ? lb4.y : lb4.w;
                        float a = pp == 0 ? wa4.x : pp == 1 ? wa4.y : pp == 2 ? wa4.z : wa4.w;
                        if constexpr (FUSED) {
                            const int ri = l * lay.r_l + (FUSED == 2 ? 0 : pp * 2);
                            lx = rp[ri + 0] + lx * (1.f / fW);
                            ly = rp[ri + 1] + ly * (1.f / fH);
                            a = __expf(a - s.x);
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
                }
#endif
                const float inv = FUSED ? 1.f / s.y : 1.f;
                // training (mvdetr_msda_forward_fused_train_f32): the softmax statistics of (query, head) -- running maximum
                // and reciprocal sum -- for the fused backward, which recomputes the weights from the raw logits
                if (FUSED && stats && ch_off == 0)
                    *reinterpret_cast<float2 *>(stats + ((cq + cell) * M + head) * 2) = make_float2(s.x, inv);
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

template <typename Cfg, int NG, int FUSED, int DEPTH>
static int launch_group2(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *off, const float *logit, const float *ref, int64_t ref_bstride,
                         SamplingLayout lay, int B, int S, int M, float *out, const int *local_hits, int opts,
                         float *stats = nullptr)
{
    constexpr int FB = TileCfg<Cfg::D, 32, 8, 16, 6>::LDS_BYTES;
    constexpr int LDS = Group2Lds<Cfg, NG>::BYTES > FB ? Group2Lds<Cfg, NG>::BYTES : FB;
    auto kernel = &msda_fwd_group2<Cfg, NG, FUSED, DEPTH>;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_group2<Cfg, NG, FUSED, DEPTH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_fwd_group2<Cfg, NG, FUSED, DEPTH>, Cfg::THREADS,
                                                         LDS) != hipSuccess || per_cu < 1)
            per_cu = 2;
        return (cus * per_cu + 7) / 8 * 8;
    });
    static const KernelResources res = kernel_resources(reinterpret_cast<const void *>(&msda_fwd_group2<Cfg, NG, FUSED, DEPTH>));
    msda_note_forward_kernel("msda_fwd_group2[pipelined taps, LDS-DMA windows]", &res);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(Cfg::THREADS), LDS, st, value, shapes, lsi, off, logit, ref, ref_bstride, lay, B, S, M, out, local_hits, opts, stats);
    return (int)hipGetLastError();
}

}  // namespace mvdetr
