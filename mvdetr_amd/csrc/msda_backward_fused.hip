// Fused MSDeformAttn training backward, the sampling half: gradients of the RAW offsets / logits -- gfx950 (MI355X).
//
// Replaces, for deformable-encoder calls of MVDeTr's shape (6 / 7 equal levels, 16-channel heads, 4 points), what autograd
// runs behind MSDeformAttn.forward in the reference: the col2im kernels' grad_sampling_loc / grad_attn_weight
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, 301-920 via func.py:30-38) AND the torch
// backward of the module arithmetic around them (ms_deform_attn.py:100-107: softmax over the L*P logits, locations =
// reference + offsets / (W, H)).  Input is what the fused forward read -- ONE raw tensor [Lq, L, M/g, (g*P*2 offsets |
// g*P logits)] from the module's single GEMM -- plus the forward's softmax statistics; output is the gradient of that raw
// tensor in the same layout.  Neither sampling_locations (135 MB at Wildtrack size) nor attention_weights (68 MB) nor
// their gradients ever exist.
//
// Per tap (query q, head m, level l, point p) with bilinear weights on the corners v00..v11 of the value map, g = grad_out
// [q, m, :], a = softmax weight, and the dots d_k = <g, v_k>:
//     d a      = (1-wy)((1-wx) d00 + wx d01) + wy((1-wx) d10 + wx d11)           (cuh:155: grad_attn_weight)
//     d x_px   = a ((1-wy)(d01 - d00) + wy (d11 - d10)),  d y_px likewise        (cuh:156-158 without the W, H factors:
//                                                                                  loc = ref + off / (W, H), so d off = d px)
//     d logit  = a (d a - D),  D = sum over all taps of the (q, m) of a * d a = <g, out[q, m, :]>   (softmax backward; the
//                                                                                  forward's output row gives D directly)
//
// Structure: msda_fwd_group2's (msda_group2_kernel.h) -- a workgroup owns a (6 x 16 tile, 128-byte slice), stages one source
// level's window in LDS per iteration (LDS-DMA), walks all NG cameras' queries over it with the tap reads as a software
// pipeline (DEPTH pairs of ds_read_b128 in flight, order pinned in the source) -- with the accumulators replaced by the
// current camera's grad_out row (16 registers, double-buffered and re-read per level: keeping all NG rows spilled) and four
// dot products per tap instead of four weighted sums.  State per (lane, camera) in LDS: softmax maximum and reciprocal sum (from the forward), D.  Results go out per (camera,
// level) -- four taps, 32 + 16 bytes per lane -- to where the inputs came from: a (query, level)'s runs of the four slices are three whole 128-byte
// lines, written by four jobs that run at the same time on one XCD.  Taps whose footprint leaves the window read zeros in
// the stream and are redone from global memory after the level's stream (rare; their stores overwrite the stream's).
//
// grad_value comes from msda_bwd_value_win<D, FUSED = 1> (msda_backward_tile.hip), which reads the same raw tensor.
#include "msda_group2_kernel.h"
#include "../../include/mvdetr_ops.h"

namespace mvdetr {

__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a, float b)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a, float b, float c, float d)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), r, (int)voff, (int)soff, 0);
}

template <typename Cfg, int NG, int DEPTH>
__global__ __launch_bounds__(Cfg::THREADS, 2) void msda_bwd_fused_sampling(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ raw, int raw_q, const float *__restrict__ ref,
    int64_t ref_bstride, const float *__restrict__ stats, const float *__restrict__ out_fwd, int B, int S, int M,
    float *__restrict__ grad_raw, int opts)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    using Lds = Group2Lds<Cfg, NG>;
    constexpr int D = Cfg::D, TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW;
    constexpr int SLICE = Cfg::SLICE, P = TILE_P, NV = Cfg::NV, NSTAGE = Cfg::NSTAGE, LCH = SLICE / 2, L = NG;
    constexpr int RPP = Cfg::ROWS_PER_PASS, NCL = Lds::NCL;
    static_assert(NV == 4 && P == 4 && D == 16, "16-channel heads: a lane holds a whole head's grad_out row");
    constexpr int HPS = 32 / D, CHUNK = HPS * P * 3;          // floats of a (query, level, slice) run: offsets | logits
    const int tid = threadIdx.x;
    const int HS = M * D / SLICE;
    const int row = M * D;
    const int l_stride = HS * CHUNK;                          // floats between a query's levels

    bool eq = true;
    for (int l = 1; l < L; ++l) eq = eq && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (!eq) {
        // not this kernel's shapes (its callers check): make the misuse loud
        for (int64_t i = (int64_t)blockIdx.x * Cfg::THREADS + tid; i < (int64_t)B * S * raw_q; i += (int64_t)gridDim.x * Cfg::THREADS)
            grad_raw[i] = __builtin_nanf("");
        return;
    }
    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const float fW = (float)Wq, fH = (float)Hq;
    const int tcols = (Wq + TW - 1) / TW, trows = (Hq + TH - 1) / TH, per_level = trows * tcols;
    const int jobs = per_level * HS * B, jobs8 = (jobs + 7) / 8;
    const int Lq = S;                                         // queries = tokens

    const int sub = tid & 1, qi = tid >> 1;
    const int qly = qi / TW, qlx = qi % TW;
    const int rot = (qlx / Cfg::TOK_PER_BANKROW) & (NV - 1);
    const int lane_off = sub * LCH;
    float2 *const st_lane = reinterpret_cast<float2 *>(win) + (tid < NCL ? tid : 0);          // (max, 1 / sum) per camera
    float *const dq_lane = win + Lds::MS + (tid < NCL ? tid : 0);                              // D per camera
    if (tid < 2 * Lds::ZPAD) win[(tid < Lds::ZPAD ? Lds::ZA : Lds::ZB - Lds::ZPAD) + tid] = 0.f;

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
        const int head = hs * HPS + sub;                      // (D == 16: a lane is a whole head)
        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qi < TH * TW && qy < Hq && qx < Wq;
        const unsigned cell = active ? (unsigned)(qy * Wq + qx) : 0u;
        // raw tensor: offsets of (query, level, head) at q * raw_q + l * l_stride + hs * CHUNK + sub * 8, logits at
        // ... + HPS * 8 + sub * 4; one 32-bit per-lane byte offset, the (camera, level) part is a scalar offset
        const unsigned vo_l = (cell * (unsigned)raw_q + (unsigned)(hs * CHUNK + sub * P * 2)) * 4u;
        const unsigned vo_w = (cell * (unsigned)raw_q + (unsigned)(hs * CHUNK + HPS * P * 2 + sub * P)) * 4u;
        const unsigned vo_r = cell * 8u;                      // reference points [L, Lq, 2]
        const float *const rawb = raw + (int64_t)b * S * raw_q;
        const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rawb), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ref + b * ref_bstride), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(grad_raw + (int64_t)b * S * raw_q, 0, 0x7fffffff, 0x00020000);
        const float *vbatch = value + (int64_t)b * S * row + hs * SLICE;
        auto cam_q = [&](int c) { return (int)lsi[c]; };     // first query of camera c inside the batch element (uniform)

        float4 na = make_float4(0, 0, 0, 0), nb = na, nw = na;
        float2 nr = make_float2(0.f, 0.f);
        auto load_cam = [&](int c, int l) {
            const unsigned so = (unsigned)(cam_q(c) * raw_q + l * l_stride) * 4u;
            na = buf_load4(rs_raw, vo_l, so);
            nb = buf_load4(rs_raw, vo_l + 16u, so);
            nw = buf_load4(rs_raw, vo_w, so);
            nr = buf_load2(rs_ref, vo_r, (unsigned)((l * Lq + cam_q(c)) * 2) * 4u);
        };

        // window shift from the tile's own taps: see msda_fwd_group
        int shift_x = 0, shift_y = 0;
        {
            const int sl = tid & 63, s_sub = sl & 1, s_qi = sl >> 1;
            const int s_qy = Y0 + s_qi / TW, s_qx = X0 + s_qi % TW;
            float sx = 0.f, sy = 0.f, sn = 0.f;
            if (s_qy < Hq && s_qx < Wq) {
                const int64_t s_q = (int64_t)cam_q(0) + (int64_t)s_qy * Wq + s_qx;
                const float *lp = rawb + s_q * raw_q + hs * CHUNK + s_sub * P * 2;
                const float4 a0 = *reinterpret_cast<const float4 *>(lp), b0 = *reinterpret_cast<const float4 *>(lp + 4);
                const float *rp = ref + b * ref_bstride + s_q * 2;
                const float mx = 0.25f * ((a0.x + a0.z) + (b0.x + b0.z)) + rp[0] * fW - 0.5f - (float)s_qx;
                const float my = 0.25f * ((a0.y + a0.w) + (b0.y + b0.w)) + rp[1] * fH - 0.5f - (float)s_qy;
                if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
            }
            if (active) load_cam(0, 0);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sx += __shfl_xor(sx, o, 64);
                sy += __shfl_xor(sy, o, 64);
                sn += __shfl_xor(sn, o, 64);
            }
            const float tx_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
            const float ty_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            if (tn > 0.f && !(opts & GROUP_OPT_NO_SHIFT)) {
                shift_x = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(tx_ / tn)));
                shift_y = max(-MVDETR_SHIFT_MAX, min(MVDETR_SHIFT_MAX, (int)rintf(ty_ / tn)));
            }
        }

        // per camera: the softmax statistics and D = <grad_out, out> of this (cell, head).  The grad_out rows themselves are
        // NOT kept across the levels (7 x 16 registers: the first version of this kernel spilled 92 dwords per lane) -- a camera's
        // row is re-read per level, one camera ahead of its taps (g_buf below): 4 more 16-byte loads per 32-tap stream.
        const unsigned vo_g = (cell * (unsigned)row + (unsigned)(head * D)) * 4u;      // this lane's row inside a camera's block
        const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(go + (int64_t)b * S * row), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            const int64_t q = (int64_t)b * S + cam_q(c) + cell;
            const float *gp = go + q * row + head * D, *op = out_fwd + q * row + head * D;
            float dq = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                float4 gv = make_float4(0, 0, 0, 0), ov = gv;
                if (active) {
                    gv = *reinterpret_cast<const float4 *>(gp + (k << 2));
                    ov = *reinterpret_cast<const float4 *>(op + (k << 2));
                }
                dq += (gv.x * ov.x + gv.y * ov.y) + (gv.z * ov.z + gv.w * ov.w);
            }
            if (tid < NCL) {
                const float2 s = active ? *reinterpret_cast<const float2 *>(stats + (q * M + head) * 2) : make_float2(0.f, 0.f);
                st_lane[Lds::st_row(c) / 2] = s;
                dq_lane[c * NCL] = dq;
            }
        }
        // grad_out rows, double-buffered by camera parity, in the lane's own read order: its k-th LDS read of a corner is
        // chunk k ^ rot (inactive lanes read cell 0's row and their results are never stored)
        float2v g_buf[2][2 * NV];
        auto load_g = [&](int c) {
            const unsigned so = (unsigned)(cam_q(c) * row) * 4u;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 gv = buf_load4(rs_go, vo_g + (unsigned)((k ^ rot) << 4), so);
                g_buf[c & 1][2 * k] = (float2v){gv.x, gv.y};
                g_buf[c & 1][2 * k + 1] = (float2v){gv.z, gv.w};
            }
        };

        const int oy = Y0 + TH / 2 - WH / 2 + shift_y, ox = X0 + TW / 2 - WW / 2 + shift_x;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
        const float *const wbase = win + lane_off;
        const float *const zbase = win + Lds::ZA + lane_off;
        int choff[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) choff[k] = (k ^ rot) << 2;

        for (int l = 0; l < L; ++l) {
            lds_barrier();                                    // (LDS only: see msda_fwd_group2)
            {
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

            unsigned mlevel = 0;
            if (active) {
                constexpr int NP = NG * P * 8;
                constexpr int R = DEPTH + 1;
                float4v ring[R][2];
                float tw[2][3];                               // (wx, wy, a) of the tap being consumed / the tap being issued
                float2v dl[2], dr[2];                         // dots with the left / right corner, rows 0 and 1
                const float *p0 = wbase;
                float4 la = na, lb = nb, wa = nw;
                float2 ra = nr;
                float aws[4] = {0.f, 0.f, 0.f, 0.f};
                float dqs[2] = {0.f, 0.f};                    // D of the camera being consumed / issued
                float gxy[2 * P], glg[P];                     // gradients of the (camera, level) being consumed
                auto consume = [&](int m) {
                    const int c = m / (P * 8), p = (m / 8) % P, r = (m / 4) % 2, k = m % 4;
                    float4v &cl = ring[m % R][0], &cr = ring[m % R][1];
                    asm volatile("" : "+v"(cl), "+v"(cr));
                    if (m % 8 == 0) dl[0] = dl[1] = dr[0] = dr[1] = (float2v){0.f, 0.f};
                    dl[r] = __builtin_elementwise_fma(g_buf[c & 1][2 * k], (float2v){cl.x, cl.y}, dl[r]);
                    dl[r] = __builtin_elementwise_fma(g_buf[c & 1][2 * k + 1], (float2v){cl.z, cl.w}, dl[r]);
                    dr[r] = __builtin_elementwise_fma(g_buf[c & 1][2 * k], (float2v){cr.x, cr.y}, dr[r]);
                    dr[r] = __builtin_elementwise_fma(g_buf[c & 1][2 * k + 1], (float2v){cr.z, cr.w}, dr[r]);
                    asm volatile("" : "+v"(dl[r]), "+v"(dr[r]));
                    if (m % 8 == 7) {
                        // ---- the tap's four dots are complete: its three gradients go out
                        const float *w = tw[(m / 8) & 1];
                        const float wx = w[0], wy = w[1], a = w[2];
                        const float d00 = dl[0].x + dl[0].y, d01 = dr[0].x + dr[0].y, d10 = dl[1].x + dl[1].y, d11 = dr[1].x + dr[1].y;
                        const float top = d00 + wx * (d01 - d00), bot = d10 + wx * (d11 - d10);
                        const float da = top + wy * (bot - top);
                        const float dx = (d01 - d00) + wy * ((d11 - d10) - (d01 - d00));
                        const float dy = (d10 - d00) + wx * ((d11 - d01) - (d10 - d00));
                        // a (camera, level)'s four taps leave together: 32 + 16 contiguous bytes per lane, three 16-byte stores
                        // instead of eight 8- / 4-byte ones (the store path is bound by requests, not bytes)
                        gxy[2 * p] = a * dx;
                        gxy[2 * p + 1] = a * dy;
                        glg[p] = a * (da - dqs[c & 1]);
                        if (p == P - 1) {
                            const unsigned so = (unsigned)(cam_q(c) * raw_q + l * l_stride) * 4u;
                            buf_store4(rs_out, vo_l, so, gxy[0], gxy[1], gxy[2], gxy[3]);
                            buf_store4(rs_out, vo_l + 16u, so, gxy[4], gxy[5], gxy[6], gxy[7]);
                            buf_store4(rs_out, vo_w, so, glg[0], glg[1], glg[2], glg[3]);
                        }
                    }
                };
                load_g(0);
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    la = na; lb = nb; wa = nw; ra = nr;
                    if (c + 1 < NG) load_cam(c + 1, l);
                    else if (l + 1 < L) load_cam(0, l + 1);
                    {
                        const float2 s = st_lane[Lds::st_row(c) / 2];
                        dqs[c & 1] = dq_lane[c * NCL];
                        aws[0] = __expf(wa.x - s.x) * s.y; aws[1] = __expf(wa.y - s.x) * s.y;
                        aws[2] = __expf(wa.z - s.x) * s.y; aws[3] = __expf(wa.w - s.x) * s.y;
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        {
                            const float lx = p == 0 ? la.x : p == 1 ? la.z : p == 2 ? lb.x : lb.z;
                            const float ly = p == 0 ? la.y : p == 1 ? la.w : p == 2 ? lb.y : lb.w;
                            // (position in two parts, common.h: the corner weights keep 5e-7 px where one fp32 number has 8e-6)
                            float x, y, fx, fy, wx1, wy1;
                            fused_px(ra.x, lx, fW, x, fx, wx1);
                            fused_px(ra.y, ly, fH, y, fy, wy1);
                            const bool in = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                            fx = in ? fx : floorf(cx);       // (finite stand-ins: a NaN position must not reach the addresses)
                            fy = in ? fy : floorf(cy);
                            mlevel |= in ? 0u : (1u << (c * P + p));
                            const int ix = (int)fx - ox, iy = (int)fy - oy;
                            float *w = tw[(c * P + p) & 1];
                            w[0] = in ? wx1 : 0.f; w[1] = in ? wy1 : 0.f; w[2] = in ? aws[p] : 0.f;
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
                            // (the previous camera's last DEPTH pairs have been consumed: its buffer is free for the next one)
                            if (p == 0 && j == DEPTH && c + 1 < NG) load_g(c + 1);
                        }
                    }
                }
#pragma unroll
                for (int m = NP - DEPTH; m < NP; ++m) consume(m);
            }

            // ---- taps of this level whose footprint left the window (rare): the same gradients from global memory; their
            // stores come after the stream's (which wrote zeros for them) in this lane's program order
            // ONE list per lane -- bit c P + p of `mlevel` -- walked with the NEXT entry's sampling data (raw offsets, logit,
            // reference point, the camera's first token) requested before the current entry's gathers: a round trip per far tap
            // of the wave's worst lane.  (Before: per camera with a miss anywhere in the wave, its sampling data and then the
            // gathers -- two dependent round trips each, seven cameras per level.)
            if (mlevel) {
                unsigned ml = mlevel;
                float2 nof = make_float2(0.f, 0.f), nrf = nof;
                float nlg = 0.f;
                int nt = 0, ncq = 0;
                auto request = [&]() {
                    nt = __ffs((int)ml) - 1;
                    ml &= ml - 1u;
                    const int c_ = nt >> 2, pp_ = nt & 3;
                    ncq = (int)lsi[c_];
                    const int64_t q_ = (int64_t)ncq + cell;                       // inside the batch element
                    nof = *reinterpret_cast<const float2 *>(rawb + q_ * raw_q + l * l_stride + hs * CHUNK + sub * P * 2 + pp_ * 2);
                    nlg = rawb[q_ * raw_q + l * l_stride + hs * CHUNK + HPS * P * 2 + sub * P + pp_];
                    nrf = *reinterpret_cast<const float2 *>(ref + b * ref_bstride + ((int64_t)l * Lq + q_) * 2);
                };
                request();
                for (;;) {
                    const float2 of_ = nof, rf_ = nrf;
                    const float lg_ = nlg;
                    const int c = nt >> 2, pp = nt & 3, cq = ncq;
                    const bool more = ml != 0u;
                    if (more) request();
                    const float2 s = st_lane[Lds::st_row(c) / 2];
                    const float dq = dq_lane[c * NCL];
                    const int64_t q = (int64_t)cq + cell;
                    float x, y, flx, fly, frx, fry;
                    fused_px(rf_.x, of_.x, fW, x, flx, frx);
                    fused_px(rf_.y, of_.y, fH, y, fly, fry);
                    const float a = __expf(lg_ - s.x) * s.y;
                    float dx = 0.f, dy = 0.f, da = 0.f;
                    if (y > -1.f && x > -1.f && y < fH && x < fW) {
                        const Footprint<float> f = footprint_split(fly, fry, flx, frx, Hq, Wq);
                        const float *r0 = vbatch + lsi[l] * row + lane_off + ((int64_t)f.y0 * Wq + f.x0) * row;
                        const float *r1 = r0 + (int64_t)Wq * row;
                        float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            const int ko = (k ^ rot) << 2;
                            const float4 c00 = load4_or_zero(r0 + ko, f.vy0 && f.vx0, vbatch), c01 = load4_or_zero(r0 + row + ko, f.vy0 && f.vx1, vbatch);
                            const float4 c10 = load4_or_zero(r1 + ko, f.vy1 && f.vx0, vbatch), c11 = load4_or_zero(r1 + row + ko, f.vy1 && f.vx1, vbatch);
                            // (rare path: the camera's grad_out row comes from memory again, g_buf has moved on)
                            const float4 gv = *reinterpret_cast<const float4 *>(go + ((int64_t)b * S + q) * row + head * D + ko);
                            const float2v ga = {gv.x, gv.y}, gb = {gv.z, gv.w};
                            d00 += (ga.x * c00.x + ga.y * c00.y) + (gb.x * c00.z + gb.y * c00.w);
                            d01 += (ga.x * c01.x + ga.y * c01.y) + (gb.x * c01.z + gb.y * c01.w);
                            d10 += (ga.x * c10.x + ga.y * c10.y) + (gb.x * c10.z + gb.y * c10.w);
                            d11 += (ga.x * c11.x + ga.y * c11.y) + (gb.x * c11.z + gb.y * c11.w);
                        }
                        const float wx = f.wx1, wy = f.wy1;
                        const float top = d00 + wx * (d01 - d00), bot = d10 + wx * (d11 - d10);
                        da = top + wy * (bot - top);
                        dx = (d01 - d00) + wy * ((d11 - d10) - (d01 - d00));
                        dy = (d10 - d00) + wx * ((d11 - d01) - (d10 - d00));
                    }
                    // (the camera differs from lane to lane: its part of the address goes into the per-lane offset)
                    const unsigned so_lane = (unsigned)(cq * raw_q + l * l_stride) * 4u;
                    buf_store2(rs_out, vo_l + pp * 8u + so_lane, 0u, a * dx, a * dy);
                    buf_store1(rs_out, vo_w + pp * 4u + so_lane, 0u, a * (da - dq));
                    if (!more) break;
                }
            }
        }
    }
}

template <typename Cfg, int NG>
static int launch_bwd_fused_sampling(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                     const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                     const float *stats, const float *out_fwd, int B, int S, int M, float *grad_raw, int opts)
{
#ifndef MVDETR_FS_DEPTH
#define MVDETR_FS_DEPTH 2
#endif
    constexpr int DEPTH = MVDETR_FS_DEPTH;
    constexpr int LDS = Group2Lds<Cfg, NG>::BYTES;
    auto kernel = &msda_bwd_fused_sampling<Cfg, NG, DEPTH>;
    static PerDevice<int> blocks_of;
    const int blocks = blocks_of.get([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_fused_sampling<Cfg, NG, DEPTH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_fused_sampling<Cfg, NG, DEPTH>, Cfg::THREADS, LDS) != hipSuccess || per_cu < 1)
            per_cu = 2;
        return (cus * per_cu + 7) / 8 * 8;
    });
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(Cfg::THREADS), LDS, st, go, value, shapes, lsi, raw, raw_q, ref,
                       ref_bstride, stats, out_fwd, B, S, M, grad_raw, opts);
    return (int)hipGetLastError();
}

int msda_backward_fused_sampling(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                                 const int64_t *lsi, const float *raw, int raw_q, const float *ref, int64_t ref_bstride,
                                 const float *stats, const float *out_fwd, int B, int S, int M, int D, int L, float *grad_raw)
{
    const int opts = GROUP_OPT_BLOCKS;
    if (D == 16 && L == 7)
        return launch_bwd_fused_sampling<GWide16, 7>(st, go, value, shapes, lsi, raw, raw_q, ref, ref_bstride, stats, out_fwd, B, S, M, grad_raw, opts);
    if (D == 16 && L == 6)
        return launch_bwd_fused_sampling<GWide16, 6>(st, go, value, shapes, lsi, raw, raw_q, ref, ref_bstride, stats, out_fwd, B, S, M, grad_raw, opts);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
