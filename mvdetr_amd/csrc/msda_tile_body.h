// Body of the LDS-tiled MSDA forward kernel (one (tile, slice, query level) per workgroup iteration) as a
// device function, so that both msda_fwd_tile (msda_forward_tile.hip) and the camera-grouped kernel's
// unequal-shapes fallback (msda_forward_group.hip) can run it.  Internal, not part of the C ABI.
#pragma once
#include "common.h"
#include "msda_tile.h"

namespace mvdetr {

typedef float float2v __attribute__((ext_vector_type(2)));

// acc += w * c for 4 consecutive channels, written on 2-wide vectors so it lowers to v_pk_fma_f32
__device__ __forceinline__ void fma4(float2v &lo, float2v &hi, float w, const float4 &c)
{
    const float2v ww = {w, w};
    lo = __builtin_elementwise_fma(ww, (float2v){c.x, c.y}, lo);
    hi = __builtin_elementwise_fma(ww, (float2v){c.z, c.w}, hi);
}

// Bank conflicts.  A ds_read_b128 is served in 16-lane groups over a 256-byte bank row; neighbouring
// queries read neighbouring tokens (128 B apart), so if every lane read chunk j of its head at step j
// the 8 queries x 2 half-slices of a group would pile onto 4 (D=16) or 2 (D=32) of the 16 slots (measured:
// 72 % of all LDS cycles were conflict cycles).  Instead lane (qx, head) reads chunk j ^ r at step j,
// r = (qx >> 1) mod chunks-per-head: same-parity neighbours then cover all slots of their head.  r is a
// per-lane constant, so accumulator j simply holds channels 4*(j^r) .. +3 for the whole kernel and only
// the final store (and the rare global-memory taps) need to know.  The LDS image itself stays linear.
// FUSED: `loc` holds the raw sampling offsets (output of the module's sampling_offsets Linear, pixels of
// the sampled level) and `aw` the raw attention logits; the kernel adds the reference points
// (`ref` [.., Lq, L, P, 2], batch stride `ref_bstride`, 0 = shared) and takes the softmax over L*P itself --
// the arithmetic of ms_deform_attn.py:100-107 -- so neither tensor is materialised.  With `level_major` the
// two raw tensors are laid out [B, Lq, L, M, P(, 2)] instead of the reference's [B, Lq, M, L, P(, 2)]: a free
// permutation of the Linear's weight rows on the module side, which puts what ONE level iteration of the
// heads of a tile needs into the same cache lines (the head-major layout spreads a line over 4 level
// iterations, between which it falls out of L2).
// FUSED: 0 = final locations / weights; 1 = raw offsets / logits + reference points [.., Lq, L, P, 2];
// 2 = the same with ONE reference point per (query, level), [.., Lq, L, 2] (MVDeTr's reference map repeats the
// point P times, mvdetr.py:49-58 with all heights 0): a quarter of the reference bytes and registers.
template <typename Cfg, int FUSED>
__device__ __forceinline__ void msda_fwd_tile_body(float *win,
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw,
    const float *__restrict__ ref, int64_t ref_bstride, SamplingLayout lay, QueryLevels qr, int B, int S,
    int M, int L, float *__restrict__ out)
{
    constexpr int D = Cfg::D, TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW;
    constexpr int SLICE = Cfg::SLICE, P = TILE_P, NV = Cfg::NV, NSTAGE = Cfg::NSTAGE, LCH = SLICE / 2;
    const int tid = threadIdx.x;
    const int HS = M * D / SLICE;                         // slices per token row
    const int row = M * D;                                // floats per value token

    // ---- the tile list: [level][tile-in-level] x head slice x batch ------------------------------
    // The queries are the tokens of levels [qr.begin, qr.end) -- all L levels for the plain encoder call, one
    // rank's cameras for the query-sharded encoder (mvdetr_amd/dist.py); `loc`, `aw`, `ref` and `out` hold
    // exactly those qr.Lq queries, in token order.  The value tokens are always those of all L levels.
    const int NQ = qr.end - qr.begin;
    const int64_t q_first = lsi[qr.begin];
    const int Lq = qr.Lq;
    int tiles_spatial = 0;
    bool equal_shapes = true;
    for (int l = 0; l < L; ++l) {
        if (l >= qr.begin && l < qr.end) tiles_spatial += tiles_of_level<Cfg>(shapes, l);
        equal_shapes = equal_shapes && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    }
    const int per_level = equal_shapes ? tiles_spatial / NQ : 0;
    const int units = per_level * HS * B;                 // (tile, slice, batch) units, equal shapes only
    const int units8 = (units + 7) / 8;                   // units per XCD
    // equal shapes: t enumerates xcd x (unit of that xcd) x query level, see the decode below
    const int total = equal_shapes ? units8 * 8 * NQ : tiles_spatial * HS * B;

    const int sub = tid & 1;                              // which half of the slice this lane owns
    const int qi = tid >> 1;
    const int qly = qi / TW, qlx = qi % TW;
    // chunk rotation: lanes whose tokens share a position inside the 256-byte bank row differ in r
    const int rot = (qlx / Cfg::TOK_PER_BANKROW) & (NV - 1);
    const int lane_off = sub * LCH;                       // floats: this lane's channels inside the slice

    // window copy: thread moves float4 `my_part` of window column `my_col`, rows my_row0 + i*ROWS_PER_PASS
    const int my_part = tid % Cfg::PARTS, my_slot = tid / Cfg::PARTS;
    const int my_row0 = my_slot / WW, my_col = my_slot % WW;
    const bool slot_ok = my_row0 < Cfg::ROWS_PER_PASS;
    float *const st_dst = win + (my_row0 * WW + my_col) * SLICE + my_part * 4;

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        // ---- decode t (wave-uniform, scalar) -----------------------------------------------------
        int lq, tin, hs, b;
        if (equal_shapes) {
            // workgroups t, t+8, t+16, ... share an XCD (and its L2).  Give XCD k a contiguous band of
            // units, and run the L query levels of one unit back to back: their source windows are
            // identical, so all but the first find them in that L2.
            const int xcd = t & 7, r = t >> 3;
            lq = qr.begin + r % NQ;
            const int unit = xcd * units8 + r / NQ;
            if (r / NQ >= units8 || unit >= units) continue;
            hs = unit % HS;
            const int u2 = unit / HS;
            tin = u2 % per_level;
            b = u2 / per_level;
        } else {
            hs = t % HS;
            const int u2 = t / HS;
            b = u2 / tiles_spatial;
            int rem = u2 % tiles_spatial;
            lq = qr.begin;
            for (;; ++lq) {
                const int n = tiles_of_level<Cfg>(shapes, lq);
                if (rem < n) break;
                rem -= n;
            }
            tin = rem;
        }
        const int Hq = (int)shapes[2 * lq], Wq = (int)shapes[2 * lq + 1];
        const int tcols = (Wq + TW - 1) / TW;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int ch0 = hs * SLICE + lane_off;              // this lane's first channel in the token row
        const int head = ch0 / D, ch_off = ch0 % D;

        const int qy = Y0 + qly, qx = X0 + qlx;
        // query index: token index minus the first token of the query levels (guarded against a caller whose
        // Lq is smaller than the levels it names)
        const int64_t q = lsi[lq] - q_first + (int64_t)qy * Wq + qx;
        const bool active = qi < TH * TW && qy < Hq && qx < Wq && q < Lq;      // (surplus lanes only copy)
        const int64_t bqm = active ? (((int64_t)b * Lq + q) * M + head) : 0;
        // sampling data of this (query, head): element (query, head, level) of `loc` starts at
        // query * lay.q_l + head * lay.h_l + level * lay.l_l floats (`aw`: the *_w strides).  The reference
        // layout [.., Lq, M, L, P(, 2)] and the level-major / column-block layouts of the fused path are all
        // instances of it; the host fills the strides in.
        const int lstep_l = lay.l_l, lstep_w = lay.l_w;
        const int64_t bq = active ? (int64_t)b * Lq + q : 0;
        const float *lp = loc + bq * lay.q_l + lay.head_l(head);
        const float *wp = aw + bq * lay.q_w + lay.head_w(head);
        const int rstep = lay.r_l;                            // floats of reference points from level to level
        const float *rp = FUSED ? ref + b * ref_bstride + (active ? q : 0) * lay.r_q : nullptr;
        const float *vbatch = value + (int64_t)b * S * row + hs * SLICE;   // this slice of token 0

        float2v acc[2 * NV];
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) acc[i] = (float2v){0.f, 0.f};
        unsigned long long miss = 0ull;

        // Where the tile's taps lie (round 3, as in msda_forward_group.hip): MSDeformAttn's offset bias is a ray per head, so
        // the far points of a ray leave a window centred on the tile as soon as the learned part adds a pixel or two, and each
        // such tap is a serialised global gather.  Every wave looks at the SAME sample -- the tile's first 32 cells x the
        // slice's two halves, source level 0 -- and arrives at the same shift (in pixels of the query level, at most +-3)
        // without exchanging anything.  Any shift gives the same results; it only decides how many taps miss the window.
        int shift_x = 0, shift_y = 0;
        {
            const int sl = tid & 63, s_sub = sl & 1, s_qi = sl >> 1;
            const int s_qy = Y0 + s_qi / TW, s_qx = X0 + s_qi % TW;
            const int s_head = (hs * SLICE + s_sub * LCH) / D;
            const int64_t s_q = lsi[lq] - q_first + (int64_t)s_qy * Wq + s_qx;
            const float W0 = (float)shapes[1], H0 = (float)shapes[0];
            float sx = 0.f, sy = 0.f, sn = 0.f;
            if (s_qi < TH * TW && s_qy < Hq && s_qx < Wq && s_q < Lq) {
                const float *slp = loc + ((int64_t)b * Lq + s_q) * lay.q_l + lay.head_l(s_head);
                const float4 a0 = *reinterpret_cast<const float4 *>(slp), b0 = *reinterpret_cast<const float4 *>(slp + 4);
                float mx = 0.25f * ((a0.x + a0.z) + (b0.x + b0.z)), my = 0.25f * ((a0.y + a0.w) + (b0.y + b0.w));
                if constexpr (FUSED) {
                    const float *srp = ref + b * ref_bstride + s_q * lay.r_q;
                    const float rx = FUSED == 2 ? srp[0] : 0.25f * ((srp[0] + srp[2]) + (srp[4] + srp[6]));
                    const float ry = FUSED == 2 ? srp[1] : 0.25f * ((srp[1] + srp[3]) + (srp[5] + srp[7]));
                    mx += rx * W0 - 0.5f;                     // raw offsets are pixels of the sampled level
                    my += ry * H0 - 0.5f;
                } else {
                    mx = mx * W0 - 0.5f;
                    my = my * H0 - 0.5f;
                }
                // displacement from the cell's own position carried to level 0, in pixels of the query level
                mx = (mx - (((float)s_qx + 0.5f) * W0 / (float)Wq - 0.5f)) * (float)Wq / W0;
                my = (my - (((float)s_qy + 0.5f) * H0 / (float)Hq - 0.5f)) * (float)Hq / H0;
                if (mx == mx && my == my && fabsf(mx) < 64.f && fabsf(my) < 64.f) { sx = mx; sy = my; sn = 1.f; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sx += __shfl_xor(sx, o, 64);
                sy += __shfl_xor(sy, o, 64);
                sn += __shfl_xor(sn, o, 64);
            }
            const float tx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
            const float ty = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sy)));
            const float tn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            if (tn > 0.f) {
                shift_x = max(-3, min(3, (int)rintf(tx / tn)));
                shift_y = max(-3, min(3, (int)rintf(ty / tn)));
            }
        }
        // window geometry of a level: origin = the tile's centre (+ the shift) carried to that level (integers)
        auto origin = [&](int l, int &oy, int &ox, int &H, int &W) {
            H = (int)shapes[2 * l];
            W = (int)shapes[2 * l + 1];
            oy = (2 * (Y0 + shift_y) + TH) * H / (2 * Hq) - WH / 2;
            ox = (2 * (X0 + shift_x) + TW) * W / (2 * Wq) - WW / 2;
        };
        // issue this thread's share of a window copy into registers (loads stay in flight): its
        // column, every ROWS_PER_PASS-th row
        float4 stage[NSTAGE];
        auto fetch_window = [&](int l) {
            int oy, ox, H, W;
            origin(l, oy, ox, H, W);
            const int gx = ox + my_col;
            const bool xok = slot_ok && (unsigned)gx < (unsigned)W;
            const float *colp = vbatch + lsi[l] * row + my_part * 4 + (xok ? gx : 0) * row;
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                // one row per pass (128-byte slices): the row index is a compile-time constant, so the row
                // offset and its bounds test are scalar (SALU) instead of 64-bit vector multiplies per load
                const int wy = Cfg::ROWS_PER_PASS == 1 ? i : my_row0 + i * Cfg::ROWS_PER_PASS;
                const int gy = oy + wy;
                stage[i] = make_float4(0, 0, 0, 0);
                if (xok && wy < WH && (unsigned)gy < (unsigned)H)
                    stage[i] = *reinterpret_cast<const float4 *>(colp + (int64_t)gy * W * row);
            }
        };

        // ---- locality probe: do this tile's taps stay near their own cell? ------------------------
        // Sampling data of the first level doubles as the probe.  If fewer than a quarter of the
        // workgroup's lanes have most of that level's taps inside the window, staging would be
        // wasted: the tile is then done entirely by the direct path below (all bits set in `miss`).
        // sampling data of one level: unfused = final (x,y) pairs and weights; fused = raw offsets, logits
        // and the reference points, turned into locations / weights where they are used
        float4 la = make_float4(0, 0, 0, 0), lb = la, wa = la, ra = la, rb = la;
        auto load_level = [&](int l, float4 &a, float4 &b2, float4 &w, float4 &r0, float4 &r1) {
            a = *reinterpret_cast<const float4 *>(lp + l * lstep_l);
            b2 = *reinterpret_cast<const float4 *>(lp + l * lstep_l + 4);
            w = *reinterpret_cast<const float4 *>(wp + l * lstep_w);
            if constexpr (FUSED == 1) {
                r0 = *reinterpret_cast<const float4 *>(rp + l * rstep);
                r1 = *reinterpret_cast<const float4 *>(rp + l * rstep + 4);
            } else if constexpr (FUSED == 2) {
                const float2 r = *reinterpret_cast<const float2 *>(rp + l * rstep);
                r0 = r1 = make_float4(r.x, r.y, r.x, r.y);
            }
        };
        // FUSED: online softmax over the L*P logits of this (query, head) -- running maximum `smax` and
        // running sum `ssum` of exp(logit - smax); the accumulators are rescaled when the maximum moves and
        // divided by the final sum at the end, so the logits are read exactly once, level by level.
        float smax = -INFINITY, ssum = 0.f;
        // raw -> (x, y) in [0,1] for level dimensions (W, H); logits are left raw here
        auto finish = [&](float4 &a, float4 &b2, const float4 &r0, const float4 &r1, float W, float H) {
            if constexpr (FUSED) {
                const float iw = 1.f / W, ih = 1.f / H;
                a = make_float4(r0.x + a.x * iw, r0.y + a.y * ih, r0.z + a.z * iw, r0.w + a.w * ih);
                b2 = make_float4(r1.x + b2.x * iw, r1.y + b2.y * ih, r1.z + b2.z * iw, r1.w + b2.w * ih);
            }
        };
        if (active) load_level(0, la, lb, wa, ra, rb);
        int hits = 0;
        {
            int oy, ox, H, W;
            origin(0, oy, ox, H, W);
            const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
            float4 pa = la, pb = lb;
            finish(pa, pb, ra, rb, (float)W, (float)H);
            const float xs[4] = {pa.x, pa.z, pb.x, pb.z}, ys[4] = {pa.y, pa.w, pb.y, pb.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float x = xs[p] * (float)W - 0.5f, y = ys[p] * (float)H - 0.5f;
                hits += (active && fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1)) ? 1 : 0;
            }
        }
        const int live = __syncthreads_count(active);
        const int tile_hits = __syncthreads_count(hits >= 2);
        const bool staged = 4 * tile_hits >= live;

        if (staged) {
            fetch_window(0);
            for (int l = 0; l < L; ++l) {
                int oy, ox, H, W;
                origin(l, oy, ox, H, W);
                __syncthreads();                          // everyone is done reading the old window
                if (slot_ok) {
#pragma unroll
                    for (int i = 0; i < NSTAGE; ++i)
                        if (my_row0 + i * Cfg::ROWS_PER_PASS < WH)
                            *reinterpret_cast<float4 *>(st_dst + i * Cfg::ROWS_PER_PASS * WW * SLICE) = stage[i];
                }
                // next level: window copy and sampling data go in flight under this level's taps
                float4 na = la, nb = lb, nw = wa, nra = ra, nrb = rb;
                if (l + 1 < L) {
                    fetch_window(l + 1);
                    if (active) load_level(l + 1, na, nb, nw, nra, nrb);
                }
                __syncthreads();

                if (active) {
                    finish(la, lb, ra, rb, (float)W, (float)H);
                    if constexpr (FUSED) {
                        // fold this level's logits into the running softmax; rescale what was accumulated
                        const float m = fmaxf(smax, fmaxf(fmaxf(wa.x, wa.y), fmaxf(wa.z, wa.w)));
                        const float sc = __expf(smax - m);               // exp(-inf) = 0 on the first level
                        wa = make_float4(__expf(wa.x - m), __expf(wa.y - m), __expf(wa.z - m), __expf(wa.w - m));
                        ssum = ssum * sc + (wa.x + wa.y) + (wa.z + wa.w);
                        smax = m;
                        const float2v scv = {sc, sc};
#pragma unroll
                        for (int i = 0; i < 2 * NV; ++i) acc[i] *= scv;
                    }
                    const float lxs[4] = {la.x, la.z, lb.x, lb.z};
                    const float lys[4] = {la.y, la.w, lb.y, lb.w};
                    const float aws[4] = {wa.x, wa.y, wa.z, wa.w};
                    const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);
                    const float fW = (float)W, fH = (float)H;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = lxs[p] * fW - 0.5f;
                        const float y = lys[p] * fH - 0.5f;
                        // footprint [floor, floor+1] inside the window <=> |x - centre| < (WW-1)/2
                        // (false for NaN/inf; the window border itself counts as outside)
                        if (fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1)) {
                            const float fx = floorf(x), fy = floorf(y);
                            const int ix = (int)fx - ox, iy = (int)fy - oy;
                            const float wx1 = x - fx, wy1 = y - fy;
                            const float a = aws[p];
                            const float ay1 = wy1 * a, ay0 = a - ay1;                 // (1 - wy1) * a
                            const float w01 = ay0 * wx1, w00 = ay0 - w01;
                            const float w11 = ay1 * wx1, w10 = ay1 - w11;
                            const float *p00 = win + __mul24(iy * WW + ix, SLICE) + lane_off;
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                const float *pk = p00 + ((k ^ rot) << 2);
                                const float4 c00 = *reinterpret_cast<const float4 *>(pk);
                                const float4 c01 = *reinterpret_cast<const float4 *>(pk + SLICE);
                                const float4 c10 = *reinterpret_cast<const float4 *>(pk + WW * SLICE);
                                const float4 c11 = *reinterpret_cast<const float4 *>(pk + WW * SLICE + SLICE);
                                fma4(acc[2 * k], acc[2 * k + 1], w00, c00);
                                fma4(acc[2 * k], acc[2 * k + 1], w01, c01);
                                fma4(acc[2 * k], acc[2 * k + 1], w10, c10);
                                fma4(acc[2 * k], acc[2 * k + 1], w11, c11);
                            }
                        } else {
                            miss |= 1ull << (l * P + p);
                        }
                        // keep the taps apart: hoisting all 4 x 16 LDS reads together costs > 256 VGPRs
                        if (Cfg::TAP_FENCE) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                la = na;
                lb = nb;
                wa = nw;
                ra = nra;
                rb = nrb;
            }
        } else {
            miss = L * P >= 64 ? ~0ull : ((1ull << (L * P)) - 1);
            if constexpr (FUSED) {
                if (active) {                             // plain two-pass statistics for the direct path
                    for (int l = 0; l < L; ++l) {
                        const float4 w = *reinterpret_cast<const float4 *>(wp + l * lstep_w);
                        smax = fmaxf(smax, fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
                    }
                    for (int l = 0; l < L; ++l) {
                        const float4 w = *reinterpret_cast<const float4 *>(wp + l * lstep_w);
                        ssum += __expf(w.x - smax) + __expf(w.y - smax) + __expf(w.z - smax) + __expf(w.w - smax);
                    }
                }
            }
        }

        if (active) {
            // ---- taps that left the window: straight from global memory (zero padding by test) ----
            while (miss) {
                const int bit = __ffsll((long long)miss) - 1;
                miss &= miss - 1;
                const int l = bit / P;
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const int pp = bit - l * P;
                float lx = lp[l * lstep_l + pp * 2 + 0], ly = lp[l * lstep_l + pp * 2 + 1], a = wp[l * lstep_w + pp];
                if constexpr (FUSED) {
                    const int ri = l * rstep + (FUSED == 2 ? 0 : pp * 2);
                    lx = rp[ri + 0] + lx * (1.f / (float)W);
                    ly = rp[ri + 1] + ly * (1.f / (float)H);
                    a = __expf(a - smax);                  // un-normalised, like the accumulators
                }
                const float x = lx * (float)W - 0.5f;
                const float y = ly * (float)H - 0.5f;
                if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
                const Footprint<float> f = footprint(y, x, H, W);
                const float *r0 = vbatch + lsi[l] * row + lane_off + ((int64_t)f.y0 * W + f.x0) * row;
                const float *r1 = r0 + (int64_t)W * row;
                const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a;
                const float w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int ko = (k ^ rot) << 2;
                    const float4 c00 = load4_or_zero(r0 + ko, f.vy0 && f.vx0, vbatch);
                    const float4 c01 = load4_or_zero(r0 + row + ko, f.vy0 && f.vx1, vbatch);
                    const float4 c10 = load4_or_zero(r1 + ko, f.vy1 && f.vx0, vbatch);
                    const float4 c11 = load4_or_zero(r1 + row + ko, f.vy1 && f.vx1, vbatch);
                    fma4(acc[2 * k], acc[2 * k + 1], w00, c00);
                    fma4(acc[2 * k], acc[2 * k + 1], w01, c01);
                    fma4(acc[2 * k], acc[2 * k + 1], w10, c10);
                    fma4(acc[2 * k], acc[2 * k + 1], w11, c11);
                }
            }
            if constexpr (FUSED) {
                const float inv = 1.f / ssum;
                const float2v invv = {inv, inv};
#pragma unroll
                for (int i = 0; i < 2 * NV; ++i) acc[i] *= invv;
            }
            float *o = out + bqm * D + ch_off;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                *reinterpret_cast<float4 *>(o + ((k ^ rot) << 2)) =
                    make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y);
        }
    }
}

}  // namespace mvdetr
