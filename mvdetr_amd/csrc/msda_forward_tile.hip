// Multi-scale deformable attention forward, LDS-tiled encoder kernel -- gfx950 (MI355X).
//
// Shape of the problem (MVDeTr's shadow transformer, and any deformable *encoder* self-attention):
// the queries ARE the value tokens (Lq == S), a query's reference point is its own cell, and the
// learned offsets are a few pixels.  So the 8x16 block of neighbouring queries of one level samples,
// in every source level, a small window around the same normalised position.  The gather kernel
// pulls every one of the 4 corners x L*P taps x 64-byte head segments through the texture-address
// path (4.3 GB of L2->L1 traffic at Wildtrack size, 67 M cache-line requests); here a workgroup
//
//   1. owns a TH x TW tile of query cells of one level and a 128-byte slice of the token row
//      (two 16-channel heads, or one 32-channel head),
//   2. for each source level copies the (TH+2R) x (TW+2R) window of that slice into LDS with
//      full-line coalesced loads (zero-filled outside the level, so zero padding costs nothing),
//   3. lets each lane -- one (query, head) pair, all D channels in registers -- take its P taps of
//      that level from LDS with ds_read_b128,
//   4. remembers (bit mask) the rare taps whose footprint falls outside the window and finishes
//      them afterwards straight from global memory.
//
// Correct for ANY sampling locations (step 4); fast when they are local.  Shapes and level
// offsets are read on the device, so the host never needs them: the grid is persistent and each
// workgroup strides over the tile list it derives from spatial_shapes.
//
// Replaces (together with msda_forward.hip) ms_deformable_im2col_gpu_kernel of the reference
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299).
#include "common.h"
#include "msda_dispatch.h"

namespace mvdetr {

constexpr int TILE_MAX_LEVELS = 16;     // 64-bit miss mask = L * P bits with P == 4
constexpr int TILE_P = 4;

template <int D, int TH_, int TW_, int R_> struct TileCfg {
    static constexpr int TH = TH_, TW = TW_, R = R_;
    static constexpr int MH = 32 / D;                 // heads per 128-byte slice
    static constexpr int SLICE = 32;                  // floats per token in LDS
    static constexpr int WH = TH + 2 * R, WW = TW + 2 * R;
    static constexpr int THREADS = TH * TW * MH;
    static constexpr int LDS_BYTES = WH * WW * SLICE * 4;
};

// Tile count of one level, recomputed by every workgroup from the device-side shapes (uniform ->
// scalar registers).
template <typename Cfg>
__device__ __forceinline__ int tiles_of_level(const int64_t *shapes, int l)
{
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    return ((H + Cfg::TH - 1) / Cfg::TH) * ((W + Cfg::TW - 1) / Cfg::TW);
}

// XOR swizzle of the 16-byte chunk index inside a token's 128-byte LDS row.  A ds_read_b128 is
// served in 16-lane groups over a 256-byte bank row; neighbouring queries read neighbouring tokens
// (stride 128 B), so without it the 8 queries x MH heads of a group pile onto 4 (D=16) or 2 (D=32)
// of the 16 slots (measured: 72 % of LDS cycles were conflict cycles).  XOR-ing with the token
// index's bits [1..] spreads same-parity tokens over all slots of their head.
template <int D> __device__ __forceinline__ int chunk_swizzle(int tok)
{
    return (tok >> 1) & (D / 4 - 1);
}

template <int D, typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void msda_fwd_tile(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw,
    int B, int S, int M, int L, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    constexpr int TH = Cfg::TH, TW = Cfg::TW, WH = Cfg::WH, WW = Cfg::WW, MH = Cfg::MH;
    constexpr int SLICE = Cfg::SLICE, P = TILE_P, NV = D / 4;
    constexpr int NSTAGE = (WH * WW * 8 + Cfg::THREADS - 1) / Cfg::THREADS;   // float4 per thread per window
    const int tid = threadIdx.x;
    const int HS = M / MH;                                // head slices per token row
    const int64_t row = (int64_t)M * D;

    // ---- the tile list: [level][tile-in-level] x head slice x batch ------------------------------
    int tiles_spatial = 0;
    bool equal_shapes = true;
    for (int l = 0; l < L; ++l) {
        tiles_spatial += tiles_of_level<Cfg>(shapes, l);
        equal_shapes = equal_shapes && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    }
    const int per_level = equal_shapes ? tiles_spatial / L : 0;
    const int64_t units = (int64_t)per_level * HS * B;   // (tile, slice, batch) units, equal shapes only
    const int64_t units8 = (units + 7) / 8;               // units per XCD
    // equal shapes: t enumerates xcd x (unit of that xcd) x level, see the decode below
    const int64_t total = equal_shapes ? units8 * 8 * L : (int64_t)tiles_spatial * HS * B;

    const int hh = tid % MH;
    const int qi = tid / MH;
    const int qly = qi / TW, qlx = qi % TW;

    // per-thread constants of the window copy: which float4s of the window this thread moves
    int st_src[NSTAGE];      // (wy * 65536 + wx) * 8 + part, or -1
    int st_dst[NSTAGE];      // float offset into win (swizzled)
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        const int idx = tid + i * Cfg::THREADS;
        const int tok = idx >> 3, part = idx & 7;
        const int wy = tok / WW, wx = tok - wy * WW;
        st_src[i] = idx < WH * WW * 8 ? ((wy << 16) | wx) : -1;
        st_dst[i] = tok * SLICE + ((part ^ chunk_swizzle<D>(tok)) << 2);
    }
    const int my_part = tid & 7;                          // idx & 7 is the same for every i (THREADS % 8 == 0)

    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        // ---- decode t (wave-uniform) ------------------------------------------------------------
        int lq, tin, hs, b;
        if (equal_shapes) {
            // workgroups t, t+8, t+16, ... share an XCD (and its L2).  Give XCD k a contiguous band of
            // units, and run the L query levels of one unit back to back: their source windows are
            // identical, so all but the first find them in that L2.
            const int64_t xcd = t & 7, r = t >> 3;
            lq = (int)(r % L);
            const int64_t unit = xcd * units8 + r / L;
            if (r / L >= units8 || unit >= units) continue;
            hs = (int)(unit % HS);
            const int64_t u2 = unit / HS;
            tin = (int)(u2 % per_level);
            b = (int)(u2 / per_level);
        } else {
            hs = (int)(t % HS);
            int64_t u2 = t / HS;
            b = (int)(u2 / tiles_spatial);
            int rem = (int)(u2 % tiles_spatial);
            lq = 0;
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
        const int m0 = hs * MH;

        const int qy = Y0 + qly, qx = X0 + qlx;
        const bool active = qy < Hq && qx < Wq;
        const int64_t q = lsi[lq] + (int64_t)qy * Wq + qx;          // query index == token index
        const int64_t bqm = active ? (((int64_t)b * S + q) * M + m0 + hh) : 0;
        const float *lp = loc + bqm * L * P * 2;
        const float *wp = aw + bqm * L * P;

        float acc[D];
#pragma unroll
        for (int i = 0; i < D; ++i) acc[i] = 0.f;
        unsigned long long miss = 0ull;

        // window geometry of a level: origin = the tile's centre carried to that level (integers)
        auto origin = [&](int l, int &oy, int &ox, int &H, int &W) {
            H = (int)shapes[2 * l];
            W = (int)shapes[2 * l + 1];
            oy = (int)(((int64_t)(2 * Y0 + TH) * H) / (2 * Hq)) - WH / 2;
            ox = (int)(((int64_t)(2 * X0 + TW) * W) / (2 * Wq)) - WW / 2;
        };
        // issue this thread's share of a window copy into registers (loads stay in flight)
        float4 stage[NSTAGE];
        auto fetch_window = [&](int l) {
            int oy, ox, H, W;
            origin(l, oy, ox, H, W);
            const float *plane = value + ((int64_t)b * S + lsi[l]) * row + (int64_t)m0 * D + my_part * 4;
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int gy = oy + (st_src[i] >> 16), gx = ox + (st_src[i] & 0xffff);
                stage[i] = make_float4(0, 0, 0, 0);
                if (st_src[i] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    stage[i] = *reinterpret_cast<const float4 *>(plane + ((int64_t)gy * W + gx) * row);
            }
        };

        // ---- locality probe: do this tile's taps stay near their own cell? ------------------------
        // Sampling data of the first level doubles as the probe.  If fewer than a quarter of the
        // workgroup's taps of that level fall inside the window, staging would be wasted: the tile is
        // then done entirely by the direct path below (all bits set in `miss`).
        float4 la = make_float4(0, 0, 0, 0), lb = la, wa = la;
        if (active) {
            la = *reinterpret_cast<const float4 *>(lp);
            lb = *reinterpret_cast<const float4 *>(lp + 4);
            wa = *reinterpret_cast<const float4 *>(wp);
        }
        int hits = 0;
        {
            int oy, ox, H, W;
            origin(0, oy, ox, H, W);
            const float xs[4] = {la.x, la.z, lb.x, lb.z}, ys[4] = {la.y, la.w, lb.y, lb.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float x = xs[p] * (float)W - 0.5f, y = ys[p] * (float)H - 0.5f;
                hits += (active && x >= (float)ox && x < (float)(ox + WW - 1) && y >= (float)oy &&
                         y < (float)(oy + WH - 1)) ? 1 : 0;
            }
        }
        const int live = __syncthreads_count(active);
        const int tile_hits = __syncthreads_count(hits >= 2);       // threads with most taps inside
        const bool staged = 4 * tile_hits >= live;

        if (staged) {
            fetch_window(0);
            for (int l = 0; l < L; ++l) {
                int oy, ox, H, W;
                origin(l, oy, ox, H, W);
                __syncthreads();                          // everyone is done reading the old window
#pragma unroll
                for (int i = 0; i < NSTAGE; ++i)
                    if (st_src[i] >= 0) *reinterpret_cast<float4 *>(win + st_dst[i]) = stage[i];
                // next level: window copy and sampling data go in flight under this level's taps
                float4 na = la, nb = lb, nw = wa;
                if (l + 1 < L) {
                    fetch_window(l + 1);
                    if (active) {
                        na = *reinterpret_cast<const float4 *>(lp + (l + 1) * P * 2);
                        nb = *reinterpret_cast<const float4 *>(lp + (l + 1) * P * 2 + 4);
                        nw = *reinterpret_cast<const float4 *>(wp + (l + 1) * P);
                    }
                }
                __syncthreads();

                if (active) {
                    const float lxs[4] = {la.x, la.z, lb.x, lb.z};
                    const float lys[4] = {la.y, la.w, lb.y, lb.w};
                    const float aws[4] = {wa.x, wa.y, wa.z, wa.w};
                    const float oxf = (float)ox, oyf = (float)oy;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float x = lxs[p] * (float)W - 0.5f;
                        const float y = lys[p] * (float)H - 0.5f;
                        const bool inwin = x >= oxf && x < oxf + (float)(WW - 1) && y >= oyf &&
                                           y < oyf + (float)(WH - 1);
                        if (inwin) {
                            const float fx = floorf(x), fy = floorf(y);
                            const int ix = (int)fx - ox, iy = (int)fy - oy;
                            const float wx1 = x - fx, wy1 = y - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                            const float a = aws[p];
                            const float w00 = wy0 * wx0 * a, w01 = wy0 * wx1 * a;
                            const float w10 = wy1 * wx0 * a, w11 = wy1 * wx1 * a;
                            const int t00 = iy * WW + ix;
                            const float *p00 = win + t00 * SLICE + hh * D;
                            const float *p10 = p00 + WW * SLICE;
                            const int s00 = chunk_swizzle<D>(t00), s01 = chunk_swizzle<D>(t00 + 1);
                            const int s10 = chunk_swizzle<D>(t00 + WW), s11 = chunk_swizzle<D>(t00 + WW + 1);
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                const float4 c00 = *reinterpret_cast<const float4 *>(p00 + ((k ^ s00) << 2));
                                const float4 c01 = *reinterpret_cast<const float4 *>(p00 + SLICE + ((k ^ s01) << 2));
                                const float4 c10 = *reinterpret_cast<const float4 *>(p10 + ((k ^ s10) << 2));
                                const float4 c11 = *reinterpret_cast<const float4 *>(p10 + SLICE + ((k ^ s11) << 2));
                                acc[4 * k + 0] += w00 * c00.x + w01 * c01.x + w10 * c10.x + w11 * c11.x;
                                acc[4 * k + 1] += w00 * c00.y + w01 * c01.y + w10 * c10.y + w11 * c11.y;
                                acc[4 * k + 2] += w00 * c00.z + w01 * c01.z + w10 * c10.z + w11 * c11.z;
                                acc[4 * k + 3] += w00 * c00.w + w01 * c01.w + w10 * c10.w + w11 * c11.w;
                            }
                        } else {
                            miss |= 1ull << (l * P + p);
                        }
                    }
                }
                la = na;
                lb = nb;
                wa = nw;
            }
        } else {
            miss = L * P >= 64 ? ~0ull : ((1ull << (L * P)) - 1);
        }

        if (active) {
            // ---- taps that left the window: straight from global memory (zero padding by test) ----
            while (miss) {
                const int bit = __ffsll((long long)miss) - 1;
                miss &= miss - 1;
                const int l = bit / P;
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const float x = lp[bit * 2 + 0] * (float)W - 0.5f;
                const float y = lp[bit * 2 + 1] * (float)H - 0.5f;
                const float a = wp[bit];
                if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
                const Footprint<float> f = footprint(y, x, H, W);
                const float *r0 = value + ((int64_t)b * S + lsi[l]) * row + (int64_t)(m0 + hh) * D +
                                  ((int64_t)f.y0 * W + f.x0) * row;
                const float *r1 = r0 + (int64_t)W * row;
                const float w00 = f.wy0 * f.wx0 * a, w01 = f.wy0 * f.wx1 * a;
                const float w10 = f.wy1 * f.wx0 * a, w11 = f.wy1 * f.wx1 * a;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const float4 z = make_float4(0, 0, 0, 0);
                    const float4 c00 = (f.vy0 && f.vx0) ? *reinterpret_cast<const float4 *>(r0 + 4 * k) : z;
                    const float4 c01 = (f.vy0 && f.vx1) ? *reinterpret_cast<const float4 *>(r0 + row + 4 * k) : z;
                    const float4 c10 = (f.vy1 && f.vx0) ? *reinterpret_cast<const float4 *>(r1 + 4 * k) : z;
                    const float4 c11 = (f.vy1 && f.vx1) ? *reinterpret_cast<const float4 *>(r1 + row + 4 * k) : z;
                    acc[4 * k + 0] += w00 * c00.x + w01 * c01.x + w10 * c10.x + w11 * c11.x;
                    acc[4 * k + 1] += w00 * c00.y + w01 * c01.y + w10 * c10.y + w11 * c11.y;
                    acc[4 * k + 2] += w00 * c00.z + w01 * c01.z + w10 * c10.z + w11 * c11.z;
                    acc[4 * k + 3] += w00 * c00.w + w01 * c01.w + w10 * c10.w + w11 * c11.w;
                }
            }
            float *o = out + bqm * D;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                *reinterpret_cast<float4 *>(o + 4 * k) =
                    make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
        }
    }
}

using Cfg16 = TileCfg<16, 8, 16, 6>;
using Cfg32 = TileCfg<32, 8, 16, 6>;

bool msda_tile_supported(int B, int S, int M, int D, int L, int Lq, int P, bool aligned16)
{
    if (!aligned16 || P != TILE_P || L > TILE_MAX_LEVELS || Lq != S || B < 1) return false;
    if (D == 16) return M % Cfg16::MH == 0;
    if (D == 32) return M % Cfg32::MH == 0;
    return false;
}

template <int D, typename Cfg>
static int launch_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *aw, int B, int S, int M, int L, float *out)
{
    static int blocks = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tile<D, Cfg>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_fwd_tile<D, Cfg>, Cfg::THREADS,
                                                         Cfg::LDS_BYTES) != hipSuccess || per_cu < 1)
            per_cu = 2;
        int n = cus * per_cu;
        return (n + 7) / 8 * 8;                          // keep the XCD interleave whole
    }();
    hipLaunchKernelGGL((msda_fwd_tile<D, Cfg>), dim3((unsigned)blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES,
                       st, value, shapes, lsi, loc, aw, B, S, M, L, out);
    return (int)hipGetLastError();
}

int msda_forward_tile(hipStream_t st, const float *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *aw, int B, int S, int M, int D, int L, int Lq,
                      int P, float *out)
{
    if (D == 16) return launch_tile<16, Cfg16>(st, value, shapes, lsi, loc, aw, B, S, M, L, out);
    if (D == 32) return launch_tile<32, Cfg32>(st, value, shapes, lsi, loc, aw, B, S, M, L, out);
    return (int)hipErrorInvalidValue;
}

}  // namespace mvdetr
