// Feature -> ground-plane homography warp -- gfx950 (MI355X) kernels + C ABI.
//
// Replaces the reference's third-party call
//     kornia.warp_perspective(imgs_feat, proj_mats, Rworld_shape, align_corners=False)
// (multiview_detector/models/mvdetr.py:194-195), which kornia (0.5.x) evaluates as three torch
// ops: normalize_homography + inverse on the 3x3s, transform_points over a normalised meshgrid
// (materialising a [N,H,W,2] grid), and F.grid_sample(bilinear, zeros).  Here it is ONE kernel:
// the 3x3 algebra and the per-pixel source coordinate are evaluated in fp64 registers from the
// caller's matrix (exact w.r.t. kornia's formula on the same inputs -- the fp32 op chain itself
// is only accurate to ~2e-4 near the horizon), the bilinear blend runs in the
// tensor dtype, and pixels whose footprint misses the source image are stored as zeros without
// touching `src`.
//
// Work mapping (wave64): a block is an 8x8 tile of destination pixels (lanes = pixels; its pre-image is a
// compact source patch, so a load instruction touches a handful of cache lines) times a group of up to 64
// channels split over the block's 4 waves.  For the channel-last output the 64x64 (pixel, channel) tile is
// transposed through LDS (65-float rows, conflict-free both ways) so the stores are 256-byte rows.
#include "common.h"
#include "../../include/mvdetr_ops.h"
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>
#include <cstring>

namespace mvdetr {

constexpr int WARP_PIX = 64;      // pixels per block (one per lane)
constexpr int WARP_CH = 64;       // channels per block
constexpr int WARP_SUB = 4;       // waves per block; each owns WARP_CH / WARP_SUB channels

struct SrcCoord {
    int y0, x0;
    double wy0, wy1, wx0, wx1;   // blend weights (rounded to the tensor dtype by the caller)
    bool v00, v01, v10, v11;
    bool any;
};

// Source sampling position of destination pixel (i, j) under kornia's convention.
template <typename T>
__device__ __forceinline__ void source_position(const T *__restrict__ Mn, int i, int j, int h, int w,
                                                double &x, double &y)
{
    // true inverse of the dst<-src homography, in fp64
    const double m0 = Mn[0], m1 = Mn[1], m2 = Mn[2], m3 = Mn[3], m4 = Mn[4], m5 = Mn[5], m6 = Mn[6],
                 m7 = Mn[7], m8 = Mn[8];
    const double c0 = m4 * m8 - m5 * m7, c1 = m5 * m6 - m3 * m8, c2 = m3 * m7 - m4 * m6;
    const double r = 1.0 / (m0 * c0 + m1 * c1 + m2 * c2);
    const double dj = (double)j, di = (double)i;
    // p = M^-1 (j, i, 1): source pixel, homogeneous
    const double px = (c0 * dj + (m2 * m7 - m1 * m8) * di + (m1 * m5 - m2 * m4)) * r;
    const double py = (c1 * dj + (m0 * m8 - m2 * m6) * di + (m2 * m3 - m0 * m5)) * r;
    const double pz = (c2 * dj + (m1 * m6 - m0 * m7) * di + (m0 * m4 - m1 * m3)) * r;
    // kornia normalises source pixels with the corner-aligned map 2p/(size-1) - 1 ...
    const double wd = w == 1 ? 1e-14 : (double)(w - 1), hd = h == 1 ? 1e-14 : (double)(h - 1);
    const double qx = 2.0 * px / wd - pz, qy = 2.0 * py / hd - pz;
    // ... divides only where |z| > 1e-8 (convert_points_from_homogeneous) ...
    const double s = fabs(pz) > 1e-8 ? 1.0 / pz : 1.0;
    // ... and hands the result to an align_corners=False sampler: ((g + 1) * size - 1) / 2
    x = ((qx * s + 1.0) * (double)w - 1.0) * 0.5;
    y = ((qy * s + 1.0) * (double)h - 1.0) * 0.5;
}

// nearest: grid_sample(mode='nearest') rounds the position half-to-even and takes that texel; here it becomes a
// bilinear footprint whose first corner carries all the weight (the call kornia.warp_perspective(..., 'nearest') of
// frameDataset.py:80)
__device__ __forceinline__ SrcCoord make_coord(double x, double y, int h, int w, int nearest)
{
    SrcCoord c;
    if (nearest) {
        x = rint(x);
        y = rint(y);
    }
    // reject far-out / non-finite positions before the int conversion
    const bool in = x > -1.0 && y > -1.0 && x < (double)w && y < (double)h;
    const double fx = floor(x), fy = floor(y);
    c.x0 = in ? (int)fx : 0;
    c.y0 = in ? (int)fy : 0;
    c.wx1 = x - fx;
    c.wy1 = y - fy;
    c.wx0 = 1.0 - c.wx1;
    c.wy0 = 1.0 - c.wy1;
    const bool vx0 = c.x0 >= 0, vx1 = c.x0 + 1 < w, vy0 = c.y0 >= 0, vy1 = c.y0 + 1 < h;
    c.v00 = in && vy0 && vx0;
    c.v01 = in && vy0 && vx1 && !nearest;
    c.v10 = in && vy1 && vx0 && !nearest;
    c.v11 = in && vy1 && vx1 && !nearest;
    c.any = c.v00 || c.v01 || c.v10 || c.v11;
    return c;
}

// Two x-adjacent source texels with one 8-byte load when both are inside the row (global loads only
// need dword alignment on gfx950), else two guarded scalar loads.
template <typename T>
__device__ __forceinline__ void load_pair(const T *p, bool v0, bool v1, T &a, T &b)
{
    if constexpr (sizeof(T) == 4) {
        if (v0 && v1) {
            float2 t;
            __builtin_memcpy(&t, p, 8);
            a = t.x;
            b = t.y;
            return;
        }
    }
    a = v0 ? p[0] : T(0);
    b = v1 ? p[1] : T(0);
}

// a destination pixel's bilinear footprint in the source view: what the channel-last kernels and (round 5) warp_fwd share
// per tile through LDS
template <typename T> struct WarpTexel {
    int o00;                     // element offset of corner (y0, x0) inside the view, in texels (not scaled by C)
    T w00, w01, w10, w11;
    int valid;                   // bit 0..3: corners 00, 01, 10, 11 inside the source
};

template <typename T>
__device__ __forceinline__ WarpTexel<T> warp_texel(const T *__restrict__ Mn, int i, int j, int h, int w, bool live, int nearest)
{
    WarpTexel<T> t;
    t.o00 = 0;
    t.valid = 0;
    t.w00 = t.w01 = t.w10 = t.w11 = T(0);
    if (live) {
        double x, y;
        source_position(Mn, i, j, h, w, x, y);
        const SrcCoord sc = make_coord(x, y, h, w, nearest);
        t.o00 = sc.y0 * w + sc.x0;
        t.w00 = T(sc.wy0 * sc.wx0);
        t.w01 = T(sc.wy0 * sc.wx1);
        t.w10 = T(sc.wy1 * sc.wx0);
        t.w11 = T(sc.wy1 * sc.wx1);
        t.valid = (sc.v00 ? 1 : 0) | (sc.v01 ? 2 : 0) | (sc.v10 ? 4 : 0) | (sc.v11 ? 8 : 0);
    }
    return t;
}

// Block id -> (view, 8x8 destination tile, channel group).  The world grid is not aligned with the image
// axes (a ground-plane row is a slanted, perspective-foreshortened line in the camera image), so 64
// consecutive pixels of one destination ROW gather from ~30 different source cache lines per load
// instruction (measured: 66 M L1 accesses per launch, i.e. the kernel ran at the texture-address rate, and
// neither fewer bytes nor a different XCD assignment changed its 100 us).  An 8x8 destination tile maps to
// a compact source patch instead: a handful of lines per load.
// (round 4 re-measured the tile shape at Wildtrack size, NCHW -> NCHW: 8 x 8 90 us, 4 x 16 102, 2 x 32 105, 16 x 4 149)
constexpr int WARP_TW = 8, WARP_TH = 8;
static_assert(WARP_TW * WARP_TH == WARP_PIX, "one lane per tile pixel");

struct WarpBlock {
    int n, i0, j0, group;
};

__device__ __forceinline__ bool warp_block(int N, int H, int W, int groups, WarpBlock &wb)
{
    const int tx = (W + WARP_TW - 1) / WARP_TW, ty = (H + WARP_TH - 1) / WARP_TH;
    int r = blockIdx.x;
    wb.group = r % groups;
    r /= groups;
    wb.j0 = (r % tx) * WARP_TW;
    r /= tx;
    wb.i0 = (r % ty) * WARP_TH;
    wb.n = r / ty;
    return wb.n < N;
}

inline int64_t warp_grid(int N, int H, int W, int groups)
{
    return (int64_t)N * ((H + WARP_TH - 1) / WARP_TH) * ((W + WARP_TW - 1) / WARP_TW) * groups;
}

template <typename T, bool NHWC>
__global__ __launch_bounds__(WARP_PIX *WARP_SUB) void warp_fwd(
    const T *__restrict__ src, const T *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, T *__restrict__ dst)
{
    __shared__ float tile[NHWC && sizeof(T) == 4 ? WARP_PIX * (WARP_CH + 1) : 1];
    // the tile's geometry once per block, by its first wave (round 5: every wave used to run the fp64 inverse, the two
    // divisions and the footprint for its own copy of the 64 pixels -- ~300 instructions in front of 16 channels x 12)
    __shared__ WarpTexel<T> tex[WARP_PIX];
    const int lane = threadIdx.x & (WARP_PIX - 1);
    const int sub = threadIdx.x / WARP_PIX;
    WarpBlock wb;
    if (!warp_block(N, H, W, (C + WARP_CH - 1) / WARP_CH, wb)) return;
    const int n = wb.n, i = wb.i0 + lane / WARP_TW, j = wb.j0 + lane % WARP_TW;
    const int64_t pix = ((int64_t)n * H + i) * W + j;
    const int cbase = wb.group * WARP_CH;
    constexpr int CPT = WARP_CH / WARP_SUB;
    const bool live = i < H && j < W;
    if (sub == 0) tex[lane] = warp_texel(Mv + (int64_t)n * 9, i, j, h, w, live, nearest);
    __syncthreads();
    const WarpTexel<T> t = tex[lane];
    const bool v00 = t.valid & 1, v01 = t.valid & 2, v10 = t.valid & 4, v11 = t.valid & 8;
    const int64_t plane = (int64_t)h * w;
#pragma unroll 4
    for (int k = 0; k < CPT; ++k) {
        const int c = cbase + sub * CPT + k;
        T val = T(0);
        if (live && c < C && t.valid) {
            const T *sp = src + ((int64_t)n * C + c) * plane + t.o00;
            T a, b, cc, d;
            load_pair(sp, v00, v01, a, b);
            load_pair(sp + w, v10, v11, cc, d);
            val = t.w00 * a + t.w01 * b + t.w10 * cc + t.w11 * d;
        }
        if constexpr (!NHWC) {
            if (live && c < C) dst[(((int64_t)n * C + c) * H + i) * W + j] = val;
        } else if constexpr (sizeof(T) == 4) {
            tile[lane * (WARP_CH + 1) + sub * CPT + k] = (float)val;
        } else {
            if (live && c < C) dst[pix * C + c] = val;
        }
    }
    if constexpr (NHWC && sizeof(T) == 4) {
        __syncthreads();
        // 64 pixels x 64 channels -> rows of 64 consecutive channels per pixel
        const int ch = threadIdx.x & (WARP_CH - 1);
#pragma unroll 4
        for (int k = 0; k < WARP_PIX / WARP_SUB; ++k) {
            const int p = k * WARP_SUB + sub;
            const int pi = wb.i0 + p / WARP_TW, pj = wb.j0 + p % WARP_TW;
            if (pi < H && pj < W && cbase + ch < C)
                dst[(((int64_t)n * H + pi) * W + pj) * C + cbase + ch] = (T)tile[p * (WARP_CH + 1) + ch];
        }
    }
}

// ---- NCHW source -> NCHW destination through LDS source patches (fp32; round 5) -------------------------------------------
// warp_fwd<NCHW> issues one 8-byte gather per (pixel, channel, corner row): 1.2 M wave-level gathers at Wildtrack size, each
// touching 8 - 16 cache lines of which it uses a few bytes -- it runs at the rate of the texture-address / L1 pipeline (85 - 90 us,
// 30 % of the HBM roofline; computing the tile's fp64 geometry once per block instead of once per wave changed nothing).  The
// destination grid is denser than the source (360 x 120 from 160 x 90), so the pre-image of an 8 x 32 destination tile is a
// compact patch of ~8 x 16 texels per channel.  Here a workgroup = (8 x 32 tile, chunk of up to 32 channels): the four waves find
// the bounding box of their footprints, copy the patch rows of every channel of the chunk to LDS with LDS-DMA (16 bytes per lane,
// x aligned down to 4 texels: whole 64-byte row pieces, every byte of a fetched line used), and every lane blends its pixel's four
// corners from LDS -- four ds_read_b32, masked by the corner's validity (zero padding and mode='nearest' need no zero-filled
// border and multiply no unloaded value) -- and stores 128-byte rows.  The chunk shrinks with the patch (LDS budget / patch size);
// a tile whose patch would leave fewer than 4 channels per chunk (extreme magnification near the horizon) gathers from memory as
// warp_fwd does, in the same launch.
#ifndef MVDETR_WARP_PP_CH
#define MVDETR_WARP_PP_CH 32
#endif
#ifndef MVDETR_WARP_PP_FLOATS
#define MVDETR_WARP_PP_FLOATS 8192          // 32 KB of LDS: 5 workgroups per CU
#endif
constexpr int WARP_PP_TH = 8, WARP_PP_TW = 32, WARP_PP_CH = MVDETR_WARP_PP_CH, WARP_PP_FLOATS = MVDETR_WARP_PP_FLOATS;
static_assert(WARP_PP_TH * WARP_PP_TW == 256, "one lane per tile pixel, four waves");

__global__ __launch_bounds__(256) void warp_fwd_nchw_patch(
    const float *__restrict__ src, const float *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, float *__restrict__ dst)
{
    extern __shared__ __attribute__((aligned(16))) float patch[];       // [channel of the chunk][patch row][PWp]
    __shared__ int box[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = (W + WARP_PP_TW - 1) / WARP_PP_TW, ty = (H + WARP_PP_TH - 1) / WARP_PP_TH;
    const int groups = (C + WARP_PP_CH - 1) / WARP_PP_CH;
    int r = blockIdx.x;
    const int group = r % groups;
    r /= groups;
    const int j0 = (r % tx) * WARP_PP_TW;
    r /= tx;
    const int i0 = (r % ty) * WARP_PP_TH, n = r / ty;
    if (n >= N) return;
    const int i = i0 + tid / WARP_PP_TW, j = j0 + tid % WARP_PP_TW;
    const bool live = i < H && j < W;
    double sx = 0.0, sy = 0.0;
    if (live) source_position(Mv + (int64_t)n * 9, i, j, h, w, sx, sy);
    SrcCoord sc = make_coord(sx, sy, h, w, nearest);
    const bool use = live && sc.any;
    const bool v00 = use && sc.v00, v01 = use && sc.v01, v10 = use && sc.v10, v11 = use && sc.v11;
    const float w00 = (float)(sc.wy0 * sc.wx0), w01 = (float)(sc.wy0 * sc.wx1), w10 = (float)(sc.wy1 * sc.wx0), w11 = (float)(sc.wy1 * sc.wx1);

    // bounding box of the VALID corners (inside the image by construction)
    int xmin = 0x3fffffff, xmax = -0x3fffffff, ymin = 0x3fffffff, ymax = -0x3fffffff;
    if (v00 || v10) { xmin = min(xmin, sc.x0); xmax = max(xmax, sc.x0); }
    if (v01 || v11) { xmin = min(xmin, sc.x0 + 1); xmax = max(xmax, sc.x0 + 1); }
    if (v00 || v01) { ymin = min(ymin, sc.y0); ymax = max(ymax, sc.y0); }
    if (v10 || v11) { ymin = min(ymin, sc.y0 + 1); ymax = max(ymax, sc.y0 + 1); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = min(xmin, __shfl_xor(xmin, o, 64));
        xmax = max(xmax, __shfl_xor(xmax, o, 64));
        ymin = min(ymin, __shfl_xor(ymin, o, 64));
        ymax = max(ymax, __shfl_xor(ymax, o, 64));
    }
    if (lane == 0) { box[wave][0] = xmin; box[wave][1] = xmax; box[wave][2] = ymin; box[wave][3] = ymax; }
    __syncthreads();
    xmin = min(min(box[0][0], box[1][0]), min(box[2][0], box[3][0]));
    xmax = max(max(box[0][1], box[1][1]), max(box[2][1], box[3][1]));
    ymin = min(min(box[0][2], box[1][2]), min(box[2][2], box[3][2]));
    ymax = max(max(box[0][3], box[1][3]), max(box[2][3], box[3][3]));

    const int c0 = group * WARP_PP_CH, cn = min(WARP_PP_CH, C - c0);
    const int64_t plane = (int64_t)h * w, oplane = (int64_t)H * W;
    float *const op = dst + ((int64_t)n * C + c0) * oplane + (int64_t)i * W + j;
    if (xmax < xmin) {
        // no pixel of the tile touches the source: zeros
        if (live)
            for (int c = 0; c < cn; ++c) op[c * oplane] = 0.f;
        return;
    }
    const int xs = xmin & ~3, PWp = ((xmax + 1 - xs) + 3) & ~3, PH = ymax - ymin + 1, PE = PH * PWp, Q = PWp >> 2;
    int chunk = WARP_PP_FLOATS / PE;
    chunk = chunk > cn ? cn : chunk;
    if (chunk < 4 && chunk < cn) {
        // the patch does not fit: per-lane gathers for this tile (warp_fwd's arithmetic)
        if (live) {
            const int64_t o00 = (int64_t)sc.y0 * w + sc.x0;
            for (int c = 0; c < cn; ++c) {
                float val = 0.f;
                if (use) {
                    const float *sp = src + ((int64_t)n * C + c0 + c) * plane + o00;
                    float a, b, cc, d;
                    load_pair(sp, v00, v01, a, b);
                    load_pair(sp + w, v10, v11, cc, d);
                    val = w00 * a + w01 * b + w10 * cc + w11 * d;
                }
                op[c * oplane] = val;
            }
        }
        return;
    }
    // this lane's corners inside a channel's patch (an invalid corner reads slot 0 and is masked)
    const int ry = sc.y0 - ymin, rx = sc.x0 - xs;
    const int r00 = v00 ? ry * PWp + rx : 0, r01 = v01 ? ry * PWp + rx + 1 : 0;
    const int r10 = v10 ? (ry + 1) * PWp + rx : 0, r11 = v11 ? (ry + 1) * PWp + rx + 1 : 0;
    // the view's channels c0.. through one buffer descriptor (the launcher checks that a view stays below 2^31 bytes)
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(src + ((int64_t)n * C + c0) * plane), 0, (int)((unsigned)cn * (unsigned)plane * 4u), 0x00020000);
    for (int cb = 0; cb < cn; cb += chunk) {
        const int cc_n = min(chunk, cn - cb), items = cc_n * PH * Q;      // 16-byte pieces of this chunk's patches
        if (cb) __syncthreads();                                          // everyone is done reading the previous chunk
        for (int base = wave * 64; base < items; base += 256) {
            const int it = base + lane;
            const int ch = it / (PH * Q), rem = it - ch * (PH * Q), row = rem / Q, q = rem - row * Q;
            const unsigned vo = it < items ? (unsigned)(((cb + ch) * (int)plane + (ymin + row) * w + xs + 4 * q) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void *)(patch + base * 4), 16, (int)vo, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's copies have landed ...
        __syncthreads();                                                  // ... and everyone's
        if (live) {
            const float *pc = patch;
            float *o = op + (int64_t)cb * oplane;
#pragma unroll 4
            for (int c = 0; c < cc_n; ++c) {
                const float a = v00 ? pc[r00] : 0.f, b = v01 ? pc[r01] : 0.f, cq = v10 ? pc[r10] : 0.f, d = v11 ? pc[r11] : 0.f;
                *o = w00 * a + w01 * b + w10 * cq + w11 * d;
                pc += PE;
                o += oplane;
            }
        }
    }
}

template <typename T, bool NHWC>
__global__ __launch_bounds__(WARP_PIX *WARP_SUB) void warp_bwd(
    const T *__restrict__ grad_dst, const T *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, T *__restrict__ grad_src)
{
    const int lane = threadIdx.x & (WARP_PIX - 1);
    const int sub = threadIdx.x / WARP_PIX;
    WarpBlock wb;
    if (!warp_block(N, H, W, (C + WARP_CH - 1) / WARP_CH, wb)) return;
    const int n = wb.n, i = wb.i0 + lane / WARP_TW, j = wb.j0 + lane % WARP_TW;
    if (i >= H || j >= W) return;
    const int64_t pix = ((int64_t)n * H + i) * W + j;
    const int cbase = wb.group * WARP_CH;
    constexpr int CPT = WARP_CH / WARP_SUB;
    double x, y;
    source_position(Mv + (int64_t)n * 9, i, j, h, w, x, y);
    const SrcCoord sc = make_coord(x, y, h, w, nearest);
    if (!sc.any) return;
    const T w00 = T(sc.wy0 * sc.wx0), w01 = T(sc.wy0 * sc.wx1);
    const T w10 = T(sc.wy1 * sc.wx0), w11 = T(sc.wy1 * sc.wx1);
    const int64_t plane = (int64_t)h * w;
    const int64_t o00 = (int64_t)sc.y0 * w + sc.x0;
    for (int k = 0; k < CPT; ++k) {
        const int c = cbase + sub * CPT + k;
        if (c >= C) break;
        const T g = NHWC ? grad_dst[pix * C + c] : grad_dst[(((int64_t)n * C + c) * H + i) * W + j];
        T *gp = grad_src + ((int64_t)n * C + c) * plane + o00;
        if (sc.v00) atomicAdd(gp, w00 * g);
        if (sc.v01) atomicAdd(gp + 1, w01 * g);
        if (sc.v10) atomicAdd(gp + w, w10 * g);
        if (sc.v11) atomicAdd(gp + w + 1, w11 * g);
    }
}

// ---- channel-last source AND destination ---------------------------------------------------------------
// The layout a channels_last trunk hands over ([N,h,w,C] memory under an NCHW-shaped tensor) and the one the
// shadow transformer's tokens want.  The NCHW-source kernel above gathers 4-byte texels, one load
// instruction per (channel, corner row): at Wildtrack size ~10 M wave-level gathers, i.e. it runs at the
// texture-address rate (93 us vs 33 us for a copy of the same bytes).  With channels innermost a corner is a
// contiguous C-vector: a lane takes 16 bytes of it, the C/4 (fp32) lanes of a pixel read whole 512-byte rows,
// and a pixel's geometry is evaluated once (fp64) by one lane and shared through LDS instead of once per
// channel group.
constexpr int WARP_CL_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(WARP_CL_THREADS) void warp_fwd_cl(
    const T *__restrict__ src, const T *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, T *__restrict__ dst)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ WarpTexel<T> tex[WARP_PIX];
    WarpBlock wb;
    if (!warp_block(N, H, W, 1, wb)) return;
    const int n = wb.n;
    if (threadIdx.x < WARP_PIX) {
        const int i = wb.i0 + threadIdx.x / WARP_TW, j = wb.j0 + threadIdx.x % WARP_TW;
        tex[threadIdx.x] = warp_texel(Mv + (int64_t)n * 9, i, j, h, w, i < H && j < W, nearest);
    }
    __syncthreads();
    const int chunks = C / VEC;                                   // 16-byte chunks per pixel
    const T *view = src + (int64_t)n * h * w * C;
    const int rowC = w * C;
    for (int item = threadIdx.x; item < WARP_PIX * chunks; item += WARP_CL_THREADS) {
        const int p = item / chunks, c = (item - p * chunks) * VEC;
        const int i = wb.i0 + p / WARP_TW, j = wb.j0 + p % WARP_TW;
        if (i >= H || j >= W) continue;
        const WarpTexel<T> t = tex[p];
        Pack<T, VEC> out = Pack<T, VEC>::zero();
        if (t.valid) {
            const T *sp = view + (int64_t)t.o00 * C + c;
            const Pack<T, VEC> z = Pack<T, VEC>::zero();
            const Pack<T, VEC> a = (t.valid & 1) ? Pack<T, VEC>::load(sp) : z;
            const Pack<T, VEC> b = (t.valid & 2) ? Pack<T, VEC>::load(sp + C) : z;
            const Pack<T, VEC> cc = (t.valid & 4) ? Pack<T, VEC>::load(sp + rowC) : z;
            const Pack<T, VEC> d = (t.valid & 8) ? Pack<T, VEC>::load(sp + rowC + C) : z;
#pragma unroll
            for (int k = 0; k < VEC; ++k) out.v[k] = t.w00 * a.v[k] + t.w01 * b.v[k] + t.w10 * cc.v[k] + t.w11 * d.v[k];
        }
        out.store(dst + (((int64_t)n * H + i) * W + j) * C + c);
    }
}

// Channel-last source -> NCHW destination (the reference's output layout, mvdetr.py:194): the same item loop as
// warp_fwd_cl, but the blended 16-byte chunks go to an LDS tile [pixel][channel] first and leave as whole 128-byte runs
// of 32 destination pixels per (channel, row) -- written channel by channel from registers they would be 4-byte
// stores 4*H*W bytes apart.  Destination tile = 2 rows x 32 columns; channels in groups of 128.
constexpr int WARP_NC_TW = 32, WARP_NC_TH = 2, WARP_NC_CG = 128, WARP_NC_LD = WARP_NC_CG + 4;
__global__ __launch_bounds__(WARP_CL_THREADS) void warp_fwd_cl_nchw(
    const float *__restrict__ src, const float *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, float *__restrict__ dst)
{
    __shared__ WarpTexel<float> tex[WARP_PIX];
    __shared__ __attribute__((aligned(16))) float ot[WARP_PIX * WARP_NC_LD];
    const int tx = (W + WARP_NC_TW - 1) / WARP_NC_TW, ty = (H + WARP_NC_TH - 1) / WARP_NC_TH;
    const int b = blockIdx.x, n = b / (tx * ty), rem = b - n * tx * ty;
    const int i0 = (rem / tx) * WARP_NC_TH, j0 = (rem % tx) * WARP_NC_TW;
    if (threadIdx.x < WARP_PIX) {
        const int i = i0 + threadIdx.x / WARP_NC_TW, j = j0 + threadIdx.x % WARP_NC_TW;
        tex[threadIdx.x] = warp_texel(Mv + (int64_t)n * 9, i, j, h, w, i < H && j < W, nearest);
    }
    __syncthreads();
    const float *view = src + (int64_t)n * h * w * C;
    const int rowC = w * C;
    const bool vec_ok = (W & 3) == 0;
    for (int cg = 0; cg < C; cg += WARP_NC_CG) {
        const int cn = min(WARP_NC_CG, C - cg), chunks = cn / 4;
        for (int item = threadIdx.x; item < WARP_PIX * chunks; item += WARP_CL_THREADS) {
            const int p = item / chunks, c = (item - p * chunks) * 4;
            const WarpTexel<float> t = tex[p];
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t.valid) {
                const float *sp = view + (int64_t)t.o00 * C + cg + c;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 a = (t.valid & 1) ? *reinterpret_cast<const float4 *>(sp) : z;
                const float4 bq = (t.valid & 2) ? *reinterpret_cast<const float4 *>(sp + C) : z;
                const float4 cc = (t.valid & 4) ? *reinterpret_cast<const float4 *>(sp + rowC) : z;
                const float4 d = (t.valid & 8) ? *reinterpret_cast<const float4 *>(sp + rowC + C) : z;
                out.x = t.w00 * a.x + t.w01 * bq.x + t.w10 * cc.x + t.w11 * d.x;
                out.y = t.w00 * a.y + t.w01 * bq.y + t.w10 * cc.y + t.w11 * d.y;
                out.z = t.w00 * a.z + t.w01 * bq.z + t.w10 * cc.z + t.w11 * d.z;
                out.w = t.w00 * a.w + t.w01 * bq.w + t.w10 * cc.w + t.w11 * d.w;
            }
            *reinterpret_cast<float4 *>(ot + p * WARP_NC_LD + c) = out;
        }
        __syncthreads();
        // (channel, tile row) segments of 32 pixels = 128 bytes, 8 lanes x float4 each
        for (int item = threadIdx.x; item < cn * WARP_NC_TH * 8; item += WARP_CL_THREADS) {
            const int part = item & 7, seg = item >> 3, r = seg % WARP_NC_TH, c = seg / WARP_NC_TH;
            const int i = i0 + r, j = j0 + part * 4;
            if (i >= H || j >= W) continue;
            const float *o = ot + (r * WARP_NC_TW + part * 4) * WARP_NC_LD + c;
            float *dp = dst + (((int64_t)n * C + cg + c) * H + i) * W + j;
            if (vec_ok && j + 4 <= W) {
                *reinterpret_cast<float4 *>(dp) = make_float4(o[0], o[WARP_NC_LD], o[2 * WARP_NC_LD], o[3 * WARP_NC_LD]);
            } else {
                for (int k = 0; k < 4 && j + k < W; ++k) dp[k] = o[k * WARP_NC_LD];
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(WARP_CL_THREADS) void warp_bwd_cl(
    const T *__restrict__ grad_dst, const T *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, T *__restrict__ grad_src)
{
    __shared__ WarpTexel<T> tex[WARP_PIX];
    WarpBlock wb;
    if (!warp_block(N, H, W, 1, wb)) return;
    const int n = wb.n;
    if (threadIdx.x < WARP_PIX) {
        const int i = wb.i0 + threadIdx.x / WARP_TW, j = wb.j0 + threadIdx.x % WARP_TW;
        tex[threadIdx.x] = warp_texel(Mv + (int64_t)n * 9, i, j, h, w, i < H && j < W, nearest);
    }
    __syncthreads();
    T *view = grad_src + (int64_t)n * h * w * C;
    const int rowC = w * C;
    // One CHANNEL per lane (not a 16-byte chunk as in the forward): the memory-side atomic units take a wave's
    // atomic instruction as one request per contiguous segment, so 64 consecutive channels of a corner are 4
    // requests, where 16-byte chunks per lane made every instruction 4-byte pieces 16 bytes apart -- 4x the
    // requests (measured 1.46 ms -> see DESIGN.md).
    for (int item = threadIdx.x; item < WARP_PIX * C; item += WARP_CL_THREADS) {
        const int p = item / C, c = item - p * C;
        const int i = wb.i0 + p / WARP_TW, j = wb.j0 + p % WARP_TW;
        if (i >= H || j >= W) continue;
        const WarpTexel<T> t = tex[p];
        if (!t.valid) continue;
        const T g = grad_dst[(((int64_t)n * H + i) * W + j) * C + c];
        T *gp = view + (int64_t)t.o00 * C + c;
        if (t.valid & 1) atomicAdd(gp, t.w00 * g);
        if (t.valid & 2) atomicAdd(gp + C, t.w01 * g);
        if (t.valid & 4) atomicAdd(gp + rowC, t.w10 * g);
        if (t.valid & 8) atomicAdd(gp + rowC + C, t.w11 * g);
    }
}

// ---- backward as a GATHER (channel-last on both sides; no atomics, deterministic) ------------------------------------
// grid_sample's backward scatters: every destination pixel adds w_k * grad to the four source texels of its bilinear
// footprint (warp_bwd / warp_bwd_cl above: one memory-side atomic request per contiguous segment, 382 us at Wildtrack
// size, 6.8 % of the roofline).  The homography is invertible, so the sum can be turned around: the destination pixels
// whose footprint touches a 2 x 2 BLOCK of source texels are those whose source position lies in
// [2bx-1, 2bx+2) x [2by-1, 2by+2), and that set is the image of this square under M (dst <- src) -- a convex
// quadrilateral of a few dozen pixels (a few hundred for the far field, where the ground plane is magnified).
//   * warp_bwd_gather: a group of G = C/4 lanes (32 at 128 channels; two groups per wave) owns one block.
//     1. Candidate scan: when the square does not cross the line that maps to infinity (the horizon: same sign of the
//        homogeneous coordinate at all four corners) its image is the hull of the four projected corners.  Far-field
//        images are long diagonal slivers whose bounding box is ~10x their area, so the scan is a SHEARED box: lines
//        (rows or columns, whichever is cheaper) u0..u1, and on line u the `len` pixels from ceil(a + s (u - uc)) on,
//        with the shear s taken from one of the hull's edges (the cheapest of five candidates).  Squares that do cross
//        the horizon get plain boxes from clipping the destination rectangle (Sutherland-Hodgman, one lane, LDS)
//        against the five linear inequalities "source position inside the square", once per sign of the homogeneous
//        coordinate.  Scans are only candidate sets: membership is decided by the exact test below.
//     2. The G lanes test G candidates at a time with the forward's own source_position / make_coord (forward and
//        backward use the same weights bit for bit), compact the hits in scan order into LDS (ballot + popcount: the
//        order of every sum is fixed -> deterministic results), then every lane walks the hits, loads its 16-byte
//        channel chunk of grad_dst (the group reads whole 512-byte pixel rows) and accumulates into the block's four
//        texels in registers.
//     grad_src is written once, with plain stores: it need not be zeroed, and each grad_dst row is read ~2.25 times
//     (once per block its footprint touches), mostly from L2.  One difference from a scatter: a hit adds 0 * g to the
//     block texels its footprint misses, so a NON-FINITE upstream gradient spreads to the other texels of the 2 x 2 block.
//   * Pixels kornia does not divide: convert_points_from_homogeneous divides by z only where |z| > 1e-8, so a destination
//     pixel with |z| <= 1e-8 samples a position unrelated to the projective map and no scan finds it.  warp_bwd_scans
//     lists such pixels (normally none) in pixel order; the gather tests the listed ones as extra candidates of every
//     block and leaves them out of its scans, so they are counted exactly once and in a fixed order as well.
template <typename T> struct WarpRec {
    int pix;                     // destination pixel index (n * H + i) * W + j
    int mask;                    // bit t: the footprint touches block texel t = 2 * (Y - 2by) + (X - 2bx)
    T w[4];                      // its weight there (the forward's T(wy * wx))
};

struct WarpScan {
    int u0, u1;                  // lines u0..u1 (inclusive); empty when u1 < u0
    int len;                     // candidates per line
    int swap;                    // 0: lines are destination rows (u = i, v = j); 1: lines are columns (u = j, v = i)
    double a, s, uc;             // line u starts at v = ceil(a + s * (u - uc))
};

// homogeneous coordinate of destination pixel (i, j) under M^-1 (the pz of source_position): the forward divides by it
// only where |pz| > 1e-8
template <typename T> __device__ __forceinline__ double source_pz(const T *__restrict__ Mn, int i, int j)
{
    const double m0 = Mn[0], m1 = Mn[1], m2 = Mn[2], m3 = Mn[3], m4 = Mn[4], m5 = Mn[5], m6 = Mn[6],
                 m7 = Mn[7], m8 = Mn[8];
    const double c0 = m4 * m8 - m5 * m7, c1 = m5 * m6 - m3 * m8, c2 = m3 * m7 - m4 * m6;
    const double r = 1.0 / (m0 * c0 + m1 * c1 + m2 * c2);
    return (c2 * (double)j + (m1 * m6 - m0 * m7) * (double)i + (m0 * m4 - m1 * m3)) * r;
}

constexpr int WARP_CLIP_MAXV = 12;      // rectangle (4) + one new vertex per clip (5), rounded up
constexpr int WARP_CLIP_SLOTS = 4 * WARP_CLIP_MAXV + 15;        // polygon double buffer + the 5 x 3 coefficients
constexpr int WARP_SCAN_THREADS = 64;
constexpr int WARP_ODD_SEGS = 1024;     // workgroups (= list segments) that look for pixels kornia does not divide

// bounding box of {(j, i) in [-1, W] x [-1, H] : e[k] . (j, i, 1) >= 0 for all k}; false when empty.  `lds` is this lane's
// scratch, slot-major over the workgroup's lanes (slot s of lane l at lds[s * WARP_SCAN_THREADS]: conflict-free);
// slots 0..4*MAXV-1 hold the polygon double buffer, the 15 coefficients follow.
__device__ inline bool warp_clip_box(double *lds, int H, int W, int &i0, int &i1, int &j0, int &j1)
{
#define SLOT(k) lds[(k) * WARP_SCAN_THREADS]
    constexpr int PX = 0, PY = WARP_CLIP_MAXV, QX = 2 * WARP_CLIP_MAXV, QY = 3 * WARP_CLIP_MAXV, E = 4 * WARP_CLIP_MAXV;
    int n = 4;
    SLOT(PX + 0) = -1.0; SLOT(PY + 0) = -1.0;
    SLOT(PX + 1) = (double)W; SLOT(PY + 1) = -1.0;
    SLOT(PX + 2) = (double)W; SLOT(PY + 2) = (double)H;
    SLOT(PX + 3) = -1.0; SLOT(PY + 3) = (double)H;
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
        int m = 0;
        const double e0 = SLOT(E + 3 * k), e1 = SLOT(E + 3 * k + 1), e2 = SLOT(E + 3 * k + 2);
#pragma unroll 1
        for (int v = 0; v < n; ++v) {
            const int u = v + 1 == n ? 0 : v + 1;
            const double ax = SLOT(PX + v), ay = SLOT(PY + v), bx = SLOT(PX + u), by = SLOT(PY + u);
            const double da = e0 * ax + e1 * ay + e2, db = e0 * bx + e1 * by + e2;
            const bool ia = da >= 0.0, ib = db >= 0.0;
            if (ia && m < WARP_CLIP_MAXV) { SLOT(QX + m) = ax; SLOT(QY + m) = ay; ++m; }
            if (ia != ib && m < WARP_CLIP_MAXV) {
                const double t = da / (da - db);
                SLOT(QX + m) = ax + t * (bx - ax);
                SLOT(QY + m) = ay + t * (by - ay);
                ++m;
            }
        }
        n = m;
        if (n == 0) return false;
#pragma unroll 1
        for (int v = 0; v < n; ++v) { SLOT(PX + v) = SLOT(QX + v); SLOT(PY + v) = SLOT(QY + v); }
    }
    double xa = SLOT(PX), xb = xa, ya = SLOT(PY), yb = ya;
#pragma unroll 1
    for (int v = 1; v < n; ++v) {
        xa = fmin(xa, SLOT(PX + v)); xb = fmax(xb, SLOT(PX + v));
        ya = fmin(ya, SLOT(PY + v)); yb = fmax(yb, SLOT(PY + v));
    }
    j0 = max(0, (int)floor(xa) - 1); j1 = min(W - 1, (int)ceil(xb) + 1);
    i0 = max(0, (int)floor(ya) - 1); i1 = min(H - 1, (int)ceil(yb) + 1);
    return j0 <= j1 && i0 <= i1;
}

__device__ __forceinline__ WarpScan warp_scan_none()
{
    WarpScan z;
    z.u0 = 0; z.u1 = -1; z.len = 0; z.swap = 0; z.a = 0.0; z.s = 0.0; z.uc = 0.0;
    return z;
}

__device__ __forceinline__ WarpScan warp_scan_box(int i0, int i1, int j0, int j1)
{
    WarpScan z;
    z.u0 = i0; z.u1 = i1; z.len = j1 - j0 + 1; z.swap = 0; z.a = (double)j0; z.s = 0.0; z.uc = 0.0;
    return z;
}

// warp_bwd_scans: one lane per 2 x 2 block of source texels -> its (up to two) candidate scans of the destination pixels
// whose source position can lie in [2bx-1, 2bx+2) x [2by-1, 2by+2).  force_clip: take the clipping path for every block
// (a test knob: MVDETR_WARP_BWD_GEOMETRY=clip).  Blocks with more than heavy_above candidates are listed apart.
template <typename T>
__global__ __launch_bounds__(WARP_SCAN_THREADS) void warp_bwd_scans(const T *__restrict__ Mv, int N, int h, int w, int H,
                                                                    int W, int force_clip, int heavy_above,
                                                                    WarpScan *__restrict__ scans, int *__restrict__ heavy_list,
                                                                    int *__restrict__ counts, int *__restrict__ odd_list)
{
    __shared__ double clip[WARP_CLIP_SLOTS * WARP_SCAN_THREADS];
    const int bw2 = (w + 1) / 2, bh2 = (h + 1) / 2;
    const int64_t nblk = (int64_t)N * bh2 * bw2;
    const int scan_wgs = (int)((nblk + WARP_SCAN_THREADS - 1) / WARP_SCAN_THREADS);
    if ((int)blockIdx.x >= scan_wgs) {
        // The last WARP_ODD_SEGS workgroups: kornia divides by z only where |z| > 1e-8 (convert_points_from_homogeneous), so
        // a destination pixel with |z| <= 1e-8 samples a position unrelated to the projective map and no scan finds it.
        // Each of these workgroups (one wave) walks a contiguous range of destination pixels IN ORDER and lists the ones
        // it finds (normally none) in its own segment: the gather then tests them as extra candidates, in a fixed order.
        const int seg = (int)blockIdx.x - scan_wgs;
        const int64_t npix = (int64_t)N * H * W, per = (npix + WARP_ODD_SEGS - 1) / WARP_ODD_SEGS;
        const int64_t p0 = seg * per, p1 = min(npix, p0 + per);
        const int lane = threadIdx.x;
        int found = 0;
        for (int64_t base = p0; base < p1; base += WARP_SCAN_THREADS) {
            const int64_t p = base + lane;
            bool odd = false;
            if (p < p1) {
                const int n = (int)(p / ((int64_t)H * W)), rem = (int)(p - (int64_t)n * H * W);
                const int i = rem / W, j = rem - i * W;
                // cheap screen first: z = zu / det with zu linear in (j, i); far from zero (a factor 4 of slack for the
                // rounding of the exact expression) -> not odd, no division
                const T *Mn = Mv + (int64_t)n * 9;
                const double m0 = Mn[0], m1 = Mn[1], m2 = Mn[2], m3 = Mn[3], m4 = Mn[4], m5 = Mn[5], m6 = Mn[6], m7 = Mn[7],
                             m8 = Mn[8];
                const double det = m0 * (m4 * m8 - m5 * m7) + m1 * (m5 * m6 - m3 * m8) + m2 * (m3 * m7 - m4 * m6);
                const double zu = (m3 * m7 - m4 * m6) * (double)j + (m1 * m6 - m0 * m7) * (double)i + (m0 * m4 - m1 * m3);
                if (!(fabs(zu) > 4e-8 * fabs(det)))
                    odd = fabs(source_pz(Mn, i, j)) <= 1e-8;                                // (NaN: neither scanned nor listed)
            }
            const uint64_t m = __ballot(odd);
            if (odd) odd_list[p0 + found + __popcll(m & ((1ull << lane) - 1ull))] = (int)p;
            found += __popcll(m);
        }
        if (lane == 0) {
            counts[2 + seg] = found;
            if (found) atomicAdd(counts + 1, found);
        }
        return;
    }
    const int64_t idx = (int64_t)blockIdx.x * WARP_SCAN_THREADS + threadIdx.x;
    if (idx >= nblk) return;
    const int n = (int)(idx / ((int64_t)bh2 * bw2)), rem = (int)(idx - (int64_t)n * bh2 * bw2);
    const int by = rem / bw2, bx = rem - by * bw2;
    const double xl = 2.0 * bx - 1.0, xh = 2.0 * bx + 2.0, yl = 2.0 * by - 1.0, yh = 2.0 * by + 2.0;
    const T *Mn = Mv + (int64_t)n * 9;
    double *const lds = clip + threadIdx.x;
    WarpScan b0, b1;
    [&] {
    const double m0 = Mn[0], m1 = Mn[1], m2 = Mn[2], m3 = Mn[3], m4 = Mn[4], m5 = Mn[5], m6 = Mn[6],
                 m7 = Mn[7], m8 = Mn[8];
    b0 = warp_scan_none();
    b1 = warp_scan_none();
    // kornia's chain: position = p * size / (size - 1) - 0.5 for the source pixel p = M^-1 (j, i, 1)
    const double wd = w == 1 ? 1e-14 : (double)(w - 1), hd = h == 1 ? 1e-14 : (double)(h - 1);
    const double ax = (double)w / wd, ay = (double)h / hd;
    const double det = m0 * (m4 * m8 - m5 * m7) + m1 * (m5 * m6 - m3 * m8) + m2 * (m3 * m7 - m4 * m6);
    const double rdet = 1.0 / det;
    // singular or non-finite matrix: the forward's positions are all inf / NaN -> zeros everywhere, no gradient
    if (!(fabs(rdet) <= 1.79e308) || !(fabs(det) <= 1.79e308)) return;
    bool fast = !force_clip;
    double X[4], Y[4];
    int pos = 0, neg = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sx = (k & 1) ? xh : xl, sy = (k & 2) ? yh : yl;
        const double qx = (sx + 0.5) / ax, qy = (sy + 0.5) / ay;                    // source pixel
        const double dx = m0 * qx + m1 * qy + m2, dy = m3 * qx + m4 * qy + m5, dz = m6 * qx + m7 * qy + m8;
        pos += dz > 0.0;
        neg += dz < 0.0;
        X[k] = dx / dz;
        Y[k] = dy / dz;
        if (!(fabs(X[k]) <= 1e300) || !(fabs(Y[k]) <= 1e300)) fast = false;
    }
    if (fast && (pos == 4 || neg == 4)) {
        // the square is on one side of the horizon: its image is the convex hull of the four corner images.  Five
        // ways to scan it: rows or columns sheared along the image of the square's x or y side, or plain rows.
        constexpr double MARGIN = 0.01;                      // pixels; fp64 rounding of the hull is ~1e-12
        double best = 1e300;
#pragma unroll 1
        for (int opt = 0; opt < 5; ++opt) {
            const int swap = opt >> 1 & (opt < 4);           // options 2, 3: lines are columns
            const int edge = opt & 1;                        // sheared along corner 0 -> 1 (x side) or 0 -> 2 (y side)
            const double Xe = edge ? X[2] : X[1], Ye = edge ? Y[2] : Y[1];
            const double du = swap ? Xe - X[0] : Ye - Y[0], dv = swap ? Ye - Y[0] : Xe - X[0];
            double sh = 0.0;
            if (opt < 4) {
                if (!(fabs(du) > 1e-3 * fabs(dv)) || !(fabs(du) > 1e-9)) continue;   // (nearly) along the lines: no shear
                sh = dv / du;
            }
            const double uc = swap ? X[0] : Y[0];
            double ua = 0, ub = 0, va = 0, vb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double u = swap ? X[k] : Y[k], v = (swap ? Y[k] : X[k]) - sh * (u - uc);
                ua = k ? fmin(ua, u) : u; ub = k ? fmax(ub, u) : u;
                va = k ? fmin(va, v) : v; vb = k ? fmax(vb, v) : v;
            }
            const double U = swap ? (double)W : (double)H;
            const double u0 = fmax(ceil(ua - MARGIN), 0.0), u1 = fmin(floor(ub + MARGIN), U - 1.0);
            WarpScan z = warp_scan_none();
            double cost = 0.0;
            if (u0 <= u1) {
                const double len = floor(vb - va + 2.0 * MARGIN) + 2.0;
                cost = (u1 - u0 + 1.0) * len;
                // (a square just below the horizon has a far-end image 1e5 .. 1e7 pixels long: the lines are clamped to the
                // grid but their length is not, so such a scan walks millions of candidates outside the grid.  The clipping
                // path below never costs more than H * W candidates: anything dearer than that is left to it.)
                if (!(cost <= (double)H * (double)W)) continue;
                z.u0 = (int)u0; z.u1 = (int)u1; z.len = (int)len; z.swap = swap; z.a = va - MARGIN; z.s = sh; z.uc = uc;
            }
            if (cost < best) { best = cost; b0 = z; }
        }
        if (best < 1e300) return;
    }
    {
        // rows of adj(M) acting on (j, i, 1) (a positive or negative multiple of M^-1: both signs are clipped)
        const double A0[3] = {m4 * m8 - m5 * m7, m2 * m7 - m1 * m8, m1 * m5 - m2 * m4};
        const double A1[3] = {m5 * m6 - m3 * m8, m0 * m8 - m2 * m6, m2 * m3 - m0 * m5};
        const double A2[3] = {m3 * m7 - m4 * m6, m1 * m6 - m0 * m7, m0 * m4 - m1 * m3};
        double big = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) big = fmax(big, fmax(fabs(A0[c]), fmax(fabs(A1[c]), fabs(A2[c]))));
        int4 c0 = make_int4(0, -1, 0, -1), c1 = c0;
        if (!(big > 0.0 && big <= 1e300)) {
            c0 = make_int4(0, H - 1, 0, W - 1);                                      // cannot reason: scan everything
        } else {
            const double sc = 1.0 / big;
#pragma unroll 1
            for (int sgn = 0; sgn < 2; ++sgn) {
                const double sg = sgn ? -sc : sc;
                constexpr int E = 4 * WARP_CLIP_MAXV;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double gx = ax * A0[c] - 0.5 * A2[c], gy = ay * A1[c] - 0.5 * A2[c], gz = A2[c];
                    lds[(E + 0 + c) * WARP_SCAN_THREADS] = sg * gz;
                    lds[(E + 3 + c) * WARP_SCAN_THREADS] = sg * (gx - xl * gz);
                    lds[(E + 6 + c) * WARP_SCAN_THREADS] = sg * (xh * gz - gx);
                    lds[(E + 9 + c) * WARP_SCAN_THREADS] = sg * (gy - yl * gz);
                    lds[(E + 12 + c) * WARP_SCAN_THREADS] = sg * (yh * gz - gy);
                }
                // (most squares that cross the horizon map outside the destination: one inequality fails at all four
                // corners of the rectangle -> nothing to clip)
                bool empty = false;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const double e0 = lds[(E + 3 * k) * WARP_SCAN_THREADS], e1 = lds[(E + 3 * k + 1) * WARP_SCAN_THREADS],
                                 e2 = lds[(E + 3 * k + 2) * WARP_SCAN_THREADS];
                    const double r00 = e2 - e0 - e1, r10 = e0 * W - e1 + e2, r11 = e0 * W + e1 * H + e2, r01 = e1 * H - e0 + e2;
                    empty = empty || (r00 < 0.0 && r10 < 0.0 && r11 < 0.0 && r01 < 0.0);
                }
                int i0, i1, j0, j1;
                if (!empty && warp_clip_box(lds, H, W, i0, i1, j0, j1)) {
                    const int4 b = make_int4(i0, i1, j0, j1);
                    if (sgn) c1 = b; else c0 = b;
                }
            }
            // two overlapping boxes would count their common pixels twice: merge them
            if (c0.x <= c0.y && c1.x <= c1.y && c0.x <= c1.y && c1.x <= c0.y && c0.z <= c1.w && c1.z <= c0.w) {
                c0 = make_int4(min(c0.x, c1.x), max(c0.y, c1.y), min(c0.z, c1.z), max(c0.w, c1.w));
                c1 = make_int4(0, -1, 0, -1);
            }
        }
        if (c0.x <= c0.y) b0 = warp_scan_box(c0.x, c0.y, c0.z, c0.w);
        if (c1.x <= c1.y) b1 = warp_scan_box(c1.x, c1.y, c1.z, c1.w);
    }
    }();
    // heavy blocks (far field: hundreds of candidates) go to a work list and are flagged in their first scan; the others are
    // found by the gather's light waves directly
    const int64_t c0 = b0.u1 >= b0.u0 ? (int64_t)(b0.u1 - b0.u0 + 1) * b0.len : 0;
    const int64_t c1 = b1.u1 >= b1.u0 ? (int64_t)(b1.u1 - b1.u0 + 1) * b1.len : 0;
    const bool is_heavy = c0 + c1 > heavy_above;
    if (is_heavy) b0.swap |= 2;
    scans[2 * idx] = b0;
    scans[2 * idx + 1] = b1;
    // one atomic per wave (25 K atomics on one address take 11 ns each: 285 us); the order inside the list does not matter
    const int lane = threadIdx.x & 63;
    const uint64_t hv = __ballot(is_heavy);
    if (is_heavy) {
        const int leader = __ffsll((unsigned long long)hv) - 1;
        int slot = 0;
        if (lane == leader) slot = atomicAdd(counts, __popcll(hv));
        slot = __shfl(slot, leader, 64) + __popcll(hv & ((1ull << lane) - 1ull));
        heavy_list[slot] = (int)idx;
    }
}
#undef SLOT

constexpr int WARP_GU = 8;          // grad_dst loads in flight per lane
constexpr int WARP_HEAVY = 128;     // blocks with more candidates than this get a whole workgroup
constexpr int WARP_HEAVY_WGS = 2048;

// The candidate rounds first, first + stride, ... (64 candidates each) of one 2 x 2 texel block, run by one wave.
// Lane = (hit stream, channel chunk): with G = 2^lgG >= C/4 lanes per stream the wave runs 64 / G streams that take the
// compacted hits round-robin (two at 128 channels).  Candidates: the block's (up to two) scans, then -- normally none --
// the listed pixels that kornia does not divide (counts[1] of them in all, counts[2 + seg] in segment seg of odd_list).
template <typename T>
__device__ __forceinline__ void warp_gather_rounds(
    const T *__restrict__ gchunk, const T (&Mn)[9], const WarpScan &sc0, const WarpScan &sc1, const int *__restrict__ counts,
    const int *__restrict__ odd_list, int64_t odd_per, int n, int bx, int by, int C, int h, int w, int H, int W, int nearest,
    int lgG, bool has_ch, int first, int stride, WarpRec<T> *mine, T (&acc)[4][16 / (int)sizeof(T)])
{
    constexpr int VEC = 16 / (int)sizeof(T);
    using Rec = WarpRec<T>;
    const int S = 64 >> lgG;
    const int lane = threadIdx.x & 63, st = lane >> lgG;
    const uint64_t below = (1ull << lane) - 1ull;

    // one round: this lane's candidate (i, j) (if `cand`), an ordinary pixel (|z| > 1e-8) or a listed odd one
    auto round = [&](bool cand, int i, int j, bool odd) {
        bool hit = false;
        Rec r;
        r.pix = 0;
        r.mask = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) r.w[t] = T(0);
        if (cand && (fabs(source_pz(Mn, i, j)) > 1e-8) != odd) {
            double x, y;
            source_position(Mn, i, j, h, w, x, y);
            const SrcCoord c = make_coord(x, y, h, w, nearest);
            const int dx = c.x0 - 2 * bx, dy = c.y0 - 2 * by;                      // corner 00 relative to the block
            if (c.any && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) {
                const T w00 = T(c.wy0 * c.wx0), w01 = T(c.wy0 * c.wx1);
                const T w10 = T(c.wy1 * c.wx0), w11 = T(c.wy1 * c.wx1);
                // corner (cy, cx) lies on block texel (dy + cy, dx + cx) when that is in {0, 1}^2
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ty = t >> 1, tx = t & 1, cy = ty - dy, cx = tx - dx;
                    if (cy >= 0 && cy <= 1 && cx >= 0 && cx <= 1) {
                        const bool ok = cy ? (cx ? c.v11 : c.v10) : (cx ? c.v01 : c.v00);
                        const T ww = cy ? (cx ? w11 : w10) : (cx ? w01 : w00);
                        if (ok) {
                            r.mask |= 1 << t;
                            r.w[t] = ww;
                        }
                    }
                }
                r.pix = ((n * H + i) * W + j) * C;                                  // element offset (< 2^31: checked on the host)
                hit = r.mask != 0;
            }
        }
        const uint64_t hits = __ballot(hit);
        const int nhit = __popcll(hits);
        if (hit) mine[__popcll(hits & below)] = r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // stream st takes hits st, st + S, ...: WARP_GU of them in flight.  (q0 + u * S < nhit is wave-uniform: whole
        // steps are skipped with a scalar branch; the one stream that runs out a hit early gets zero weights.)
        for (int q0 = 0; q0 < nhit; q0 += WARP_GU * S) {
            Pack<T, VEC> g[WARP_GU];
#pragma unroll
            for (int u = 0; u < WARP_GU; ++u) {               // the loads first (only the offsets are read here) ...
                const int off = mine[min(q0 + u * S + st, nhit - 1)].pix;          // (clamped: branch-free, so they overlap)
                g[u] = has_ch ? Pack<T, VEC>::load(gchunk + off) : Pack<T, VEC>::zero();
            }
#pragma unroll
            for (int u = 0; u < WARP_GU; ++u) {               // ... then the weights, re-read from LDS
                if (q0 + u * S < nhit) {
                    const int qq = q0 + u * S + st;
                    const Rec rr = mine[min(qq, nhit - 1)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const T wt = qq < nhit ? rr.w[t] : T(0);           // (texels the footprint misses have weight 0)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[t][v] += wt * g[u].v[v];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (int bsel = 0; bsel < 2; ++bsel) {
        const WarpScan &sc = bsel ? sc1 : sc0;
        const int swap = sc.swap & 1;
        const int cnt = sc.u1 >= sc.u0 && sc.len > 0 ? (sc.u1 - sc.u0 + 1) * sc.len : 0;
        const int V = swap ? H : W;
        const float rlen = 1.0f / (float)sc.len;
        const bool small = cnt < (1 << 22);                   // k * rlen is then within one of the quotient
        for (int base = first * 64; base < cnt; base += stride * 64) {
            const int k = base + lane;
            int i = 0, j = 0;
            bool cand = false;
            if (k < cnt) {
                int du;
                if (small) {
                    du = (int)((float)k * rlen);
                    du -= du * sc.len > k;
                    du += (du + 1) * sc.len <= k;
                } else {
                    du = k / sc.len;
                }
                const int u = sc.u0 + du;
                const int v = (int)ceil(sc.a + sc.s * ((double)u - sc.uc)) + (k - du * sc.len);
                i = swap ? v : u;
                j = swap ? u : v;
                cand = v >= 0 && v < V;
            }
            round(cand, i, j, false);
        }
    }
    if (counts[1] != 0) {
        // pixels kornia does not divide (listed by warp_bwd_scans, segment by segment in pixel order)
        const int64_t npix = (int64_t)H * W;                  // per view
        for (int seg = 0; seg < WARP_ODD_SEGS; ++seg) {
            const int cnt = counts[2 + seg];
            for (int base = first * 64; base < cnt; base += stride * 64) {
                const int k = base + lane;
                int i = 0, j = 0;
                bool cand = false;
                if (k < cnt) {
                    const int p = odd_list[seg * odd_per + k];
                    const int pn = (int)(p / npix), rem = (int)(p - pn * npix);
                    i = rem / W;
                    j = rem - i * W;
                    cand = pn == n;
                }
                round(cand, i, j, true);
            }
        }
    }
    // the streams' partial sums, in a fixed order
    for (int off = 1 << lgG; off < 64; off <<= 1)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[t][v] += __shfl_xor(acc[t][v], off, 64);
}

// Workgroups [0, heavy_wgs): the heavy blocks (far field: hundreds of candidates; listed by warp_bwd_scans), one
// workgroup per (block, channel group) at a time, its four waves taking the candidate rounds in turn and adding their
// partial sums in wave order.  The other workgroups: one WAVE per (block, channel group), which returns at once if the
// block is flagged heavy.  counts = [heavy blocks, odd pixels, odd pixels per segment ...].
template <typename T>
__global__ __launch_bounds__(256) void warp_bwd_gather(
    const T *__restrict__ grad_dst, const T *__restrict__ Mv, const WarpScan *__restrict__ scans,
    const int *__restrict__ heavy_list, const int *__restrict__ counts, const int *__restrict__ odd_list, int N, int C, int h,
    int w, int H, int W, int nearest, int lgG, int cgroups, int heavy_wgs, T *__restrict__ grad_src)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ WarpRec<T> recs[4][64];
    __shared__ T part[3][64][4 * VEC];                        // heavy path: the partial sums of waves 1..3
    const int G = 1 << lgG;
    const int tid = threadIdx.x, lane = tid & 63, lg = lane & (G - 1), st = lane >> lgG;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: block, view, matrix and scans live in SGPRs)
    const int bw2 = (w + 1) / 2, bh2 = (h + 1) / 2;
    const int64_t nblk = (int64_t)N * bh2 * bw2;
    const bool heavy = (int)blockIdx.x < heavy_wgs;
    const int64_t nitems = heavy ? (int64_t)counts[0] * cgroups : nblk * cgroups;
    for (int64_t item = heavy ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x - heavy_wgs) * 4 + wv; item < nitems;
         item += heavy ? (int64_t)heavy_wgs : nitems) {
        const int64_t blk = heavy ? (int64_t)heavy_list[item / cgroups] : item / cgroups;
        const int cgi = (int)(item % cgroups);
        const WarpScan sc0 = scans[2 * blk], sc1 = scans[2 * blk + 1];
        if (!heavy && (sc0.swap & 2)) return;                 // a heavy block: the workgroups above take it
        const int n = (int)(blk / ((int64_t)bh2 * bw2));
        const int rem = (int)(blk - (int64_t)n * bh2 * bw2);
        const int by = rem / bw2, bx = rem - by * bw2;
        const int chunk = cgi * G + lg;
        const bool has_ch = chunk < C / VEC;
        T Mn[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Mn[k] = Mv[(int64_t)n * 9 + k];
        T acc[4][VEC];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[t][v] = T(0);
        const int64_t odd_per = ((int64_t)N * H * W + WARP_ODD_SEGS - 1) / WARP_ODD_SEGS;
        warp_gather_rounds<T>(grad_dst + (int64_t)chunk * VEC, Mn, sc0, sc1, counts, odd_list, odd_per, n, bx, by, C, h, w, H,
                              W, nearest, lgG, has_ch, heavy ? wv : 0, heavy ? 4 : 1, recs[wv], acc);
        if (heavy) {
            if (wv > 0 && st == 0)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) part[wv - 1][lg][t * VEC + v] = acc[t][v];
            __syncthreads();
            if (wv == 0 && st == 0)
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[t][v] += part[o][lg][t * VEC + v];
            __syncthreads();
        }
        if (has_ch && st == 0 && (!heavy || wv == 0)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int Y = 2 * by + (t >> 1), X = 2 * bx + (t & 1);
                if (Y < h && X < w) {
                    Pack<T, VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = acc[t][v];
                    o.store(grad_src + (((int64_t)n * h + Y) * w + X) * C + (int64_t)chunk * VEC);
                }
            }
        }
    }
}

// Which kernel the last warp call of this process launched (tests assert the layout routes; bench.py reports it).  Not
// thread-local: autograd runs the backward on its own thread.
static std::atomic<const char *> g_warp_last_kernel{"none"};

static int warp_bwd_impl()
{
    // MVDETR_WARP_BWD_IMPL = gather (default) | scatter: the atomic kernels, kept for comparison and as the fallback for
    // shapes the gather does not take
    const char *e = getenv("MVDETR_WARP_BWD_IMPL");
    return e && !strcmp(e, "scatter") ? 1 : 0;
}

// Scratch of the gather backward, one buffer per (device, stream), grown on demand and kept for the life of the process: every
// use is enqueued on that stream, so reuse is ordered; calls on different streams get different buffers (thread-safe).
// Not for HIP graph capture (the allocation would be captured once and the pointer reused across replays): capturing callers
// use the two-step form (mvdetr_warp_backward_plan_* into their own buffer).  Streams that have been destroyed leave their
// entry behind; mvdetr_warp_release_scratch() drops every entry (call it when no warp backward is in flight).
// (tag, key): the caller's version tag and the shapes of the plan the scratch currently holds (0: none)
struct WarpScratchEntry { char *ptr; size_t size; uint64_t tag; int64_t key[6]; };
static std::mutex g_warp_scratch_mu;
static std::map<std::pair<int, hipStream_t>, WarpScratchEntry> g_warp_scratch;
// `tag` != 0: the caller's promise that calls with the same tag (on this device and stream, with the same shapes) have the same
// matrices; `reuse` is set when the scratch already holds the plan of such a call -- the geometry pass is then skipped.
static char *warp_stream_scratch(hipStream_t st, size_t bytes, hipError_t &rc, uint64_t tag = 0, const int64_t *key = nullptr,
                                 bool *reuse = nullptr)
{
    using Entry = WarpScratchEntry;
    std::mutex &mu = g_warp_scratch_mu;
    auto &cache = g_warp_scratch;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    Entry &e = cache[{dev, st}];
    if (reuse) *reuse = false;
    if (e.size < bytes) {
        if (e.ptr) (void)hipFreeAsync(e.ptr, st);              // (after the work already queued on this stream)
        e.ptr = nullptr;
        e.size = 0;
        e.tag = 0;
        const size_t want = bytes + bytes / 4;
        rc = hipMallocAsync(reinterpret_cast<void **>(&e.ptr), want, st);
        if (rc != hipSuccess) { e.ptr = nullptr; return nullptr; }
        e.size = want;
    }
    if (tag && key && e.tag == tag && !memcmp(e.key, key, sizeof(e.key))) {
        if (reuse) *reuse = true;
    } else {
        e.tag = key ? tag : 0;                               // (an untagged call overwrites the plan: forget the tag)
        if (key) memcpy(e.key, key, sizeof(e.key));
    }
    return e.ptr;
}

// Layout of the gather backward's geometry ("plan"): [counts: heavy, odd, odd per segment | scans: 2 per 2 x 2 block | heavy-
// block list | odd-pixel list].  It depends on the matrices and the shapes only, not on the gradient.
struct WarpPlanLayout {
    size_t count_bytes, scan_bytes, heavy_bytes, odd_bytes;
    int64_t nblk, npix;
    size_t total() const { return count_bytes + scan_bytes + heavy_bytes + odd_bytes; }
};
static WarpPlanLayout warp_plan_layout(int N, int h, int w, int H, int W)
{
    WarpPlanLayout L;
    L.nblk = (int64_t)N * ((h + 1) / 2) * ((w + 1) / 2);
    L.npix = (int64_t)N * H * W;
    L.count_bytes = ((2 + WARP_ODD_SEGS) * sizeof(int) + 15) / 16 * 16;
    L.scan_bytes = (size_t)L.nblk * 2 * sizeof(WarpScan);
    L.heavy_bytes = ((size_t)L.nblk * sizeof(int) + 15) / 16 * 16;
    const int64_t odd_per = (L.npix + WARP_ODD_SEGS - 1) / WARP_ODD_SEGS;
    L.odd_bytes = (size_t)odd_per * WARP_ODD_SEGS * sizeof(int);
    return L;
}

// does the gather take these shapes?  (lgG / cgroups: lanes per hit stream, channel groups per block)
template <typename T> static bool warp_gather_shapes(int N, int C, int h, int w, int H, int W, int &lgG, int &cgroups, int64_t &wgs)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    if (C <= 0 || C % VEC) return false;
    const int chunks = C / VEC;
    lgG = 0;
    while ((1 << lgG) < chunks && lgG < 6) ++lgG;
    const int G = 1 << lgG;
    cgroups = (chunks + G - 1) / G;
    const WarpPlanLayout L = warp_plan_layout(N, h, w, H, W);
    wgs = (L.nblk * cgroups + 3) / 4;
    return wgs <= 0x7fffffffLL && L.npix * C <= 0x7fffffffLL;
}

// the geometry pass: candidate scans of every block, the heavy-block list, the pixels kornia does not divide
template <typename T>
static int warp_bwd_plan_launch(hipStream_t st, const T *Mv, int N, int h, int w, int H, int W, char *plan)
{
    const WarpPlanLayout L = warp_plan_layout(N, h, w, H, W);
    const char *ge = getenv("MVDETR_WARP_BWD_GEOMETRY");
    const int force_clip = ge && !strcmp(ge, "clip");
    int *counts = reinterpret_cast<int *>(plan);
    WarpScan *scans = reinterpret_cast<WarpScan *>(plan + L.count_bytes);
    int *heavy_list = reinterpret_cast<int *>(plan + L.count_bytes + L.scan_bytes);
    int *odd_list = reinterpret_cast<int *>(plan + L.count_bytes + L.scan_bytes + L.heavy_bytes);
    // the two running counters start from zero: one 8-byte stream write (a command-processor packet, no fill kernel)
    hipError_t rc = hipStreamWriteValue64(st, counts, 0, 0);
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        rc = hipMemsetAsync(counts, 0, 2 * sizeof(int), st);
        if (rc != hipSuccess) return (int)rc;
    }
    const char *he = getenv("MVDETR_WARP_BWD_HEAVY");        // (test knob: 0 = every block with candidates is "heavy")
    const int heavy_above = he ? atoi(he) : WARP_HEAVY;
    const int64_t scan_wgs = (L.nblk + WARP_SCAN_THREADS - 1) / WARP_SCAN_THREADS;
    hipLaunchKernelGGL((warp_bwd_scans<T>), dim3((unsigned)(scan_wgs + WARP_ODD_SEGS)), dim3(WARP_SCAN_THREADS), 0, st, Mv, N,
                       h, w, H, W, force_clip, heavy_above, scans, heavy_list, counts, odd_list);
    return (int)hipGetLastError();
}

// the gradient from a plan of the same matrices and shapes (the plan is only read)
template <typename T>
static int warp_bwd_planned_launch(hipStream_t st, const T *grad_dst, const T *Mv, const char *plan, int N, int C, int h,
                                   int w, int H, int W, int nearest, T *grad_src)
{
    int lgG, cgroups;
    int64_t wgs;
    if (!warp_gather_shapes<T>(N, C, h, w, H, W, lgG, cgroups, wgs)) return (int)hipErrorNotSupported;
    const WarpPlanLayout L = warp_plan_layout(N, h, w, H, W);
    const int *counts = reinterpret_cast<const int *>(plan);
    const WarpScan *scans = reinterpret_cast<const WarpScan *>(plan + L.count_bytes);
    const int *heavy_list = reinterpret_cast<const int *>(plan + L.count_bytes + L.scan_bytes);
    const int *odd_list = reinterpret_cast<const int *>(plan + L.count_bytes + L.scan_bytes + L.heavy_bytes);
    const char *hw = getenv("MVDETR_WARP_BWD_HEAVY_WGS");
    const int heavy_wgs = hw ? (atoi(hw) > 0 ? atoi(hw) : 1) : WARP_HEAVY_WGS;
    hipLaunchKernelGGL((warp_bwd_gather<T>), dim3((unsigned)(wgs + heavy_wgs)), dim3(256), 0, st, grad_dst, Mv, scans,
                       heavy_list, counts, odd_list, N, C, h, w, H, W, nearest, lgG, cgroups, heavy_wgs, grad_src);
    return (int)hipGetLastError();
}

template <typename T>
static int warp_bwd_gather_launch(hipStream_t st, const T *grad_dst, const T *Mv, int N, int C, int h, int w, int H,
                                  int W, int nearest, T *grad_src, uint64_t tag)
{
    int lgG, cgroups;
    int64_t wgs;
    if (!warp_gather_shapes<T>(N, C, h, w, H, W, lgG, cgroups, wgs)) return (int)hipErrorNotSupported;
    // the plan of this call in stream-ordered scratch (kept per stream between calls: hipMallocAsync + hipFreeAsync cost the
    // host ~10 us per call, more than the launches).  A caller-supplied version tag of the matrices lets consecutive calls
    // reuse the plan itself (mvdetr_warp_perspective_backward_tagged_*): the gather is then the only launch.  Callers who own
    // a plan buffer use mvdetr_warp_backward_plan_* / mvdetr_warp_perspective_backward_planned_* instead.
    hipError_t rc = hipSuccess;
    const char *ge = getenv("MVDETR_WARP_BWD_GEOMETRY"), *he = getenv("MVDETR_WARP_BWD_HEAVY");
    const int64_t key[6] = {N, h, w, ((int64_t)H << 32) | (unsigned)W, (int64_t)sizeof(T),
                            (int64_t)(ge && !strcmp(ge, "clip")) | ((int64_t)(he ? atoi(he) + 1 : 0) << 8)};
    bool reuse = false;
    char *scratch = warp_stream_scratch(st, warp_plan_layout(N, h, w, H, W).total(), rc, tag, key, &reuse);
    if (!scratch) return (int)rc;
    if (!reuse) {
        const int r1 = warp_bwd_plan_launch<T>(st, Mv, N, h, w, H, W, scratch);
        if (r1) {
            (void)warp_stream_scratch(st, 0, rc, 0, key, nullptr);      // (no plan was built: forget the tag)
            return r1;
        }
    }
    return warp_bwd_planned_launch<T>(st, grad_dst, Mv, scratch, N, C, h, w, H, W, nearest, grad_src);
}

template <typename T>
static int warp_entry(bool backward, void *stream, const T *a, const T *Mv, int N, int C, int h, int w,
                      int H, int W, int nhwc, T *o, uint64_t tag = 0)
{
    if (N < 0 || C < 0 || h <= 0 || w <= 0 || H < 0 || W < 0) return (int)hipErrorInvalidValue;
    if (nhwc & ~7) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)N * H * W;
    if (backward) {
        // grad_src is OVERWRITTEN: the gather stores every element, the scatter kernels start from zeros
        const int64_t nsrc = (int64_t)N * C * h * w;
        if (nsrc == 0) return 0;
        if (!o) return (int)hipErrorInvalidValue;
        if (npix == 0) return (int)hipMemsetAsync(o, 0, (size_t)nsrc * sizeof(T), st);
    }
    if (npix == 0 || C == 0) return 0;
    if (!a || !Mv || !o) return (int)hipErrorInvalidValue;
    const int nearest = (nhwc >> 2) & 1;                   // bit 2: mode='nearest'
    nhwc &= 3;
    const size_t src_bytes = (size_t)N * C * h * w * sizeof(T);
    if (nhwc & 2) {
        // channel-last source: implemented for channel-last destinations, whole 16-byte chunks per pixel and a
        // view that fits 32-bit element offsets
        constexpr int VEC = 16 / (int)sizeof(T);
        if (C % VEC || !aligned(a, 16) || !aligned(o, 16) || (int64_t)h * w * C > 0x7fffffffLL)
            return (int)hipErrorNotSupported;
        if (!(nhwc & 1)) {
            // channel-last source, NCHW destination: forward, fp32 only
            if constexpr (sizeof(T) == 4) {
                if (!backward) {
                    const int64_t nb2 = (int64_t)N * ((H + WARP_NC_TH - 1) / WARP_NC_TH) * ((W + WARP_NC_TW - 1) / WARP_NC_TW);
                    if (nb2 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
                    hipLaunchKernelGGL(warp_fwd_cl_nchw, dim3((unsigned)nb2), dim3(WARP_CL_THREADS), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
                    g_warp_last_kernel = "warp_fwd_cl_nchw";
                    return (int)hipGetLastError();
                }
            }
            return (int)hipErrorNotSupported;
        }
        const int64_t nb = warp_grid(N, H, W, 1);
        if (nb > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        if (!backward) {
            hipLaunchKernelGGL((warp_fwd_cl<T>), dim3((unsigned)nb), dim3(WARP_CL_THREADS), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
            g_warp_last_kernel = "warp_fwd_cl";
            return (int)hipGetLastError();
        }
        if (!warp_bwd_impl()) {
            const int rc = warp_bwd_gather_launch<T>(st, a, Mv, N, C, h, w, H, W, nearest, o, tag);
            if (rc != (int)hipErrorNotSupported) {
                g_warp_last_kernel = "warp_bwd_gather";
                return rc;
            }
        }
        hipError_t rc = hipMemsetAsync(o, 0, src_bytes, st);
        if (rc != hipSuccess) return (int)rc;
        hipLaunchKernelGGL((warp_bwd_cl<T>), dim3((unsigned)nb), dim3(WARP_CL_THREADS), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        g_warp_last_kernel = "warp_bwd_cl";
        return (int)hipGetLastError();
    }
    const int groups = (C + WARP_CH - 1) / WARP_CH;
    const int64_t blocks = warp_grid(N, H, W, groups);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(WARP_PIX * WARP_SUB);
    if (!backward) {
        // NCHW source: 8x8-tile gather kernel for both destination layouts (91 us / 28 % at Wildtrack size for
        // NCHW -> NCHW, the literal layouts of the kornia call; every replacement tried was slower)
        if constexpr (sizeof(T) == 4) {
            // fp32 NCHW -> NCHW (the literal kornia layouts): LDS source patches where whole 16-byte row pieces can be copied
            // (rows of a multiple of 4 texels, aligned base, a view below 2 GiB); MVDETR_WARP_FWD_NCHW=gather keeps the gather kernel
            static const bool patch_ok = [] { const char *e = getenv("MVDETR_WARP_FWD_NCHW"); return !(e && !strcmp(e, "gather")); }();
            if (!nhwc && patch_ok && w % 4 == 0 && aligned(a, 16) && (int64_t)C * h * w * 4 < 0x7fffffffLL) {
                const int64_t pb = (int64_t)N * ((H + WARP_PP_TH - 1) / WARP_PP_TH) * ((W + WARP_PP_TW - 1) / WARP_PP_TW) *
                                   ((C + WARP_PP_CH - 1) / WARP_PP_CH);
                if (pb <= 0x7fffffffLL) {
                    hipLaunchKernelGGL(warp_fwd_nchw_patch, dim3((unsigned)pb), dim3(256), WARP_PP_FLOATS * 4, st, a, Mv, N, C, h, w, H, W, nearest, o);
                    g_warp_last_kernel = "warp_fwd_nchw_patch";
                    return (int)hipGetLastError();
                }
            }
        }
        if (nhwc) hipLaunchKernelGGL((warp_fwd<T, true>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        else hipLaunchKernelGGL((warp_fwd<T, false>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        g_warp_last_kernel = nhwc ? "warp_fwd<NHWC>" : "warp_fwd<NCHW>";
    } else {
        hipError_t rc = hipMemsetAsync(o, 0, src_bytes, st);
        if (rc != hipSuccess) return (int)rc;
        if (nhwc) hipLaunchKernelGGL((warp_bwd<T, true>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        else hipLaunchKernelGGL((warp_bwd<T, false>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        g_warp_last_kernel = nhwc ? "warp_bwd<NHWC>" : "warp_bwd<NCHW>";
    }
    return (int)hipGetLastError();
}

// [n, rows, cols] -> [n, cols, rows] through a 64 x 64 LDS tile (reads and writes both in 256-byte runs): turns an NCHW
// feature map into the channel-last layout the fast warp kernels read (rows = C, cols = h*w) and back
// (rows = h*w, cols = C).
template <typename T>
__global__ __launch_bounds__(256) void transpose_tiles(const T *__restrict__ src, int rows, int cols, T *__restrict__ dst)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int64_t base = (int64_t)blockIdx.z * rows * cols;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * k][tx] = src[base + (int64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[base + (int64_t)c * rows + r] = tile[tx][ty + 4 * k];
    }
}

template <typename T> static int transpose_entry(void *stream, const T *src, int n, int rows, int cols, T *dst)
{
    if (n < 0 || rows < 0 || cols < 0) return (int)hipErrorInvalidValue;
    if ((int64_t)n * rows * cols == 0) return 0;
    if (!src || !dst || n > 65535 || (rows + 63) / 64 > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((transpose_tiles<T>), dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)n), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, rows, cols, dst);
    return (int)hipGetLastError();
}

// ---- the two-step gradient (plan + planned call) behind the C ABI -----------------------------------------------------
static int64_t plan_bytes(int n, int channels, int src_h, int src_w, int dst_h, int dst_w, int elem_size)
{
    if (n <= 0 || channels <= 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return 0;
    int lgG, cgroups;
    int64_t wgs;
    const bool ok = elem_size == 4   ? warp_gather_shapes<float>(n, channels, src_h, src_w, dst_h, dst_w, lgG, cgroups, wgs)
                    : elem_size == 8 ? warp_gather_shapes<double>(n, channels, src_h, src_w, dst_h, dst_w, lgG, cgroups, wgs)
                                     : false;
    if (!ok || (int64_t)src_h * src_w * channels > 0x7fffffffLL || warp_bwd_impl()) return 0;
    return (int64_t)warp_plan_layout(n, src_h, src_w, dst_h, dst_w).total();
}

template <typename T>
static int plan_entry(void *stream, const T *M, int n, int channels, int src_h, int src_w, int dst_h, int dst_w, void *plan)
{
    if (!M || !plan || (reinterpret_cast<uintptr_t>(plan) & 15)) return (int)hipErrorInvalidValue;
    if (!plan_bytes(n, channels, src_h, src_w, dst_h, dst_w, (int)sizeof(T))) return (int)hipErrorNotSupported;
    return warp_bwd_plan_launch<T>(reinterpret_cast<hipStream_t>(stream), M, n, src_h, src_w, dst_h, dst_w,
                                           static_cast<char *>(plan));
}
template <typename T>
static int planned_entry(void *stream, const T *grad_dst, const T *M, const void *plan, int n, int channels, int src_h, int src_w,
                         int dst_h, int dst_w, int layout_nhwc, T *grad_src)
{
    // channel-last on both sides (bits 0 and 1), bilinear or nearest (bit 2): the layouts the gather exists for
    if ((layout_nhwc & ~7) || (layout_nhwc & 3) != 3 || !grad_dst || !M || !plan || !grad_src) return (int)hipErrorInvalidValue;
    if (!plan_bytes(n, channels, src_h, src_w, dst_h, dst_w, (int)sizeof(T)) ||
        !aligned(grad_dst, 16) || !aligned(grad_src, 16))
        return (int)hipErrorNotSupported;
    g_warp_last_kernel = "warp_bwd_gather[planned]";
    return warp_bwd_planned_launch<T>(reinterpret_cast<hipStream_t>(stream), grad_dst, M, static_cast<const char *>(plan), n,
                                              channels, src_h, src_w, dst_h, dst_w, (layout_nhwc >> 2) & 1, grad_src);
}

}  // namespace mvdetr


extern "C" {

int mvdetr_warp_release_scratch(void)
{
    std::lock_guard<std::mutex> lock(mvdetr::g_warp_scratch_mu);
    int prev = 0, rc = 0;
    (void)hipGetDevice(&prev);
    for (auto &kv : mvdetr::g_warp_scratch) {
        if (!kv.second.ptr) continue;
        (void)hipSetDevice(kv.first.first);
        const hipError_t e = hipFree(kv.second.ptr);           // (synchronises with the device: nothing is still using it)
        if (e != hipSuccess && !rc) rc = (int)e;
    }
    mvdetr::g_warp_scratch.clear();
    (void)hipSetDevice(prev);
    return rc;
}

const char *mvdetr_warp_last_kernel(void) { return mvdetr::g_warp_last_kernel.load(); }

int mvdetr_transpose_f32(void *stream, const float *src, int n, int rows, int cols, float *dst)
{
    return mvdetr::transpose_entry<float>(stream, src, n, rows, cols, dst);
}
int mvdetr_transpose_f64(void *stream, const double *src, int n, int rows, int cols, double *dst)
{
    return mvdetr::transpose_entry<double>(stream, src, n, rows, cols, dst);
}


int mvdetr_warp_perspective_forward_f32(void *stream, const float *src, const float *M, int n,
                                        int channels, int src_h, int src_w, int dst_h, int dst_w,
                                        int layout_nhwc, float *dst)
{
    return mvdetr::warp_entry<float>(false, stream, src, M, n, channels, src_h, src_w, dst_h, dst_w,
                                     layout_nhwc, dst);
}

int mvdetr_warp_perspective_forward_f64(void *stream, const double *src, const double *M, int n,
                                        int channels, int src_h, int src_w, int dst_h, int dst_w,
                                        int layout_nhwc, double *dst)
{
    return mvdetr::warp_entry<double>(false, stream, src, M, n, channels, src_h, src_w, dst_h, dst_w,
                                      layout_nhwc, dst);
}

int mvdetr_warp_perspective_backward_f32(void *stream, const float *grad_dst, const float *M, int n,
                                         int channels, int src_h, int src_w, int dst_h, int dst_w,
                                         int layout_nhwc, float *grad_src)
{
    return mvdetr::warp_entry<float>(true, stream, grad_dst, M, n, channels, src_h, src_w, dst_h, dst_w,
                                     layout_nhwc, grad_src);
}

int mvdetr_warp_perspective_backward_tagged_f32(void *stream, const float *grad_dst, const float *M, int n, int channels,
                                                int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                uint64_t matrices_tag, float *grad_src)
{
    return mvdetr::warp_entry<float>(true, stream, grad_dst, M, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, grad_src,
                                     matrices_tag);
}
int mvdetr_warp_perspective_backward_tagged_f64(void *stream, const double *grad_dst, const double *M, int n, int channels,
                                                int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                uint64_t matrices_tag, double *grad_src)
{
    return mvdetr::warp_entry<double>(true, stream, grad_dst, M, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, grad_src,
                                      matrices_tag);
}

int64_t mvdetr_warp_backward_plan_bytes(int n, int channels, int src_h, int src_w, int dst_h, int dst_w, int elem_size)
{
    return mvdetr::plan_bytes(n, channels, src_h, src_w, dst_h, dst_w, elem_size);
}

int mvdetr_warp_backward_plan_f32(void *stream, const float *M, int n, int channels, int src_h, int src_w, int dst_h, int dst_w,
                                  void *plan)
{
    return mvdetr::plan_entry<float>(stream, M, n, channels, src_h, src_w, dst_h, dst_w, plan);
}
int mvdetr_warp_backward_plan_f64(void *stream, const double *M, int n, int channels, int src_h, int src_w, int dst_h, int dst_w,
                                  void *plan)
{
    return mvdetr::plan_entry<double>(stream, M, n, channels, src_h, src_w, dst_h, dst_w, plan);
}
int mvdetr_warp_perspective_backward_planned_f32(void *stream, const float *grad_dst, const float *M, const void *plan, int n,
                                                 int channels, int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                 float *grad_src)
{
    return mvdetr::planned_entry<float>(stream, grad_dst, M, plan, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, grad_src);
}
int mvdetr_warp_perspective_backward_planned_f64(void *stream, const double *grad_dst, const double *M, const void *plan, int n,
                                                 int channels, int src_h, int src_w, int dst_h, int dst_w, int layout_nhwc,
                                                 double *grad_src)
{
    return mvdetr::planned_entry<double>(stream, grad_dst, M, plan, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, grad_src);
}

int mvdetr_warp_perspective_backward_f64(void *stream, const double *grad_dst, const double *M,
                                         int n, int channels, int src_h, int src_w, int dst_h,
                                         int dst_w, int layout_nhwc, double *grad_src)
{
    return mvdetr::warp_entry<double>(true, stream, grad_dst, M, n, channels, src_h, src_w, dst_h, dst_w,
                                      layout_nhwc, grad_src);
}

}  // extern "C"
