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

// Block id -> (view, 8x8 destination tile, channel group).  The world grid is not aligned with the image
// axes (a ground-plane row is a slanted, perspective-foreshortened line in the camera image), so 64
// consecutive pixels of one destination ROW gather from ~30 different source cache lines per load
// instruction (measured: 66 M L1 accesses per launch, i.e. the kernel ran at the texture-address rate, and
// neither fewer bytes nor a different XCD assignment changed its 100 us).  An 8x8 destination tile maps to
// a compact source patch instead: a handful of lines per load.
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
    const int lane = threadIdx.x & (WARP_PIX - 1);
    const int sub = threadIdx.x / WARP_PIX;
    WarpBlock wb;
    if (!warp_block(N, H, W, (C + WARP_CH - 1) / WARP_CH, wb)) return;
    const int n = wb.n, i = wb.i0 + lane / WARP_TW, j = wb.j0 + lane % WARP_TW;
    const int64_t pix = ((int64_t)n * H + i) * W + j;
    const int cbase = wb.group * WARP_CH;
    constexpr int CPT = WARP_CH / WARP_SUB;
    const bool live = i < H && j < W;
    SrcCoord sc;
    sc.any = false;
    if (live) {
        double x, y;
        source_position(Mv + (int64_t)n * 9, i, j, h, w, x, y);
        sc = make_coord(x, y, h, w, nearest);
    }
    const T w00 = T(sc.wy0 * sc.wx0), w01 = T(sc.wy0 * sc.wx1);
    const T w10 = T(sc.wy1 * sc.wx0), w11 = T(sc.wy1 * sc.wx1);
    const int64_t plane = (int64_t)h * w;
    const int64_t o00 = (int64_t)sc.y0 * w + sc.x0;
#pragma unroll 4
    for (int k = 0; k < CPT; ++k) {
        const int c = cbase + sub * CPT + k;
        T val = T(0);
        if (live && c < C && sc.any) {
            const T *sp = src + ((int64_t)n * C + c) * plane + o00;
            T a, b, cc, d;
            load_pair(sp, sc.v00, sc.v01, a, b);
            load_pair(sp + w, sc.v10, sc.v11, cc, d);
            val = w00 * a + w01 * b + w10 * cc + w11 * d;
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

// ---- NCHW source -> NCHW destination through LDS source patches (fp32) -------------------------------------------
// The gather kernel above issues one 8-byte gather per (pixel, channel, corner row) -- 10 M wave-level gathers at
// Wildtrack size, each touching a dozen cache lines -- and stores 32-byte row pieces (8x8 tiles): it runs at the
// texture-address rate (90 us, 28 % of the roofline).  The destination grid is ~3x denser than the source, so here a
// workgroup owns an 8 x 32 destination tile (lanes = pixels, x fastest: 128-byte store runs), finds the bounding box
// of the tile's source footprints (a compact patch: a few hundred texels), and per chunk of channels copies the
// patch rows to LDS with coalesced row loads (texels outside the image are stored as zeros: zero padding needs no
// per-corner test afterwards), then every lane blends its four corners from LDS (two ds_read2_b32).  The channel
// chunk adapts to the patch: WARP_PATCH_FLOATS / patch texels, at most 8; a patch that does not fit at all (extreme
// magnification) takes per-lane gathers for that tile.
constexpr int WARP_P_TH = 8, WARP_P_TW = 32, WARP_PATCH_FLOATS = 8192, WARP_P_CC = 8;

__global__ __launch_bounds__(256) void warp_fwd_nchw_patch(
    const float *__restrict__ src, const float *__restrict__ Mv, int N, int C, int h, int w, int H, int W,
    int nearest, float *__restrict__ dst)
{
    __shared__ float patch[WARP_PATCH_FLOATS];
    __shared__ int box[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = (W + WARP_P_TW - 1) / WARP_P_TW, ty = (H + WARP_P_TH - 1) / WARP_P_TH;
    int r = blockIdx.x;
    const int j0 = (r % tx) * WARP_P_TW;
    r /= tx;
    const int i0 = (r % ty) * WARP_P_TH, n = r / ty;
    const int i = i0 + tid / WARP_P_TW, j = j0 + tid % WARP_P_TW;
    const bool live = i < H && j < W;
    SrcCoord sc;
    sc.any = false;
    sc.x0 = sc.y0 = 0;
    if (live) {
        double x, y;
        source_position(Mv + (int64_t)n * 9, i, j, h, w, x, y);
        sc = make_coord(x, y, h, w, nearest);
    }
    const bool use = live && sc.any;
    float w00 = use && sc.v00 ? (float)(sc.wy0 * sc.wx0) : 0.f, w01 = use && sc.v01 ? (float)(sc.wy0 * sc.wx1) : 0.f;
    float w10 = use && sc.v10 ? (float)(sc.wy1 * sc.wx0) : 0.f, w11 = use && sc.v11 ? (float)(sc.wy1 * sc.wx1) : 0.f;

    // bounding box of the footprints [x0, x0+1] x [y0, y0+1] (x0 in [-1, w-1]: may stick out of the image by one)
    int xmin = use ? sc.x0 : 0x3fffffff, xmax = use ? sc.x0 + 1 : -0x3fffffff;
    int ymin = use ? sc.y0 : 0x3fffffff, ymax = use ? sc.y0 + 1 : -0x3fffffff;
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
    const int64_t plane = (int64_t)h * w, oplane = (int64_t)H * W;
    float *const op = dst + (int64_t)n * C * oplane + (int64_t)i * W + j;
    if (xmax < xmin) {                                        // the whole tile misses the image
        if (live)
            for (int c = 0; c < C; ++c) op[c * oplane] = 0.f;
        return;
    }
    const int PW = xmax - xmin + 1, PH = ymax - ymin + 1;
    const int64_t P = (int64_t)PW * PH;
    const float *const sview = src + (int64_t)n * C * plane;
    if (P > WARP_PATCH_FLOATS) {
        // magnification too large for a patch: per-lane gathers (the formulation of warp_fwd)
        if (live) {
            const int64_t o00 = (int64_t)sc.y0 * w + sc.x0;
            for (int c = 0; c < C; ++c) {
                float val = 0.f;
                if (use) {
                    const float *sp = sview + c * plane + o00;
                    float a, b, cc, d;
                    load_pair(sp, sc.v00, sc.v01, a, b);
                    load_pair(sp + w, sc.v10, sc.v11, cc, d);
                    val = w00 * a + w01 * b + w10 * cc + w11 * d;
                }
                op[c * oplane] = val;
            }
        }
        return;
    }
    const int Pi = (int)P;
    const int CC = min(WARP_P_CC, WARP_PATCH_FLOATS / Pi);
    const int o00 = use ? (sc.y0 - ymin) * PW + (sc.x0 - xmin) : 0;
    for (int c0 = 0; c0 < C; c0 += CC) {
        const int nc = min(CC, C - c0);
        // patch rows of nc channels: wave `wave` takes rows wave, wave + 4, ... of the nc * PH rows; lanes = columns
        for (int rr = wave; rr < nc * PH; rr += 4) {
            const int ch = rr / PH, py = rr - ch * PH, sy = ymin + py;
            const float *srow = sview + (int64_t)(c0 + ch) * plane + (int64_t)sy * w;
            float *prow = patch + ch * Pi + py * PW;
            const bool yok = (unsigned)sy < (unsigned)h;
            for (int px = lane; px < PW; px += 64) {
                const int sx = xmin + px;
                prow[px] = yok && (unsigned)sx < (unsigned)w ? srow[sx] : 0.f;
            }
        }
        __syncthreads();
        if (live) {
            const float *pp = patch + o00;
#pragma unroll 4
            for (int ch = 0; ch < nc; ++ch) {
                const float *q = pp + ch * Pi;
                op[(int64_t)(c0 + ch) * oplane] = w00 * q[0] + w01 * q[1] + w10 * q[PW] + w11 * q[PW + 1];
            }
        }
        __syncthreads();
    }
}

template <typename T>
static int warp_entry(bool backward, void *stream, const T *a, const T *Mv, int N, int C, int h, int w,
                      int H, int W, int nhwc, T *o)
{
    if (N < 0 || C < 0 || h <= 0 || w <= 0 || H < 0 || W < 0) return (int)hipErrorInvalidValue;
    const int64_t npix = (int64_t)N * H * W;
    if (npix == 0 || C == 0) return 0;
    if (!a || !Mv || !o) return (int)hipErrorInvalidValue;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nhwc & ~7) return (int)hipErrorInvalidValue;
    const int nearest = (nhwc >> 2) & 1;                   // bit 2: mode='nearest'
    nhwc &= 3;
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
                    return (int)hipGetLastError();
                }
            }
            return (int)hipErrorNotSupported;
        }
        const int64_t nb = warp_grid(N, H, W, 1);
        if (nb > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        if (!backward)
            hipLaunchKernelGGL((warp_fwd_cl<T>), dim3((unsigned)nb), dim3(WARP_CL_THREADS), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        else
            hipLaunchKernelGGL((warp_bwd_cl<T>), dim3((unsigned)nb), dim3(WARP_CL_THREADS), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        return (int)hipGetLastError();
    }
    if constexpr (sizeof(T) == 4) {
        if (!backward && !nhwc && C > 1) {
            // NCHW -> NCHW, fp32: LDS source patches
            const int64_t nb3 = (int64_t)N * ((H + WARP_P_TH - 1) / WARP_P_TH) * ((W + WARP_P_TW - 1) / WARP_P_TW);
            if (nb3 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
            hipLaunchKernelGGL(warp_fwd_nchw_patch, dim3((unsigned)nb3), dim3(256), 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
            return (int)hipGetLastError();
        }
    }
    const int groups = (C + WARP_CH - 1) / WARP_CH;
    const int64_t blocks = warp_grid(N, H, W, groups);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(WARP_PIX * WARP_SUB);
    if (!backward) {
        if (nhwc) hipLaunchKernelGGL((warp_fwd<T, true>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        else hipLaunchKernelGGL((warp_fwd<T, false>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
    } else {
        if (nhwc) hipLaunchKernelGGL((warp_bwd<T, true>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
        else hipLaunchKernelGGL((warp_bwd<T, false>), grid, block, 0, st, a, Mv, N, C, h, w, H, W, nearest, o);
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

}  // namespace mvdetr

extern "C" {

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

int mvdetr_warp_perspective_backward_f64(void *stream, const double *grad_dst, const double *M,
                                         int n, int channels, int src_h, int src_w, int dst_h,
                                         int dst_w, int layout_nhwc, double *grad_src)
{
    return mvdetr::warp_entry<double>(true, stream, grad_dst, M, n, channels, src_h, src_w, dst_h, dst_w,
                                      layout_nhwc, grad_src);
}

}  // extern "C"
