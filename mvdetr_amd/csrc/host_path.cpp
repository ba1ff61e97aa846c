// CPU path of libmvdetr_ops.so: multi-scale deformable attention forward / backward and the perspective warp on
// host tensors (plain C++17, std::thread; no GPU, no PyTorch).
//
// The reference's extension has none -- ms_deform_attn_cpu.cpp:17-41 are stubs that raise "Not implement on cpu" --
// so its model code cannot run the deform_trans path without CUDA, and BASELINE.json's configs[0] ("PyTorch CPU-only")
// exists only for --world_feat conv.  This file makes CPU tensors work behind the same Python face.
// It is product code, separate from the test suite's checker under oracle/ (nothing here includes or calls it); the
// arithmetic follows the reference kernels' definitions:
//   forward   ms_deformable_im2col_gpu_kernel + ms_deform_attn_im2col_bilinear       (ms_deform_im2col_cuda.cuh:237-299, 33-84)
//   backward  ms_deformable_col2im_* + ms_deform_attn_col2im_bilinear                 (cuh:301-920, 87-158)
//   warp      kornia.warp_perspective(bilinear, zeros, align_corners=False)           (call site mvdetr.py:194)
// Work decomposition (different from the GPU kernels on purpose -- no atomics, deterministic results):
//   forward   threads own contiguous ranges of (batch, query);
//   backward  threads own (batch, head) pairs: grad_value[b, :, m, :] is then written by exactly one thread;
//   warp      forward: threads own destination rows; backward: threads own (view, channel) planes of grad_src.
#include "../../include/mvdetr_ops.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

int host_threads()
{
    static const int n = [] {
        if (const char *e = getenv("MVDETR_HOST_THREADS")) return std::max(1, atoi(e));
        if (const char *e = getenv("OMP_NUM_THREADS")) return std::max(1, atoi(e));        // the reference pins this to 1 (main.py:3)
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(hw ? hw : 1, 64);
    }();
    return n;
}

// fn(first, last) over [0, n) split into contiguous chunks, one thread each
template <typename Fn> void parallel_ranges(int64_t n, Fn fn)
{
    const int t = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(n, 1));
    if (t <= 1) {
        fn((int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(t);
    for (int i = 0; i < t; ++i) pool.emplace_back([=] { fn(n * i / t, n * (i + 1) / t); });
    for (auto &th : pool) th.join();
}

// bilinear footprint of a sampling point given in pixel units (loc * size - 0.5): integer corner, weights, validity
template <typename T> struct Tap {
    int y0, x0;
    T w[4];             // corner weights (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    bool ok[4];
    T fy, fx;           // fractional parts
};

template <typename T> inline bool make_tap(T y, T x, int H, int W, Tap<T> &t)
{
    if (!(y > T(-1) && x > T(-1) && y < T(H) && x < T(W))) return false;      // cuh:288 (NaN fails too)
    const T yl = std::floor(y), xl = std::floor(x);
    t.y0 = (int)yl;
    t.x0 = (int)xl;
    t.fy = y - yl;
    t.fx = x - xl;
    const T hy = T(1) - t.fy, hx = T(1) - t.fx;
    t.w[0] = hy * hx;
    t.w[1] = hy * t.fx;
    t.w[2] = t.fy * hx;
    t.w[3] = t.fy * t.fx;
    const bool y0 = t.y0 >= 0, y1 = t.y0 + 1 <= H - 1, x0 = t.x0 >= 0, x1 = t.x0 + 1 <= W - 1;
    t.ok[0] = y0 && x0;
    t.ok[1] = y0 && x1;
    t.ok[2] = y1 && x0;
    t.ok[3] = y1 && x1;
    return true;
}

bool bad(int B, int S, int M, int D, int L, int Lq, int P) { return B < 0 || S < 0 || M < 1 || D < 1 || L < 1 || Lq < 0 || P < 1; }

template <typename T>
int msda_forward_host(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw, int B, int S, int M,
                      int D, int L, int Lq, int P, T *out)
{
    if (bad(B, S, M, D, L, Lq, P) || !shapes || !lsi || (!out && B > 0 && Lq > 0)) return 1;
    const int64_t row = (int64_t)M * D;
    parallel_ranges((int64_t)B * Lq, [=](int64_t first, int64_t last) {
        std::vector<T> acc(D);
        for (int64_t bq = first; bq < last; ++bq) {
            const int64_t b = bq / Lq;
            for (int m = 0; m < M; ++m) {
                std::fill(acc.begin(), acc.end(), T(0));
                const T *lp = loc + (bq * M + m) * (int64_t)L * P * 2;
                const T *wp = aw + (bq * M + m) * (int64_t)L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                    const T *plane = value + (b * S + lsi[l]) * row + (int64_t)m * D;
                    for (int p = 0; p < P; ++p) {
                        Tap<T> t;
                        if (!make_tap(lp[(l * P + p) * 2 + 1] * T(H) - T(0.5), lp[(l * P + p) * 2] * T(W) - T(0.5), H, W, t)) continue;
                        const T a = wp[l * P + p];
                        for (int k = 0; k < 4; ++k) {
                            if (!t.ok[k]) continue;
                            const T *v = plane + ((int64_t)(t.y0 + (k >> 1)) * W + t.x0 + (k & 1)) * row;
                            const T wk = t.w[k] * a;
                            for (int c = 0; c < D; ++c) acc[c] += wk * v[c];
                        }
                    }
                }
                std::copy(acc.begin(), acc.end(), out + bq * row + (int64_t)m * D);
            }
        }
    });
    return 0;
}

template <typename T>
int msda_backward_host(const T *go, const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw, int B,
                       int S, int M, int D, int L, int Lq, int P, T *gv, T *gl, T *ga)
{
    if (bad(B, S, M, D, L, Lq, P) || !shapes || !lsi) return 1;
    const int64_t row = (int64_t)M * D;
    // grad_value is accumulated: the caller hands it over zeroed (like the reference's zeros_like, cu:121)
    parallel_ranges((int64_t)B * M, [=](int64_t first, int64_t last) {
        for (int64_t bm = first; bm < last; ++bm) {
            const int64_t b = bm / M;
            const int m = (int)(bm % M);
            for (int64_t q = 0; q < Lq; ++q) {
                const int64_t bq = b * Lq + q;
                const T *g = go + bq * row + (int64_t)m * D;
                const T *lp = loc + (bq * M + m) * (int64_t)L * P * 2;
                const T *wp = aw + (bq * M + m) * (int64_t)L * P;
                T *glp = gl + (bq * M + m) * (int64_t)L * P * 2;
                T *gap = ga + (bq * M + m) * (int64_t)L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                    const T *plane = value + (b * S + lsi[l]) * row + (int64_t)m * D;
                    T *gplane = gv + (b * S + lsi[l]) * row + (int64_t)m * D;
                    for (int p = 0; p < P; ++p) {
                        T d_a = 0, d_x = 0, d_y = 0;
                        Tap<T> t;
                        if (make_tap(lp[(l * P + p) * 2 + 1] * T(H) - T(0.5), lp[(l * P + p) * 2] * T(W) - T(0.5), H, W, t)) {
                            const T a = wp[l * P + p];
                            // d(weight_k)/dy, d(weight_k)/dx of the four corners (cuh:115-152)
                            const T dy[4] = {-(T(1) - t.fx), -t.fx, T(1) - t.fx, t.fx};
                            const T dx[4] = {-(T(1) - t.fy), T(1) - t.fy, -t.fy, t.fy};
                            for (int k = 0; k < 4; ++k) {
                                if (!t.ok[k]) continue;
                                const int64_t tok = ((int64_t)(t.y0 + (k >> 1)) * W + t.x0 + (k & 1)) * row;
                                const T *v = plane + tok;
                                T *gvk = gplane + tok;
                                T dot = 0;
                                const T wk = t.w[k] * a;
                                for (int c = 0; c < D; ++c) {
                                    dot += g[c] * v[c];
                                    gvk[c] += wk * g[c];
                                }
                                d_a += t.w[k] * dot;
                                d_y += dy[k] * dot;
                                d_x += dx[k] * dot;
                            }
                            d_x *= a * T(W);
                            d_y *= a * T(H);
                        }
                        gap[l * P + p] = d_a;
                        glp[(l * P + p) * 2] = d_x;
                        glp[(l * P + p) * 2 + 1] = d_y;
                    }
                }
            }
        }
    });
    return 0;
}

// 3x3 inverse in double; false if singular
bool invert3(const double *m, double *inv)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(std::fabs(det) > 0)) return false;
    const double r = 1.0 / det;
    inv[0] = c00 * r;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * r;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    inv[3] = c01 * r;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * r;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    inv[6] = c02 * r;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * r;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * r;
    return true;
}

// where destination pixel (i, j) samples the source, in source pixel units of grid_sample(align_corners=False):
// kornia normalises with (size - 1) and samples with align_corners=False, hence the size / (size - 1) factor
// (the same quirk the HIP kernel reproduces)
inline void source_position(const double *inv, int i, int j, int sh, int sw, double &y, double &x, bool &finite)
{
    // p = M^-1 (j, i, 1), homogeneous; normalised like kornia does (2 p / (size - 1) - 1, scaled by z), divided only
    // where |z| > 1e-8, then mapped by the align_corners=False sampler: ((g + 1) * size - 1) / 2
    const double px = inv[0] * j + inv[1] * i + inv[2], py = inv[3] * j + inv[4] * i + inv[5], pz = inv[6] * j + inv[7] * i + inv[8];
    const double wd = sw == 1 ? 1e-14 : (double)(sw - 1), hd = sh == 1 ? 1e-14 : (double)(sh - 1);
    const double qx = 2.0 * px / wd - pz, qy = 2.0 * py / hd - pz;
    const double s = std::fabs(pz) > 1e-8 ? 1.0 / pz : 1.0;
    x = ((qx * s + 1.0) * sw - 1.0) * 0.5;
    y = ((qy * s + 1.0) * sh - 1.0) * 0.5;
    finite = std::isfinite(x) && std::isfinite(y);
}

// mode: 0 bilinear, 1 nearest.  layout bit 0: dst NHWC, bit 1: src NHWC
template <typename T, bool BACKWARD>
int warp_host(const T *src_or_gdst, const T *mats, int n, int C, int sh, int sw, int dh, int dw, int layout, int mode, T *dst_or_gsrc)
{
    if (n < 0 || C < 1 || sh < 1 || sw < 1 || dh < 1 || dw < 1) return 1;
    const bool dst_nhwc = layout & 1, src_nhwc = layout & 2;
    auto s_idx = [=](int v, int c, int y, int x) { return src_nhwc ? (((int64_t)v * sh + y) * sw + x) * C + c : (((int64_t)v * C + c) * sh + y) * sw + x; };
    auto d_idx = [=](int v, int c, int y, int x) { return dst_nhwc ? (((int64_t)v * dh + y) * dw + x) * C + c : (((int64_t)v * C + c) * dh + y) * dw + x; };
    std::vector<double> inv((size_t)n * 9);
    std::vector<char> good(n);
    for (int v = 0; v < n; ++v) {
        double m[9];
        for (int k = 0; k < 9; ++k) m[k] = (double)mats[v * 9 + k];
        good[v] = invert3(m, &inv[(size_t)v * 9]);
    }
    const double *invp = inv.data();
    const char *goodp = good.data();
    // forward: units are destination rows; backward: units are (view, channel) planes of grad_src (race-free scatter)
    const int64_t units = BACKWARD ? (int64_t)n * C : (int64_t)n * dh;
    parallel_ranges(units, [=](int64_t first, int64_t last) {
        for (int64_t u = first; u < last; ++u) {
            const int v = BACKWARD ? (int)(u / C) : (int)(u / dh);
            const int c_only = BACKWARD ? (int)(u % C) : -1;
            const int i0 = BACKWARD ? 0 : (int)(u % dh), i1 = BACKWARD ? dh : i0 + 1;
            for (int i = i0; i < i1; ++i)
                for (int j = 0; j < dw; ++j) {
                    double y, x;
                    bool fin = goodp[v];
                    if (fin) source_position(invp + (size_t)v * 9, i, j, sh, sw, y, x, fin);
                    int ys[4], xs[4], nk = 0;
                    double ws[4];
                    if (fin) {
                        if (mode == 1) {
                            const double ry = std::nearbyint(y), rx = std::nearbyint(x);          // grid_sample 'nearest' rounds half to even
                            if (ry >= 0 && ry < sh && rx >= 0 && rx < sw) { ys[0] = (int)ry; xs[0] = (int)rx; ws[0] = 1.0; nk = 1; }
                        } else if (y > -1 && x > -1 && y < sh && x < sw) {
                            const double yl = std::floor(y), xl = std::floor(x), fy = y - yl, fx = x - xl;
                            const int yy[2] = {(int)yl, (int)yl + 1}, xx[2] = {(int)xl, (int)xl + 1};
                            const double wy[2] = {1 - fy, fy}, wx[2] = {1 - fx, fx};
                            for (int a = 0; a < 2; ++a)
                                for (int b = 0; b < 2; ++b)
                                    if (yy[a] >= 0 && yy[a] < sh && xx[b] >= 0 && xx[b] < sw) { ys[nk] = yy[a]; xs[nk] = xx[b]; ws[nk] = wy[a] * wx[b]; ++nk; }
                        }
                    }
                    if constexpr (!BACKWARD) {
                        for (int c = 0; c < C; ++c) {
                            double acc = 0;
                            for (int k = 0; k < nk; ++k) acc += ws[k] * (double)src_or_gdst[s_idx(v, c, ys[k], xs[k])];
                            dst_or_gsrc[d_idx(v, c, i, j)] = (T)acc;
                        }
                    } else {
                        const T g = src_or_gdst[d_idx(v, c_only, i, j)];
                        for (int k = 0; k < nk; ++k) dst_or_gsrc[s_idx(v, c_only, ys[k], xs[k])] += (T)(ws[k] * (double)g);
                    }
                }
        }
    });
    return 0;
}

}  // namespace

extern "C" {

int mvdetr_msda_forward_host_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const float *sampling_loc, const float *attn_weight, int batch, int spatial_size, int num_heads,
                                 int channels, int num_levels, int num_query, int num_point, float *out)
{
    return msda_forward_host(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch, spatial_size, num_heads,
                             channels, num_levels, num_query, num_point, out);
}
int mvdetr_msda_forward_host_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const double *sampling_loc, const double *attn_weight, int batch, int spatial_size, int num_heads,
                                 int channels, int num_levels, int num_query, int num_point, double *out)
{
    return msda_forward_host(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch, spatial_size, num_heads,
                             channels, num_levels, num_query, num_point, out);
}
int mvdetr_msda_backward_host_f32(const float *grad_output, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const float *sampling_loc, const float *attn_weight, int batch,
                                  int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                                  float *grad_value, float *grad_sampling_loc, float *grad_attn_weight)
{
    return msda_backward_host(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch, spatial_size,
                              num_heads, channels, num_levels, num_query, num_point, grad_value, grad_sampling_loc, grad_attn_weight);
}
int mvdetr_msda_backward_host_f64(const double *grad_output, const double *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start_index, const double *sampling_loc, const double *attn_weight, int batch,
                                  int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point,
                                  double *grad_value, double *grad_sampling_loc, double *grad_attn_weight)
{
    return msda_backward_host(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, batch, spatial_size,
                              num_heads, channels, num_levels, num_query, num_point, grad_value, grad_sampling_loc, grad_attn_weight);
}
int mvdetr_warp_perspective_forward_host_f32(const float *src, const float *mats, int n, int channels, int src_h, int src_w, int dst_h,
                                             int dst_w, int layout_nhwc, int mode, float *dst)
{
    return warp_host<float, false>(src, mats, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, mode, dst);
}
int mvdetr_warp_perspective_forward_host_f64(const double *src, const double *mats, int n, int channels, int src_h, int src_w,
                                             int dst_h, int dst_w, int layout_nhwc, int mode, double *dst)
{
    return warp_host<double, false>(src, mats, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, mode, dst);
}
int mvdetr_warp_perspective_backward_host_f32(const float *grad_dst, const float *mats, int n, int channels, int src_h, int src_w,
                                              int dst_h, int dst_w, int layout_nhwc, int mode, float *grad_src)
{
    return warp_host<float, true>(grad_dst, mats, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, mode, grad_src);
}
int mvdetr_warp_perspective_backward_host_f64(const double *grad_dst, const double *mats, int n, int channels, int src_h, int src_w,
                                              int dst_h, int dst_w, int layout_nhwc, int mode, double *grad_src)
{
    return warp_host<double, true>(grad_dst, mats, n, channels, src_h, src_w, dst_h, dst_w, layout_nhwc, mode, grad_src);
}

}  // extern "C"
