/* Type-generic body of the C oracle; included twice by oracle.c with
 *   T      = float | double
 *   FN(x)  = x##_f32 | x##_f64
 * TEST INFRASTRUCTURE ONLY (see oracle.c).  */

/* One bilinear tap with grid_sample(bilinear, zeros, align_corners=False) semantics, the sampler
 * the reference's CPU fallback calls (ms_deform_attn_func.py:55-56) and the one kornia ends in
 * (mvdetr.py:194-195).  `plane` points at element (y=0,x=0) of the wanted channel;
 * consecutive x are `sx` elements apart, consecutive y `sy`.  Each corner is bounds-checked on its
 * own (zero padding), like ms_deform_attn_im2col_bilinear (ms_deform_im2col_cuda.cuh:33-84). */
static inline T FN(tap)(const T *plane, int64_t H, int64_t W, int64_t sy, int64_t sx, T y, T x)
{
    T fy = FLOOR(y), fx = FLOOR(x);
    int64_t y0 = (int64_t)fy, x0 = (int64_t)fx, y1 = y0 + 1, x1 = x0 + 1;
    T wx1 = x - fx, wy1 = y - fy, wx0 = (T)1 - wx1, wy0 = (T)1 - wy1;
    T acc = 0;
    if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) acc += plane[y0 * sy + x0 * sx] * (wy0 * wx0);
    if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) acc += plane[y0 * sy + x1 * sx] * (wy0 * wx1);
    if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) acc += plane[y1 * sy + x0 * sx] * (wy1 * wx0);
    if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) acc += plane[y1 * sy + x1 * sx] * (wy1 * wx1);
    return acc;
}

/* grid_sample's un-normalisation for align_corners=False: ((g + 1) * size - 1) / 2 with
 * g = 2*loc - 1 (ms_deform_attn_func.py:47).  Algebraically loc*size - 0.5, the form the
 * reference's CUDA kernel uses (ms_deform_im2col_cuda.cuh:285-286). */
static inline T FN(unnorm)(T loc, int64_t size)
{
    T g = (T)2 * loc - (T)1;
    return ((g + (T)1) * (T)size - (T)1) / (T)2;
}

/* Forward.  value [B,S,M,D], shapes [L,2]=(H,W), lsi [L], loc [B,Lq,M,L,P,2]=(x,y), aw [B,Lq,M,L,P]
 * -> out [B,Lq,M*D].   Restates ms_deform_attn_core_pytorch (ms_deform_attn_func.py:41-61). */
int FN(oracle_msda_forward)(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                            const T *aw, int64_t B, int64_t S, int64_t M, int64_t D, int64_t L,
                            int64_t Lq, int64_t P, T *out)
{
    int64_t BQ = B * Lq;
#pragma omp parallel for schedule(static)
    for (int64_t bq = 0; bq < BQ; ++bq) {
        int64_t b = bq / Lq;
        for (int64_t m = 0; m < M; ++m) {
            const T *l_ptr = loc + ((bq * M + m) * L * P) * 2;
            const T *w_ptr = aw + (bq * M + m) * L * P;
            T *o = out + (bq * M + m) * D;
            for (int64_t c = 0; c < D; ++c) o[c] = 0;
            for (int64_t l = 0; l < L; ++l) {
                int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
                const T *plane = value + ((b * S + lsi[l]) * M + m) * D;
                for (int64_t p = 0; p < P; ++p) {
                    T x = FN(unnorm)(l_ptr[(l * P + p) * 2 + 0], W);
                    T y = FN(unnorm)(l_ptr[(l * P + p) * 2 + 1], H);
                    T a = w_ptr[l * P + p];
                    for (int64_t c = 0; c < D; ++c)
                        o[c] += FN(tap)(plane + c, H, W, W * M * D, M * D, y, x) * a;
                }
            }
        }
    }
    return 0;
}

/* Backward: analytic gradients of the forward above w.r.t. value, loc and aw.
 * Formulas as in ms_deform_attn_col2im_bilinear (ms_deform_im2col_cuda.cuh:87-159):
 *   grad_value[corner] += w_corner * aw * go
 *   grad_aw            = sum_c go * tap
 *   grad_loc           = (W * sum_c d tap/dx * aw * go,  H * sum_c d tap/dy * aw * go)
 * Parallel over (b, m): a head's slice of grad_value is private to it, so the result is
 * deterministic.  Outputs are fully written (no pre-zeroing needed). */
int FN(oracle_msda_backward)(const T *grad_out, const T *value, const int64_t *shapes,
                             const int64_t *lsi, const T *loc, const T *aw, int64_t B, int64_t S,
                             int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P, T *grad_value,
                             T *grad_loc, T *grad_aw)
{
    for (int64_t i = 0; i < B * S * M * D; ++i) grad_value[i] = 0;
    int64_t BM = B * M;
#pragma omp parallel for schedule(static)
    for (int64_t bm = 0; bm < BM; ++bm) {
        int64_t b = bm / M, m = bm % M;
        for (int64_t q = 0; q < Lq; ++q) {
            int64_t bq = b * Lq + q;
            const T *go = grad_out + (bq * M + m) * D;
            for (int64_t l = 0; l < L; ++l) {
                int64_t H = shapes[2 * l], W = shapes[2 * l + 1];
                int64_t sy = W * M * D, sx = M * D;
                const T *plane = value + ((b * S + lsi[l]) * M + m) * D;
                T *gplane = grad_value + ((b * S + lsi[l]) * M + m) * D;
                for (int64_t p = 0; p < P; ++p) {
                    int64_t t = ((bq * M + m) * L + l) * P + p;
                    T x = FN(unnorm)(loc[t * 2 + 0], W);
                    T y = FN(unnorm)(loc[t * 2 + 1], H);
                    T a = aw[t];
                    T fy = FLOOR(y), fx = FLOOR(x);
                    int64_t y0 = (int64_t)fy, x0 = (int64_t)fx, y1 = y0 + 1, x1 = x0 + 1;
                    T wx1 = x - fx, wy1 = y - fy, wx0 = (T)1 - wx1, wy0 = (T)1 - wy1;
                    int v00 = y0 >= 0 && y0 < H && x0 >= 0 && x0 < W;
                    int v01 = y0 >= 0 && y0 < H && x1 >= 0 && x1 < W;
                    int v10 = y1 >= 0 && y1 < H && x0 >= 0 && x0 < W;
                    int v11 = y1 >= 0 && y1 < H && x1 >= 0 && x1 < W;
                    T g_a = 0, g_x = 0, g_y = 0;
                    for (int64_t c = 0; c < D; ++c) {
                        T g = go[c];
                        T p00 = v00 ? plane[y0 * sy + x0 * sx + c] : (T)0;
                        T p01 = v01 ? plane[y0 * sy + x1 * sx + c] : (T)0;
                        T p10 = v10 ? plane[y1 * sy + x0 * sx + c] : (T)0;
                        T p11 = v11 ? plane[y1 * sy + x1 * sx + c] : (T)0;
                        g_a += g * (p00 * wy0 * wx0 + p01 * wy0 * wx1 + p10 * wy1 * wx0 + p11 * wy1 * wx1);
                        g_x += g * a * ((p01 - p00) * wy0 + (p11 - p10) * wy1);
                        g_y += g * a * ((p10 - p00) * wx0 + (p11 - p01) * wx1);
                        if (v00) gplane[y0 * sy + x0 * sx + c] += wy0 * wx0 * a * g;
                        if (v01) gplane[y0 * sy + x1 * sx + c] += wy0 * wx1 * a * g;
                        if (v10) gplane[y1 * sy + x0 * sx + c] += wy1 * wx0 * a * g;
                        if (v11) gplane[y1 * sy + x1 * sx + c] += wy1 * wx1 * a * g;
                    }
                    grad_aw[t] = g_a;
                    grad_loc[t * 2 + 0] = (T)W * g_x;
                    grad_loc[t * 2 + 1] = (T)H * g_y;
                }
            }
        }
    }
    return 0;
}

/* ---- kornia.warp_perspective (kornia 0.5.x algorithm; PARITY UNPINNED, see oracle.c) ---------- */

static void FN(mat3_mul)(const T *a, const T *b, T *o)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

static void FN(mat3_inv)(const T *m, T *o)
{
    T c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    T det = m[0] * c0 + m[1] * c1 + m[2] * c2, r = (T)1 / det;
    o[0] = c0 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c1 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c2 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
}

/* src-normalised <- dst-normalised 3x3 for one view: inv( N_dst * M * inv(N_src) ), where
 * N(h,w) = [[2/(w-1),0,-1],[0,2/(h-1),-1],[0,0,1]] (kornia normalize_homography). */
static void FN(src_from_dst_norm)(const T *Mv, int64_t h, int64_t w, int64_t H, int64_t W, T *o)
{
    T eps = (T)1e-14;
    T wd = w == 1 ? eps : (T)(w - 1), hd = h == 1 ? eps : (T)(h - 1);
    T Wd = W == 1 ? eps : (T)(W - 1), Hd = H == 1 ? eps : (T)(H - 1);
    T ns[9] = {(T)2 / wd, 0, -1, 0, (T)2 / hd, -1, 0, 0, 1};
    T nd[9] = {(T)2 / Wd, 0, -1, 0, (T)2 / Hd, -1, 0, 0, 1};
    T nsi[9], t0[9], t1[9];
    FN(mat3_inv)(ns, nsi);
    FN(mat3_mul)(Mv, nsi, t0);
    FN(mat3_mul)(nd, t0, t1);
    FN(mat3_inv)(t1, o);
}

static inline T FN(linspace_m1_p1)(int64_t i, int64_t n)
{
    /* torch.linspace(-1, 1, n)[i]: symmetric evaluation from both ends */
    if (n == 1) return (T)-1;
    T step = (T)2 / (T)(n - 1);
    return i < n / 2 ? (T)-1 + step * (T)i : (T)1 - step * (T)(n - 1 - i);
}

/* src [N,C,h,w], Mv [N,9] (dst pixel <- src pixel), out [N,C,H,W]. */
int FN(oracle_warp_perspective)(const T *src, const T *Mv, int64_t N, int64_t C, int64_t h, int64_t w,
                                int64_t H, int64_t W, T *out)
{
    int64_t NH = N * H;
#pragma omp parallel for schedule(static)
    for (int64_t ni = 0; ni < NH; ++ni) {
        int64_t n = ni / H, i = ni % H;
        T A[9];
        FN(src_from_dst_norm)(Mv + n * 9, h, w, H, W, A);
        T yn = FN(linspace_m1_p1)(i, H);
        for (int64_t j = 0; j < W; ++j) {
            T xn = FN(linspace_m1_p1)(j, W);
            T px = A[0] * xn + A[1] * yn + A[2];
            T py = A[3] * xn + A[4] * yn + A[5];
            T pz = A[6] * xn + A[7] * yn + A[8];
            T s = FABS(pz) > (T)1e-8 ? (T)1 / pz : (T)1;
            T x = ((px * s + (T)1) * (T)w - (T)1) / (T)2;
            T y = ((py * s + (T)1) * (T)h - (T)1) / (T)2;
            for (int64_t c = 0; c < C; ++c)
                out[((n * C + c) * H + i) * W + j] = FN(tap)(src + (n * C + c) * h * w, h, w, w, 1, y, x);
        }
    }
    return 0;
}

/* Gradient of the warp w.r.t. src (the only differentiable input on the path): scatter of
 * grad_out through the same bilinear weights.  grad_src is fully written. */
int FN(oracle_warp_perspective_backward)(const T *grad_out, const T *Mv, int64_t N, int64_t C,
                                         int64_t h, int64_t w, int64_t H, int64_t W, T *grad_src)
{
    for (int64_t i = 0; i < N * C * h * w; ++i) grad_src[i] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        T A[9];
        FN(src_from_dst_norm)(Mv + n * 9, h, w, H, W, A);
        for (int64_t i = 0; i < H; ++i) {
            T yn = FN(linspace_m1_p1)(i, H);
            for (int64_t j = 0; j < W; ++j) {
                T xn = FN(linspace_m1_p1)(j, W);
                T px = A[0] * xn + A[1] * yn + A[2];
                T py = A[3] * xn + A[4] * yn + A[5];
                T pz = A[6] * xn + A[7] * yn + A[8];
                T s = FABS(pz) > (T)1e-8 ? (T)1 / pz : (T)1;
                T x = ((px * s + (T)1) * (T)w - (T)1) / (T)2;
                T y = ((py * s + (T)1) * (T)h - (T)1) / (T)2;
                T fy = FLOOR(y), fx = FLOOR(x);
                int64_t y0 = (int64_t)fy, x0 = (int64_t)fx, y1 = y0 + 1, x1 = x0 + 1;
                T wx1 = x - fx, wy1 = y - fy, wx0 = (T)1 - wx1, wy0 = (T)1 - wy1;
                for (int64_t c = 0; c < C; ++c) {
                    T g = grad_out[((n * C + c) * H + i) * W + j];
                    T *gs = grad_src + (n * C + c) * h * w;
                    if (y0 >= 0 && y0 < h && x0 >= 0 && x0 < w) gs[y0 * w + x0] += g * wy0 * wx0;
                    if (y0 >= 0 && y0 < h && x1 >= 0 && x1 < w) gs[y0 * w + x1] += g * wy0 * wx1;
                    if (y1 >= 0 && y1 < h && x0 >= 0 && x0 < w) gs[y1 * w + x0] += g * wy1 * wx0;
                    if (y1 >= 0 && y1 < h && x1 >= 0 && x1 < w) gs[y1 * w + x1] += g * wy1 * wx1;
                }
            }
        }
    }
    return 0;
}
