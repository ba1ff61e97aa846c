// EXPERIMENT, not part of libmvdetr_ops.so (round 2; measured slower than the 8x8-tile gather kernel: 112 us best case,
// 504 us in the form below, against 91 us -- DESIGN.md section 4.4).  Kept for the record: it needs the helpers of
// mvdetr_amd/csrc/warp_perspective.hip (source_position, make_coord, load_pair) to compile.
// Known defect: nearest mode blends all four LDS corners with zeroed weights (0 * Inf = NaN).
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

