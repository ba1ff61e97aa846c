// EXPERIMENT (round 3), not part of libmvdetr_ops.so.  grad_value by sorting the corner hits inside the tile and gathering,
// instead of msda_bwd_value_win's per-channel LDS atomics.  Parity-green (tests/test_msda_gpu.py backward tests) but SLOWER:
// whole backward 1,059 us against 729 us at Wildtrack size (this kernel ~725 us against 394 us).  Phase stamps of one
// workgroup (tools/experiments/vs_trace.py, -DMVDETR_VS_TRACE), per source level: count 1.7 us (with the sampling data
// prefetched a level ahead and the far taps done cooperatively; 10-15 us before), scan 0.4, place 1.8, GATHER 13.7, store +
// flush 5 -- the gather is a chain of dependent LDS reads (entry -> grad_out row) per (token, quad) lane, lists of 16
// entries on average and ~50 at worst in a wave, 7 lists per lane; running the 7 lists of a lane in lock-step (the
// version below) made it 29 us because every list then runs to the wave's longest.  Its LDS-bandwidth floor is ~3 us per
// level; reaching it needs balanced work per lane (segmented reduction over the entry array), which was not built.
// To try it: copy next to mvdetr_amd/csrc/msda_backward_tile.hip, add to the Makefile, and call msda_backward_value_sort()
// before msda_backward_value_tile() in msda_backward.hip (it returns hipErrorNotSupported for shapes it does not take).
// Multi-scale deformable attention backward, grad_value for deformable-ENCODER calls: SORT inside the tile, then GATHER
// -- gfx950 (MI355X).
//
// msda_bwd_value_win (msda_backward_tile.hip) adds every tap's four corners, channel pair by channel pair, into an LDS window
// with 64-bit fixed-point atomics: 32 ds_add_u64 per tap and 16-channel slice, and with taps displaced by a pixel or two a
// conflicted LDS atomic costs ~4x the conflict-free one -- the kernel waits on the LDS atomic unit (394 us at Wildtrack
// size).  Which window token a corner lands on does not depend on the channel, so here the atomics are spent on the
// GEOMETRY only and the channels are plain FMAs (the warp gradient's lesson, warp_perspective.hip):
//
//   * a job is (tile of 4 x 16 cells, head); the grad_out rows of the tile's cells for ALL query cameras (L x 64 rows of D
//     floats) are staged in LDS once and serve every source level;
//   * per source level l: (A) every (camera, cell) lane computes its four taps' footprints in level l's 16 x 28-token window
//     and COUNTS the corner hits per token (4 integer LDS atomics per tap, channel-independent); (scan) exclusive prefix
//     over the 448 tokens; (B) the same lanes PLACE (row, weight x attention) entries into their token's list (one returning
//     integer atomic per corner); (gather) a lane per (token, 4-channel quad) walks its token's list and accumulates
//     weight * grad_out row from LDS -- plain FMAs, four entries in flight; (flush) the window goes to grad_value with
//     fp32 atomics as whole D-float runs, zeros skipped, like msda_bwd_value_win's;
//   * taps that leave the window go straight to memory as fp32 atomics, so the result is right for any locations.
// Plain fp32 throughout (no fixed point, no bounds pass): non-finite inputs propagate as they do with fp32 atomics.
// The order of a token's sum follows the atomic slots, i.e. it is not deterministic -- like the reference's atomicAdd.
//
// Unequal level shapes / far-flung taps (locality probe) are detected on the device; the launch then runs the lane-group
// backward for all three gradients itself, exactly like msda_bwd_value_win, and the sampling kernel stands down.
//
// Replaces (with msda_backward.hip) ms_deformable_col2im_cuda's grad_value accumulation
// (multiview_detector/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-152,301-920).
#include "common.h"
#include "msda_dispatch.h"
#include "msda_tile.h"
#include "msda_backward_lanes.h"
#include <stdlib.h>
#include <string.h>

#ifdef MVDETR_VS_TRACE
// tuning aid (never in the shipped build): 100 MHz wall-clock stamps of one workgroup, first job
__device__ unsigned long long g_vs_trace[256];
extern "C" int mvdetr_debug_vs_trace(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vs_trace), n * sizeof(unsigned long long));
}
#define VTRACE(i) do { if (blockIdx.x == 8 && threadIdx.x == 0 && t == (int)blockIdx.x && (i) < 256) g_vs_trace[(i)] = wall_clock64(); } while (0)
#else
#define VTRACE(i) do { } while (0)
#endif

namespace mvdetr {

constexpr int VS_TH = 4, VS_TW = 16, VS_R = 6, VS_WH = VS_TH + 2 * VS_R, VS_WW = VS_TW + 2 * VS_R, VS_NTOK = VS_WH * VS_WW;
constexpr int VS_CELLS = VS_TH * VS_TW, VS_MAXL = 8, VS_THREADS = 256;
constexpr int VS_ITEMS = (VS_MAXL * VS_CELLS + VS_THREADS - 1) / VS_THREADS;        // (camera, cell) items per lane

// LDS layout for L query cameras (dynamic: 75 KB at D = 16, L = 7 -> two workgroups per CU)
template <int D> struct VsLds {
    int go, cnt, start, entw, entq, bytes;
    __host__ __device__ explicit VsLds(int L)
    {
        const int ne = L * VS_CELLS * TILE_P * 4;                                   // corner entries per level, at most
        go = 0;                                                                    // staged grad_out rows [L * CELLS][D]
        cnt = go + L * VS_CELLS * D * 4;
        start = cnt + VS_NTOK * 4;
        entw = start + (VS_NTOK + 4) * 4;                                          // weights; later the output tile [NTOK][D]
        entq = entw + (ne * 4 > VS_NTOK * D * 4 ? ne * 4 : VS_NTOK * D * 4);
        bytes = entq + ne * 2;
    }
};

template <int D>
__global__ __launch_bounds__(VS_THREADS, 2) void msda_bwd_value_sort(
    const float *__restrict__ go, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ aw, int B, int S, int M,
    int L, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw,
    const int *__restrict__ local_hits)
{
    constexpr int TH = VS_TH, TW = VS_TW, WH = VS_WH, WW = VS_WW, NTOK = VS_NTOK, CELLS = VS_CELLS, P = TILE_P;
    constexpr int THREADS = VS_THREADS, NQ = D / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const VsLds<D> lay(L);
    float *const go_s = reinterpret_cast<float *>(smem + lay.go);
    int *const cnt = reinterpret_cast<int *>(smem + lay.cnt);
    int *const start = reinterpret_cast<int *>(smem + lay.start);
    float *const ent_w = reinterpret_cast<float *>(smem + lay.entw);
    unsigned short *const ent_q = reinterpret_cast<unsigned short *>(smem + lay.entq);
    float *const out_s = ent_w;                               // [NTOK][D], after the gather has read the weights
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)M * D;

    bool equal = true;
    for (int l = 1; l < L; ++l) equal = equal && shapes[2 * l] == shapes[0] && shapes[2 * l + 1] == shapes[1];
    if (local_hits && *local_hits * 2 < MSDA_PROBE_SAMPLES) equal = false;
    if (!equal) {
        // not this kernel's case: the lane-group backward (msda_backward_lanes.h) does all three gradients here, and the
        // sampling kernel, which sees the same shapes, stands down
        const int64_t total = (int64_t)B * S * M * D;
        for (int64_t base = (int64_t)blockIdx.x * THREADS; base < total; base += (int64_t)gridDim.x * THREADS)
            msda_bwd_lanes_body<float, 1, D, true>(base + tid, go, value, shapes, lsi, loc, aw, B, S, M, D, L, S, P,
                                                   grad_value, grad_loc, grad_aw);
        return;
    }

    const int Hq = (int)shapes[0], Wq = (int)shapes[1];
    const int tcols = (Wq + TW - 1) / TW, per_level = ((Hq + TH - 1) / TH) * tcols;
    const int jobs = per_level * M * B, jobs8 = (jobs + 7) / 8;
    const float fW = (float)Wq, fH = (float)Hq;
    const int nitems = L * CELLS;

    for (int t = blockIdx.x; t < jobs8 * 8; t += gridDim.x) {
        const int job = (t & 7) * jobs8 + (t >> 3);          // XCD k takes a contiguous band of jobs
        if ((t >> 3) >= jobs8 || job >= jobs) continue;
        const int head = job % M, u2 = job / M;               // the heads of a tile run back to back: same sampling rows
        const int tin = u2 % per_level, b = u2 / per_level;
        const int Y0 = (tin / tcols) * TH, X0 = (tin % tcols) * TW;
        const int oy = Y0 + TH / 2 - WH / 2, ox = X0 + TW / 2 - WW / 2;
        const float cx = (float)ox + 0.5f * (WW - 1), cy = (float)oy + 0.5f * (WH - 1);

        // this lane's (camera, cell) items: item = camera * CELLS + cell
        int64_t q_of[VS_ITEMS];
        bool act[VS_ITEMS];
#pragma unroll
        for (int it = 0; it < VS_ITEMS; ++it) {
            const int item = tid + it * THREADS, cam = item / CELLS, c = item - cam * CELLS;
            const int qy = Y0 + c / TW, qx = X0 + c % TW;
            act[it] = item < nitems && qy < Hq && qx < Wq;
            q_of[it] = act[it] ? (int64_t)b * S + lsi[cam] + (int64_t)qy * Wq + qx : 0;
        }
        VTRACE(0);
        __syncthreads();                                      // the previous job is done with go_s / out_s
        // stage the grad_out rows: row `item` = D floats of (camera, cell), zeros for cells outside the level
        for (int i = tid; i < nitems * NQ; i += THREADS) {
            const int item = i / NQ, quad = i - item * NQ, cam = item / CELLS, c = item - cam * CELLS;
            const int qy = Y0 + c / TW, qx = X0 + c % TW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qy < Hq && qx < Wq)
                v = *reinterpret_cast<const float4 *>(go + ((int64_t)b * S + lsi[cam] + (int64_t)qy * Wq + qx) * row + head * D + quad * 4);
            *reinterpret_cast<float4 *>(go_s + item * D + quad * 4) = v;
        }

        // sampling data of (item, level): loaded one level ahead (level l + 1's loads are issued before level l's lists are
        // placed and gathered, so their latency is off the critical path)
        float4 nla[VS_ITEMS], nlb[VS_ITEMS], nwa[VS_ITEMS];
        auto load_level = [&](int l) {
#pragma unroll
            for (int it = 0; it < VS_ITEMS; ++it) {
                const int64_t e0 = ((q_of[it] * M + head) * L + l) * P;          // (inactive items read query 0's data, unused)
                nla[it] = *reinterpret_cast<const float4 *>(loc + e0 * 2);
                nlb[it] = *reinterpret_cast<const float4 *>(loc + e0 * 2 + 4);
                nwa[it] = *reinterpret_cast<const float4 *>(aw + e0);
            }
        };
        load_level(0);

        for (int l = 0; l < L; ++l) {
            const int64_t level_base = ((int64_t)b * S + lsi[l]) * row + head * D;
            for (int i = tid; i < NTOK; i += THREADS) cnt[i] = 0;
            __syncthreads();                                  // counters are zero; (first level: go_s has landed)
            VTRACE(1 + l * 8);

            // ---- A: footprints of this lane's taps, corner hits counted per window token ----
            int tok[VS_ITEMS][P];                             // -1: not in the window
            float w4[VS_ITEMS][P][4];
#pragma unroll
            for (int it = 0; it < VS_ITEMS; ++it) {
                const float4 la = nla[it], lb = nlb[it], wa = nwa[it];
                const float xs[4] = {la.x * fW - 0.5f, la.z * fW - 0.5f, lb.x * fW - 0.5f, lb.z * fW - 0.5f};
                const float ys[4] = {la.y * fH - 0.5f, la.w * fH - 0.5f, lb.y * fH - 0.5f, lb.w * fH - 0.5f};
                const float as[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float x = xs[p], y = ys[p], a = as[p];
                    tok[it][p] = -1;
                    const bool in_window = fabsf(x - cx) < 0.5f * (WW - 1) && fabsf(y - cy) < 0.5f * (WH - 1);
                    if (act[it] && in_window) {
                        const float fx = floorf(x), fy = floorf(y);
                        const int tk = ((int)fy - oy) * WW + ((int)fx - ox);
                        const float wx1 = x - fx, wy1 = y - fy;
                        const float ay1 = wy1 * a, ay0 = a - ay1;
                        const float w01 = ay0 * wx1, w00 = ay0 - w01, w11 = ay1 * wx1, w10 = ay1 - w11;
                        tok[it][p] = tk;
                        w4[it][p][0] = w00; w4[it][p][1] = w01; w4[it][p][2] = w10; w4[it][p][3] = w11;
                        __hip_atomic_fetch_add(cnt + tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cnt + tk + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cnt + tk + WW, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cnt + tk + WW + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // taps outside the window but inside the level: straight to memory, the WAVE working on one of them at
                    // a time with lanes = (corner, channel) -- one atomic instruction per 16 channels of a tap
                    const bool miss = act[it] && !in_window && y > -1.f && x > -1.f && y < fH && x < fW;
                    unsigned long long pend = __ballot(miss);
                    while (pend) {
                        const int src = __ffsll((long long)pend) - 1;
                        pend &= pend - 1;
                        const float sx = __shfl(x, src, 64), sy = __shfl(y, src, 64), sa = __shfl(a, src, 64);
                        const int sitem = __shfl(tid + it * THREADS, src, 64);
                        const Footprint<float> f = footprint(sy, sx, Hq, Wq);
                        const int corner = (tid & 63) >> 4, c16 = tid & 15;
                        const bool cv = corner == 0 ? (f.vy0 && f.vx0) : corner == 1 ? (f.vy0 && f.vx1) : corner == 2 ? (f.vy1 && f.vx0) : (f.vy1 && f.vx1);
                        const float cw = (corner & 2 ? f.wy1 : f.wy0) * (corner & 1 ? f.wx1 : f.wx0) * sa;
                        float *pc = grad_value + level_base + ((int64_t)(f.y0 + (corner >> 1)) * Wq + f.x0 + (corner & 1)) * row;
                        if (cv)
                            for (int ch = c16; ch < D; ch += 16) atomicAdd(pc + ch, cw * go_s[sitem * D + ch]);
                    }
                }
            }
            VTRACE(2 + l * 8);
            __syncthreads();
            VTRACE(3 + l * 8);

            // ---- scan: start[t] = first entry of token t; cnt becomes the running cursor ----
            if (tid < 64) {
                constexpr int PER = NTOK / 64;                // 7 tokens per lane
                static_assert(NTOK % 64 == 0, "tokens per scan lane");
                int local[PER], sum = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) { local[j] = cnt[tid * PER + j]; sum += local[j]; }
                int incl = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (tid >= o) incl += v;
                }
                int run = incl - sum;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    start[tid * PER + j] = run;
                    cnt[tid * PER + j] = run;
                    run += local[j];
                }
                if (tid == 63) start[NTOK] = run;
            }
            __syncthreads();
            VTRACE(4 + l * 8);

            if (l + 1 < L) load_level(l + 1);
            // ---- B: place (row, weight) entries into their token's list ----
#pragma unroll
            for (int it = 0; it < VS_ITEMS; ++it) {
                const unsigned short item = (unsigned short)(tid + it * THREADS);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int tk = tok[it][p];
                    if (tk < 0) continue;
                    const int s0 = __hip_atomic_fetch_add(cnt + tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int s1 = __hip_atomic_fetch_add(cnt + tk + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int s2 = __hip_atomic_fetch_add(cnt + tk + WW, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int s3 = __hip_atomic_fetch_add(cnt + tk + WW + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ent_q[s0] = item; ent_w[s0] = w4[it][p][0];
                    ent_q[s1] = item; ent_w[s1] = w4[it][p][1];
                    ent_q[s2] = item; ent_w[s2] = w4[it][p][2];
                    ent_q[s3] = item; ent_w[s3] = w4[it][p][3];
                }
            }
            __syncthreads();
            VTRACE(5 + l * 8);

            // ---- gather: lane = (token, quad) for GI (token, quad) items; the GI lists advance TOGETHER, two entries each
            //      per step, so that GI independent chains of LDS reads are in flight (a single list is a chain of
            //      dependent reads: entry -> grad_out row) ----
            constexpr int GI = (NTOK * NQ + THREADS - 1) / THREADS;
            float4 acc[GI];
            int e_at[GI], e_end[GI];
            int longest = 0;
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int gi = tid + k * THREADS;
                const int tk = gi < NTOK * NQ ? gi / NQ : 0;
                e_at[k] = gi < NTOK * NQ ? start[tk] : 0;
                e_end[k] = gi < NTOK * NQ ? start[tk + 1] : 0;
                longest = max(longest, e_end[k] - e_at[k]);
            }
            const float *gq = go_s + ((tid % NQ) * 4);        // (THREADS is a multiple of NQ: the quad is the same for every k)
            for (int step = 0; step < longest; step += 2) {
                float w[GI][2];
                int r[GI][2];
#pragma unroll
                for (int k = 0; k < GI; ++k)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = e_at[k] + step + u;
                        const bool on = e < e_end[k];
                        const int ee = on ? e : 0;
                        r[k][u] = ent_q[ee];
                        w[k][u] = on ? ent_w[ee] : 0.f;
                    }
#pragma unroll
                for (int k = 0; k < GI; ++k)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float4 g = *reinterpret_cast<const float4 *>(gq + r[k][u] * D);
                        acc[k].x += w[k][u] * g.x;
                        acc[k].y += w[k][u] * g.y;
                        acc[k].z += w[k][u] * g.z;
                        acc[k].w += w[k][u] * g.w;
                    }
            }
            VTRACE(6 + l * 8);
            __syncthreads();                                  // every list has been read: ent_w becomes the output tile
#pragma unroll
            for (int k = 0; k < GI; ++k) {
                const int gi = tid + k * THREADS;
                if (gi < NTOK * NQ) *reinterpret_cast<float4 *>(out_s + gi * 4) = acc[k];
            }
            __syncthreads();
            VTRACE(7 + l * 8);

            // ---- flush: D-float runs of the touched tokens, fp32 atomics ----
            for (int i = tid; i < NTOK * D; i += THREADS) {
                const float v = out_s[i];
                if (v != 0.f) {
                    const int tk = i / D, ch = i - tk * D;
                    const int gy = oy + tk / WW, gx = ox + tk % WW;
                    // corners outside the level were accumulated like any other and are dropped here (zero padding)
                    if ((unsigned)gy < (unsigned)Hq && (unsigned)gx < (unsigned)Wq)
                        atomicAdd(grad_value + level_base + ((int64_t)gy * Wq + gx) * row + ch, v);
                }
            }
            VTRACE(8 + l * 8);
            __syncthreads();                                  // out_s / cnt are rewritten by the next level
        }
    }
}

template <int D>
static int launch_value_sort(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    const int LDS = VsLds<D>(L).bytes;
    static int blocks_for[VS_MAXL + 1] = {};                  // (benign race: every thread computes the same value)
    if (!blocks_for[L]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_value_sort<D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, VsLds<D>(VS_MAXL).bytes);
        int dev = 0, cus = 256, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, msda_bwd_value_sort<D>, VS_THREADS, LDS) != hipSuccess || per_cu < 1)
            per_cu = 1;
        blocks_for[L] = (cus * per_cu + 7) / 8 * 8;
    }
    const int blocks = blocks_for[L];
    hipLaunchKernelGGL((msda_bwd_value_sort<D>), dim3((unsigned)blocks), dim3(VS_THREADS), LDS, st, go, value, shapes, lsi,
                       loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    return (int)hipGetLastError();
}

// hipErrorNotSupported: not this kernel's shape (the caller falls back to msda_backward_value_tile)
int msda_backward_value_sort(hipStream_t st, const float *go, const float *value, const int64_t *shapes,
                             const int64_t *lsi, const float *loc, const float *aw, int B, int S, int M, int D, int L,
                             float *grad_value, float *grad_loc, float *grad_aw, const int *local_hits)
{
    static const bool enabled = [] { const char *e = getenv("MVDETR_MSDA_BWD_VALUE"); return !(e && !strcmp(e, "window")); }();
    if (!enabled || L > VS_MAXL) return (int)hipErrorNotSupported;
    if (D == 16) return launch_value_sort<16>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    if (D == 32) return launch_value_sort<32>(st, go, value, shapes, lsi, loc, aw, B, S, M, L, grad_value, grad_loc, grad_aw, local_hits);
    return (int)hipErrorNotSupported;
}

}  // namespace mvdetr
