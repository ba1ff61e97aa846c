// Micro-experiment (round 4): LDS integer-atomic throughput of accumulation patterns that are bank-conflict-free BY
// CONSTRUCTION, against the production pattern of msda_bwd_value_win (lanes = cells, channel-major planes, taps displaced by
// iid noise: ~3.6x the conflict-free cost).
//
// Token-major window [tok][8 x u64] = 64 B per token (16 banks), window rows WW tokens with WW = 2 (mod 4): the two rows of a
// bilinear footprint (tokens t, t+1 | t+WW, t+WW+1) are two 128-byte spans that start 2 bank-groups apart, i.e. ONE tap's 4
// corners x 8 channel pairs cover all 64 banks exactly once wherever the tap lies; a wave instruction = 2 taps = 2 passes,
// the minimum for 512 bytes.
//   hipcc -O3 --offload-arch=gfx950 lds_atomic_rate2.hip -o /tmp/lar2 && /tmp/lar2
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int WW = 46, WH = 16, NTOK = WW * WH;       // 736 tokens x 64 B = 47,104 B

__device__ __forceinline__ unsigned hash(unsigned a, unsigned b)
{
    unsigned h = (a * 2654435761u) ^ (b * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return h;
}

// MODE 0: production (lanes = 2 rows x 32 cells, channel-major planes of 708 qwords, displaced taps, u64)
// MODE 1: 2 taps x 4 corners x 8 pairs, token-major, WW = 46, random taps anywhere in the window (u64)
// MODE 2: as 1 with WW = 44 (rows NOT complementary: shows what the row stride buys)
// MODE 3: 8 taps x 8 pairs (one corner per instruction), token-major, random taps (u64)
// MODE 4: 1 tap x 4 corners x 16 channels, u32, token-major 64 B per token, WW = 46 (1 pass)
// MODE 5: MODE 1 + the per-instruction LDS read of (weight, address) and the fixed-point packing VALU work
// MODE 6 / 7: fp32 atomics (ds_add_f32; build with -munsafe-fp-atomics), MODE 4's addresses; 7 adds the entry read + multiply
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ long long win64[];
    __shared__ float2 tapinfo[256];
    for (int i = threadIdx.x; i < NTOK * 8; i += 256) win64[i] = 0;
    if (threadIdx.x < 256) tapinfo[threadIdx.x] = make_float2(0.25f + threadIdx.x * 1e-3f, 0.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // addresses (qword / dword index) and tap-info slots of the 8 instructions of an iteration, computed ONCE: the loop below
    // is the atomics (+ MODE 5's read and packing) only, so that VALU work does not hide the LDS rate
    int addr[8], slot[8];
    const int wv = wave + 4 * blockIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (MODE == 0) {
            const unsigned h = hash(lane + 64 * wv, u);
            const int jx = (int)(h % 7) - 3, jy = (int)((h >> 8) % 5) - 2;
            addr[u] = ((u & 7) * 708 + (6 + (lane >> 5) + jy) * 44 + 6 + (lane & 31) + jx) & 8191;
        } else if (MODE == 1 || MODE == 2 || MODE == 5) {
            constexpr int W = MODE == 2 ? 44 : WW;
            const int tap = lane >> 5, corner = (lane >> 3) & 3, pair = lane & 7;
            const unsigned h = hash(tap + 2 * wv, u);
            const int x0 = (int)(h % (W - 1)), y0 = (int)((h >> 10) % (WH - 1));
            addr[u] = ((y0 + (corner >> 1)) * W + x0 + (corner & 1)) * 8 + pair;
            slot[u] = ((u * 4 + wave) * 2 + tap) * 4 + corner;
        } else if (MODE == 3) {
            const unsigned h = hash((lane >> 3) + 8 * wv, u);
            addr[u] = (int)(h % NTOK) * 8 + (lane & 7);
        } else {
            const int corner = lane >> 4, ch = lane & 15;
            slot[u] = (u * 4 + wave) * 8 + corner;
            const unsigned h = hash(wv, u);
            const int x0 = (int)(h % (WW - 1)), y0 = (int)((h >> 10) % (WH - 1));
            addr[u] = ((y0 + (corner >> 1)) * WW + x0 + (corner & 1)) * 16 + ch;
        }
    }
    const long long t0 = clock64();
    long long v = 1 + lane;
    float g0 = 1.f + lane, g1 = 2.f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 5) {
                const float2 ti = tapinfo[slot[u]];
                const float a = ti.x * g0, b = ti.x * g1;
                const int lo = __float2int_rn(a), hi = __float2int_rn(b) + (lo >> 31);
                v = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
            }
            if (MODE == 6 || MODE == 7) {
                // fp32 atomics (needs -munsafe-fp-atomics for the native ds_add_f32; without it hipcc emits a CAS loop)
                float fv = g0;
                if (MODE == 7) fv = tapinfo[slot[u]].x * g0;
                __hip_atomic_fetch_add(reinterpret_cast<float *>(win64) + addr[u], fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 4)
                __hip_atomic_fetch_add(reinterpret_cast<int *>(win64) + addr[u], (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                __hip_atomic_fetch_add(&win64[addr[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0);
    if (win64[threadIdx.x] == 12345) out[0] = 0;
}

template <int MODE> void run(const char *name, float *d_out, int wgs_per_cu)
{
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    const int lds = NTOK * 64;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)wgs_per_cu * 4 * iters * 8;
    printf("%-66s %d wg/CU %8.3f ms  %.2f ns per wave-atomic per CU\n", name, wgs_per_cu, ms, ms * 1e6 / instr_per_cu);
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 1024 * 4);
    for (int w = 2; w <= 3; ++w) {
        run<0>("u64 production: lanes = cells, channel planes, displaced taps", d_out, w);
        run<1>("u64 2 taps x 4 corners x 8 pairs, token-major, WW=46", d_out, w);
        run<2>("u64 2 taps x 4 corners x 8 pairs, token-major, WW=44", d_out, w);
        run<3>("u64 8 taps x 8 pairs (one corner), token-major", d_out, w);
        run<4>("u32 1 tap x 4 corners x 16 channels, token-major, WW=46", d_out, w);
        run<5>("u64 2 taps x 4 x 8, WW=46 + tapinfo read + fixed-point packing", d_out, w);
        run<6>("f32 1 tap x 4 corners x 16 channels, token-major (ds_add_f32)", d_out, w);
        run<7>("f32 1 tap x 4 x 16 + tapinfo read + multiply", d_out, w);
    }
    return 0;
}
