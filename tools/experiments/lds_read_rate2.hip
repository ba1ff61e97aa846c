// Micro-experiment (round 4): ds_read_b128 throughput of the forward's tap reads on a [18 x 28 tokens][32 floats] window when
// every tap is displaced independently (SURVEY 8d's microbenchmark input), for two lane layouts:
//   A  two lanes per (cell, 128-byte slice), each reading the four 16-byte chunks of its half at the four corners, chunk order
//      rotated by the cell's position (msda_fwd_group2's layout: conflict-free while neighbouring cells' taps stay neighbours)
//   B  eight lanes per (cell, slice), each reading ONE chunk at the four corners: the eight lanes of a tap cover a token's whole
//      slice, so a quarter-wave reads two whole tokens wherever they lie
// Same bytes per tap (4 corners x 128 B).  hipcc -O3 --offload-arch=gfx950 lds_read_rate2.hip -o /tmp/lrr2 && /tmp/lrr2
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int WH = 18, WW = 28, SLICE = 32, NTOK = WH * WW;

__device__ __forceinline__ unsigned hashu(unsigned a, unsigned b)
{
    unsigned h = (a * 2654435761u) ^ (b * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return h;
}

// MODE 0 / 1: layout A without / with displacement; MODE 2 / 3: layout B without / with displacement
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    for (int i = threadIdx.x; i < NTOK * SLICE; i += 256) win[i] = (float)i;
    __syncthreads();
    const int tid = threadIdx.x;
    constexpr bool B8 = MODE >= 2, DISP = MODE & 1;
    const int cell = B8 ? tid >> 3 : tid >> 1;                 // 32 (B) or 128 (A) cells per workgroup pass
    const int part = B8 ? tid & 7 : tid & 1;
    const int cx = cell % 16, cy = (cell / 16) % 6;            // position in the 6 x 16 tile
    const int rot = (cx / 2) & 3;
    // token offsets of 8 precomputed taps (floats), displaced by +-3 x +-3 tokens when DISP
    int base[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const unsigned h = hashu(cell + 1000 * blockIdx.x, u);
        const int jx = DISP ? (int)(h % 7) - 3 : 0, jy = DISP ? (int)((h >> 8) % 7) - 3 : 0;
        base[u] = ((6 + cy + jy) * WW + 6 + cx + jx) * SLICE;
    }
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int off = base[u];
            asm volatile("" : "+v"(off));                        // (the reads must stay inside the loop)
            const float *p = win + off;
            if (B8) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 v = *reinterpret_cast<const float4 *>(p + (c & 1) * SLICE + (c >> 1) * WW * SLICE + part * 4);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float4 v = *reinterpret_cast<const float4 *>(p + (c & 1) * SLICE + (c >> 1) * WW * SLICE + part * 16 + ((kk ^ rot) << 2));
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}

template <int MODE> void run(const char *name, float *d_out)
{
    const int blocks = 512, lds = NTOK * SLICE * 4;
    const int iters = MODE >= 2 ? 4000 : 1000;                  // same bytes per workgroup: B reads a quarter as much per lane-iteration
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 8 * (MODE >= 2 ? 4 : 16) * 16;
    printf("%-64s %8.3f ms  %7.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, ms, bytes / (ms * 1e-3) / 2.4e9 / 256);
}

int main()
{
    float *d_out;
    (void)hipMalloc(&d_out, 64);
    run<0>("A: 2 lanes per tap, rotated chunks, taps undisplaced", d_out);
    run<1>("A: 2 lanes per tap, rotated chunks, taps displaced +-3 px", d_out);
    run<2>("B: 8 lanes per tap, one chunk each, taps undisplaced", d_out);
    run<3>("B: 8 lanes per tap, one chunk each, taps displaced +-3 px", d_out);
    return 0;
}
