// Micro-experiment: throughput of LDS fp32 atomics (ds_add_f32, no return) on gfx950 under three address
// patterns.  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o /tmp/lar && /tmp/lar
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const int *tok, int iters)
{
    __shared__ long long win64[8192];
    float *win = reinterpret_cast<float *>(win64);
    for (int i = threadIdx.x; i < 16384; i += 256) win[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    long long t0 = clock64();
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int a;
            const int r = tok[(it * 8 + u) * 4 + (lane >> 4)];      // pseudo-random token per 16-lane group
            if (MODE == 0) a = ((it * 8 + u) * 64 + lane) & 16383;                 // consecutive dwords
            else if (MODE == 1) a = ((r * 32) + (lane & 15) + ((lane >> 4) & 1) * 16) & 16383;   // 4 groups x 16 ch at random tokens
            else a = (lane * 32 + u) & 16383;                                      // stride-32: one bank pair
            if (MODE == 4 || MODE == 5) {   // integer atomics: 4x16 random groups / lanes = queries (random tokens, same channel)
                a = MODE == 4 ? ((r * 32) + (lane & 15) + ((lane >> 4) & 1) * 16) & 16383
                              : ((tok[(it * 8 + u) * 4 + (lane & 3)] + lane) * 32 + u) & 16383;
                __hip_atomic_fetch_add(reinterpret_cast<int *>(&win[a]), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 6) {         // 64-bit integer atomics, lanes = queries at random tokens
                a = (((tok[(it * 8 + u) * 4 + (lane & 3)] + lane) * 32 + u * 2) & 16382);
                __hip_atomic_fetch_add(&win64[a >> 1], (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 7) {         // 64-bit integer atomics, consecutive qwords (conflict-free)
                a = (((it * 8 + u) * 64 + lane) * 2) & 16382;
                __hip_atomic_fetch_add(&win64[a >> 1], (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 8) {         // 32-bit integer atomics, consecutive dwords
                a = ((it * 8 + u) * 64 + lane) & 16383;
                __hip_atomic_fetch_add(reinterpret_cast<int *>(&win[a]), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 9 || MODE == 10 || MODE == 11 || MODE == 12) {
                // the production pattern: lanes = 2 rows of 32 cells (row stride 44 tokens), each cell's tap displaced
                // by a few tokens (hash of lane / iteration), one channel plane per instruction
                unsigned hsh = (lane * 2654435761u) ^ ((it * 8 + u) * 40503u);
                hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
                int jx = (int)(hsh % 7) - 3, jy = (int)((hsh >> 8) % 5) - 2;
                if (MODE == 12) { jx = jx > 100 ? 1 : 0; jy = jy > 100 ? 1 : 0; }     // same VALU work, no displacement
                const int tokn = (6 + (lane >> 5) + jy) * 44 + 6 + (lane & 31) + jx;
                if (MODE == 9 || MODE == 12) {
                    a = ((u & 15) * 708 + tokn) & 16383;
                    __hip_atomic_fetch_add(reinterpret_cast<int *>(&win[a]), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (MODE == 10) {
                    a = ((u & 7) * 708 + tokn) & 8191;
                    __hip_atomic_fetch_add(&win64[a], (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    a = ((u & 15) * 708 + (6 + (lane >> 5)) * 44 + 6 + (lane & 31)) & 16383;     // no displacement
                    __hip_atomic_fetch_add(reinterpret_cast<int *>(&win[a]), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else if (MODE == 3) {            // read-modify-write without atomics, same addresses as MODE 1
                a = ((r * 32) + (lane & 15) + ((lane >> 4) & 1) * 16) & 16383;
                win[a] += v;
            } else {
                __hip_atomic_fetch_add(&win[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0);
    if (win[threadIdx.x] == 12345.f) out[0] = 0;
}

template <int MODE> void run(const char *name, float *d_out, int *d_tok)
{
    const int iters = 2000, blocks = 512;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_tok, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_tok, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[512]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    // per CU: 2 workgroups x 4 waves; wave-instructions per CU = 8 * iters * 8
    const double instr_per_cu = 8.0 * iters * 8;
    printf("%-40s %8.3f ms  clock64 ticks/block %.0f   ~%.1f ns per wave-instr per CU  (%.1f G lane-atomics/s chip)\n", name, ms,
           h[1], ms * 1e6 / instr_per_cu, 512.0 * 4 * iters * 8 * 64 / (ms * 1e-3) / 1e9);
}

int main()
{
    float *d_out; int *d_tok;
    hipMalloc(&d_out, 512 * 4);
    const int n = 2000 * 8 * 4;
    int *h = (int *)malloc(n * 4);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % 500; }
    hipMalloc(&d_tok, n * 4); hipMemcpy(d_tok, h, n * 4, hipMemcpyHostToDevice);
    run<0>("ds_add_f32 consecutive dwords", d_out, d_tok);
    run<1>("ds_add_f32 4x16-channel groups, random", d_out, d_tok);
    run<2>("ds_add_f32 stride-32 (same banks)", d_out, d_tok);
    run<3>("plain read-add-write, 4x16 groups", d_out, d_tok);
    run<4>("ds_add_u32 4x16-channel groups, random", d_out, d_tok);
    run<5>("ds_add_u32 lanes = random tokens, 1 chan", d_out, d_tok);
    run<6>("ds_add_u64 lanes = random tokens", d_out, d_tok);
    run<7>("ds_add_u64 consecutive qwords", d_out, d_tok);
    run<8>("ds_add_u32 consecutive dwords", d_out, d_tok);
    run<11>("ds_add_u32 2 rows x 32 cells, undisplaced", d_out, d_tok);
    run<9>("ds_add_u32 2 rows x 32 cells, +-3 x +-2 px", d_out, d_tok);
    run<12>("ds_add_u32 2 rows x 32, hash computed, undisplaced", d_out, d_tok);
    run<10>("ds_add_u64 2 rows x 32 cells, +-3 x +-2 px", d_out, d_tok);
    return 0;
}
