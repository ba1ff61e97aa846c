// LDS read rate per CU by instruction width (gfx950), wall-clock (hipEvents) over a long kernel: every lane reads
// conflict-free, linearly increasing addresses; DEPTH independent reads in flight per wait.
//   hipcc --offload-arch=gfx950 -O3 lds_read_rate.hip -o lds_read_rate && ./lds_read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BYTES, int DEPTH>
__global__ void rd(float *out, int iters)
{
    extern __shared__ float4 sm[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    const char *base = reinterpret_cast<const char *>(sm) + wave * 1024;
    for (int i = 0; i < iters; ++i) {
        const char *p = base + ((i & 7) * 4096);
        if constexpr (BYTES == 16) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const float4 *>(p + lane * 16 + d * 1024 * 0 + (d & 3) * 16384);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
        } else if constexpr (BYTES == 8) {
            float2 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const float2 *>(p + lane * 8 + (d & 3) * 16384);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
        } else {
            float v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const float *>(p + lane * 4 + (d & 3) * 16384);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d];
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int BYTES, int DEPTH>
void run(float *out, int waves)
{
    const int iters = 4000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rd<BYTES, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd<BYTES, DEPTH><<<256, waves * 64, 65536>>>(out, 10);
    (void)hipEventRecord(e0);
    rd<BYTES, DEPTH><<<256, waves * 64, 65536>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)iters * DEPTH * waves;                 // wave-level reads per CU
    printf("ds_read_b%-3d depth %2d waves/CU %2d: %8.1f us  %6.2f ns per wave-read per CU  = %6.1f B/ns/CU (%5.1f B/clk at 2.4 GHz)\n", BYTES * 8, DEPTH,
           waves, ms * 1e3, ms * 1e6 / reads, reads * 64 * BYTES / (ms * 1e6), reads * 64 * BYTES / (ms * 1e6) / 2.4);
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int waves : {4, 8, 12, 16}) {
        run<16, 4>(out, waves);
        run<16, 16>(out, waves);
        run<8, 16>(out, waves);
        run<4, 16>(out, waves);
    }
    return 0;
}
