// Calibration microbenchmarks for the MSDA forward kernels (gfx950): VALU issue cost of the instructions the tap loop is
// made of (v_fmac_f32, v_fmac_f32_dpp, v_pk_fma_f32, v_add_u32_dpp) and the ds_read_b128 rate / latency, per CU, for
// 4..16 waves per CU.   hipcc --offload-arch=gfx950 -O3 valu_lds_rate.hip -o valu_lds_rate && ./valu_lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ void valu(float *out, long long *cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, w = 0.999f, c = 1.0001f;
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            asm volatile(REP16("v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(c));
        } else if constexpr (KIND == 1) {
            asm volatile(REP16("v_fmac_f32_dpp %0, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(c));
        } else if constexpr (KIND == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {b0, b1}, x3 = {b2, b3}, ww = {w, w}, cc = {c, c};
            asm volatile(REP16("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(ww), "v"(cc));
            a0 = x0.x; a1 = x0.y; a2 = x1.x; a3 = x1.y; b0 = x2.x; b1 = x2.y; b2 = x3.x; b3 = x3.y;
        } else if constexpr (KIND == 3) {
            int i0 = (int)a0, i1 = (int)a1, i2 = (int)a2, i3 = (int)a3, k = 3;
            asm volatile(REP16("v_add_u32_dpp %0, %4, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %4, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
            a0 = i0; a1 = i1; a2 = i2; a3 = i3;
        } else if constexpr (KIND == 4) {
            asm volatile(REP16("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(c));
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ds_read_b128: every lane reads 16 bytes; MODE 0: conflict-free (lane*16), 1: random tokens (quad layout, parity roles),
// 2: random tokens without roles.  DEPTH reads in flight per wait.
template <int DEPTH>
__global__ void lds(float *out, long long *cyc, int iters, int mode)
{
    extern __shared__ float4 sm[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned rng = threadIdx.x * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        int idx[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            rng = rng * 1664525u + 1013904223u;
            if (mode == 0) idx[d] = ((lane + d * 64 + i * 7) & 8191);
            else {
                // quad = (lane>>2); all 4 lanes of a quad read one 64-byte half token: token t, half h=(lane>>2)&1
                unsigned q = __shfl(rng, lane & ~3) >> 8;
                int t = q % 500;
                const int c8 = lane >> 3, role = (c8 >> 1) & 1;
                if (mode == 1) t = (t & ~1) | role;
                idx[d] = t * 8 + ((lane >> 2) & 1) * 4 + (lane & 3);
            }
        }
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = sm[idx[d]];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<long long> h(256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const char *names[] = {"v_fmac_f32", "v_fmac_f32_dpp", "v_pk_fma_f32", "v_add_u32_dpp", "v_fma_f32(vop3)"};
    const int iters = 200;
    for (int kind = 0; kind < 5; ++kind)
        for (int waves : {4, 8, 12, 16}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) valu<0><<<256, waves * 64>>>(out, cyc, iters);
                if (kind == 1) valu<1><<<256, waves * 64>>>(out, cyc, iters);
                if (kind == 2) valu<2><<<256, waves * 64>>>(out, cyc, iters);
                if (kind == 3) valu<3><<<256, waves * 64>>>(out, cyc, iters);
                if (kind == 4) valu<4><<<256, waves * 64>>>(out, cyc, iters);
                hipEventRecord(e1);
                hipDeviceSynchronize();
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double c = 0; for (auto x : h) c += x; c /= 256;
            const double instr_per_simd = (double)iters * 64 * waves / 4;
            printf("%-16s waves/CU %2d: %8.0f clk, %6.2f clk per wave-instr per SIMD; kernel %.1f us -> %.2f ns per wave-instr per SIMD, tick = %.2f ns\n", names[kind], waves, c, c / instr_per_simd, ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / c);
        }
    for (int mode = 0; mode < 3; ++mode)
        for (int depth : {4, 8, 16})
            for (int waves : {4, 8, 12, 16}) {
                const int it = 400;
                for (int rep = 0; rep < 2; ++rep) {
                    if (depth == 4) lds<4><<<256, waves * 64, 131072>>>(out, cyc, it, mode);
                    if (depth == 8) lds<8><<<256, waves * 64, 131072>>>(out, cyc, it, mode);
                    if (depth == 16) lds<16><<<256, waves * 64, 131072>>>(out, cyc, it, mode);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
                double c = 0; for (auto x : h) c += x; c /= 256;
                const double reads = (double)it * depth * waves;
                printf("ds_read_b128 mode %d depth %2d waves/CU %2d: %8.0f clk, %6.2f clk per wave-read per CU (%5.1f B/clk/CU)\n", mode, depth, waves, c,
                       c / reads, reads * 1024 / c);
            }
    return 0;
}
