// Issue cost of the VALU instructions the MSDA backward's scatter / dot stream is made of (gfx950), per SIMD, at 1 - 4 waves
// per SIMD, measured with HIP events (ns per wave instruction per SIMD; four independent chains per wave).
//   hipcc --offload-arch=gfx950 -O3 valu_mix_rate.hip -o valu_mix_rate && ./valu_mix_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, w = 0.999f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, pw = {w, w};
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0)
            asm volatile(REP8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
        else if constexpr (KIND == 1)
            asm volatile(REP8("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %5\n v_cvt_rpi_i32_f32 %2, %6\n v_cvt_rpi_i32_f32 %3, %7\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if constexpr (KIND == 2)
            asm volatile(REP8("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if constexpr (KIND == 3)
            asm volatile(REP8("v_ashrrev_i32 %0, 31, %0\n v_add_u32 %1, %1, %0\n v_ashrrev_i32 %2, 31, %2\n v_add_u32 %3, %3, %2\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
        else if constexpr (KIND == 4)
            asm volatile(REP8("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        else if constexpr (KIND == 5)
            asm volatile(REP8("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        else if constexpr (KIND == 6)
            asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
        else if constexpr (KIND == 7)
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
        else if constexpr (KIND == 8)
            asm volatile(REP8("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
        else if constexpr (KIND == 9)
            asm volatile(REP8("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s20, 5\n v_readlane_b32 s21, %2, 3\n v_writelane_b32 %3, s21, 5\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "s20", "s21");
        else if constexpr (KIND == 10)
            asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        else if constexpr (KIND == 12)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "s20", "s21", "s22", "s23");
        else if constexpr (KIND == 13)
            asm volatile(REP8("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "vcc");
        else if constexpr (KIND == 14)
            asm volatile(REP8("v_mul_hi_i32 %0, %0, %1\n v_mul_hi_i32 %1, %1, %2\n v_mul_hi_i32 %2, %2, %3\n v_mul_hi_i32 %3, %3, %0\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
        else if constexpr (KIND == 15)
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n") : "+v"(p0), "+v"(p1) : "v"(pw));
        else if constexpr (KIND == 16)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "vcc");
        else if constexpr (KIND == 17)
            asm volatile(REP8("v_cmp_gt_f32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cmp_gt_f32_e64 s[22:23], %1, %4\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "s20", "s21", "s22", "s23");
        else if constexpr (KIND == 18)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w) : "vcc");
        else if constexpr (KIND == 11)
            asm volatile(REP8("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 + p0.x + p1.y;
}

template <int KIND> void run(const char *name, float *out)
{
    const int iters = 4000;
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, out, 10);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wps), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x iters x 32 instructions
        printf("  %d w/SIMD: %5.2f ns", wps, ms * 1e6 / ((double)wps * iters * 32));
    }
    printf("   (per wave instruction per SIMD)\n");
}

int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * 8 * 4);
    run<0>("v_mul_f32", out);
    run<7>("v_fma_f32", out);
    run<1>("v_cvt_rpi_i32_f32", out);
    run<2>("v_cvt_i32_f32", out);
    run<3>("v_ashrrev_i32 / v_add_u32 (dependent pairs)", out);
    run<8>("v_add_u32", out);
    run<10>("v_mov_b32", out);
    run<6>("v_cndmask_b32 (vcc)", out);
    run<4>("v_add_f32_dpp quad_perm (+ s_nop 1 per 4)", out);
    run<5>("v_add_f32_dpp row_half_mirror (+ s_nop 1 per 4)", out);
    run<13>("v_cndmask_b32 (vcc), independent", out);
    run<12>("v_cndmask_b32_e64 (sgpr pair), independent", out);
    run<16>("v_cmp vcc + v_cndmask vcc (pairs)", out);
    run<17>("v_cmp_e64 sgpr + v_cndmask_e64 sgpr (pairs)", out);
    run<18>("v_cmp_gt_f32 vcc", out);
    run<14>("v_mul_hi_i32", out);
    run<15>("v_pk_mul_f32", out);
    run<9>("v_readlane_b32 + v_writelane_b32", out);
    run<11>("v_mad_u32_u24 x2 + v_mul_lo_u32 x2", out);
    return 0;
}
