// FETCH_SIZE / WRITE_SIZE calibration kernels (tools/copy_calibration.py): device copies of one buffer with 4, 8 and 16 bytes
// per lane, and the backward kernels' sparse pattern -- 8 useful bytes per lane out of every 256 (a line per lane).  The buffers
// are 1 GiB each: four times the 256 MB Infinity Cache, so that the reads come from HBM.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <typename V>
__global__ __launch_bounds__(256) void copy_w(const V *__restrict__ src, V *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// reads 8 bytes of every `stride` bytes (one lane per piece), writes them densely
__global__ __launch_bounds__(256) void read_sparse8(const char *__restrict__ src, float2 *__restrict__ dst, size_t pieces, size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = *reinterpret_cast<const float2 *>(src + i * stride);
}
// writes 16 bytes of every `stride` bytes (the sampling kernels' per-level pieces)
__global__ __launch_bounds__(256) void write_sparse16(float4 *__restrict__ dst, size_t pieces, size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (size_t)gridDim.x * blockDim.x)
        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(dst) + i * stride) = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    char *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    if (hipMemset(a, 1, bytes) != hipSuccess || hipMemset(b, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy_w<float>, dim3(8192), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
        hipLaunchKernelGGL(copy_w<float2>, dim3(8192), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, bytes / 8);
        hipLaunchKernelGGL(copy_w<float4>, dim3(8192), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16);
        hipLaunchKernelGGL(read_sparse8, dim3(8192), dim3(256), 0, 0, (const char *)a, (float2 *)b, bytes / 256, (size_t)256);
        hipLaunchKernelGGL(read_sparse8, dim3(8192), dim3(256), 0, 0, (const char *)a, (float2 *)b, bytes / 64, (size_t)64);
        hipLaunchKernelGGL(write_sparse16, dim3(8192), dim3(256), 0, 0, (float4 *)b, bytes / 128, (size_t)128);
        hipLaunchKernelGGL(write_sparse16, dim3(8192), dim3(256), 0, 0, (float4 *)b, bytes / 48, (size_t)48);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("copy_w<4|8|16>: 1 GiB read + 1 GiB written each; read_sparse8 stride 256: %zu pieces (one 128-B line each = %.0f MiB of lines, "
           "%.0f MiB useful) ; stride 64: %zu pieces (every 64-B half line: 1 GiB of lines) ; write_sparse16 stride 128: %zu pieces ; stride 48: %zu pieces\n",
           bytes / 256, (double)(bytes / 256) * 128 / 1048576.0, (double)(bytes / 256) * 8 / 1048576.0, bytes / 64, bytes / 128, bytes / 48);
    return 0;
}
