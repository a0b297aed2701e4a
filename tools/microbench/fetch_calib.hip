// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the load widths the engine uses (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a 16-B-per-lane streaming read ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three streaming copies of a buffer far larger than the
// Infinity Cache with 4, 8 and 16 bytes per lane; the byte counts are known exactly, the counters come from
// `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` around this program (tools/calibrate_fetch.sh).
// build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// the register-tile SOR's shape: 8-byte loads of 512-B row segments, 2 KB apart (a 128-px-wide tile of a 512-px image)
__global__ __launch_bounds__(256) void k_copy_tile8(const float2* __restrict__ src, float2* __restrict__ dst, size_t rows)
{
    const int ln = threadIdx.x & 63, seg = blockIdx.y;                 // 4 segments of 64 float2 per 256-float2 row
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4)
        dst[r * 256 + seg * 64 + ln] = src[r * 256 + seg * 64 + ln];
}

int main()
{
    const size_t bytes = (size_t)1 << 30;      // 1 GiB read + 1 GiB written per kernel
    void *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_copy<float>, dim3(8192), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4);
    hipLaunchKernelGGL(k_copy<float2>, dim3(8192), dim3(256), 0, 0, (const float2*)a, (float2*)b, bytes / 8);
    hipLaunchKernelGGL(k_copy<float4>, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
    hipLaunchKernelGGL(k_copy_tile8, dim3(2048, 4), dim3(256), 0, 0, (const float2*)a, (float2*)b, bytes / 2048);
    CK(hipDeviceSynchronize());
    printf("bytes_per_kernel %zu\n", bytes);
    return 0;
}
