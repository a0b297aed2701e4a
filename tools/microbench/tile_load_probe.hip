// How fast can ONE resident 1024-thread block per CU pull a 128 x 64 tile of 8 fp32 planes (262 KB) out of memory, and does
// compute of the previous block overlap it?  The register-tile SOR kernel (teeflow_sor_rt.hip.h) spends ~13 us of a 22-us
// block on exactly this.  Variants: 8-byte loads in the kernel's own mapping, 16-byte loads (32 lanes per row, two rows per
// wave-instruction), with and without a dependent-FMA "compute phase" of a given length.
// build: hipcc --offload-arch=gfx950 -O3 tile_load_probe.hip -o tile_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>   // 0: float2 per lane (lane = 2 columns, wave = one 128-px row), 1: float4 per lane (wave = two rows)
__global__ __launch_bounds__(1024) void k_tile(const float* __restrict__ in, float* __restrict__ out, int W, int H, int pitch, size_t plane, size_t pairstride,
                                               int hl, int spin)
{
    extern __shared__ float pad[];       // forces one block per CU
    const int x0 = blockIdx.x * (128 - 2 * hl), y0 = blockIdx.y * (64 - 2 * hl);
    const size_t po = (size_t)blockIdx.z * pairstride;
    float acc = 0.f;
    if (MODE == 0) {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, gx = x0 + 2 * ln;
        float2 v[8][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = y0 + wv * 4 + k;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                v[p][k] = make_float2(0, 0);
                if (gy < H && gx < W) v[p][k] = *reinterpret_cast<const float2*>(in + p * plane + po + (size_t)gy * pitch + gx);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += v[p][k].x + v[p][k].y;
    } else {
        // thread t covers columns 4*(t%32).., rows (t/32) and (t/32)+32
        const int c = threadIdx.x & 31, r = threadIdx.x >> 5, gx = x0 + 4 * c;
        float4 v[8][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gy = y0 + r + 32 * k;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                v[p][k] = make_float4(0, 0, 0, 0);
                if (gy < H && gx < W) v[p][k] = *reinterpret_cast<const float4*>(in + p * plane + po + (size_t)gy * pitch + gx);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += v[p][k].x + v[p][k].y + v[p][k].z + v[p][k].w;
    }
    for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);     // dependent chain: ~spin * 5-7 cycles per wave, 4 waves per SIMD
    if (acc == 123.456f) pad[threadIdx.x] = acc;
    // write the core of two planes, as the solver does
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, gx = x0 + 2 * ln;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gy = y0 + wv * 4 + k;
        if (gy >= y0 + hl && gy < y0 + 64 - hl && gy < H && gx >= x0 + hl && gx < x0 + 128 - hl && gx < W) {
            *reinterpret_cast<float2*>(out + po + (size_t)gy * pitch + gx) = make_float2(acc, acc);
            *reinterpret_cast<float2*>(out + plane + po + (size_t)gy * pitch + gx) = make_float2(acc, acc);
        }
    }
}

int main()
{
    const int W = 512, H = 512, pitch = 512, B = 64, hl = 10;
    const size_t pairstride = (size_t)pitch * H, plane = pairstride * B;
    float *in, *out;
    CK(hipMalloc(&in, 8 * plane * sizeof(float)));
    CK(hipMalloc(&out, 2 * plane * sizeof(float)));
    CK(hipMemset(in, 0, 8 * plane * sizeof(float)));
    const dim3 grid(1 + (W - 128 + 107) / 108, 1 + (H - 64 + 43) / 44, B);
    const double tile_bytes = 128.0 * 64 * 8 * 4, blocks = (double)grid.x * grid.y * grid.z;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int mode = 0; mode < 2; ++mode)
        for (int lds : {100 * 1024, 40 * 1024})
            for (int spin : {0, 1000, 2000, 4000}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(a));
                    if (mode == 0) hipLaunchKernelGGL(k_tile<0>, grid, dim3(1024), lds, 0, in, out, W, H, pitch, plane, pairstride, hl, spin);
                    else hipLaunchKernelGGL(k_tile<1>, grid, dim3(1024), lds, 0, in, out, W, H, pitch, plane, pairstride, hl, spin);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("mode %d (%s) blocks/CU %d spin %4d: %8.1f us  %6.2f us per block-round  %5.2f TB/s of tile loads\n", mode, mode ? "16-B loads" : " 8-B loads",
                       lds > 50 * 1024 ? 1 : 2, spin, best * 1e3, best * 1e3 / (blocks / 256.0), blocks * tile_bytes / (best * 1e-3) / 1e12);
            }
    return 0;
}
