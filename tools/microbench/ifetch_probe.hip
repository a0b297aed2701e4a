// Is a long straight-line VALU body limited by instruction fetch?  The same v_fma_f32 / v_fma_f64 / v_pk_fma_f32 mix in a loop
// whose body is BODY x 24 instructions (8 bytes each): small bodies replay from the instruction buffer / cache, large ones
// stream from the shared instruction cache.  Prints cycles per wave-instruction at 1, 2, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BODY>
__global__ __launch_bounds__(768) void k(float* out, int trips)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float c = 1.0001f; const f2 c2 = {1.0001f, 0.9999f}; const double cd = 1.0001;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < BODY; ++u) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %3, %0\n v_fma_f64 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %5, %2" : "+v"(a##i), "+v"(d##i), "+v"(p##i) : "v"(c), "v"(cd), "v"(c2));
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                                                 (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int BODY>
static void run(float* out, int cus)
{
    const long long instr_target = 24LL * 64 * 4000;
    const int trips = (int)(instr_target / (24LL * BODY));
    for (int wps = 1; wps <= 3; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<BODY>, dim3(cus), dim3(256 * wps), 0, 0, out, 2);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<BODY>, dim3(cus), dim3(256 * wps), 0, 0, out, trips);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)trips * 24 * BODY * wps;
        printf("body %5d instructions (%6.1f KB): waves/SIMD %d : %5.2f cycles per wave-instruction\n", 24 * BODY, 24 * BODY * 8 / 1024.0, wps,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    run<2>(out, pr.multiProcessorCount); run<16>(out, pr.multiProcessorCount); run<64>(out, pr.multiProcessorCount);
    run<128>(out, pr.multiProcessorCount); run<256>(out, pr.multiProcessorCount); run<512>(out, pr.multiProcessorCount);
    return 0;
}
