// What does v_div_fmas_f32 do when VCC is set?  (The published pseudo-code is ambiguous about the scale.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* s0, const float* s1, const float* s2, float* d1, float* d0, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        d1[i] = __builtin_amdgcn_div_fmasf(s0[i], s1[i], s2[i], true);
        d0[i] = __builtin_amdgcn_div_fmasf(s0[i], s1[i], s2[i], false);
    }
}
int main()
{
    const int n = 24;
    float h0[n], h1[n], h2[n], r1[n], r0[n];
    int e[] = {0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 100, 120, -1, -31, -32, -33, -63, -64, -65, -96, -100, -120, -126};
    for (int i = 0; i < n; ++i) { h0[i] = 0.f; h1[i] = 0.f; h2[i] = ldexpf(1.5f, e[i]); }
    float *a, *b, *c, *d, *f;
    hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&c, 4 * n); hipMalloc(&d, 4 * n); hipMalloc(&f, 4 * n);
    hipMemcpy(a, h0, 4 * n, hipMemcpyHostToDevice); hipMemcpy(b, h1, 4 * n, hipMemcpyHostToDevice); hipMemcpy(c, h2, 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c, d, f, n);
    hipMemcpy(r1, d, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(r0, f, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("S2 = 1.5*2^%-5d  vcc=1 -> %-14g (log2 ratio %6.1f)   vcc=0 -> %g\n", e[i], r1[i], log2(fabs((double)r1[i] / h2[i])), r0[i]);
    // rounding check: fma result 2^-60*(1+2^-23+...) scaled into the denormal range
    return 0;
}
