// Upper bound for a barrier-free tvl1_iter: the real quad arithmetic (two fused iterations: U, P, U, P + both error terms)
// on data streamed from HBM, neighbours through wave shuffles instead of LDS + barriers, one wave marching down a band.
// NOT a correct solver (vertical neighbours are faked with the thread's previous row) -- it measures what the arithmetic,
// the loads/stores and the shuffles cost at 1..3 waves per SIMD.   build: see README.md
#include "../../tee_optical_flow_amd/csrc/teeflow_kernels.hip.h"
#include <cstdio>
#include <vector>

template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_probe(const float* __restrict__ in, float* __restrict__ out, int rows, int pitch, size_t plane, float l_t, float theta, float taut,
             unsigned long long* __restrict__ err, int wrap)
{
    const int lane = threadIdx.x, x = lane * 4;
    const size_t base = (size_t)blockIdx.x * rows * pitch;           // each wave owns `rows` rows of a 256-px band
    float pu1[4] = {0, 0, 0, 0}, pu2[4] = {0, 0, 0, 0}, pp12[4] = {0, 0, 0, 0}, pp22[4] = {0, 0, 0, 0};
    double accA = 0.0, accB = 0.0;
    unsigned inw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) inw[j] = opaque_u(~0u);
    const unsigned keep[4] = {inw[0], inw[1], inw[2], inw[3]};
    for (int r = 0; r < rows; ++r) {
        const size_t row = base + (size_t)(r & wrap) * pitch + x;       // wrap = 7: the wave re-reads 8 rows (cache-resident: compute bound)
        QuadU qu;
        float4 v;
        v = ld4(in + 0 * plane + row); UNPACK4(qu.u1k, v)
        v = ld4(in + 1 * plane + row); UNPACK4(qu.u2k, v)
        v = ld4(in + 2 * plane + row); UNPACK4(qu.wx, v)
        v = ld4(in + 3 * plane + row); UNPACK4(qu.wy, v)
        v = ld4(in + 4 * plane + row); UNPACK4(qu.r, v)
        v = ld4(in + 5 * plane + row); UNPACK4(qu.p11, v)
        v = ld4(in + 6 * plane + row); UNPACK4(qu.p12, v)
        v = ld4(in + 7 * plane + row); UNPACK4(qu.p21, v)
        v = ld4(in + 8 * plane + row); UNPACK4(qu.p22, v)
#pragma unroll
        for (int i = 0; i < 4; ++i) { qu.p12u[i] = pp12[i]; qu.p22u[i] = pp22[i]; }
        qu.l11 = __shfl_up(qu.p11[3], 1, 64); qu.l21 = __shfl_up(qu.p21[3], 1, 64);
        float u1a[4], u2a[4], u1b[4], u2b[4], q11[4], q12[4], q21[4], q22[4], s11[4], s12[4], s21[4], s22[4];
        tv_u_quad_pk(l_t, theta, qu, r == 0, lane == 0, u1a, u2a);
        accA += tv_err_quad_pk(u1a, qu.u1k, u2a, qu.u2k, keep);
        float ux1[4], uy1[4], ux2[4], uy2[4];
        const float rr1 = __shfl_down(u1a[0], 1, 64), rr2 = __shfl_down(u2a[0], 1, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u1a[i + 1] : rr1, e2 = i < 3 ? u2a[i + 1] : rr2;
            ux1[i] = mask_f(e1 - u1a[i], inw[i + 1]); ux2[i] = mask_f(e2 - u2a[i], inw[i + 1]);
            uy1[i] = mask_f(u1a[i] - pu1[i], inw[0]); uy2[i] = mask_f(u2a[i] - pu2[i], inw[0]);
        }
        tv_p_quad_pk(taut, ux1, uy1, ux2, uy2, qu.p11, qu.p12, qu.p21, qu.p22, q11, q12, q21, q22);
        QuadU q2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q2.u1k[i] = u1a[i]; q2.u2k[i] = u2a[i]; q2.wx[i] = qu.wx[i]; q2.wy[i] = qu.wy[i]; q2.r[i] = qu.r[i];
            q2.p11[i] = q11[i]; q2.p12[i] = q12[i]; q2.p21[i] = q21[i]; q2.p22[i] = q22[i]; q2.p12u[i] = pp12[i]; q2.p22u[i] = pp22[i];
        }
        q2.l11 = __shfl_up(q11[3], 1, 64); q2.l21 = __shfl_up(q21[3], 1, 64);
        tv_u_quad_pk(l_t, theta, q2, r == 0, lane == 0, u1b, u2b);
        accB += tv_err_quad_pk(u1b, u1a, u2b, u2a, keep);
        const float t1 = __shfl_down(u1b[0], 1, 64), t2 = __shfl_down(u2b[0], 1, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u1b[i + 1] : t1, e2 = i < 3 ? u2b[i + 1] : t2;
            ux1[i] = mask_f(e1 - u1b[i], inw[i + 1]); ux2[i] = mask_f(e2 - u2b[i], inw[i + 1]);
            uy1[i] = mask_f(u1b[i] - pu1[i], inw[0]); uy2[i] = mask_f(u2b[i] - pu2[i], inw[0]);
        }
        tv_p_quad_pk(taut, ux1, uy1, ux2, uy2, q11, q12, q21, q22, s11, s12, s21, s22);
        st4(out + 0 * plane + row, PACK4(u1b)); st4(out + 1 * plane + row, PACK4(u2b));
        st4(out + 2 * plane + row, PACK4(s11)); st4(out + 3 * plane + row, PACK4(s12));
        st4(out + 4 * plane + row, PACK4(s21)); st4(out + 5 * plane + row, PACK4(s22));
#pragma unroll
        for (int i = 0; i < 4; ++i) { pu1[i] = u1b[i]; pu2[i] = u2b[i]; pp12[i] = s12[i]; pp22[i] = s22[i]; }
    }
    const unsigned long long a = (unsigned long long)accA + (unsigned long long)accB;
    if (a == 0x123456789ull) err[0] = a;
}

__global__ void k_fill(float* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761ull) % 2001ull) * 1e-3f - 1.0f;
}

template <int WAVES>
static void run(const float* in, float* out, int rows, int pitch, size_t plane, unsigned long long* err, int cus, int wrap)
{
    const int waves_total = cus * 4 * WAVES;                       // exactly fills the machine at WAVES per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<WAVES>, dim3(waves_total), dim3(64), 0, 0, in, out, 4, pitch, plane, 0.045f, 0.3f, 0.8333f, err, wrap);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<WAVES>, dim3(waves_total), dim3(64), 0, 0, in, out, rows, pitch, plane, 0.045f, 0.3f, 0.8333f, err, wrap);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double px_it = (double)waves_total * rows * 256 * 2;     // pixel-iterations
    printf("%s waves/SIMD %d: %7.3f ms for %d rows per wave -> %6.1f Gpx-it/s, %5.0f cycles per wave-row (2.4 GHz), HBM traffic if streamed %.2f TB/s\n",
           wrap == 7 ? "cache-resident" : "streaming     ", WAVES, ms, rows, px_it / ms * 1e-6, ms * 1e-3 * 2.4e9 / rows, px_it / 2 * 60 / (ms * 1e-3) * 1e-12);
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, pitch = 256, rows = 512;
    const size_t waves_max = (size_t)cus * 4 * 3, plane = waves_max * rows * pitch;
    float *in, *out; unsigned long long* err;
    (void)hipMalloc(&in, plane * 9 * sizeof(float)); (void)hipMalloc(&out, plane * 6 * sizeof(float)); (void)hipMalloc(&err, 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, plane * 9);
    (void)hipDeviceSynchronize();
    for (int wrap : {0x7fffffff, 7}) {
        run<1>(in, out, rows, pitch, plane, err, cus, wrap);
        run<2>(in, out, rows, pitch, plane, err, cus, wrap);
        run<3>(in, out, rows, pitch, plane, err, cus, wrap);
    }
    printf("reference: k_iter2_rows full level-0 launches run at ~8000 cycles per wave-step of 2 rows x 256 px x 2 iterations at 3 waves/SIMD\n");
    return 0;
}
