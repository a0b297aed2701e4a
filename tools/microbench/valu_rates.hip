// Issue-rate microbenchmark for the VALU instructions the tvl1_iter kernel is made of (gfx950).
// Each kernel runs N dependent-free instructions per loop trip on 8 independent register sets; reports cycles per
// wave-instruction per SIMD at 1..3 resident waves per SIMD.   build: hipcc --offload-arch=gfx950 -O2 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void k(float* out, int trips)
{
    asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    asm volatile("s_mov_b64 s[20:21], 0x3333" ::: "s20", "s21");
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float c = 1.0001f; const f2 c2 = {1.0001f, 0.9999f}; const double cd = 1.0001;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p##i) : "v"(c2));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(c2));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d##i));
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 9) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(c2));
                REP8(X)
#undef X
            } else if (KIND == 10) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(a##i));
                REP8(X)
#undef X
            } else if (KIND == 11) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 12) {   // pk_fma with three distinct VGPR-pair sources
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p##i) : "v"(c2), "v"(p7));
                REP8(X)
#undef X
            } else if (KIND == 13) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c) : );
                REP8(X)
#undef X
            } else if (KIND == 14) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(c) : "s20","s21");
                REP8(X)
#undef X
            } else if (KIND == 15) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a##i), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 16) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(a##i), "v"(c) : "s20","s21");
                REP8(X)
#undef X
            } else if (KIND == 17) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 18) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 19) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 20) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 21) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 22) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 23) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d##i) : "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 24) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a##i) : "v"(d##i));
                REP8(X)
#undef X
            } else if (KIND == 25) {
#define X(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (KIND == 26) {
#define X(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (KIND == 27) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 28) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d##i) : "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 29) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
                REP8(X)
#undef X
            } else if (KIND == 30) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 31) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(a7));
                REP8(X)
#undef X
            } else if (KIND == 32) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 33) {
#define X(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 34) {
#define X(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 35) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 36) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 37) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(c) : "s20","s21");
                REP8(X)
#undef X
            } else if (KIND == 38) {
#define X(i) asm volatile("s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 40) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(a7), "v"(a6));
                REP8(X)
#undef X
            } else if (KIND == 41) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_fma_f64 %1, %1, %3, %1" : "+v"(p##i), "+v"(d##i) : "v"(c2), "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 42) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %0\n v_fma_f64 %1, %1, %3, %1" : "+v"(a##i), "+v"(d##i) : "v"(c), "v"(cd));
                REP8(X)
#undef X
            } else if (KIND == 43) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_mul_f32 %1, %1, %3" : "+v"(p##i), "+v"(a##i) : "v"(c2), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 44) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %0\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(a##i), "+v"(p##i.x) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 45) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 50) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d##i) : "v"(d7), "v"(d6));
                REP8(X)
#undef X
            } else if (KIND == 51) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d##i) : "v"(d7), "v"(d6));
                REP8(X)
#undef X
            } else if (KIND == 52) {
#define X(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d##i) : "v"(d7), "v"(d6));
                REP8(X)
#undef X
            } else if (KIND == 53) {
#define X(i) asm volatile("v_fma_f64 %0, -%1, %2, 0.5" : "=v"(d##i) : "v"(d7), "v"(d6));
                REP8(X)
#undef X
            } else if (KIND == 54) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(p##i) : "v"(p7), "v"(p6));
                REP8(X)
#undef X
            } else if (KIND == 55) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1\n v_fma_f64 %0, %0, %0, %0" : "=v"(d##i) : "v"(a##i));
                REP8(X)
#undef X
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x +
                                                 p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

static int g_trips = 4000;

template <int KIND>
static void run(const char* name, float* out)
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const double clk = pr.clockRate * 1e3;     // Hz
    const int trips = g_trips;
    for (int wps = 1; wps <= 4; wps += (wps == 1 ? 1 : (wps == 2 ? 2 : 1))) {   // 1, 2, 4 waves per SIMD
        const int threads = 256 * wps;         // one block per CU, 4*wps waves
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 0, 0, out, trips);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)trips * 64 * wps;
        printf("%-34s waves/SIMD %d : %6.2f cycles per wave-instruction (nominal clock %.2f GHz)\n", name, wps,
               ms * 1e-3 * clk / instr_per_simd, clk * 1e-9);
        if (wps == 4) break;
    }
}

int main(int argc, char** argv)
{
    if (argc > 1) g_trips = atoi(argv[1]);      // long runs (e.g. 400000) show sustained clocks under power limits
    const bool only_heavy = argc > 2;
    float* out; hipMalloc(&out, 1 << 24);
    if (only_heavy) {
        run<0>("v_fma_f32", out); run<3>("v_fma_f64", out); run<1>("v_pk_fma_f32 (2 distinct srcs)", out); run<41>("PAIR v_pk_fma_f32 + v_fma_f64 alternating", out);
        run<5>("v_rsq_f64", out); run<7>("v_mov_b32", out);
        run<50>("v_fma_f64 three distinct VGPR-pair sources", out); run<51>("v_mul_f64 two distinct sources, distinct dest", out); run<52>("v_fmac_f64 distinct sources", out); run<53>("v_fma_f64 with inline constant", out); run<54>("v_pk_fma_f32 distinct sources + neg", out); run<55>("PAIR v_cvt_f64_f32 -> dependent v_fma_f64", out);
        hipFree(out);
        return 0;
    }
    run<0>("v_fma_f32", out); run<8>("v_mul_f32", out); run<1>("v_pk_fma_f32 (2 distinct srcs)", out);
    run<12>("v_pk_fma_f32 (3 distinct srcs)", out); run<2>("v_pk_mul_f32", out); run<9>("v_pk_add_f32", out);
    run<3>("v_fma_f64", out); run<11>("v_mul_f64", out); run<10>("v_cvt_f64_f32", out); run<4>("v_rcp_f32", out);
    run<5>("v_rsq_f64", out); run<6>("v_div_fixup_f32", out); run<7>("v_mov_b32", out); run<13>("v_cndmask_b32", out);
    run<14>("v_cndmask_b32_e64 (sgpr mask)", out); run<15>("v_cmp_lt_f32 -> vcc", out); run<16>("v_cmp_lt_f32_e64 -> sgpr", out); run<17>("v_max_f32", out); run<18>("v_add_f32", out); run<19>("v_add_u32", out); run<20>("v_and_b32", out); run<21>("v_bfi_b32", out); run<22>("v_add_f64", out); run<23>("v_max_f64", out); run<24>("v_cvt_f32_f64", out); run<25>("v_rndne_f32", out); run<26>("v_cvt_u32_f32", out); run<27>("v_med3_f32", out); run<28>("v_lshl_add_u64", out); run<29>("v_sqrt_f32", out); run<30>("v_mov_b32 dpp row_shr:1", out); run<31>("v_cndmask_b32 vcc (src=other vgpr)", out); run<32>("v_fmac_f32", out); run<33>("v_div_scale_f32", out); run<34>("v_div_fmas_f32", out);
    run<35>("PAIR v_cmp->vcc + v_cndmask vcc", out); run<36>("v_cndmask_b32_e64 (vcc as sgpr pair)", out); run<37>("PAIR v_cmp_e64->sgpr + v_cndmask_e64", out); run<38>("PAIR s_mov vcc + v_cndmask vcc", out);
    run<40>("v_fma_f32 three distinct VGPR sources", out); run<41>("PAIR v_pk_fma_f32 + v_fma_f64 alternating", out); run<42>("PAIR v_fma_f32 + v_fma_f64 alternating", out); run<43>("PAIR v_pk_fma_f32 + v_mul_f32 alternating", out); run<44>("PAIR v_fma_f32 + v_cndmask_e64 alternating", out); run<45>("QUAD dependent mul/add chain (4 instr, 8 chains)", out);
    hipFree(out);
    return 0;
}
