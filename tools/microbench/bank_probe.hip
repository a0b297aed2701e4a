// VGPR bank conflicts?  The same instruction with hand-picked registers: sources in one bank (register numbers equal mod 4)
// against sources spread over the banks.  8 independent destinations, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
             "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", \
             "v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147"

template <int KIND>
__global__ __launch_bounds__(768) void k(float* out, int trips)
{
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0)        // f32 fma, three sources in bank 0
                asm volatile("v_fma_f32 v100, v104, v108, v112\n v_fma_f32 v101, v104, v108, v112\n v_fma_f32 v102, v104, v108, v112\n v_fma_f32 v103, v104, v108, v112\n"
                             "v_fma_f32 v116, v104, v108, v112\n v_fma_f32 v117, v104, v108, v112\n v_fma_f32 v118, v104, v108, v112\n v_fma_f32 v119, v104, v108, v112" ::: CLOB);
            else if (KIND == 1)   // f32 fma, sources in banks 0,1,2
                asm volatile("v_fma_f32 v100, v104, v109, v114\n v_fma_f32 v101, v104, v109, v114\n v_fma_f32 v102, v104, v109, v114\n v_fma_f32 v103, v104, v109, v114\n"
                             "v_fma_f32 v116, v104, v109, v114\n v_fma_f32 v117, v104, v109, v114\n v_fma_f32 v118, v104, v109, v114\n v_fma_f32 v119, v104, v109, v114" ::: CLOB);
            else if (KIND == 2)   // f64 fma, three source pairs all starting in bank 0
                asm volatile("v_fma_f64 v[100:101], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[102:103], v[104:105], v[108:109], v[112:113]\n"
                             "v_fma_f64 v[116:117], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[118:119], v[104:105], v[108:109], v[112:113]\n"
                             "v_fma_f64 v[120:121], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[122:123], v[104:105], v[108:109], v[112:113]\n"
                             "v_fma_f64 v[124:125], v[104:105], v[108:109], v[112:113]\n v_fma_f64 v[126:127], v[104:105], v[108:109], v[112:113]" ::: CLOB);
            else if (KIND == 3)   // f64 fma, source pairs starting in banks 0, 2, 0
                asm volatile("v_fma_f64 v[100:101], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[102:103], v[104:105], v[110:111], v[112:113]\n"
                             "v_fma_f64 v[116:117], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[118:119], v[104:105], v[110:111], v[112:113]\n"
                             "v_fma_f64 v[120:121], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[122:123], v[104:105], v[110:111], v[112:113]\n"
                             "v_fma_f64 v[124:125], v[104:105], v[110:111], v[112:113]\n v_fma_f64 v[126:127], v[104:105], v[110:111], v[112:113]" ::: CLOB);
            else if (KIND == 4)   // pk fma, three source pairs starting in bank 0
                asm volatile("v_pk_fma_f32 v[100:101], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[102:103], v[104:105], v[108:109], v[112:113]\n"
                             "v_pk_fma_f32 v[116:117], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[118:119], v[104:105], v[108:109], v[112:113]\n"
                             "v_pk_fma_f32 v[120:121], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[122:123], v[104:105], v[108:109], v[112:113]\n"
                             "v_pk_fma_f32 v[124:125], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[126:127], v[104:105], v[108:109], v[112:113]" ::: CLOB);
            else if (KIND == 5)   // pk fma, source pairs starting in banks 0, 2, 0
                asm volatile("v_pk_fma_f32 v[100:101], v[104:105], v[110:111], v[112:113]\n v_pk_fma_f32 v[102:103], v[104:105], v[110:111], v[112:113]\n"
                             "v_pk_fma_f32 v[116:117], v[104:105], v[110:111], v[112:113]\n v_pk_fma_f32 v[118:119], v[104:105], v[110:111], v[112:113]\n"
                             "v_pk_fma_f32 v[120:121], v[104:105], v[110:111], v[112:113]\n v_pk_fma_f32 v[122:123], v[104:105], v[110:111], v[112:113]\n"
                             "v_pk_fma_f32 v[124:125], v[104:105], v[110:111], v[112:113]\n v_pk_fma_f32 v[126:127], v[104:105], v[110:111], v[112:113]" ::: CLOB);
            else if (KIND == 6)   // dependent chain through ONE register: latency of v_fma_f32
                asm volatile("v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100\n"
                             "v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100\n v_fma_f32 v100, v100, v108, v100" ::: CLOB);
            else if (KIND == 7)   // dependent chain: latency of v_fma_f64
                asm volatile("v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]\n v_fma_f64 v[100:101], v[100:101], v[108:109], v[100:101]" ::: CLOB);
            else if (KIND == 8)   // dependent chain: latency of v_pk_fma_f32
                asm volatile("v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n"
                             "v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]\n v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[100:101]" ::: CLOB);
        }
    }
    if (trips < 0) out[0] = 1.f;
}

template <int KIND>
static void run(const char* name, float* out, int cus)
{
    const int trips = 4000;
    for (int wps = 1; wps <= 3; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(256 * wps), 0, 0, out, 2);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(256 * wps), 0, 0, out, trips);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s waves/SIMD %d : %6.2f cycles per wave-instruction\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)trips * 64 * wps));
    }
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* out; (void)hipMalloc(&out, 1 << 20);
    run<0>("v_fma_f32     sources all in bank 0", out, cus);
    run<1>("v_fma_f32     sources in banks 0,1,2", out, cus);
    run<2>("v_fma_f64     source pairs all starting in bank 0", out, cus);
    run<3>("v_fma_f64     source pairs starting in banks 0,2,0", out, cus);
    run<4>("v_pk_fma_f32  source pairs all starting in bank 0", out, cus);
    run<5>("v_pk_fma_f32  source pairs starting in banks 0,2,0", out, cus);
    run<6>("v_fma_f32     fully dependent chain (latency)", out, cus);
    run<7>("v_fma_f64     fully dependent chain (latency)", out, cus);
    run<8>("v_pk_fma_f32  fully dependent chain (latency)", out, cus);
    return 0;
}
