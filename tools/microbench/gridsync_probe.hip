// What does it cost a set of co-resident 1024-thread blocks (one per CU) to exchange 32 bytes per thread with their neighbours through
// global memory in the middle of a kernel?  A persistent form of the register-tile SOR kernel (teeflow_sor_rt.hip.h) would keep a tile's
// coefficients in registers over all 25 sweeps of a fixed-point iteration and trade only du, dv with the neighbouring tiles every S sweeps
// instead of ending the kernel and loading all 8 planes again; that pays only if one exchange costs well under the ~11 us it saves.
// Each step: every thread writes 4 float2 into buffer (i+1)&1, the T blocks of a group meet at a counter, every thread reads the 4 float2
// the NEXT tile's same thread wrote and checks them.  Variants:
//   0: plain stores / loads, __threadfence() (agent-scope release / acquire: L2 write-back + invalidate across the 8 XCDs) around the counter
//   1: agent-scope relaxed atomic stores / loads of the data (write-through / L2-bypassing), no fences
//   4: plain stores / loads and the counter, no fences (the meeting alone; values may be stale)
//   5: as 1, but no counter: every block raises its own flag (one 128-byte line each) and waits for the flags of 8 other tiles of its group
//   6: as 1 with each group's counter on its own 4-KB page
//   7: plain stores, ONE agent-scope release fence per block (lane 0 of wave 0, after the block's barrier) before its flag, sc1 loads
//   2: no meeting at all (time of the stores + loads alone; the check is expected to fail)
//   3: nothing but the compute stand-in
// build: hipcc --offload-arch=gfx950 -O3 gridsync_probe.hip -o gridsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_sync(unsigned long long* buf, unsigned* counters, unsigned* errs, int T, int steps, int spin, unsigned base, int slp)
{
    extern __shared__ float pad[];
    const int tile = blockIdx.x, grp = blockIdx.y, G = gridDim.y, t = threadIdx.x;
    const size_t per_buf = (size_t)G * T * 1024 * 4;
    unsigned bad = 0, timeouts = 0;
    float acc = (float)t;
    for (int i = 0; i < steps; ++i) {
        for (int k = 0; k < spin; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
        if (MODE == 3) continue;
        unsigned long long* wr = buf + (size_t)((i + 1) & 1) * per_buf + (size_t)(grp * T + tile) * 4096 + t;
        const unsigned long long val = ((unsigned long long)(i + 1) << 32) | (unsigned)(tile * 1024 + t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 1 || MODE == 5 || MODE == 6) __hip_atomic_store(wr + 1024 * k, val + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else wr[1024 * k] = val + k;
        }
        if (MODE == 0) __threadfence();
        if (MODE == 1 || MODE >= 5) __builtin_amdgcn_s_waitcnt(0);          // every store of this wave acknowledged
        __syncthreads();
        if (MODE == 5 || MODE == 7) {
            if (t < 64) {
                unsigned* flags = counters + 4096;                       // [block][32] : one line per block
                if (MODE == 7 && t == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (t == 0) __hip_atomic_store(flags + (size_t)(grp * T + tile) * 32, base + (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int nb = (tile + 1 + (t & 7)) % T;                  // 8 other tiles (fewer when the group is small)
                int n = 0;
                while (true) {
                    const unsigned f = __hip_atomic_load(flags + (size_t)(grp * T + nb) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((int)(f - (base + (unsigned)(i + 1))) >= 0)) break;
                    if (++n > (1 << 20)) { ++timeouts; break; }
                    if (slp == 0) __builtin_amdgcn_s_sleep(1); else if (slp == 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
                }
            }
            __syncthreads();
        } else if (MODE != 2) {
            unsigned* ctr = MODE == 6 ? counters + 8192 + (size_t)grp * 1024 : counters + grp;
            if (t == 0) {
                const unsigned target = base + (unsigned)T * (unsigned)(i + 1);
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int n = 0;
                while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                    if (++n > (1 << 20)) { ++timeouts; break; }            // every wave reaches the end whatever happens
                    if (slp == 0) __builtin_amdgcn_s_sleep(1); else if (slp == 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
                }
            }
            __syncthreads();
            if (MODE == 0) __threadfence();
        }
        const int nt = (tile + 1) % T;
        const unsigned long long* rd = buf + (size_t)((i + 1) & 1) * per_buf + (size_t)(grp * T + nt) * 4096 + t;
        unsigned long long got[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) got[k] = (MODE == 1 || MODE >= 5) ? __hip_atomic_load(rd + 1024 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rd[1024 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) bad += got[k] != ((((unsigned long long)(i + 1)) << 32) | (unsigned)(nt * 1024 + t)) + k;
        acc += (float)(got[0] & 1);
    }
    if (acc == 123.456f) pad[t] = acc;
    if (bad) atomicAdd(errs, bad);
    if (timeouts) atomicAdd(errs + 1, timeouts);
}

template <int MODE>
static int run(const char* name, int T, int G, int steps, int spin, int slp, unsigned long long* buf, unsigned* counters, unsigned* errs, float* base_ms)
{
    CK(hipMemset(counters, 0, (1 << 20) * sizeof(unsigned)));
    CK(hipMemset(errs, 0, 2 * sizeof(unsigned)));
    CK(hipFuncSetAttribute((const void*)k_sync<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned base = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_sync<MODE>, dim3(T, G), dim3(1024), 100 * 1024, 0, buf, counters, errs, T, steps, spin, base, slp);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        base += (MODE == 5 || MODE == 7) ? (unsigned)steps : (unsigned)T * steps;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned h[2];
    CK(hipMemcpy(h, errs, sizeof(h), hipMemcpyDeviceToHost));
    if (MODE == 3) *base_ms = best;
    printf("  %-44s %8.2f us/step  (+%6.2f over compute)  wrong values %u, timeouts %u\n", name, 1e3 * best / steps, 1e3 * (best - *base_ms) / steps, h[0], h[1]);
    return 0;
}

int main(int argc, char** argv)
{
    const int steps = 400;
    unsigned long long* buf; unsigned *counters, *errs;
    CK(hipMalloc(&buf, (size_t)2 * 256 * 1024 * 4 * 8));
    CK(hipMalloc(&counters, (1 << 20) * sizeof(unsigned)));
    CK(hipMalloc(&errs, 2 * sizeof(unsigned)));
    const int shapes[][2] = {{60, 4}, {21, 12}};
    for (auto& sh : shapes)
        for (int spin : {0, 320}) {
            float base_ms = 0;
            printf("%d tiles x %d groups = %d blocks, compute stand-in %d dependent FMAs\n", sh[0], sh[1], sh[0] * sh[1], spin);
            if (run<3>("compute only", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<2>("stores + loads, no meeting", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (spin == 0 && run<0>("plain data, __threadfence around the counter", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<4>("plain data + counter, no fences (stale ok)", sh[0], sh[1], steps, spin, 1, buf, counters, errs, &base_ms)) return 1;
            if (run<1>("agent-scope atomic data, poll every 64 clk", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<6>("same, counters 4 KB apart, poll every 64 clk", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<5>("own flag + 8 neighbours' flags, poll 64 clk", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<7>("plain stores + one release fence per block", sh[0], sh[1], steps, spin, 0, buf, counters, errs, &base_ms)) return 1;
            if (run<5>("own flag + 8 neighbours' flags, poll 512 clk", sh[0], sh[1], steps, spin, 1, buf, counters, errs, &base_ms)) return 1;
            if (run<1>("agent-scope atomic data, poll every 512 clk", sh[0], sh[1], steps, spin, 1, buf, counters, errs, &base_ms)) return 1;
        }
    return 0;
}
