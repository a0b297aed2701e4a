// teeflow_wase.hip.h -- SURVEY.md rows a7 / f2: "WASE" background compensation on the device.
//
// Reference (/root/reference/optical_flow/calculate_optical_flow.py:647-660):
//     masked = flow * mask_dict['bkgd']          # flow f32 [H,W,2], bkgd bool [N,H,W,2]  ->  f32 [N,H,W,2]
//     background = np.mean(masked[masked != 0])  # ONE scalar over all N frames' masks and both components
//     return flow - background
// per frame pair, i.e. O(N*H*W) numpy work per pair and O(N^2) per study.  The scalar is a float32 numpy reduction, so
// its bits depend on numpy's summation order; this file reproduces that order exactly (checked against numpy 1.26 and
// 2.2 in tests/test_wase_cpu.py through a pure-python restatement, and against numpy itself on the GPU box):
//   * masked[masked != 0] is the C-order (n,h,w,c) compaction of the products that compare unequal to 0 (NaN counts);
//   * np.add.reduce walks the contiguous result in pieces of 8192 elements (the ufunc buffer) and ADDS one
//     pairwise sum per piece to a running float32 total that starts at 0;
//   * pairwise sum of n elements: n < 8 sequential from 0; n <= 128: eight strided accumulators r[j] = a[j],
//     r[j] += a[i+j], combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n%8 tail sequentially; else split at
//     n/2 rounded down to a multiple of 8, left + right;
//   * np.mean divides in float64 (float32 sum / intp count) and casts the quotient to float32.
#pragma once
#include "teeflow_kernels.hip.h"

#define WASE_CHUNK 2048        // elements per compaction block (256 threads x 8 consecutive elements)
#define NP_BUFSIZE 8192        // np.getbufsize(): elements per inner-loop call of the reduction
#define NP_PW_BLOCK 128        // numpy's PW_BLOCKSIZE
#define WASE_MAX_LEAVES 128    // a piece of <= 8192 elements has at most 8192/64 leaves

__device__ __forceinline__ float wase_product(float f, uint8_t m) { return f * (m ? 1.0f : 0.0f); }

// pass 1: how many products of frame n, chunk c are non-zero
__global__ __launch_bounds__(256) void k_wase_count(const float* __restrict__ flow, const uint8_t* __restrict__ mask, size_t hw2, int C,
                                                    unsigned* __restrict__ cnt)
{
    __shared__ unsigned sw[4];
    const int c = blockIdx.x, n = blockIdx.y;
    const size_t base = (size_t)c * WASE_CHUNK + (size_t)threadIdx.x * 8;
    const uint8_t* m = mask + (size_t)n * hw2;
    unsigned k = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (base + i < hw2) k += wase_product(flow[base + i], m[base + i]) != 0.0f ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(size_t)n * C + c] = sw[0] + sw[1] + sw[2] + sw[3];
}

// pass 2: exclusive scan of the n block counts (one block; n is frames x chunks, a few 10^4); off[n] = total
__global__ __launch_bounds__(1024) void k_wase_scan(const unsigned* __restrict__ cnt, size_t n, u64* __restrict__ off)
{
    __shared__ u64 sw[16];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t b = 0; b < n; b += 1024) {
        const size_t i = b + threadIdx.x;
        const u64 v = i < n ? cnt[i] : 0;
        u64 x = v;                                           // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u64 y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = x;
        __syncthreads();
        u64 wbase = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += sw[w];
        const u64 c0 = carry;
        if (i < n) off[i] = c0 + wbase + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + wbase + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = carry;
}

// pass 3: the compaction itself (stable: element order inside a chunk = thread order x 8 consecutive elements)
__global__ __launch_bounds__(256) void k_wase_scatter(const float* __restrict__ flow, const uint8_t* __restrict__ mask, size_t hw2, int C,
                                                      const u64* __restrict__ off, float* __restrict__ a)
{
    __shared__ unsigned sw[4];
    const int c = blockIdx.x, n = blockIdx.y;
    const size_t base = (size_t)c * WASE_CHUNK + (size_t)threadIdx.x * 8;
    const uint8_t* m = mask + (size_t)n * hw2;
    float v[8];
    unsigned k = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = base + i < hw2 ? wase_product(flow[base + i], m[base + i]) : 0.0f;
        k += v[i] != 0.0f ? 1u : 0u;
    }
    unsigned x = k;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += sw[w];
    u64 p = off[(size_t)n * C + c] + wbase + x - k;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (v[i] != 0.0f) a[p++] = v[i];
}

TF_HD inline int np_pw_split(int n) { int n2 = n / 2; return n2 - n2 % 8; }

// pass 4: numpy's pairwise sum of every 8192-element piece of the compacted array.  Thread 0 lists the leaves of the
// piece's split tree, 8-lane groups sum the leaves (one lane per strided accumulator), thread 0 adds them up the tree.
__global__ __launch_bounds__(512) void k_wase_piece_sums(const float* __restrict__ a, const u64* __restrict__ total, float* __restrict__ s)
{
    __shared__ int loff[WASE_MAX_LEAVES], ln[WASE_MAX_LEAVES];
    __shared__ float lsum[WASE_MAX_LEAVES];
    __shared__ int nleaves;
    const u64 M = *total;
    const u64 p0 = (u64)blockIdx.x * NP_BUFSIZE;
    if (p0 >= M) return;                                              // block-uniform
    const int n = (int)(M - p0 < NP_BUFSIZE ? M - p0 : NP_BUFSIZE);
    const float* piece = a + p0;
    if (threadIdx.x == 0) {
        int so[16], sn[16], sp = 0, nl = 0;
        so[0] = 0; sn[0] = n; sp = 1;
        while (sp > 0) {
            --sp;
            const int o = so[sp], m = sn[sp];
            if (m <= NP_PW_BLOCK) { loff[nl] = o; ln[nl] = m; ++nl; continue; }
            const int n2 = np_pw_split(m);
            so[sp] = o + n2; sn[sp] = m - n2; ++sp;                   // right child below the left one: left is listed first
            so[sp] = o; sn[sp] = n2; ++sp;
        }
        nleaves = nl;
    }
    __syncthreads();
    const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
    for (int l0 = 0; l0 < nleaves; l0 += 64) {                        // every lane takes part in the shuffles
        const int l = l0 + grp;
        const bool on = l < nleaves;
        const int m = on ? ln[l] : 0;
        const float* x = piece + (on ? loff[l] : 0);
        float r = 0.f;
        if (m >= 8) {
            r = x[j];
            const int lim = m - m % 8;
            for (int i = 8; i < lim; i += 8) r += x[i + j];
        }
        const float r1 = r + __shfl_xor(r, 1, 64);                    // (r0+r1) (r2+r3) (r4+r5) (r6+r7)
        const float r2 = r1 + __shfl_xor(r1, 2, 64);                  // (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
        float res = r2 + __shfl_xor(r2, 4, 64);
        if (on && j == 0) {
            if (m < 8) { res = 0.f; for (int i = 0; i < m; ++i) res += x[i]; }
            else for (int i = m - m % 8; i < m; ++i) res += x[i];
            lsum[l] = res;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // post-order walk of the same tree; leaves are consumed left to right
        int sn[16], st[16], sp = 0, k = 0;
        float sl[16];
        float val = 0.f;
        bool have = false;
        sn[0] = n; st[0] = 0; sp = 1;
        while (sp > 0) {
            const int t = sp - 1;
            if (have) {                                               // a child of the node on top has just been evaluated
                have = false;
                if (st[t] == 1) { sl[t] = val; st[t] = 2; sn[sp] = sn[t] - np_pw_split(sn[t]); st[sp] = 0; ++sp; }
                else { val = sl[t] + val; have = true; --sp; }
                continue;
            }
            if (sn[t] <= NP_PW_BLOCK) { val = lsum[k++]; have = true; --sp; continue; }
            st[t] = 1; sn[sp] = np_pw_split(sn[t]); st[sp] = 0; ++sp;
        }
        s[blockIdx.x] = val;
    }
}

// pass 5: running float32 total over the pieces, mean in float64, cast to float32 (np.mean of an empty selection: nan)
__global__ void k_wase_finish(const float* __restrict__ s, const u64* __restrict__ total, float* __restrict__ bg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u64 M = *total;
    const u64 np = (M + NP_BUFSIZE - 1) / NP_BUFSIZE;
    float res = 0.f;
    for (u64 i = 0; i < np; ++i) res += s[i];
    *bg = (float)((double)res / (double)M);
}

// flow[p] = (flow[p] - background[p]) * scale      (reference :659, then the unit scale of :600)
__global__ __launch_bounds__(256) void k_wase_apply(float* __restrict__ flows, const float* __restrict__ bg, size_t hw2, float scale)
{
    const int p = blockIdx.y;
    const float b = bg[p];
    float* f = flows + (size_t)p * hw2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw2; i += (size_t)gridDim.x * 256) f[i] = (f[i] - b) * scale;
}
