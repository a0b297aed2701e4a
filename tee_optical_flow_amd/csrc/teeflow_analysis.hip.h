// teeflow_analysis.hip.h -- SURVEY.md row f1: radial / longitudinal projection of the flow and the per-frame statistics
// the reference derives from it (/root/reference/optical_flow/analysis.py:89-212: radial_vecgrid, calc_proj_mag,
// calculate_comp_magnitude, calc_bidirectional_hist).  The reference builds [N,H,W,2] float64 unit-vector grids in Python
// loops and calls np.percentile / np.histogram per frame; here the projection is one pass over the flow, the histogram a
// second, and the percentiles' order statistics an exact 4-pass radix select -- all in float64 with the reference's
// operation order (bit-identical to tests/golden/reference_analysis.npz, which the reference's own code produced).
#pragma once
#include "teeflow_kernels.hip.h"

// order-preserving map double -> u64 (total order of finite values; -0.0 < +0.0 is irrelevant: zeros are excluded)
__device__ __forceinline__ u64 f64_key(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double f64_unkey(u64 k)
{
    const u64 b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    union { u64 u; double d; } c; c.u = b;
    return c.d;
}

// rad/long planes [N][H][W] (f64), global min/max keys of both, per-frame non-zero counts
__global__ __launch_bounds__(256) void k_radlong_project(const float* __restrict__ flow, const double* __restrict__ cent, int H, int W,
                                                         double* __restrict__ rad, double* __restrict__ lon, u64* __restrict__ mm /* [4]: rad min,max, long min,max */,
                                                         unsigned long long* __restrict__ cnt /* [N][2] */)
{
    __shared__ u64 s[4][4];
    __shared__ unsigned sc[4][2];
    const int n = blockIdx.y;
    const size_t npx = (size_t)H * W;
    const double cH = cent[2 * n], cW = cent[2 * n + 1];
    u64 k[4] = {~0ull, 0ull, ~0ull, 0ull};
    unsigned c0 = 0, c1 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const int hh = (int)(i / W), ww = (int)(i - (size_t)hh * W);
        const double dh = cH - (double)hh, dw = cW - (double)ww;
        const double nrm = sqrt(dh * dh + dw * dw);
        double u0 = dh / nrm, u1 = dw / nrm;
        u0 = u0 == u0 ? u0 : 0.0; u1 = u1 == u1 ? u1 : 0.0;         // nan_to_num(nan=0): the centroid pixel itself
        const float2 f = reinterpret_cast<const float2*>(flow)[(size_t)n * npx + i];
        const double r = (double)f.x * u0 + (double)f.y * u1;         // sum(OF * unitvec): note OF[...,0] (x) meets the ROW component
        const double l = (double)f.x * u1 + (double)f.y * (-1.0 * u0);
        rad[(size_t)n * npx + i] = r;
        lon[(size_t)n * npx + i] = l;
        const u64 kr = f64_key(r), kl = f64_key(l);
        k[0] = kr < k[0] ? kr : k[0]; k[1] = kr > k[1] ? kr : k[1];
        k[2] = kl < k[2] ? kl : k[2]; k[3] = kl > k[3] ? kl : k[3];
        c0 += r != 0.0; c1 += l != 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 o = __shfl_down(k[j], off, 64);
            k[j] = (j & 1) ? (o > k[j] ? o : k[j]) : (o < k[j] ? o : k[j]);
        }
        c0 += __shfl_down(c0, off, 64); c1 += __shfl_down(c1, off, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 4; ++j) s[wv][j] = k[j]; sc[wv][0] = c0; sc[wv][1] = c1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < 4; ++w2) {
            for (int j = 0; j < 4; ++j) k[j] = (j & 1) ? (s[w2][j] > k[j] ? s[w2][j] : k[j]) : (s[w2][j] < k[j] ? s[w2][j] : k[j]);
            c0 += sc[w2][0]; c1 += sc[w2][1];
        }
        atomicMin(&mm[0], k[0]); atomicMax(&mm[1], k[1]); atomicMin(&mm[2], k[2]); atomicMax(&mm[3], k[3]);
        atomicAdd(&cnt[2 * n], (unsigned long long)c0); atomicAdd(&cnt[2 * n + 1], (unsigned long long)c1);
    }
}

// np.histogram(frame[frame != 0], bins=nbins, range=(first, last)) per frame: estimate, then numpy's edge fix-ups
__global__ __launch_bounds__(256) void k_radlong_hist(const double* __restrict__ v, size_t npx, const double* __restrict__ edges, int nbins,
                                                      unsigned long long* __restrict__ freq /* [N][nbins] */)
{
    const int n = blockIdx.y;
    const double first = edges[0], last = edges[nbins];
    const double denom = last - first;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const double x = v[(size_t)n * npx + i];
        if (x == 0.0 || !(x >= first && x <= last)) continue;
        int idx = (int)(((x - first) / denom) * (double)nbins);
        if (idx >= nbins) idx = nbins - 1;
        if (idx < 0) idx = 0;
        if (x < edges[idx]) --idx;
        else if (idx != nbins - 1 && x >= edges[idx + 1]) ++idx;
        atomicAdd(&freq[(size_t)n * nbins + idx], 1ull);
    }
}

// one 16-bit digit pass of the radix select: histogram of the digit among the non-zero values whose higher digits equal the
// slot's prefix (up to NSLOT order statistics per frame are selected at once)
#define RL_NSLOT 4
__global__ __launch_bounds__(256) void k_radlong_sel_hist(const double* __restrict__ v, size_t npx, int shift, const u64* __restrict__ prefix /* [N][NSLOT] */,
                                                          const int* __restrict__ active /* [N][NSLOT] */, unsigned* __restrict__ hist /* [N][NSLOT][65536] */)
{
    const int n = blockIdx.y;
    u64 pf[RL_NSLOT]; int ac[RL_NSLOT];
#pragma unroll
    for (int j = 0; j < RL_NSLOT; ++j) { pf[j] = prefix[n * RL_NSLOT + j]; ac[j] = active[n * RL_NSLOT + j]; }
    const u64 himask = shift == 48 ? 0ull : (~0ull << (shift + 16));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const double x = v[(size_t)n * npx + i];
        if (x == 0.0) continue;
        const u64 key = f64_key(x);
        const unsigned digit = (unsigned)((key >> shift) & 0xFFFFull);
#pragma unroll
        for (int j = 0; j < RL_NSLOT; ++j)
            if (ac[j] && (key & himask) == (pf[j] & himask)) atomicAdd(&hist[((size_t)n * RL_NSLOT + j) * 65536 + digit], 1u);
    }
}

// find the digit bucket that contains the slot's remaining rank; extend the prefix, reduce the rank
__global__ __launch_bounds__(256) void k_radlong_sel_scan(const unsigned* __restrict__ hist, int shift, u64* __restrict__ prefix, long long* __restrict__ rank,
                                                          const int* __restrict__ active)
{
    __shared__ unsigned long long part[256];
    const int slot = blockIdx.x;                       // n * NSLOT + j
    if (!active[slot]) return;
    const unsigned* hrow = hist + (size_t)slot * 65536;
    unsigned long long loc = 0;
    for (int b = 0; b < 256; ++b) loc += hrow[threadIdx.x * 256 + b];
    part[threadIdx.x] = loc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long r = rank[slot];
        int t = 0;
        while (t < 255 && (long long)part[t] <= r) { r -= (long long)part[t]; ++t; }
        int b = 0;
        while (b < 255 && (long long)hrow[t * 256 + b] <= r) { r -= (long long)hrow[t * 256 + b]; ++b; }
        const u64 digit = (u64)(t * 256 + b);
        prefix[slot] |= digit << shift;
        rank[slot] = r;
    }
}
