// teeflow_saliency.hip.h -- cv2.saliency.StaticSaliencyFineGrained.computeSaliency() on the device: the frame preprocessing the
// reference applies when no_saliency=False (/root/reference/optical_flow/calculate_optical_flow.py:559-560, :586).
// Steps and arithmetic types are oracle/saliency_oracle.c's (which says what of opencv-contrib it restates and that nothing pins it):
//   gray (BGR2GRAY weights on the channels as handed over) -> 3x3 Gaussian twice (8-bit fixed point, reflect-101) -> float integral
//   image -> six centre-surround scales (on / off maps, truncated) summed -> each sum scaled by its own maximum -> (on + off)
//   scaled by the larger maximum.
// All of it is byte / integer work except the integral image (one float rounding per element, in raster order down each column)
// and the surround means (float, fixed expression order); HBM-bound, a few passes over N x H x W.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sal {

__device__ __forceinline__ int refl101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

// (uchar)double as x86 does it: cvttsd2si to int32 (0x80000000 when not representable), low byte
__device__ __forceinline__ uint8_t u8_from_f64(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) return 0;
    return (uint8_t)(uint32_t)(int)v;
}

__global__ __launch_bounds__(256) void k_sal_gray(const uint8_t* __restrict__ src, int channels, size_t n, uint8_t* __restrict__ gray)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (channels == 1) { gray[i] = src[i]; return; }
    const int c0 = src[i * 3], c1 = src[i * 3 + 1], c2 = src[i * 3 + 2];
    gray[i] = (uint8_t)((c0 * 3735 + c1 * 19235 + c2 * 9798 + (1 << 14)) >> 15);
}

// grid (ceil(W/256), H, N)
__global__ __launch_bounds__(256) void k_sal_blur3(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t base = (size_t)blockIdx.z * H * W;
    const uint8_t* r0 = src + base + (size_t)refl101(y - 1, H) * W;
    const uint8_t* r1 = src + base + (size_t)y * W;
    const uint8_t* r2 = src + base + (size_t)refl101(y + 1, H) * W;
    const int xl = refl101(x - 1, W), xr = refl101(x + 1, W);
    const int h0 = r0[xl] + 2 * r0[x] + r0[xr], h1 = r1[xl] + 2 * r1[x] + r1[xr], h2 = r2[xl] + 2 * r2[x] + r2[xr];
    dst[base + (size_t)y * W + x] = (uint8_t)((h0 + 2 * h1 + h2 + 8) >> 4);
}

// Row prefix sums (exact integers): one wave per row, 64 columns per step.  grid (H, N), block 64.
__global__ __launch_bounds__(64) void k_sal_rowprefix(const uint8_t* __restrict__ gray, int H, int W, int* __restrict__ P)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const size_t base = ((size_t)blockIdx.y * H + y) * W;
    int carry = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        int v = x < W ? (int)gray[base + x] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(v, off, 64);
            if (lane >= off) v += t;
        }
        if (x < W) P[base + x] = v + carry;
        carry += __shfl(v, 63, 64);
    }
}

// integral(gray, CV_32F): I is (H+1) x (W+1) per frame; column x+1 accumulates the row prefixes downwards, one rounding per row.
// grid (ceil((W+1)/256), N)
__global__ __launch_bounds__(256) void k_sal_integral(const int* __restrict__ P, int H, int W, float* __restrict__ I)
{
    const int xs = blockIdx.x * 256 + threadIdx.x;      // column of the integral image
    if (xs > W) return;
    const int SW = W + 1;
    float* If = I + (size_t)blockIdx.y * (H + 1) * SW;
    const int* Pf = P + (size_t)blockIdx.y * H * W;
    If[xs] = 0.f;
    if (xs == 0) { for (int y = 0; y < H; ++y) If[(size_t)(y + 1) * SW] = 0.f; return; }
    float t = 0.f;
    for (int y = 0; y < H; ++y) {
        t = t + (float)Pf[(size_t)y * W + xs - 1];
        If[(size_t)(y + 1) * SW + xs] = t;
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float surround_mean(const float* __restrict__ I, int SW, int SH, int x, int y, int nb, int center)
{
    const int x1 = clampi(x - nb + 1, 0, SW - 1), y1 = clampi(y - nb + 1, 0, SH - 1);
    const int x2 = clampi(x + nb + 1, 0, SW - 1), y2 = clampi(y + nb + 1, 0, SH - 1);
    float v = I[(size_t)y2 * SW + x2] + I[(size_t)y1 * SW + x1] - I[(size_t)y2 * SW + x1] - I[(size_t)y1 * SW + x2];
    v = (v - (float)center) / (float)((x2 - x1) * (y2 - y1) - 1);
    return v;
}

__device__ __forceinline__ void block_max2(int a, int b, int* dst)
{
    __shared__ int sa[4], sb[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a2 = __shfl_down(a, off, 64), b2 = __shfl_down(b, off, 64);
        a = a2 > a ? a2 : a; b = b2 > b ? b2 : b;
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { a = sa[w] > a ? sa[w] : a; b = sb[w] > b ? sb[w] : b; }
        atomicMax(dst, a); atomicMax(dst + 1, b);
    }
}

// The per-frame maxima are atomicMax words every block of the frame meets on: SAL_ROWS rows per block (one atomic pair per 256 x 8
// pixels instead of per 256) and one 128-byte line per frame (SAL_MX ints: frames do not share a line) -- with one row per block and
// four ints per frame the atomics were ~0.8 of the two kernels' time at 512 x 512 (133 k of them serialised on nine cache lines).
#define SAL_ROWS 8
#define SAL_MX 32

// six scales summed: mon / moff (uint16), and the per-frame maxima of the two sums -> mx[f][0..1].  grid (ceil(W/256), ceil(H/SAL_ROWS), N)
__global__ __launch_bounds__(256) void k_sal_scales(const uint8_t* __restrict__ gray, const float* __restrict__ I, int H, int W,
                                                    uint16_t* __restrict__ mon, uint16_t* __restrict__ moff, int* __restrict__ mx)
{
    const int x = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    int mxon = 0, mxoff = 0;
    if (x < W) {
        const float* If = I + (size_t)f * (H + 1) * (W + 1);
        const int nbs[6] = {12, 24, 48, 28, 56, 112};
        const int yend = min(H, (int)(blockIdx.y + 1) * SAL_ROWS);
#pragma unroll 1
        for (int y = blockIdx.y * SAL_ROWS; y < yend; ++y) {            // (unrolled eightfold this kernel was 52 KB of code)
            const size_t i = ((size_t)f * H + y) * W + x;
            const int g = gray[i];
            int son = 0, soff = 0;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float value = surround_mean(If, W + 1, H + 1, x, y, nbs[s], g);
                const float on = (float)g - value, off = value - (float)g;
                if (on > 0) son += u8_from_f64((double)on);
                if (off > 0) soff += u8_from_f64((double)off);
            }
            mon[i] = (uint16_t)son; moff[i] = (uint16_t)soff;
            mxon = son > mxon ? son : mxon; mxoff = soff > mxoff ? soff : mxoff;
        }
    }
    block_max2(mxon, mxoff, mx + SAL_MX * f);
}

// each sum scaled to 0..255 by its maximum; maxima of the two scaled maps -> mx[f][2..3].  Same grid.
__global__ __launch_bounds__(256) void k_sal_mix_scales(const uint16_t* __restrict__ mon, const uint16_t* __restrict__ moff, int H, int W,
                                                        uint8_t* __restrict__ ion, uint8_t* __restrict__ ioff, int* __restrict__ mx)
{
    const int x = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    int ma = 0, mb = 0;
    if (x < W) {
        const float max_on = (float)mx[SAL_MX * f], max_off = (float)mx[SAL_MX * f + 1];
        const int yend = min(H, (int)(blockIdx.y + 1) * SAL_ROWS);
        for (int y = blockIdx.y * SAL_ROWS; y < yend; ++y) {
            const size_t i = ((size_t)f * H + y) * W + x;
            const int a = u8_from_f64(255. * (double)((float)mon[i] / max_on));
            const int b = u8_from_f64(255. * (double)((float)moff[i] / max_off));
            ion[i] = (uint8_t)a; ioff[i] = (uint8_t)b;
            ma = a > ma ? a : ma; mb = b > mb ? b : mb;
        }
    }
    block_max2(ma, mb, mx + SAL_MX * f + 2);
}

__global__ __launch_bounds__(256) void k_sal_mix_onoff(const uint8_t* __restrict__ ion, const uint8_t* __restrict__ ioff, int H, int W,
                                                       const int* __restrict__ mx, uint8_t* __restrict__ out, float* __restrict__ outf)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const size_t i = ((size_t)f * H + y) * W + x;
    const int m = mx[SAL_MX * f + 2] > mx[SAL_MX * f + 3] ? mx[SAL_MX * f + 2] : mx[SAL_MX * f + 3];
    const uint8_t v = u8_from_f64(255. * (double)(float)((int)ion[i] + (int)ioff[i]) / (double)(float)m);
    // outf: what computeSaliency() hands back in opencv-contrib 4.x, `dst.convertTo(saliencyMap, CV_32F, 1.0f / 255.0f)` -- values in [0, 1]
    if (outf) outf[i] = (float)v * (1.0f / 255.0f);
    else out[i] = v;
}

}  // namespace sal
