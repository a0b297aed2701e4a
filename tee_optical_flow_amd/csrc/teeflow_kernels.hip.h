// teeflow_kernels.hip.h -- gfx950 (MI355X) kernels of the DualTVL1 engine.
//
// What they compute is OpenCV's CPU DualTVL1 as the reference reaches it through
// /root/reference/optical_flow/calculate_optical_flow.py:577-578, 642.
// Arithmetic contract: every float expression is evaluated in the written order with IEEE
// single/double operations and NO fused multiply-add (build flag -ffp-contract=off), so results
// are bit-identical to the CPU oracle in oracle/tvl1_oracle.c (which tests/ compare against).
//
// Data layout in HBM: every image-like quantity is a stack of fp32 planes [pair|frame][h][pitch],
// pitch = round_up(w, 32) floats (128-B rows, float4-aligned), plane = pitch*h.  Per level the
// engine keeps: the frame pyramid (I0/I1 share frames in sequence mode), three per-warp constants
// (I1wx, I1wy, rho_c -- |grad|^2 is recomputed, I1x/I1y are never materialised) and the
// ping-pong state u1,u2 (x2) and p11,p12,p21,p22 (x2).
//
// All kernels are HBM-bandwidth-bound stencils (no MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define TF_HD __host__ __device__
#include "median_net.h"

typedef unsigned long long u64;

struct Geom {
    int w, h, pitch;
    long long plane;   // floats per frame plane of the pyramid (pitch*h of THIS level)
    long long splane;  // floats between consecutive pairs in the state/constant buffers (level-0 plane at
                       // every level, so pairs whose ping-pong parity differs never overlap across levels)
};

struct PairCtl {
    int ubase;  // which of the two u buffers holds this pair's current flow at stage start
    int pbase;  // same for the dual variable
};

struct StateBufs {
    float* u1[2]; float* u2[2];
    float* p11[2]; float* p12[2]; float* p21[2]; float* p22[2];
};

#define ERR_SCALE_F 1073741824.0f  // 2^30: per-pixel convergence term -> exact integer (oracle D1)
#define ERR_CAP_F 4096.0f

#define UNPACK4(dst, v) { dst[0] = (v).x; dst[1] = (v).y; dst[2] = (v).z; dst[3] = (v).w; }
#define PACK4(a) make_float4((a)[0], (a)[1], (a)[2], (a)[3])

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

// active(pair, it): did iteration it-1 leave error > threshold?  Slots are zeroed per stage, so a pair
// that stopped earlier reads 0 and stays stopped.
__device__ __forceinline__ bool pair_active(const u64* errb, int it, double thr_q)
{
    if (it == 0) return true;
    return (double)errb[it - 1] > thr_q;
}

// ---------------------------------------------------------------------------------------------
// u8 -> f32 (cv::Mat::convertTo(CV_32F, 1.0)): dense [F][H][W] bytes -> pitched fp32 planes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* __restrict__ src, float* __restrict__ dst, Geom g)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= g.w) return;
    dst[(size_t)f * g.plane + (size_t)y * g.pitch + x] = (float)src[((size_t)f * g.h + y) * g.w + x];
}

// CV_32FC1 frames.  cv2's DualTVL1 brings them to the 0..255 range, convertTo(CV_32F, 255.0) = one fp32 multiply per pixel (scale255);
// cv2's DeepFlow takes them as they are, convertTo(CV_32F) without a factor = a copy.
__global__ __launch_bounds__(256) void k_f32_to_level0(const float* __restrict__ src, float* __restrict__ dst, Geom g, int scale255)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= g.w) return;
    const float v = src[((size_t)f * g.h + y) * g.w + x];
    dst[(size_t)f * g.plane + (size_t)y * g.pitch + x] = scale255 ? v * 255.0f : v;
}

// ---------------------------------------------------------------------------------------------
// cv::resize INTER_LINEAR on CV_32FC1 (see oracle orc_resize_linear for the border rules)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float resize_px(const float* __restrict__ S, int sw, int sh, int spitch,
                                           int dx, int dy, double scale_x, double scale_y)
{
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= (float)sx;
    bool tail = false;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx + 1 >= sw) { tail = true; if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; } }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= (float)sy;
    const float b0 = 1.f - fy, b1 = fy;
    const int r0 = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const int r1 = sy + 1 >= 0 ? (sy + 1 < sh ? sy + 1 : sh - 1) : 0;
    const float* S0 = S + (size_t)r0 * spitch;
    const float* S1 = S + (size_t)r1 * spitch;
    float t0, t1;
    if (tail) { t0 = S0[sx]; t1 = S1[sx]; }
    else {
        const float a1 = fx, a0 = 1.f - fx;
        t0 = S0[sx] * a0 + S0[sx + 1] * a1;
        t1 = S1[sx] * a0 + S1[sx + 1] * a1;
    }
    return t0 * b0 + t1 * b1;
}

// cv::cuda::resize INTER_LINEAR as its resize_linear kernel samples (TF_VARIANT_CUDA only; oracle orc_resize_cuda, [UPSTREAM-FROM-MEMORY]):
// no half-pixel shift, replicate at the far edges, four weighted taps accumulated in float in upstream's order
__device__ __forceinline__ float resize_px_cuda(const float* __restrict__ S, int sw, int sh, int spitch, int dx, int dy, float scale_x, float scale_y)
{
    const float src_x = (float)dx * scale_x, src_y = (float)dy * scale_y;
    int x1 = cv_floor_f(src_x), y1 = cv_floor_f(src_y);
    x1 = x1 > sw - 1 ? sw - 1 : x1; y1 = y1 > sh - 1 ? sh - 1 : y1;
    const int x2 = x1 + 1, y2 = y1 + 1, x2r = x2 < sw - 1 ? x2 : sw - 1, y2r = y2 < sh - 1 ? y2 : sh - 1;
    const float* S1 = S + (size_t)y1 * spitch;
    const float* S2 = S + (size_t)y2r * spitch;
    float out = 0.f;
    out = out + S1[x1] * (((float)x2 - src_x) * ((float)y2 - src_y));
    out = out + S1[x2r] * ((src_x - (float)x1) * ((float)y2 - src_y));
    out = out + S2[x1] * (((float)x2 - src_x) * (src_y - (float)y1));
    out = out + S2[x2r] * ((src_x - (float)x1) * (src_y - (float)y1));
    return out;
}

// pyramid: one level down for every frame (cuda_sampling: the CUDA class's cuda::resize rule, scale = (float)(1/fx))
__global__ __launch_bounds__(256) void k_pyr_down(const float* __restrict__ src, Geom gs, float* __restrict__ dst, Geom gd,
                                                  double scale_x, double scale_y, int cuda_sampling = 0)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (dx >= gd.w || dy >= gd.h) return;
    dst[(size_t)f * gd.plane + (size_t)dy * gd.pitch + dx] = cuda_sampling
        ? resize_px_cuda(src + (size_t)f * gs.plane, gs.w, gs.h, gs.pitch, dx, dy, (float)scale_x, (float)scale_y)
        : resize_px(src + (size_t)f * gs.plane, gs.w, gs.h, gs.pitch, dx, dy, scale_x, scale_y);
}

// flow: coarse level -> next finer level, times 1/scaleStep (resize + multiply of DualTVL1::calc)
__global__ __launch_bounds__(256) void k_flow_up(StateBufs sb, const PairCtl* __restrict__ ctl, Geom gs, Geom gd,
                                                 double scale_x, double scale_y, float mul, int cuda_sampling = 0)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (dx >= gd.w || dy >= gd.h) return;
    const int uc = ctl[b].ubase & 1;
    const size_t di = (size_t)b * gd.splane + (size_t)dy * gd.pitch + dx;
    if (cuda_sampling) {
        sb.u1[uc ^ 1][di] = resize_px_cuda(sb.u1[uc] + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, (float)scale_x, (float)scale_y) * mul;
        sb.u2[uc ^ 1][di] = resize_px_cuda(sb.u2[uc] + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, (float)scale_x, (float)scale_y) * mul;
        return;
    }
    sb.u1[uc ^ 1][di] = resize_px(sb.u1[uc] + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, scale_x, scale_y) * mul;
    sb.u2[uc ^ 1][di] = resize_px(sb.u2[uc] + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, scale_x, scale_y) * mul;
}

// ---------------------------------------------------------------------------------------------
// Warp stage (tvl1flow.cpp: buildFlowMap + 3x cv::remap INTER_CUBIC/BORDER_CONSTANT + calcGradRho).
// The centred gradient of I1 is evaluated on the fly from a clamped 6x6 patch (bit-identical to
// gradient-then-remap: each tap's gradient is the same 0.5f*(next-prev) of the same two pixels),
// so I1x/I1y never exist in HBM.  Output: I1wx, I1wy, rho_c.
// ---------------------------------------------------------------------------------------------
struct WarpArgs {
    const float* pyr;       // this level's frame planes
    int off0, off1;         // pair b uses frames off0+b (I0) and off1+b (I1)
    StateBufs sb;
    const PairCtl* ctl;
    const float* tab;       // [32][4] bicubic coefficients (A = -0.75)
    float *wx, *wy, *rho;
    Geom g;
};

// one output pixel of the warp stage; every pointer is already offset to the pair's plane (I0 / I1: to its frames)
__device__ __forceinline__ void warp_px(const float* stab, const float* __restrict__ I0, const float* __restrict__ I1,
                                        const float* __restrict__ gu1, const float* __restrict__ gu2,
                                        float* __restrict__ owx, float* __restrict__ owy, float* __restrict__ orho,
                                        int W, int H, int pitch, int x, int y)
{
    const size_t idx = (size_t)y * pitch + x;
    const float u1 = gu1[idx], u2 = gu2[idx];
    const float mx = (float)x + u1, my = (float)y + u2;
    const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
    const float* wxp = stab + (sx & 31) * 4;
    const float* wyp = stab + (sy & 31) * 4;
    int ixs = sx >> 5, iys = sy >> 5;
    ixs = clampi(ixs, -32768, 32767); iys = clampi(iys, -32768, 32767);   // saturate_cast<short>
    const int ix = ixs - 1, iy = iys - 1;
    float vI = 0.f, vX = 0.f, vY = 0.f;
    if (!(ix >= W || ix + 4 <= 0 || iy >= H || iy + 4 <= 0)) {
        float P[6][6];
        unsigned xo[6], yo[6];                             // unsigned 32-bit BYTE offsets from the frame base (a plane is < 2^24 px):
                                                           // the loads take the scalar-base + 32-bit-offset form, no 64-bit address math
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            xo[i] = (unsigned)clampi(ix - 1 + i, 0, W - 1) * 4u;
            yo[i] = (unsigned)clampi(iy - 1 + i, 0, H - 1) * (unsigned)pitch * 4u;
        }
        const char* base1 = reinterpret_cast<const char*>(I1);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 6; ++i) P[j][i] = *reinterpret_cast<const float*>(base1 + (yo[j] + xo[i]));
        float wgt[16];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) wgt[k1 * 4 + k2] = wyp[k1] * wxp[k2];
        const unsigned width1 = (unsigned)(W - 3 > 0 ? W - 3 : 0), height1 = (unsigned)(H - 3 > 0 ? H - 3 : 0);
        if ((unsigned)ix < width1 && (unsigned)iy < height1) {
            // interior: each source row summed left to right, rows accumulated in order
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                float rI = P[k1 + 1][1] * wgt[k1 * 4];
                float rX = (0.5f * (P[k1 + 1][2] - P[k1 + 1][0])) * wgt[k1 * 4];
                float rY = (0.5f * (P[k1 + 2][1] - P[k1][1])) * wgt[k1 * 4];
#pragma unroll
                for (int k2 = 1; k2 < 4; ++k2) {
                    rI = rI + P[k1 + 1][k2 + 1] * wgt[k1 * 4 + k2];
                    rX = rX + (0.5f * (P[k1 + 1][k2 + 2] - P[k1 + 1][k2])) * wgt[k1 * 4 + k2];
                    rY = rY + (0.5f * (P[k1 + 2][k2 + 1] - P[k1][k2 + 1])) * wgt[k1 * 4 + k2];
                }
                if (k1 == 0) { vI = rI; vX = rX; vY = rY; }
                else { vI += rI; vX += rX; vY += rY; }
            }
        } else {
            // partially outside: constant border 0, valid taps accumulated one by one
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                const int yi = iy + k1;
                if (yi < 0 || yi >= H) continue;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int xj = ix + k2;
                    if (xj < 0 || xj >= W) continue;
                    vI += P[k1 + 1][k2 + 1] * wgt[k1 * 4 + k2];
                    vX += (0.5f * (P[k1 + 1][k2 + 2] - P[k1 + 1][k2])) * wgt[k1 * 4 + k2];
                    vY += (0.5f * (P[k1 + 2][k2 + 1] - P[k1][k2 + 1])) * wgt[k1 * 4 + k2];
                }
            }
        }
    }
    owx[idx] = vX;
    owy[idx] = vY;
    orho[idx] = ((vI - vX * u1) - vY * u2) - I0[idx];
}

__global__ __launch_bounds__(256) void k_warp(WarpArgs a)
{
    __shared__ float stab[128];
    if (threadIdx.x < 128) stab[threadIdx.x] = a.tab[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    if (x >= W || y >= H) return;
    const int uc = a.ctl[b].ubase & 1;
    const size_t po = (size_t)b * a.g.splane;
    warp_px(stab, a.pyr + (size_t)(a.off0 + b) * a.g.plane, a.pyr + (size_t)(a.off1 + b) * a.g.plane, a.sb.u1[uc] + po, a.sb.u2[uc] + po,
            a.wx + po, a.wy + po, a.rho + po, W, H, pitch, x, y);
}

// ---- cv2.cuda.OpticalFlowDual_TVL1 variant (oracle variant 1) -----------------------------------
// centeredGradientKernel: 0.5 * (next - prev) with replicate at the border, for every frame of a level
__global__ __launch_bounds__(256) void k_grad(const float* __restrict__ src, float* __restrict__ gx, float* __restrict__ gy, Geom g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.w || y >= g.h) return;
    const size_t fo = (size_t)blockIdx.z * g.plane;
    const float* S = src + fo;
    const size_t i = (size_t)y * g.pitch + x;
    const int xp = x + 1 < g.w ? x + 1 : g.w - 1, xm = x > 0 ? x - 1 : 0, yp = y + 1 < g.h ? y + 1 : g.h - 1, ym = y > 0 ? y - 1 : 0;
    gx[fo + i] = 0.5f * (S[(size_t)y * g.pitch + xp] - S[(size_t)y * g.pitch + xm]);
    gy[fo + i] = 0.5f * (S[(size_t)yp * g.pitch + x] - S[(size_t)ym * g.pitch + x]);
}

__device__ __forceinline__ float cuda_bicubic_coeff(float x_)
{
    const float x = fabsf(x_);
    if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

struct WarpCudaArgs {
    WarpArgs w;
    const float *gx, *gy;      // centred gradient of this level's frames (same layout as w.pyr)
};

// warpBackwardKernel: weight-normalised Catmull-Rom taps over ceil(w-2)..floor(w+2) with clamp addressing, on I1, I1x, I1y
__global__ __launch_bounds__(256) void k_warp_cuda(WarpCudaArgs A)
{
    const WarpArgs& a = A.w;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    if (x >= W || y >= H) return;
    const int uc = a.ctl[b].ubase & 1;
    const size_t po = (size_t)b * a.g.splane, idx = (size_t)y * pitch + x;
    const size_t f1 = (size_t)(a.off1 + b) * a.g.plane;
    const float* __restrict__ I0 = a.pyr + (size_t)(a.off0 + b) * a.g.plane;
    const float* __restrict__ I1 = a.pyr + f1;
    const float* __restrict__ I1x = A.gx + f1;
    const float* __restrict__ I1y = A.gy + f1;
    const float u1v = a.sb.u1[uc][po + idx], u2v = a.sb.u2[uc][po + idx];
    const float wx = (float)x + u1v, wy = (float)y + u2v;
    const int xmin = (int)ceilf(wx - 2.0f), xmax = (int)floorf(wx + 2.0f);
    const int ymin = (int)ceilf(wy - 2.0f), ymax = (int)floorf(wy + 2.0f);
    float sum = 0.0f, sumx = 0.0f, sumy = 0.0f, wsum = 0.0f;
    for (int cy = ymin; cy <= ymax; ++cy) {
        const float wyc = cuda_bicubic_coeff(wy - (float)cy);
        const size_t row = (size_t)clampi(cy, 0, H - 1) * pitch;
        for (int cx = xmin; cx <= xmax; ++cx) {
            const float wt = cuda_bicubic_coeff(wx - (float)cx) * wyc;
            const size_t j = row + clampi(cx, 0, W - 1);
            sum += wt * I1[j];
            sumx += wt * I1x[j];
            sumy += wt * I1y[j];
            wsum += wt;
        }
    }
    const float coeff = 1.0f / wsum;
    const float I1w = sum * coeff, gxv = sumx * coeff, gyv = sumy * coeff;
    a.wx[po + idx] = gxv;
    a.wy[po + idx] = gyv;
    a.rho[po + idx] = ((I1w - gxv * u1v) - gyv * u2v) - I0[idx];
}

// ---------------------------------------------------------------------------------------------
// k_warp_lds: same arithmetic as k_warp, but the 36 taps of a pixel come from an LDS copy of the I1 tile plus a margin
// of M pixels (k_warp is L1/TA-bound on its 36 scalar gathers per pixel; ds_read_b32 from a staged tile is ~7x cheaper).
// A pixel whose 6x6 footprint leaves the staged region (|flow| > ~M) falls back to clamped global loads, so the result
// never depends on M.  The staged array holds I1 at UNclamped coordinates with replicate content, which is exactly what
// the clamped patch loads of k_warp read.  All global loads of a thread are in flight together (see median_stage).
// ---------------------------------------------------------------------------------------------
#define WL_TW 64
#define WL_TH 16

__device__ __forceinline__ void warp_accumulate(const float (&P)[6][6], const float* wxp, const float* wyp, int ix, int iy, int W, int H,
                                                float& vI, float& vX, float& vY)
{
    float wgt[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) wgt[k1 * 4 + k2] = wyp[k1] * wxp[k2];
    const unsigned width1 = (unsigned)(W - 3 > 0 ? W - 3 : 0), height1 = (unsigned)(H - 3 > 0 ? H - 3 : 0);
    vI = vX = vY = 0.f;
    if ((unsigned)ix < width1 && (unsigned)iy < height1) {
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            float rI = P[k1 + 1][1] * wgt[k1 * 4];
            float rX = (0.5f * (P[k1 + 1][2] - P[k1 + 1][0])) * wgt[k1 * 4];
            float rY = (0.5f * (P[k1 + 2][1] - P[k1][1])) * wgt[k1 * 4];
#pragma unroll
            for (int k2 = 1; k2 < 4; ++k2) {
                rI = rI + P[k1 + 1][k2 + 1] * wgt[k1 * 4 + k2];
                rX = rX + (0.5f * (P[k1 + 1][k2 + 2] - P[k1 + 1][k2])) * wgt[k1 * 4 + k2];
                rY = rY + (0.5f * (P[k1 + 2][k2 + 1] - P[k1][k2 + 1])) * wgt[k1 * 4 + k2];
            }
            if (k1 == 0) { vI = rI; vX = rX; vY = rY; }
            else { vI += rI; vX += rX; vY += rY; }
        }
    } else {
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const int yi = iy + k1;
            if (yi < 0 || yi >= H) continue;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int xj = ix + k2;
                if (xj < 0 || xj >= W) continue;
                vI += P[k1 + 1][k2 + 1] * wgt[k1 * 4 + k2];
                vX += (0.5f * (P[k1 + 1][k2 + 2] - P[k1 + 1][k2])) * wgt[k1 * 4 + k2];
                vY += (0.5f * (P[k1 + 2][k2 + 1] - P[k1][k2 + 1])) * wgt[k1 * 4 + k2];
            }
        }
    }
}

template <int M>
__global__ __launch_bounds__(256) void k_warp_lds(WarpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stab = smem;                  // [128] bicubic table
    float* S = smem + 128;               // [SH][SW] staged I1 (16-byte aligned rows)
    // horizontal margin M + 4 on both sides (the region then starts on a float4 boundary), vertical M + 3 above / M + 4 below
    constexpr int MX = M + 4, SW = WL_TW + 2 * MX, SH = WL_TH + 2 * M + 7, QW = SW / 4, NQ = QW * SH, NV = (NQ + 255) / 256;
    static_assert(M % 4 == 0, "margin classes are multiples of 4");
    const int b = blockIdx.z;
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    const int x0 = blockIdx.x * WL_TW, y0 = blockIdx.y * WL_TH;
    const int rx0 = x0 - MX, ry0 = y0 - M - 3;
    const float* __restrict__ I0 = a.pyr + (size_t)(a.off0 + b) * a.g.plane;
    const float* __restrict__ I1 = a.pyr + (size_t)(a.off1 + b) * a.g.plane;
    const int uc = a.ctl[b].ubase & 1;
    const size_t po = (size_t)b * a.g.splane;
    const int lx = threadIdx.x & 63, x = x0 + lx, ty = threadIdx.x >> 6;
    // everything this thread reads from global memory is requested before the first wait: the flow and I0 of its four
    // pixels, then its share of the staged region (float4 where the quad lies inside the image, clamped scalars at the border)
    float u1r[4], u2r[4], i0r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + ty + 4 * r;
        u1r[r] = u2r[r] = i0r[r] = 0.f;
        if (x < W && y < H) {
            const size_t idx = (size_t)y * pitch + x;
            u1r[r] = a.sb.u1[uc][po + idx]; u2r[r] = a.sb.u2[uc][po + idx]; i0r[r] = I0[idx];
        }
    }
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NQ) {
            const int ly = i / QW, gx = rx0 + 4 * (i - ly * QW);
            const float* row = I1 + (size_t)clampi(ry0 + ly, 0, H - 1) * pitch;
            if (gx >= 0 && gx + 3 < W) v[k] = *reinterpret_cast<const float4*>(row + gx);
            else v[k] = make_float4(row[clampi(gx, 0, W - 1)], row[clampi(gx + 1, 0, W - 1)], row[clampi(gx + 2, 0, W - 1)], row[clampi(gx + 3, 0, W - 1)]);
        }
    }
    if (threadIdx.x < 128) stab[threadIdx.x] = a.tab[threadIdx.x];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NQ) *reinterpret_cast<float4*>(S + 4 * i) = v[k];
    }
    __syncthreads();
    if (x >= W) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + ty + 4 * r;
        if (y >= H) break;
        const size_t idx = (size_t)y * pitch + x;
        const float u1 = u1r[r], u2 = u2r[r];
        const float mx = (float)x + u1, my = (float)y + u2;
        const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
        const float* wxp = stab + (sx & 31) * 4;
        const float* wyp = stab + (sy & 31) * 4;
        const int ix = clampi(sx >> 5, -32768, 32767) - 1, iy = clampi(sy >> 5, -32768, 32767) - 1;   // saturate_cast<short>
        float vI = 0.f, vX = 0.f, vY = 0.f;
        if (!(ix >= W || ix + 4 <= 0 || iy >= H || iy + 4 <= 0)) {
            float P[6][6];
            const int px = ix - 1 - rx0, py = iy - 1 - ry0;
            if (px >= 0 && py >= 0 && px + 6 <= SW && py + 6 <= SH) {
                const float* Sp = S + py * SW + px;
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int i = 0; i < 6; ++i) P[j][i] = Sp[j * SW + i];
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float* row = I1 + (size_t)clampi(iy - 1 + j, 0, H - 1) * pitch;
#pragma unroll
                    for (int i = 0; i < 6; ++i) P[j][i] = row[clampi(ix - 1 + i, 0, W - 1)];
                }
            }
            warp_accumulate(P, wxp, wyp, ix, iy, W, H, vI, vX, vY);
        }
        a.wx[po + idx] = vX;
        a.wy[po + idx] = vY;
        a.rho[po + idx] = ((vI - vX * u1) - vY * u2) - i0r[r];
    }
}

// ---------------------------------------------------------------------------------------------
// cv::medianBlur(u, u, KS) on both flow planes, BORDER_REPLICATE, for pairs still iterating.
// Tile 64x16 outputs per 256-thread block, staged through LDS with its halo; grid.z = 2*B.
// ---------------------------------------------------------------------------------------------
struct MedArgs {
    StateBufs sb;
    const PairCtl* ctl;
    const u64* err; int errstride; int it; double thr_q; int utog;
    Geom g;
};

// Stage one 64 x 16 output tile of a plane with its halo (replicate border) in LDS.  Every load of a thread is issued
// before its first LDS write: one memory round trip per block instead of one per 256 staged values (the staging loop used
// to be ten dependent load -> wait -> write trips, several times the 1.5 k cycles the selection network takes).  Tiles
// whose 64 columns lie inside the image take float4 loads for the body and scalar loads for the 2R halo columns.
template <int KS>
__device__ __forceinline__ void median_stage(float (*t)[64 + 2 * (KS / 2)], const float* __restrict__ src, int x0, int y0, int W, int H, int pitch)
{
    constexpr int R = KS / 2, LW = 64 + 2 * R, LH = 16 + 2 * R;
    const int tid = threadIdx.x;
    if (x0 + 64 <= W) {
        constexpr int NV = (LH * 16 + 255) / 256;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k, ly = i >> 4, q = i & 15;
            if (i < LH * 16) v[k] = *reinterpret_cast<const float4*>(src + (size_t)clampi(y0 - R + ly, 0, H - 1) * pitch + x0 + 4 * q);
        }
        const bool halo = tid < LH * 2 * R;
        const int hly = tid / (2 * R), hc = tid % (2 * R), hlx = hc < R ? hc : 64 + hc;
        float hv = 0.f;
        if (halo) hv = src[(size_t)clampi(y0 - R + hly, 0, H - 1) * pitch + clampi(x0 - R + hlx, 0, W - 1)];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k, ly = i >> 4, q = i & 15;
            if (i < LH * 16) {
                float* d = &t[ly][R + 4 * q];
                if constexpr (R % 2 == 0) {            // 8-byte aligned: two ds_write_b64
                    *reinterpret_cast<float2*>(d) = make_float2(v[k].x, v[k].y);
                    *reinterpret_cast<float2*>(d + 2) = make_float2(v[k].z, v[k].w);
                } else { d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w; }
            }
        }
        if (halo) t[hly][hlx] = hv;
    } else {
        constexpr int NS = (LH * LW + 255) / 256;
        float sv[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = tid + 256 * k, ly = i / LW, lx = i % LW;
            if (i < LH * LW) sv[k] = src[(size_t)clampi(y0 - R + ly, 0, H - 1) * pitch + clampi(x0 - R + lx, 0, W - 1)];
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = tid + 256 * k;
            if (i < LH * LW) (&t[0][0])[i] = sv[k];
        }
    }
    __syncthreads();
}

// median of a staged (TH+2R) x (TW+2R) tile: 5x5 -> each thread produces FOUR horizontally adjacent outputs from one 5x8
// window (tf_median25_row4: shared column sorts and merges, 76 min/max/med3 per output instead of 198); 3x3 -> one
// output per thread and row as before.
template <int KS, int LW>
__device__ __forceinline__ void median_tile(const float (*t)[LW], float* __restrict__ dst, int x0, int y0, int W, int H, int pitch)
{
    if constexpr (KS == 5) {
        const int qx = threadIdx.x & 15, ly = threadIdx.x >> 4;        // 16 quads x 16 rows = the 64 x 16 tile
        const int x = x0 + 4 * qx, y = y0 + ly;
        if (x < W && y < H) {
            float col[8][5], out[4];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float4 lo = *reinterpret_cast<const float4*>(&t[ly + r][4 * qx]);
                const float4 hi = *reinterpret_cast<const float4*>(&t[ly + r][4 * qx + 4]);
                col[0][r] = lo.x; col[1][r] = lo.y; col[2][r] = lo.z; col[3][r] = lo.w;
                col[4][r] = hi.x; col[5][r] = hi.y; col[6][r] = hi.z; col[7][r] = hi.w;
            }
            tf_median25_row4(col, out);
            float* o = dst + (size_t)y * pitch + x;
            if (x + 3 < W) *reinterpret_cast<float4*>(o) = make_float4(out[0], out[1], out[2], out[3]);
            else
#pragma unroll
                for (int i = 0; i < 4; ++i) if (x + i < W) o[i] = out[i];
        }
    } else {
        const int lx = threadIdx.x & 63, x = x0 + lx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ly = (threadIdx.x >> 6) + 4 * r, y = y0 + ly;
            if (x < W && y < H) {
                float p[KS * KS];
#pragma unroll
                for (int j = 0; j < KS; ++j)
#pragma unroll
                    for (int i = 0; i < KS; ++i) p[j * KS + i] = t[ly + j][lx + i];
                dst[(size_t)y * pitch + x] = tf_median9(p);
            }
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256) void k_median(MedArgs a)
{
    constexpr int R = KS / 2, TWm = 64, THm = 16, LW = TWm + 2 * R, LH = THm + 2 * R;
    __shared__ __attribute__((aligned(16))) float t[LH][LW];
    const int b = blockIdx.z >> 1, plane = blockIdx.z & 1;
    if (!pair_active(a.err + (size_t)b * a.errstride, a.it, a.thr_q)) return;
    const int uc = (a.ctl[b].ubase ^ a.utog) & 1;
    const size_t po = (size_t)b * a.g.splane;
    const float* __restrict__ src = (plane ? a.sb.u2[uc] : a.sb.u1[uc]) + po;
    float* __restrict__ dst = (plane ? a.sb.u2[uc ^ 1] : a.sb.u1[uc ^ 1]) + po;
    const int x0 = blockIdx.x * TWm, y0 = blockIdx.y * THm, W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    median_stage<KS>(t, src, x0, y0, W, H, pitch);
    median_tile<KS, LW>(t, dst, x0, y0, W, H, pitch);
}

// ---------------------------------------------------------------------------------------------
// tvl1_iter: ONE fused inner iteration of procOneScale for a tile:
//   estimateV (threshold TH) -> divergence(p) -> estimateU (+ convergence term) ->
//   forwardGradient(u') -> estimateDualVariables
// 256 threads = 16 quads x 16 rows compute u' on a 64x16 px region (float4 per thread); the block
// OUTPUTS the 60x15 sub-tile whose forward differences it can form from that region (u' goes
// through LDS for the x+1 / y+1 neighbours).  u and p are ping-ponged so neighbouring tiles always
// read the previous iterate.  Algorithmic traffic: 9 plane reads + 6 plane writes = 60 B/px.
// The convergence sum is accumulated exactly (uint64 of rint(t*2^30)), one atomic per block.
// ---------------------------------------------------------------------------------------------
#define IT_TW 64
#define IT_TH 16
#define IT_OW 60
#define IT_OH 15

struct IterArgs {
    const float *wx, *wy, *rho;
    StateBufs sb;
    const PairCtl* ctl;
    u64* err; int errstride; int it; double thr_q;
    int utog, ptog, pzero;
    Geom g;
    float l_t, theta, taut;
    int* host_slot;   // host-mapped word: block 0 publishes how many pairs are still iterating at this launch
    int B;
    int variant;      // 0 = cv2.optflow CPU DualTVL1; 1 = cv2.cuda.OpticalFlowDual_TVL1 stop rule
    double thr_d;     // epsilon^2 * area in double: the CUDA class keeps scaledEpsilon, error and prevError in double (variant 1)
};

// Block 0 / wave 0 tells the host how many of the B pairs enter iteration `it` active, through fine-grained
// host memory.  The host reads it a few launches later (never blocking the stream) to stop enqueuing a stage.
__device__ __forceinline__ void publish_active_count(const IterArgs& a)
{
    if (a.host_slot && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) {
        int c = 0;
        for (int b2 = threadIdx.x; b2 < a.B; b2 += 64)
            c += pair_active(a.err + (size_t)b2 * a.errstride, a.it, a.thr_q) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (threadIdx.x == 0) __hip_atomic_store(a.host_slot, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- exact arithmetic helpers shared by the three tvl1_iter forms ----------------------------------------
// All of them return the SAME bits as the plain C expressions in the oracle; they only drop work that the
// generic lowering does for operand ranges that cannot occur here.  tests/test_gpu_kernels.py compares the
// results with the oracle bit for bit.

// oracle D2: (float)sqrt((double)a*a + (double)b*b).  a*a, b*b are exact in double, so fma(a,a,b*b) is the same
// single rounding as the sum of the two products.  The square root is the Goldschmidt sequence hipcc emits for
// sqrt(double) (correctly rounded), without its ldexp rescaling for x < 2^-767: x is 0 or >= 2^-298 here.
__device__ __forceinline__ float hypot_exact(float a, float b)
{
    const double ad = (double)a, bd = (double)b;
    const double x = __builtin_fma(ad, ad, bd * bd);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    g = (x == 0.0 || x == __builtin_inf()) ? x : g;
    return (float)g;
}

// rint(x) for 0 <= x < 2^43 as uint64 (x = min(t,4096)*2^30): split at 2^32, both halves exact
__device__ __forceinline__ u64 rint_u64(float x)
{
    const float v = __builtin_rintf(x);
    const float hi = __builtin_floorf(v * 0x1p-32f);
    const float lo = __builtin_fmaf(hi, -0x1p32f, v);
    return ((u64)(unsigned)hi << 32) | (u64)(unsigned)lo;
}

// estimateV + divergence + estimateU for the 4 pixels of one quad.  `ytop` is uniform per row; only the first pixel of
// the first quad (x == 0) has no left neighbour.
struct QuadU {
    float u1k[4], u2k[4], wx[4], wy[4], r[4];       // current flow and warp constants
    float p11[4], p12[4], p21[4], p22[4];           // dual variable at the pixel
    float p12u[4], p22u[4];                         // ... one row up
    float l11, l21;                                 // ... p11/p21 of the pixel left of the quad
};

__device__ __forceinline__ void tv_u_quad(float l_t, float theta, const QuadU& q, bool ytop, bool x0, float* u1n, float* u2n)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // estimateV, branch-free: the three cases of the thresholding step become selects (the quotient is computed
        // in every lane and discarded where it does not apply; straight-line code lets the 4 pixels interleave)
        const float Ix2 = q.wx[i] * q.wx[i], Iy2 = q.wy[i] * q.wy[i];
        const float grad = Ix2 + Iy2;
        const float rho = q.r[i] + (q.wx[i] * q.u1k[i] + q.wy[i] * q.u2k[i]);
        const float lg = l_t * grad;
        const bool c1 = rho < -lg, c2 = rho > lg, c3 = grad > FLT_EPSILON;
        const float fi = -rho / grad;
        const float k = c1 ? l_t : (c2 ? -l_t : fi);
        const bool any = c1 || c2 || c3;
        const float d1 = any ? k * q.wx[i] : 0.f, d2 = any ? k * q.wy[i] : 0.f;
        const float v1 = q.u1k[i] + d1, v2 = q.u2k[i] + d2;
        const float p11l = i == 0 ? q.l11 : q.p11[i - 1], p21l = i == 0 ? q.l21 : q.p21[i - 1];
        // divergence: backward differences with upstream's first-row / first-column forms
        float div1, div2;
        if (!ytop) {
            div1 = (q.p11[i] - p11l) + (q.p12[i] - q.p12u[i]); div2 = (q.p21[i] - p21l) + (q.p22[i] - q.p22u[i]);
            if (i == 0) {
                const float b1 = (q.p11[i] + q.p12[i]) - q.p12u[i], b2 = (q.p21[i] + q.p22[i]) - q.p22u[i];
                div1 = x0 ? b1 : div1; div2 = x0 ? b2 : div2;
            }
        } else {
            div1 = (q.p11[i] - p11l) + q.p12[i]; div2 = (q.p21[i] - p21l) + q.p22[i];
            if (i == 0) {
                const float b1 = q.p11[i] + q.p12[i], b2 = q.p21[i] + q.p22[i];
                div1 = x0 ? b1 : div1; div2 = x0 ? b2 : div2;
            }
        }
        u1n[i] = v1 + theta * div1;
        u2n[i] = v2 + theta * div2;
    }
}

__device__ __forceinline__ u64 tv_err_q(float u1n, float u1k, float u2n, float u2k)
{
    const float e1 = u1n - u1k, e2 = u2n - u2k;
    const float t = e1 * e1 + e2 * e2;
    return rint_u64(fminf(t, ERR_CAP_F) * ERR_SCALE_F);
}

// estimateDualVariables for the 4 pixels of a quad, both flow components.  (A hand-rolled division sharing the
// reciprocal between the two quotients by the same 1 + taut*|grad u| was bit-exact but 15 % SLOWER: the wave vote that
// guards its operand range splits the basic block and stops the 16 divisions from interleaving.)
__device__ __forceinline__ void tv_p_quad(float taut, const float* u1x, const float* u1y, const float* u2x, const float* u2y,
                                          const float* p11, const float* p12, const float* p21, const float* p22,
                                          float* o11, float* o12, float* o21, float* o22)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float ng1 = 1.0f + taut * hypot_exact(u1x[i], u1y[i]);
        const float ng2 = 1.0f + taut * hypot_exact(u2x[i], u2y[i]);
        o11[i] = (p11[i] + taut * u1x[i]) / ng1; o12[i] = (p12[i] + taut * u1y[i]) / ng1;
        o21[i] = (p21[i] + taut * u2x[i]) / ng2; o22[i] = (p22[i] + taut * u2y[i]) / ng2;
    }
}

// ---- packed forms used by k_iter2_rows ------------------------------------------------------------------------
// The iteration kernel is VALU-bound (non-packed fp32 issues one wave64 instruction per 4 cycles), so the pointwise part
// of the quad helpers is written on float2 values spanning two NEIGHBOURING PIXELS: they sit in adjacent registers of the
// dwordx4 loads, so v_pk_mul/add/fma_f32 apply without shuffles.  Same operations in the same order as the scalar forms
// above, hence the same bits.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// IEEE a/b as hipcc lowers it (v_rcp_f32, one Newton step on the reciprocal, two on the quotient, v_div_fixup_f32), with
// two changes that keep the bits: (1) the refined reciprocal is shared by the quotients that have the same denominator;
// (2) instead of v_div_scale's case analysis, numerator AND denominator are always scaled by 2^64 -- what v_div_scale does
// for a tiny numerator.  The quotient is unchanged, a power of two commutes with every rounding in the chain, and the
// remainders fma(-b, q, a) cannot underflow.  A quotient in the denormal range is rounded once, by the last fma, like
// v_div_fmas does.  Valid for 2^-24 < |b| < 2^60 and |a| < 2^60 -- here b is 1 + taut*|grad u| >= 1 or |grad I|^2 in
// (2^-23, 2^24) and |a| stays far below 2^60.  `rcp2s` returns 1/(b*2^64).
__device__ __forceinline__ f2 rcp2s(f2 bs)
{
    const f2 r = mk2(__builtin_amdgcn_rcpf(bs.x), __builtin_amdgcn_rcpf(bs.y));
    const f2 e = fma2(-bs, r, mk2(1.0f, 1.0f));
    return fma2(e, r, r);
}
__device__ __forceinline__ f2 div2s(f2 a, f2 b, f2 bs, f2 rs)
{
    const f2 as = a * 0x1p64f;
    f2 q = as * rs;
    f2 e = fma2(-bs, q, as);
    q = fma2(e, rs, q);
    e = fma2(-bs, q, as);
    q = fma2(e, rs, q);
    return mk2(__builtin_amdgcn_div_fixupf(q.x, b.x, a.x), __builtin_amdgcn_div_fixupf(q.y, b.y, a.y));
}

// hypot_exact without the x == inf test (a, b are floats: x <= 2^257) and with the x == 0 case folded into a clamp
// (the smallest non-zero x is 2^-298; sqrt(2^-400) converts to 0.0f like sqrt(0))
__device__ __forceinline__ float hypot_exact2(float a, float b)
{
    const double ad = (double)a, bd = (double)b;
    const double x = __builtin_fmax(__builtin_fma(ad, ad, bd * bd), 0x1p-400);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (float)g;
}

// estimateV + divergence + estimateU for two neighbouring pixels; `first` = this is the pair that holds pixel x == 0 of
// the row when x0 is set (only that pixel uses the first-column form of the divergence)
__device__ __forceinline__ void tv_u_pair(float l_t, float theta, f2 u1k, f2 u2k, f2 wx, f2 wy, f2 rc, f2 p11, f2 p12, f2 p21,
                                          f2 p22, f2 p12u, f2 p22u, float l11, float l21, bool ytop, bool x0, f2& u1n, f2& u2n)
{
    const f2 Ix2 = wx * wx, Iy2 = wy * wy;
    const f2 grad = Ix2 + Iy2;
    const f2 rho = rc + (wx * u1k + wy * u2k);
    const f2 lg = l_t * grad;
    const f2 grads = grad * 0x1p64f;
    const f2 fi = div2s(-rho, grad, grads, rcp2s(grads));
    const bool c1x = rho.x < -lg.x, c2x = rho.x > lg.x, c3x = grad.x > FLT_EPSILON;
    const bool c1y = rho.y < -lg.y, c2y = rho.y > lg.y, c3y = grad.y > FLT_EPSILON;
    const f2 k = mk2(c1x ? l_t : (c2x ? -l_t : fi.x), c1y ? l_t : (c2y ? -l_t : fi.y));
    const f2 kd1 = k * wx, kd2 = k * wy;
    const bool anyx = c1x || c2x || c3x, anyy = c1y || c2y || c3y;
    const f2 d1 = mk2(anyx ? kd1.x : 0.f, anyy ? kd1.y : 0.f), d2 = mk2(anyx ? kd2.x : 0.f, anyy ? kd2.y : 0.f);
    const f2 v1 = u1k + d1, v2 = u2k + d2;
    const f2 dx1 = mk2(p11.x - l11, p11.y - p11.x), dx2 = mk2(p21.x - l21, p21.y - p21.x);
    f2 div1, div2_;
    if (!ytop) {
        div1 = dx1 + (p12 - p12u); div2_ = dx2 + (p22 - p22u);
        const float b1 = (p11.x + p12.x) - p12u.x, b2 = (p21.x + p22.x) - p22u.x;
        div1.x = x0 ? b1 : div1.x; div2_.x = x0 ? b2 : div2_.x;
    } else {
        div1 = dx1 + p12; div2_ = dx2 + p22;
        const float b1 = p11.x + p12.x, b2 = p21.x + p22.x;
        div1.x = x0 ? b1 : div1.x; div2_.x = x0 ? b2 : div2_.x;
    }
    u1n = v1 + theta * div1;
    u2n = v2 + theta * div2_;
}

__device__ __forceinline__ void tv_u_quad_pk(float l_t, float theta, const QuadU& q, bool ytop, bool x0, float* u1n, float* u2n)
{
    f2 a1, a2, b1, b2;
    tv_u_pair(l_t, theta, mk2(q.u1k[0], q.u1k[1]), mk2(q.u2k[0], q.u2k[1]), mk2(q.wx[0], q.wx[1]), mk2(q.wy[0], q.wy[1]),
              mk2(q.r[0], q.r[1]), mk2(q.p11[0], q.p11[1]), mk2(q.p12[0], q.p12[1]), mk2(q.p21[0], q.p21[1]),
              mk2(q.p22[0], q.p22[1]), mk2(q.p12u[0], q.p12u[1]), mk2(q.p22u[0], q.p22u[1]), q.l11, q.l21, ytop, x0, a1, a2);
    tv_u_pair(l_t, theta, mk2(q.u1k[2], q.u1k[3]), mk2(q.u2k[2], q.u2k[3]), mk2(q.wx[2], q.wx[3]), mk2(q.wy[2], q.wy[3]),
              mk2(q.r[2], q.r[3]), mk2(q.p11[2], q.p11[3]), mk2(q.p12[2], q.p12[3]), mk2(q.p21[2], q.p21[3]),
              mk2(q.p22[2], q.p22[3]), mk2(q.p12u[2], q.p12u[3]), mk2(q.p22u[2], q.p22u[3]), q.p11[1], q.p21[1], ytop, false, b1, b2);
    u1n[0] = a1.x; u1n[1] = a1.y; u1n[2] = b1.x; u1n[3] = b1.y;
    u2n[0] = a2.x; u2n[1] = a2.y; u2n[2] = b2.x; u2n[3] = b2.y;
}

// convergence terms of a quad, added to a double accumulator: every term is an integer below 2^43, so the sum is exact
// while it stays below 2^53 (the caller folds the accumulator into a u64 every 256 steps)
// `keep[i]` is all-ones for a pixel that counts and 0 for one that does not (halo rows, columns >= W).  Masks, not
// selects: a v_cndmask whose VCC was produced by the scalar unit (row predicate AND column predicate) costs ~23 cycles on
// gfx950 against 2.5 for a v_and.
__device__ __forceinline__ unsigned opaque_u(unsigned m) { asm volatile("" : "+v"(m)); return m; }
__device__ __forceinline__ float mask_f(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

__device__ __forceinline__ double tv_err_quad_pk(const float* u1n, const float* u1k, const float* u2n, const float* u2k,
                                                 const unsigned* keep)
{
    const f2 e1a = mk2(u1n[0], u1n[1]) - mk2(u1k[0], u1k[1]), e2a = mk2(u2n[0], u2n[1]) - mk2(u2k[0], u2k[1]);
    const f2 e1b = mk2(u1n[2], u1n[3]) - mk2(u1k[2], u1k[3]), e2b = mk2(u2n[2], u2n[3]) - mk2(u2k[2], u2k[3]);
    const f2 ta = e1a * e1a + e2a * e2a, tb = e1b * e1b + e2b * e2b;
    const float t[4] = {ta.x, ta.y, tb.x, tb.y};
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = __builtin_rintf(fminf(t[i], ERR_CAP_F) * ERR_SCALE_F);
        acc += (double)mask_f(v, keep[i]);
    }
    return acc;
}

__device__ __forceinline__ void tv_p_pair(float taut, f2 u1x, f2 u1y, f2 u2x, f2 u2y, f2 p11, f2 p12, f2 p21, f2 p22,
                                          f2& o11, f2& o12, f2& o21, f2& o22)
{
    const f2 g1 = mk2(hypot_exact2(u1x.x, u1y.x), hypot_exact2(u1x.y, u1y.y));
    const f2 g2 = mk2(hypot_exact2(u2x.x, u2y.x), hypot_exact2(u2x.y, u2y.y));
    const f2 ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
    const f2 ns1 = ng1 * 0x1p64f, ns2 = ng2 * 0x1p64f;
    const f2 r1 = rcp2s(ns1), r2 = rcp2s(ns2);
    o11 = div2s(p11 + taut * u1x, ng1, ns1, r1); o12 = div2s(p12 + taut * u1y, ng1, ns1, r1);
    o21 = div2s(p21 + taut * u2x, ng2, ns2, r2); o22 = div2s(p22 + taut * u2y, ng2, ns2, r2);
}

__device__ __forceinline__ void tv_p_quad_pk(float taut, const float* u1x, const float* u1y, const float* u2x, const float* u2y,
                                             const float* p11, const float* p12, const float* p21, const float* p22,
                                             float* o11, float* o12, float* o21, float* o22)
{
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        f2 a, b, c, d;
        tv_p_pair(taut, mk2(u1x[h], u1x[h + 1]), mk2(u1y[h], u1y[h + 1]), mk2(u2x[h], u2x[h + 1]), mk2(u2y[h], u2y[h + 1]),
                  mk2(p11[h], p11[h + 1]), mk2(p12[h], p12[h + 1]), mk2(p21[h], p21[h + 1]), mk2(p22[h], p22[h + 1]), a, b, c, d);
        o11[h] = a.x; o11[h + 1] = a.y; o12[h] = b.x; o12[h + 1] = b.y;
        o21[h] = c.x; o21[h + 1] = c.y; o22[h] = d.x; o22[h + 1] = d.y;
    }
}

__global__ __launch_bounds__(256) void k_iter(IterArgs a)
{
    __shared__ __attribute__((aligned(16))) float su1[IT_TH][IT_TW + 4];
    __shared__ __attribute__((aligned(16))) float su2[IT_TH][IT_TW + 4];
    __shared__ u64 sred[4];
    publish_active_count(a);
    const int b = blockIdx.z;
    u64* errb = a.err + (size_t)b * a.errstride;
    if (!pair_active(errb, a.it, a.thr_q)) return;   // block-uniform
    const PairCtl c = a.ctl[b];
    const int uc = (c.ubase ^ a.utog) & 1, pc = (c.pbase ^ a.ptog) & 1;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    const int x = blockIdx.x * IT_OW + tx * 4, y = blockIdx.y * IT_OH + ty;
    const bool inr = x < W && y < H;
    const size_t po = (size_t)b * a.g.splane;
    const size_t row = po + (size_t)y * pitch + x;

    float u1n[4], u2n[4], p11c[4], p12c[4], p21c[4], p22c[4];
    u64 q = 0;
    const bool outr = inr && tx < 15 && ty < 15;
    if (inr) {
        const float4 u1q = ld4(a.sb.u1[uc] + row), u2q = ld4(a.sb.u2[uc] + row);
        const float4 wxq = ld4(a.wx + row), wyq = ld4(a.wy + row), rq = ld4(a.rho + row);
        float4 a11 = make_float4(0, 0, 0, 0), a12 = a11, a21 = a11, a22 = a11, up12 = a11, up22 = a11;
        float l11 = 0.f, l21 = 0.f;
        if (!a.pzero) {
            a11 = ld4(a.sb.p11[pc] + row); a12 = ld4(a.sb.p12[pc] + row);
            a21 = ld4(a.sb.p21[pc] + row); a22 = ld4(a.sb.p22[pc] + row);
            if (y > 0) { up12 = ld4(a.sb.p12[pc] + row - pitch); up22 = ld4(a.sb.p22[pc] + row - pitch); }
            if (x > 0) { l11 = a.sb.p11[pc][row - 1]; l21 = a.sb.p21[pc][row - 1]; }
        }
        QuadU qu;
        UNPACK4(qu.u1k, u1q) UNPACK4(qu.u2k, u2q) UNPACK4(qu.wx, wxq) UNPACK4(qu.wy, wyq) UNPACK4(qu.r, rq)
        UNPACK4(qu.p11, a11) UNPACK4(qu.p12, a12) UNPACK4(qu.p21, a21) UNPACK4(qu.p22, a22)
        UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
        qu.l11 = l11; qu.l21 = l21;
        UNPACK4(p11c, a11) UNPACK4(p12c, a12) UNPACK4(p21c, a21) UNPACK4(p22c, a22)
        tv_u_quad(a.l_t, a.theta, qu, y == 0, x == 0, u1n, u2n);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (outr && x + i < W) q += tv_err_q(u1n[i], qu.u1k[i], u2n[i], qu.u2k[i]);
        st4(&su1[ty][tx * 4], make_float4(u1n[0], u1n[1], u1n[2], u1n[3]));
        st4(&su2[ty][tx * 4], make_float4(u2n[0], u2n[1], u2n[2], u2n[3]));
    }
    __syncthreads();
    if (outr) {
        float o11[4], o12[4], o21[4], o22[4], u1x[4], u1y[4], u2x[4], u2y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xi = x + i;
            // forwardGradient of u' (0 in the last column / row)
            const float r1 = i < 3 ? u1n[i + 1] : su1[ty][tx * 4 + 4];
            const float r2 = i < 3 ? u2n[i + 1] : su2[ty][tx * 4 + 4];
            u1x[i] = xi < W - 1 ? r1 - u1n[i] : 0.f;
            u2x[i] = xi < W - 1 ? r2 - u2n[i] : 0.f;
            u1y[i] = y < H - 1 ? su1[ty + 1][tx * 4 + i] - u1n[i] : 0.f;
            u2y[i] = y < H - 1 ? su2[ty + 1][tx * 4 + i] - u2n[i] : 0.f;
        }
        tv_p_quad(a.taut, u1x, u1y, u2x, u2y, p11c, p12c, p21c, p22c, o11, o12, o21, o22);
        st4(a.sb.u1[uc ^ 1] + row, make_float4(u1n[0], u1n[1], u1n[2], u1n[3]));
        st4(a.sb.u2[uc ^ 1] + row, make_float4(u2n[0], u2n[1], u2n[2], u2n[3]));
        st4(a.sb.p11[pc ^ 1] + row, make_float4(o11[0], o11[1], o11[2], o11[3]));
        st4(a.sb.p12[pc ^ 1] + row, make_float4(o12[0], o12[1], o12[2], o12[3]));
        st4(a.sb.p21[pc ^ 1] + row, make_float4(o21[0], o21[1], o21[2], o21[3]));
        st4(a.sb.p22[pc ^ 1] + row, make_float4(o22[0], o22[1], o22[2], o22[3]));
    }
    // exact convergence sum: wave shuffle reduction -> 4 partials in LDS -> one atomic per block
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 tot = sred[0] + sred[1] + sred[2] + sred[3];
        atomicAdd(&errb[a.it], tot);
    }
}

// ---------------------------------------------------------------------------------------------
// tvl1_iter, row-strip form (single-iteration variant; W <= 2048): same arithmetic as k_iter,
// different traffic shape.  A block owns R full-width rows of one pair and marches down them RY rows
// per step (thread = one float4 quad of one row).  Full-width rows mean every 128-B line of every plane
// is fetched exactly once per launch (no x halo; the 64x16 tiles of k_iter start at 240-B offsets and
// fetch ~1.8x the algorithmic bytes at the fabric), and the only re-computation is one halo row per
// strip for the forward difference in y.  Neighbour exchange (p12/p22 of the row above, p11/p21 of the
// quad to the left, u' of the quad to the right and of the row below) goes through LDS; the dual
// update of a row is deferred by one step until the u' row below it exists.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_iter_rows(IterArgs a, int R, int QX, int RY)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LW = QX * 4 + 4;
    u64* sred = reinterpret_cast<u64*>(smem);      // 8 x u64 (one per wave, blocks have up to 512 threads)
    float* su1 = smem + 32;                        // [2][RY][LW]   u1' rows of this / the previous step
    float* su2 = su1 + 2 * RY * LW;
    float* sp12 = su2 + 2 * RY * LW;               // [2][RY][LW]   old p12 rows (the row below reads them)
    float* sp22 = sp12 + 2 * RY * LW;
    float* sp11w = sp22 + 2 * RY * LW;             // [RY][QX]      last element of each quad of old p11
    float* sp21w = sp11w + RY * QX;

    publish_active_count(a);
    const int b = blockIdx.z;
    u64* errb = a.err + (size_t)b * a.errstride;
    if (!pair_active(errb, a.it, a.thr_q)) return;   // block-uniform
    const PairCtl c = a.ctl[b];
    const int uc = (c.ubase ^ a.utog) & 1, pc = (c.pbase ^ a.ptog) & 1;
    const int tid = threadIdx.x;
    const int ty = tid / QX, tx = tid - ty * QX;
    const bool lane_on = ty < RY;
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    const int x = tx * 4;
    const int y0 = blockIdx.x * R;
    const int nsteps = R / RY;
    const size_t po = (size_t)b * a.g.splane;

    const float* __restrict__ gu1 = a.sb.u1[uc] + po;
    const float* __restrict__ gu2 = a.sb.u2[uc] + po;
    const float* __restrict__ g11 = a.sb.p11[pc] + po;
    const float* __restrict__ g12 = a.sb.p12[pc] + po;
    const float* __restrict__ g21 = a.sb.p21[pc] + po;
    const float* __restrict__ g22 = a.sb.p22[pc] + po;
    const float* __restrict__ gwx = a.wx + po;
    const float* __restrict__ gwy = a.wy + po;
    const float* __restrict__ grh = a.rho + po;

    // state of the previous step's row, waiting for the u' row below it
    float pu1[4] = {0, 0, 0, 0}, pu2[4] = {0, 0, 0, 0}, q11[4], q12[4], q21[4], q22[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q11[i] = q12[i] = q21[i] = q22[i] = 0.f;
    bool prev_out = false;
    int prev_y = 0;
    u64 q = 0;

    for (int s = 0; s <= nsteps; ++s) {
        const int cur = s & 1;
        const int y = y0 + s * RY + ty;
        const bool valid = lane_on && y < H && (s < nsteps || ty == 0);   // step nsteps = the halo row (u' only)
        const bool is_out = lane_on && s < nsteps && y < H;
        const size_t row = (size_t)y * pitch + x;
        float4 u1q, u2q, wxq, wyq, rq, a11, a12, a21, a22;
        u1q = u2q = wxq = wyq = rq = a11 = a12 = a21 = a22 = make_float4(0, 0, 0, 0);
        if (valid) {
            u1q = ld4(gu1 + row); u2q = ld4(gu2 + row);
            wxq = ld4(gwx + row); wyq = ld4(gwy + row); rq = ld4(grh + row);
            if (!a.pzero) { a11 = ld4(g11 + row); a12 = ld4(g12 + row); a21 = ld4(g21 + row); a22 = ld4(g22 + row); }
            st4(sp12 + (cur * RY + ty) * LW + x, a12);
            st4(sp22 + (cur * RY + ty) * LW + x, a22);
            sp11w[ty * QX + tx] = a11.w;
            sp21w[ty * QX + tx] = a21.w;
        }
        __syncthreads();
        float u1n[4] = {0, 0, 0, 0}, u2n[4] = {0, 0, 0, 0};
        if (valid) {
            float4 up12 = make_float4(0, 0, 0, 0), up22 = up12;
            if (y > 0) {
                if (ty > 0) { up12 = ld4(sp12 + (cur * RY + ty - 1) * LW + x); up22 = ld4(sp22 + (cur * RY + ty - 1) * LW + x); }
                else if (s > 0) { up12 = ld4(sp12 + ((cur ^ 1) * RY + RY - 1) * LW + x); up22 = ld4(sp22 + ((cur ^ 1) * RY + RY - 1) * LW + x); }
                else if (!a.pzero) { up12 = ld4(g12 + row - pitch); up22 = ld4(g22 + row - pitch); }
            }
            float l11 = 0.f, l21 = 0.f;
            if (tx > 0) { l11 = sp11w[ty * QX + tx - 1]; l21 = sp21w[ty * QX + tx - 1]; }
            QuadU qu;
            UNPACK4(qu.u1k, u1q) UNPACK4(qu.u2k, u2q) UNPACK4(qu.wx, wxq) UNPACK4(qu.wy, wyq) UNPACK4(qu.r, rq)
            UNPACK4(qu.p11, a11) UNPACK4(qu.p12, a12) UNPACK4(qu.p21, a21) UNPACK4(qu.p22, a22)
            UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
            qu.l11 = l11; qu.l21 = l21;
            tv_u_quad(a.l_t, a.theta, qu, y == 0, x == 0, u1n, u2n);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (is_out && x + i < W) q += tv_err_q(u1n[i], qu.u1k[i], u2n[i], qu.u2k[i]);
            st4(su1 + (cur * RY + ty) * LW + x, make_float4(u1n[0], u1n[1], u1n[2], u1n[3]));
            st4(su2 + (cur * RY + ty) * LW + x, make_float4(u2n[0], u2n[1], u2n[2], u2n[3]));
            // keep this row's old p for its deferred dual update (after the previous row's update below)
        }
        __syncthreads();
        if (prev_out) {
            // dual update of the previous step's row: forward differences of u' (0 in the last column / row)
            const float* d1 = ty < RY - 1 ? su1 + ((cur ^ 1) * RY + ty + 1) * LW + x : su1 + (cur * RY) * LW + x;
            const float* d2 = ty < RY - 1 ? su2 + ((cur ^ 1) * RY + ty + 1) * LW + x : su2 + (cur * RY) * LW + x;
            const bool lastrow = prev_y >= H - 1;
            float4 dn1 = make_float4(0, 0, 0, 0), dn2 = dn1;
            if (!lastrow) { dn1 = ld4(d1); dn2 = ld4(d2); }
            float r1 = 0.f, r2 = 0.f;
            if (x + 4 < W) { r1 = su1[((cur ^ 1) * RY + ty) * LW + x + 4]; r2 = su2[((cur ^ 1) * RY + ty) * LW + x + 4]; }
            const float dv1[4] = {dn1.x, dn1.y, dn1.z, dn1.w}, dv2[4] = {dn2.x, dn2.y, dn2.z, dn2.w};
            float o11[4], o12[4], o21[4], o22[4], u1x[4], u1y[4], u2x[4], u2y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xi = x + i;
                const float n1 = i < 3 ? pu1[i + 1] : r1, n2 = i < 3 ? pu2[i + 1] : r2;
                u1x[i] = xi < W - 1 ? n1 - pu1[i] : 0.f;
                u2x[i] = xi < W - 1 ? n2 - pu2[i] : 0.f;
                u1y[i] = !lastrow ? dv1[i] - pu1[i] : 0.f;
                u2y[i] = !lastrow ? dv2[i] - pu2[i] : 0.f;
            }
            tv_p_quad(a.taut, u1x, u1y, u2x, u2y, q11, q12, q21, q22, o11, o12, o21, o22);
            const size_t prow = po + (size_t)prev_y * pitch + x;
            st4(a.sb.u1[uc ^ 1] + prow, make_float4(pu1[0], pu1[1], pu1[2], pu1[3]));
            st4(a.sb.u2[uc ^ 1] + prow, make_float4(pu2[0], pu2[1], pu2[2], pu2[3]));
            st4(a.sb.p11[pc ^ 1] + prow, make_float4(o11[0], o11[1], o11[2], o11[3]));
            st4(a.sb.p12[pc ^ 1] + prow, make_float4(o12[0], o12[1], o12[2], o12[3]));
            st4(a.sb.p21[pc ^ 1] + prow, make_float4(o21[0], o21[1], o21[2], o21[3]));
            st4(a.sb.p22[pc ^ 1] + prow, make_float4(o22[0], o22[1], o22[2], o22[3]));
        }
        // rotate: this step's row becomes the pending one
        prev_out = is_out;
        prev_y = y;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pu1[i] = u1n[i]; pu2[i] = u2n[i]; }
        q11[0] = a11.x; q11[1] = a11.y; q11[2] = a11.z; q11[3] = a11.w;
        q12[0] = a12.x; q12[1] = a12.y; q12[2] = a12.z; q12[3] = a12.w;
        q21[0] = a21.x; q21[1] = a21.y; q21[2] = a21.z; q21[3] = a21.w;
        q22[0] = a22.x; q22[1] = a22.y; q22[2] = a22.z; q22[3] = a22.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) {
        u64 tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sred[w];
        atomicAdd(&errb[a.it], tot);
    }
}

// ---------------------------------------------------------------------------------------------
// tvl1_iter x2: TWO inner iterations per launch inside the row march (time skewing).  The strip's rows flow
// through a 3-stage pipeline, one row group per step:
//     stage 1 (group s)   : load row, u1 = U(u0, p0)                         [iteration `it`]
//     stage 2 (group s-1) : p1 = P(p0, u1) ; u2 = U(u1, p1)                   [`it` dual, `it+1` primal]
//     stage 3 (group s-2) : p2 = P(p1, u2) ; store u2, p2                     [`it+1` dual]
// so the 9 input planes are read once and the 6 state planes written once per TWO iterations (30 B/px per
// iteration instead of 60).  Extra work: one halo row above and two below each strip.
//
// Stopping stays exact.  A launch covers iterations (it, it+1); both error sums are accumulated.  If a pair
// met the threshold already at `it`, the launch overshot by one iteration; the state it read is still
// intact in the other ping-pong half, so the NEXT launch re-runs that pair in REPLAY mode: iteration `it`
// alone, from the previous launch's source buffers into its destination buffers (no error accumulation).
//   active_at(j)  = j == 0 || (err[j-2] > thr && err[j-1] > thr)           (j even; slots are zeroed per stage)
//   NORMAL  at it : it < total && active_at(it)
//   REPLAY  at it : it >= 2 && active_at(it-2) && !(err[it-2] > thr)
// ---------------------------------------------------------------------------------------------
#define M_EXIT 0
#define M_NORMAL 1
#define M_REPLAY 2

__device__ __forceinline__ int pair_mode2(const u64* e, int it, int total, double thr)
{
    const bool a1 = it >= 1 ? (double)e[it - 1] > thr : true;
    const bool a2 = it >= 2 ? (double)e[it - 2] > thr : true;
    if (it < total && (it == 0 || (a2 && a1))) return M_NORMAL;
    if (it >= 2 && !a2) {
        const int j = it - 2;
        const bool act = j == 0 || ((double)e[j - 2] > thr && (double)e[j - 1] > thr);
        if (act) return M_REPLAY;
    }
    return M_EXIT;
}

// Stop rule of cv2.cuda.OpticalFlowDual_TVL1 (cudaoptflow tvl1flow.cpp procOneScale, restated in oracle/tvl1_oracle.c
// variant 1): one loop of `total` iterations; the error sum is looked at only on odd iterations n, and only once the
// running prevError (last seen error, minus the threshold for every iteration without a look) has dropped below the
// threshold.  A stop can therefore only follow an odd iteration = the second one of a launch: no REPLAY in this variant.
// Replays the recurrence over iterations [0, it): M_NORMAL if the pair still iterates at launch `it`, else M_EXIT with
// *n_it = iterations executed.
__device__ __forceinline__ int pair_mode_cuda(const u64* e, int it, int total, double thr, int* n_it)
{
    // cudaoptflow procOneScale keeps scaledEpsilon / error / prevError in double ([UPSTREAM-FROM-MEMORY]; a float
    // `prevError -= scaledEpsilon` could flip the iteration at which the sum is next consulted)
    double prev = 0.0;
    for (int n = 0; n < it; ++n) {
        const bool calc = thr > 0.0 && (n & 1) && prev < thr;
        if (calc) {
            const double err = (double)e[n] * 0x1p-30;
            prev = err;
            if (!(err > thr)) { if (n_it) *n_it = n + 1; return M_EXIT; }
        } else prev -= thr;
    }
    if (n_it) *n_it = total;
    return it < total ? M_NORMAL : M_EXIT;
}

__device__ __forceinline__ int pair_mode(const u64* e, int it, int total, double thr_q, int variant, double thr_d)
{
    return variant ? pair_mode_cuda(e, it, total, thr_d, nullptr) : pair_mode2(e, it, total, thr_q);
}

// Work items of a tvl1_iter launch when the strips are sized ON THE DEVICE from the number of pairs that still
// iterate (`n`): one round of at most `slots` resident blocks (slots = CUs x blocks per CU), each marching a strip that is
// as long as that allows -- a lock-step batch loses a third of its time otherwise (a launch with 1024 blocks on 768 slots
// takes two rounds, one with 300 takes as long as one with 768).  Returns rows per strip and the strip count.
TF_HD inline void strip_rule_min(int n, int H, int minrows, int slots, int* R, int* S)
{
    if (n < 1) n = 1;
    if (minrows < 1) minrows = 1;
    const int k = (n + slots - 1) / slots;                 // rounds
    int s = (int)(((long long)k * slots) / n);
    const int smax = H / minrows > 0 ? H / minrows : 1;    // no strip shorter than `minrows` rows (each pays ~3 halo rows)
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    int r = (H + s - 1) / s;
    *R = r;
    *S = (H + r - 1) / r;
}
TF_HD inline void strip_rule(int n, int H, int RY, int slots, int* R, int* S)
{
    strip_rule_min(n, H, 4 * RY, slots, R, S);            // at least 4 steps per strip
}

struct Iter2Args {
    IterArgs a;                           // a.it = first iteration of the launch (even), a.utog/ptog/pzero for it
    int utog_prev, ptog_prev, pzero_prev; // the same three for the previous launch (used by REPLAY blocks)
    int total;                            // inner*outer
};

__device__ __forceinline__ void publish_active_count2(const Iter2Args& A)
{
    const IterArgs& a = A.a;
    if (a.host_slot && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) {
        int c = 0;
        for (int b2 = threadIdx.x; b2 < a.B; b2 += 64)
            c += pair_mode(a.err + (size_t)b2 * a.errstride, a.it, A.total, a.thr_q, a.variant, a.thr_d) != M_EXIT ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (threadIdx.x == 0) __hip_atomic_store(a.host_slot, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// `slots` > 0: grid = (max work items, 1, 1) and every block finds its (pair, strip) among the pairs that still iterate;
// `slots` == 0: grid = (strips, 1, pairs) with the fixed strip length R.
// What one block of the two-iteration row march needs to know: which pair and strip, the strip length, the block shape
// and the pair's mode.  The lock-step launch derives it from launch-uniform arguments + the pair's error slots
// (k_iter2_rows), the free-running scheduler from the pair's own state (teeflow_sched.hip.h).
struct Iter2Blk {
    int b, strip, R, QX, RY;
    bool replay, pzero;
    int uc, pc;          // ping-pong halves the block READS u / p from (it writes the other ones)
    int it;              // first of the two iterations (error slots it, it+1)
    u64* errb;           // the pair's error slots
    int W, H, pitch;     // geometry of the pair's level
    long long splane;    // floats between consecutive pairs in the state / constant buffers
};

__device__ __forceinline__ void iter2_rows_body(const IterArgs& a, const Iter2Blk& k, float* smem);

__global__ __launch_bounds__(512) void k_iter2_rows(Iter2Args A, int R, int QX, int RY, int slots)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const IterArgs& a = A.a;
    u64* sred = reinterpret_cast<u64*>(smem);      // 2 x 8 x u64 = 128 B (one per wave and error sum)

    publish_active_count2(A);
    int b = blockIdx.z, strip = blockIdx.x;
    if (slots > 0) {
        // 64-pair masks of the pairs that are not in EXIT mode (sred is free until the end of the kernel; B <= 1024)
        const int nchunk = (a.B + 63) >> 6, nw = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
        for (int c = wv; c < nchunk; c += nw) {
            const int pb = c * 64 + ln;
            const bool on = pb < a.B && pair_mode(a.err + (size_t)pb * a.errstride, a.it, A.total, a.thr_q, a.variant, a.thr_d) != M_EXIT;
            const u64 m = __ballot(on);
            if (ln == 0) sred[c] = m;
        }
        __syncthreads();
        int nact = 0;
        for (int c = 0; c < nchunk; ++c) nact += __popcll(sred[c]);
        int S;
        strip_rule(nact, a.g.h, RY, slots, &R, &S);
        const int item = blockIdx.x;
        if (item >= nact * S) return;                      // block-uniform
        int k = item / S;
        strip = item - k * S;
        int c = 0;
        u64 m = sred[0];
        while (k >= __popcll(m)) { k -= __popcll(m); m = sred[++c]; }
        for (; k > 0; --k) m &= m - 1;                      // drop the k lowest set bits
        b = c * 64 + (__ffsll((long long)m) - 1);
        __syncthreads();                                    // sred is reused for the error sums below
    }
    u64* errb = a.err + (size_t)b * a.errstride;
    const int mode = pair_mode(errb, a.it, A.total, a.thr_q, a.variant, a.thr_d);   // block-uniform
    if (mode == M_EXIT) return;
    const bool replay = mode == M_REPLAY;
    const PairCtl c = a.ctl[b];
    const int utog = replay ? A.utog_prev : a.utog, ptog = replay ? A.ptog_prev : a.ptog;
    Iter2Blk blk;
    blk.b = b; blk.strip = strip; blk.R = R; blk.QX = QX; blk.RY = RY; blk.replay = replay;
    blk.pzero = (replay ? A.pzero_prev : a.pzero) != 0;
    blk.uc = (c.ubase ^ utog) & 1; blk.pc = (c.pbase ^ ptog) & 1;
    blk.it = a.it; blk.errb = errb;
    blk.W = a.g.w; blk.H = a.g.h; blk.pitch = a.g.pitch; blk.splane = a.g.splane;
    iter2_rows_body(a, blk, smem);
}

__device__ __forceinline__ void iter2_rows_body(const IterArgs& a, const Iter2Blk& k, float* smem)
{
    const int b = k.b, strip = k.strip, R = k.R, QX = k.QX, RY = k.RY, uc = k.uc, pc = k.pc;
    const bool replay = k.replay, pzero = k.pzero;
    u64* errb = k.errb;
    const int LW = QX * 4 + 4;
    u64* sred = reinterpret_cast<u64*>(smem);      // 2 x 8 x u64 = 128 B (one per wave and error sum)
    float* U1a = smem + 32;                        // [2][RY][LW]  u1 (first iterate) plane 1 / 2
    float* U1b = U1a + 2 * RY * LW;
    float* U2a = U1b + 2 * RY * LW;                // [2][RY][LW]  u2 (second iterate)
    float* U2b = U2a + 2 * RY * LW;
    float* B12 = U2b + 2 * RY * LW;                // [RY+1][LW]   rolling rows of p1_12 / p1_22 (p0's row above is
                                                   //              re-read from global/L2: keeps LDS at 3 blocks per CU)
    float* B22 = B12 + (RY + 1) * LW;
    float* B11w = B22 + (RY + 1) * LW;             // [RY][QX]     last element of each quad of p1_11 / p1_21
    float* B21w = B11w + RY * QX;
    const int tid = threadIdx.x;
    const int ty = tid / QX, tx = tid - ty * QX;
    const bool lane_on = ty < RY;
    const int W = k.W, H = k.H, pitch = k.pitch;
    const int x = tx * 4;
    const int y0 = strip * R;
    // the first primal update covers rows y0-1 .. y0+R+1; groups of RY rows start at y0-1 (R need not be a multiple of RY)
    const int ngroups = (R + 3 + RY - 1) / RY;
    const size_t po = (size_t)b * (size_t)k.splane;
    const int RB = RY + 1;

    const float* __restrict__ gu1 = a.sb.u1[uc] + po;
    const float* __restrict__ gu2 = a.sb.u2[uc] + po;
    const float* __restrict__ g11 = a.sb.p11[pc] + po;
    const float* __restrict__ g12 = a.sb.p12[pc] + po;
    const float* __restrict__ g21 = a.sb.p21[pc] + po;
    const float* __restrict__ g22 = a.sb.p22[pc] + po;
    const float* __restrict__ gwx = a.wx + po;
    const float* __restrict__ gwy = a.wy + po;
    const float* __restrict__ grh = a.rho + po;
    float* __restrict__ ou1 = a.sb.u1[uc ^ 1] + po;
    float* __restrict__ ou2 = a.sb.u2[uc ^ 1] + po;
    float* __restrict__ o11 = a.sb.p11[pc ^ 1] + po;
    float* __restrict__ o12 = a.sb.p12[pc ^ 1] + po;
    float* __restrict__ o21 = a.sb.p21[pc ^ 1] + po;
    float* __restrict__ o22 = a.sb.p22[pc ^ 1] + po;

    // row predicates (absolute row index)
    const int yu1_lo = y0 - 1, yu1_hi = y0 + R + 1, yp1_hi = y0 + R, yout_hi = y0 + R - 1;

    // pipeline registers.  Written only under the predicate (s1_valid / s2_valid) they are later read under, so they
    // need no initial value and the predicated-off lanes need no zero fill.
    float s1_u1[4], s1_u2[4], s1_wx[4], s1_wy[4], s1_r[4], s1_11[4], s1_12[4], s1_21[4], s1_22[4];   // stage1 -> stage2
    float s2_u1[4], s2_u2[4], s2_11[4], s2_12[4], s2_21[4], s2_22[4];                                  // stage2 -> stage3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1_u1[i] = s1_u2[i] = s1_wx[i] = s1_wy[i] = s1_r[i] = s1_11[i] = s1_12[i] = s1_21[i] = s1_22[i] = 0.f;
        s2_u1[i] = s2_u2[i] = s2_11[i] = s2_12[i] = s2_21[i] = s2_22[i] = 0.f;
    }
    bool s1_valid = false, s2_valid = false;
    // column masks: inw[j] = all-ones iff column x + j lies inside the image (j = 0..4; x itself always does)
    unsigned inw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) inw[j] = opaque_u(x + j < W ? ~0u : 0u);
    int ring = ty + 1;                                    // (r2 mod RB) for r2 = (s-1)*RY + ty, kept incrementally
    u64 qA = 0, qB = 0;
    double accA = 0.0, accB = 0.0;

    for (int s = 0; s < ngroups + 2; ++s) {
        // ================= stage 1: group s, iteration `it` primal =================
        const int r1 = s * RY + ty;                       // linear row counter inside the strip's pipeline
        const int y = y0 - 1 + r1;
        const bool v1 = lane_on && s < ngroups && y >= 0 && y < H && y >= yu1_lo && y <= yu1_hi;
        const size_t row = (size_t)y * pitch + x;
        float4 u1q, u2q, wxq, wyq, rq, a11, a12, a21, a22;
        u1q = u2q = wxq = wyq = rq = a11 = a12 = a21 = a22 = make_float4(0, 0, 0, 0);
        if (v1) {
            u1q = ld4(gu1 + row); u2q = ld4(gu2 + row);
            wxq = ld4(gwx + row); wyq = ld4(gwy + row); rq = ld4(grh + row);
            if (!pzero) { a11 = ld4(g11 + row); a12 = ld4(g12 + row); a21 = ld4(g21 + row); a22 = ld4(g22 + row); }
        }
        float n_u1[4] = {0, 0, 0, 0}, n_u2[4] = {0, 0, 0, 0};
        float c11[4], c12[4], c21[4], c22[4], wxv[4], wyv[4], rv[4];
        UNPACK4(c11, a11) UNPACK4(c12, a12) UNPACK4(c21, a21) UNPACK4(c22, a22) UNPACK4(wxv, wxq) UNPACK4(wyv, wyq) UNPACK4(rv, rq)
        if (v1) {
            float4 up12 = make_float4(0, 0, 0, 0), up22 = up12;
            if (y > 0 && !pzero) { up12 = ld4(g12 + row - pitch); up22 = ld4(g22 + row - pitch); }
            // dual variable of the pixel left of the quad: one dword each straight from global memory (the line is in
            // L1/L2, the neighbouring lane loads it as part of its float4) -- no LDS exchange, no barrier
            float l11 = 0.f, l21 = 0.f;
            if (tx > 0 && !pzero) { l11 = g11[row - 1]; l21 = g21[row - 1]; }
            QuadU qu;
            UNPACK4(qu.u1k, u1q) UNPACK4(qu.u2k, u2q) UNPACK4(qu.wx, wxq) UNPACK4(qu.wy, wyq) UNPACK4(qu.r, rq)
            UNPACK4(qu.p11, a11) UNPACK4(qu.p12, a12) UNPACK4(qu.p21, a21) UNPACK4(qu.p22, a22)
            UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
            qu.l11 = l11; qu.l21 = l21;
            const bool isout = y >= y0 && y <= yout_hi;
            tv_u_quad_pk(a.l_t, a.theta, qu, y == 0, x == 0, n_u1, n_u2);
            const unsigned mrow = opaque_u(!replay && isout ? ~0u : 0u);
            const unsigned keep[4] = {inw[0] & mrow, inw[1] & mrow, inw[2] & mrow, inw[3] & mrow};
            accA += tv_err_quad_pk(n_u1, qu.u1k, n_u2, qu.u2k, keep);
            st4(U1a + ((s & 1) * RY + ty) * LW + x, PACK4(n_u1));
            st4(U1b + ((s & 1) * RY + ty) * LW + x, PACK4(n_u2));
        }
        __syncthreads();
        // ================= stage 2: group s-1: iteration `it` dual, then `it+1` primal =================
        const int yb = y - RY;
        const bool v2 = s1_valid && yb <= yp1_hi;           // s1_valid already implies in-image and >= yu1_lo
        float p1_11[4] = {0, 0, 0, 0}, p1_12[4] = {0, 0, 0, 0}, p1_21[4] = {0, 0, 0, 0}, p1_22[4] = {0, 0, 0, 0};
        if (v2) {
            const int bp = (s - 1) & 1;
            const bool lastrow = yb >= H - 1;
            float4 dn1 = make_float4(0, 0, 0, 0), dn2 = dn1;
            if (!lastrow) {
                const float* d1 = ty < RY - 1 ? U1a + (bp * RY + ty + 1) * LW + x : U1a + ((s & 1) * RY) * LW + x;
                const float* d2 = ty < RY - 1 ? U1b + (bp * RY + ty + 1) * LW + x : U1b + ((s & 1) * RY) * LW + x;
                dn1 = ld4(d1); dn2 = ld4(d2);
            }
            float rr1 = 0.f, rr2 = 0.f;
            if (x + 4 < W) { rr1 = U1a[(bp * RY + ty) * LW + x + 4]; rr2 = U1b[(bp * RY + ty) * LW + x + 4]; }
            const unsigned mnl = opaque_u(lastrow ? 0u : ~0u);
            float dv1[4], dv2[4], u1x[4], u1y[4], u2x[4], u2y[4];
            UNPACK4(dv1, dn1) UNPACK4(dv2, dn2)
            {   // the thread's own quad of the first iterate lives in LDS since stage 1 of the previous step
                const float4 o1 = ld4(U1a + (bp * RY + ty) * LW + x), o2 = ld4(U1b + (bp * RY + ty) * LW + x);
                UNPACK4(s1_u1, o1) UNPACK4(s1_u2, o2)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e1 = i < 3 ? s1_u1[i + 1] : rr1, e2 = i < 3 ? s1_u2[i + 1] : rr2;
                u1x[i] = mask_f(e1 - s1_u1[i], inw[i + 1]);       // 0 in the last column (x + i == W - 1) and beyond
                u2x[i] = mask_f(e2 - s1_u2[i], inw[i + 1]);
                u1y[i] = mask_f(dv1[i] - s1_u1[i], mnl);          // 0 in the last row
                u2y[i] = mask_f(dv2[i] - s1_u2[i], mnl);
            }
            tv_p_quad_pk(a.taut, u1x, u1y, u2x, u2y, s1_11, s1_12, s1_21, s1_22, p1_11, p1_12, p1_21, p1_22);
            if (replay) {
                if (yb >= y0 && yb <= yout_hi) {
                    const size_t prow = (size_t)yb * pitch + x;
                    st4(ou1 + prow, PACK4(s1_u1)); st4(ou2 + prow, PACK4(s1_u2));
                    st4(o11 + prow, PACK4(p1_11)); st4(o12 + prow, PACK4(p1_12));
                    st4(o21 + prow, PACK4(p1_21)); st4(o22 + prow, PACK4(p1_22));
                }
            } else {
                st4(B12 + ring * LW + x, PACK4(p1_12));
                st4(B22 + ring * LW + x, PACK4(p1_22));
                B11w[ty * QX + tx] = p1_11[3];
                B21w[ty * QX + tx] = p1_21[3];
            }
        }
        bool v2u = false;
        float m_u1[4] = {0, 0, 0, 0}, m_u2[4] = {0, 0, 0, 0};
        if (!replay) {
            __syncthreads();
            v2u = v2 && yb >= y0;                           // rows y0 .. y0+R get the second primal update
            if (v2u) {
                float4 up12 = make_float4(0, 0, 0, 0), up22 = up12;
                if (yb > 0) {
                    const int ri = ring > 0 ? ring - 1 : RB - 1;        // (r2 - 1) mod RB
                    up12 = ld4(B12 + ri * LW + x); up22 = ld4(B22 + ri * LW + x);
                }
                float l11 = 0.f, l21 = 0.f;
                if (tx > 0) { l11 = B11w[ty * QX + tx - 1]; l21 = B21w[ty * QX + tx - 1]; }
                QuadU qu;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qu.u1k[i] = s1_u1[i]; qu.u2k[i] = s1_u2[i]; qu.wx[i] = s1_wx[i]; qu.wy[i] = s1_wy[i]; qu.r[i] = s1_r[i];
                    qu.p11[i] = p1_11[i]; qu.p12[i] = p1_12[i]; qu.p21[i] = p1_21[i]; qu.p22[i] = p1_22[i];
                }
                UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
                qu.l11 = l11; qu.l21 = l21;
                const bool isout = yb <= yout_hi;
                tv_u_quad_pk(a.l_t, a.theta, qu, yb == 0, x == 0, m_u1, m_u2);
                const unsigned mrow = opaque_u(isout ? ~0u : 0u);
                const unsigned keep[4] = {inw[0] & mrow, inw[1] & mrow, inw[2] & mrow, inw[3] & mrow};
                accB += tv_err_quad_pk(m_u1, s1_u1, m_u2, s1_u2, keep);
                st4(U2a + (((s - 1) & 1) * RY + ty) * LW + x, PACK4(m_u1));
                st4(U2b + (((s - 1) & 1) * RY + ty) * LW + x, PACK4(m_u2));
            }
            __syncthreads();
            // ================= stage 3: group s-2: iteration `it+1` dual, store =================
            const int yc = y - 2 * RY;
            if (s2_valid && yc <= yout_hi) {
                const int bq = s & 1;                       // (s-2)&1
                const bool lastrow = yc >= H - 1;
                float4 dn1 = make_float4(0, 0, 0, 0), dn2 = dn1;
                if (!lastrow) {
                    const float* d1 = ty < RY - 1 ? U2a + (bq * RY + ty + 1) * LW + x : U2a + ((bq ^ 1) * RY) * LW + x;
                    const float* d2 = ty < RY - 1 ? U2b + (bq * RY + ty + 1) * LW + x : U2b + ((bq ^ 1) * RY) * LW + x;
                    dn1 = ld4(d1); dn2 = ld4(d2);
                }
                float rr1 = 0.f, rr2 = 0.f;
                if (x + 4 < W) { rr1 = U2a[(bq * RY + ty) * LW + x + 4]; rr2 = U2b[(bq * RY + ty) * LW + x + 4]; }
                const unsigned mnl = opaque_u(lastrow ? 0u : ~0u);
                float dv1[4], dv2[4], r11[4], r12[4], r21[4], r22[4], u1x[4], u1y[4], u2x[4], u2y[4];
                UNPACK4(dv1, dn1) UNPACK4(dv2, dn2)
                {   // own quad of the second iterate: in LDS since stage 2 of the previous step
                    const float4 o1 = ld4(U2a + (bq * RY + ty) * LW + x), o2 = ld4(U2b + (bq * RY + ty) * LW + x);
                    UNPACK4(s2_u1, o1) UNPACK4(s2_u2, o2)
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e1 = i < 3 ? s2_u1[i + 1] : rr1, e2 = i < 3 ? s2_u2[i + 1] : rr2;
                    u1x[i] = mask_f(e1 - s2_u1[i], inw[i + 1]);
                    u2x[i] = mask_f(e2 - s2_u2[i], inw[i + 1]);
                    u1y[i] = mask_f(dv1[i] - s2_u1[i], mnl);
                    u2y[i] = mask_f(dv2[i] - s2_u2[i], mnl);
                }
                tv_p_quad_pk(a.taut, u1x, u1y, u2x, u2y, s2_11, s2_12, s2_21, s2_22, r11, r12, r21, r22);
                const size_t prow = (size_t)yc * pitch + x;
                st4(ou1 + prow, PACK4(s2_u1)); st4(ou2 + prow, PACK4(s2_u2));
                st4(o11 + prow, PACK4(r11)); st4(o12 + prow, PACK4(r12));
                st4(o21 + prow, PACK4(r21)); st4(o22 + prow, PACK4(r22));
            }
        }
        else __syncthreads();       // REPLAY blocks: keep the next step's stage-1 LDS writes behind this step's stage-2 reads
        // ================= rotate the pipeline registers =================
        s2_valid = v2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s2_11[i] = p1_11[i]; s2_12[i] = p1_12[i]; s2_21[i] = p1_21[i]; s2_22[i] = p1_22[i];
            s1_wx[i] = wxv[i]; s1_wy[i] = wyv[i]; s1_r[i] = rv[i];
            s1_11[i] = c11[i]; s1_12[i] = c12[i]; s1_21[i] = c21[i]; s1_22[i] = c22[i];
        }
        s1_valid = v1;
        ring += RY; ring = ring >= RB ? ring - RB : ring;
        if ((s & 255) == 255) { qA += (u64)accA; qB += (u64)accB; accA = accB = 0.0; }   // keep the double sums exact
    }
    if (!replay) {
        qA += (u64)accA; qB += (u64)accB;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { qA += __shfl_down(qA, off, 64); qB += __shfl_down(qB, off, 64); }
        __syncthreads();
        if ((tid & 63) == 0) { sred[tid >> 6] = qA; sred[8 + (tid >> 6)] = qB; }
        __syncthreads();
        if (tid == 0) {
            u64 ta = 0, tb = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ta += sred[w]; tb += sred[8 + w]; }
            atomicAdd(&errb[k.it], ta);
            atomicAdd(&errb[k.it + 1], tb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tvl1_iter, two iterations per launch on TILES: the form for launches too small for the row strips (a single pair, a
// handful of pairs) and for levels wider than 2048 px.  Same schedule, modes (NORMAL / REPLAY / EXIT) and arithmetic as
// k_iter2_rows; a block stages a 64 x 16 region (16 quads x 16 rows, region origin 4 px left of and 1 row above its
// outputs) and produces the 52 x 13 outputs whose dependency cone (1 px left / 1 row up for the primal updates, 1 px
// right / 1 row down for the dual updates, twice) stays inside the region:
//     U1 on quads 0..15, rows 0..15 | P1 on quads 0..14, rows 0..14 | U2 on quads 1..14, rows 1..14 | P2, store: quads 1..13, rows 1..13
// Halving the number of dependent launches is what matters here: a single 512 x 512 pair spends its 4 ms in ~390 launches.
// ---------------------------------------------------------------------------------------------
#define T2_OW 52
#define T2_OH 13

__global__ __launch_bounds__(256) void k_iter2_tile(Iter2Args A)
{
    constexpr int LW = 68;
    __shared__ __attribute__((aligned(16))) float sA[16][LW], sB[16][LW];      // u1 / u2 of the first, then of the second iterate
    __shared__ __attribute__((aligned(16))) float s12[16][LW], s22[16][LW];    // p1_12 / p1_22
    __shared__ float s11w[16][16], s21w[16][16];                               // last element of each quad of p1_11 / p1_21
    __shared__ u64 sred[8];
    const IterArgs& a = A.a;
    publish_active_count2(A);
    const int b = blockIdx.z;
    u64* errb = a.err + (size_t)b * a.errstride;
    const int mode = pair_mode(errb, a.it, A.total, a.thr_q, a.variant, a.thr_d);   // block-uniform
    if (mode == M_EXIT) return;
    const bool replay = mode == M_REPLAY;
    const PairCtl c = a.ctl[b];
    const int utog = replay ? A.utog_prev : a.utog, ptog = replay ? A.ptog_prev : a.ptog;
    const bool pzero = (replay ? A.pzero_prev : a.pzero) != 0;
    const int uc = (c.ubase ^ utog) & 1, pc = (c.pbase ^ ptog) & 1;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    const int x = (int)blockIdx.x * T2_OW - 4 + tx * 4, y = (int)blockIdx.y * T2_OH - 1 + ty;
    const bool inr = x >= 0 && x < W && y >= 0 && y < H;
    const bool outq = inr && tx >= 1 && tx <= 13 && ty >= 1 && ty <= 13;        // this thread's quad is an output of the block
    const size_t po = (size_t)b * a.g.splane;
    const size_t row = po + (size_t)(inr ? y : 0) * pitch + (inr ? x : 0);

    unsigned inw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) inw[j] = opaque_u(x + j < W ? ~0u : 0u);
    const unsigned mnl = opaque_u(y >= H - 1 ? 0u : ~0u);
    const unsigned mout = opaque_u(outq ? ~0u : 0u);
    const unsigned keep[4] = {inw[0] & mout, inw[1] & mout, inw[2] & mout, inw[3] & mout};

    float u0_1[4], u0_2[4], wx[4], wy[4], rc[4], p0_11[4], p0_12[4], p0_21[4], p0_22[4];
    float u1_1[4], u1_2[4];
    double accA = 0.0, accB = 0.0;
    // ---- first primal update on the whole region ----
    if (inr) {
        const float4 u1q = ld4(a.sb.u1[uc] + row), u2q = ld4(a.sb.u2[uc] + row);
        const float4 wxq = ld4(a.wx + row), wyq = ld4(a.wy + row), rq = ld4(a.rho + row);
        float4 a11 = make_float4(0, 0, 0, 0), a12 = a11, a21 = a11, a22 = a11, up12 = a11, up22 = a11;
        float l11 = 0.f, l21 = 0.f;
        if (!pzero) {
            a11 = ld4(a.sb.p11[pc] + row); a12 = ld4(a.sb.p12[pc] + row);
            a21 = ld4(a.sb.p21[pc] + row); a22 = ld4(a.sb.p22[pc] + row);
            if (y > 0) { up12 = ld4(a.sb.p12[pc] + row - pitch); up22 = ld4(a.sb.p22[pc] + row - pitch); }
            if (x > 0) { l11 = a.sb.p11[pc][row - 1]; l21 = a.sb.p21[pc][row - 1]; }
        }
        QuadU qu;
        UNPACK4(qu.u1k, u1q) UNPACK4(qu.u2k, u2q) UNPACK4(qu.wx, wxq) UNPACK4(qu.wy, wyq) UNPACK4(qu.r, rq)
        UNPACK4(qu.p11, a11) UNPACK4(qu.p12, a12) UNPACK4(qu.p21, a21) UNPACK4(qu.p22, a22)
        UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
        qu.l11 = l11; qu.l21 = l21;
        UNPACK4(u0_1, u1q) UNPACK4(u0_2, u2q) UNPACK4(wx, wxq) UNPACK4(wy, wyq) UNPACK4(rc, rq)
        UNPACK4(p0_11, a11) UNPACK4(p0_12, a12) UNPACK4(p0_21, a21) UNPACK4(p0_22, a22)
        tv_u_quad_pk(a.l_t, a.theta, qu, y == 0, x == 0, u1_1, u1_2);
        if (!replay) accA += tv_err_quad_pk(u1_1, u0_1, u1_2, u0_2, keep);
        st4(&sA[ty][tx * 4], PACK4(u1_1));
        st4(&sB[ty][tx * 4], PACK4(u1_2));
    }
    __syncthreads();
    // ---- first dual update: quads 0..14, rows 0..14 (the right / lower neighbour of the first iterate is in the region) ----
    float p1_11[4], p1_12[4], p1_21[4], p1_22[4];
    const bool vP1 = inr && tx <= 14 && ty <= 14;
    if (vP1) {
        const float4 dn1 = ld4(&sA[ty + 1][tx * 4]), dn2 = ld4(&sB[ty + 1][tx * 4]);     // masked where y is the last image row
        const float rr1 = sA[ty][tx * 4 + 4], rr2 = sB[ty][tx * 4 + 4];                 // masked where the column is the last one
        float dv1[4], dv2[4], u1x[4], u1y[4], u2x[4], u2y[4];
        UNPACK4(dv1, dn1) UNPACK4(dv2, dn2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u1_1[i + 1] : rr1, e2 = i < 3 ? u1_2[i + 1] : rr2;
            u1x[i] = mask_f(e1 - u1_1[i], inw[i + 1]); u2x[i] = mask_f(e2 - u1_2[i], inw[i + 1]);
            u1y[i] = mask_f(dv1[i] - u1_1[i], mnl); u2y[i] = mask_f(dv2[i] - u1_2[i], mnl);
        }
        tv_p_quad_pk(a.taut, u1x, u1y, u2x, u2y, p0_11, p0_12, p0_21, p0_22, p1_11, p1_12, p1_21, p1_22);
        if (replay) {
            if (outq) {
                st4(a.sb.u1[uc ^ 1] + row, PACK4(u1_1)); st4(a.sb.u2[uc ^ 1] + row, PACK4(u1_2));
                st4(a.sb.p11[pc ^ 1] + row, PACK4(p1_11)); st4(a.sb.p12[pc ^ 1] + row, PACK4(p1_12));
                st4(a.sb.p21[pc ^ 1] + row, PACK4(p1_21)); st4(a.sb.p22[pc ^ 1] + row, PACK4(p1_22));
            }
        } else {
            st4(&s12[ty][tx * 4], PACK4(p1_12));
            st4(&s22[ty][tx * 4], PACK4(p1_22));
            s11w[ty][tx] = p1_11[3];
            s21w[ty][tx] = p1_21[3];
        }
    }
    if (replay) return;                                                          // block-uniform
    __syncthreads();
    // ---- second primal update: quads 1..14, rows 1..14 ----
    float u2_1[4], u2_2[4];
    const bool vU2 = vP1 && tx >= 1 && ty >= 1;
    if (vU2) {
        const float4 up12 = ld4(&s12[ty - 1][tx * 4]), up22 = ld4(&s22[ty - 1][tx * 4]);  // unused in the first image row
        QuadU qu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qu.u1k[i] = u1_1[i]; qu.u2k[i] = u1_2[i]; qu.wx[i] = wx[i]; qu.wy[i] = wy[i]; qu.r[i] = rc[i];
            qu.p11[i] = p1_11[i]; qu.p12[i] = p1_12[i]; qu.p21[i] = p1_21[i]; qu.p22[i] = p1_22[i];
        }
        UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
        qu.l11 = s11w[ty][tx - 1]; qu.l21 = s21w[ty][tx - 1];                   // unused for the first image column
        tv_u_quad_pk(a.l_t, a.theta, qu, y == 0, x == 0, u2_1, u2_2);
        accB += tv_err_quad_pk(u2_1, u1_1, u2_2, u1_2, keep);
        st4(&sA[ty][tx * 4], PACK4(u2_1));                                      // the first iterate's LDS copy was last read before the barrier above
        st4(&sB[ty][tx * 4], PACK4(u2_2));
    }
    __syncthreads();
    // ---- second dual update and store: quads 1..13, rows 1..13 ----
    if (outq) {
        const float4 dn1 = ld4(&sA[ty + 1][tx * 4]), dn2 = ld4(&sB[ty + 1][tx * 4]);
        const float rr1 = sA[ty][tx * 4 + 4], rr2 = sB[ty][tx * 4 + 4];
        float dv1[4], dv2[4], u1x[4], u1y[4], u2x[4], u2y[4], o11[4], o12[4], o21[4], o22[4];
        UNPACK4(dv1, dn1) UNPACK4(dv2, dn2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u2_1[i + 1] : rr1, e2 = i < 3 ? u2_2[i + 1] : rr2;
            u1x[i] = mask_f(e1 - u2_1[i], inw[i + 1]); u2x[i] = mask_f(e2 - u2_2[i], inw[i + 1]);
            u1y[i] = mask_f(dv1[i] - u2_1[i], mnl); u2y[i] = mask_f(dv2[i] - u2_2[i], mnl);
        }
        tv_p_quad_pk(a.taut, u1x, u1y, u2x, u2y, p1_11, p1_12, p1_21, p1_22, o11, o12, o21, o22);
        st4(a.sb.u1[uc ^ 1] + row, PACK4(u2_1)); st4(a.sb.u2[uc ^ 1] + row, PACK4(u2_2));
        st4(a.sb.p11[pc ^ 1] + row, PACK4(o11)); st4(a.sb.p12[pc ^ 1] + row, PACK4(o12));
        st4(a.sb.p21[pc ^ 1] + row, PACK4(o21)); st4(a.sb.p22[pc ^ 1] + row, PACK4(o22));
    }
    u64 qA = (u64)accA, qB = (u64)accB;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { qA += __shfl_down(qA, off, 64); qB += __shfl_down(qB, off, 64); }
    if ((threadIdx.x & 63) == 0) { sred[threadIdx.x >> 6] = qA; sred[4 + (threadIdx.x >> 6)] = qB; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&errb[a.it], sred[0] + sred[1] + sred[2] + sred[3]);
        atomicAdd(&errb[a.it + 1], sred[4] + sred[5] + sred[6] + sred[7]);
    }
}

// one 64 x 16 output tile of one flow plane: stage the tile + halo (replicate border) in LDS, then the selection network
template <int KS>
__device__ __forceinline__ void median_block(float (*t)[64 + 2 * (KS / 2)], const float* __restrict__ src, float* __restrict__ dst,
                                             int x0, int y0, int W, int H, int pitch)
{
    constexpr int R = KS / 2, LW = 64 + 2 * R;
    median_stage<KS>(t, src, x0, y0, W, H, pitch);
    median_tile<KS, LW>(t, dst, x0, y0, W, H, pitch);
}

// median for the two-iterations-per-launch schedule: a pair takes part iff it is in NORMAL mode at `it`
template <int KS>
__global__ __launch_bounds__(256) void k_median2(MedArgs a, int total)
{
    constexpr int R = KS / 2, TWm = 64, THm = 16, LW = TWm + 2 * R, LH = THm + 2 * R;
    __shared__ __attribute__((aligned(16))) float t[LH][LW];
    const int b = blockIdx.z >> 1, plane = blockIdx.z & 1;
    if (pair_mode2(a.err + (size_t)b * a.errstride, a.it, total, a.thr_q) != M_NORMAL) return;
    const int uc = (a.ctl[b].ubase ^ a.utog) & 1;
    const size_t po = (size_t)b * a.g.splane;
    const float* __restrict__ src = (plane ? a.sb.u2[uc] : a.sb.u1[uc]) + po;
    float* __restrict__ dst = (plane ? a.sb.u2[uc ^ 1] : a.sb.u1[uc ^ 1]) + po;
    const int x0 = blockIdx.x * TWm, y0 = blockIdx.y * THm, W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    median_block<KS>(t, src, dst, x0, y0, W, H, pitch);
}

// stage end for the two-iterations-per-launch schedule: a pair took part in ceil(n_it/2) launches
__global__ void k_stage_end2(const u64* __restrict__ err, int errstride, PairCtl* ctl, int* iters, int B,
                             int total, int inner, int median_on, double thr_q, int level, int warp, int nlev, int warps,
                             int variant, double thr_d)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const u64* e = err + (size_t)b * errstride;
    int n_it = total;
    if (variant) (void)pair_mode_cuda(e, total, total, thr_d, &n_it);
    else
        for (int j = 0; j < total; ++j)
            if (!((double)e[j] > thr_q)) { n_it = j + 1; break; }
    const int n_out = variant ? 0 : (n_it > 0 ? (n_it - 1) / inner + 1 : 0);      // the CUDA variant has no outer loop / median
    const int launches = (n_it + 1) / 2;
    PairCtl c = ctl[b];
    c.ubase = (c.ubase + launches + (median_on ? n_out : 0)) & 1;
    c.pbase = (c.pbase + launches) & 1;
    ctl[b] = c;
    int* o = iters + (((size_t)b * nlev + level) * warps + warp) * 2;
    o[0] = n_it; o[1] = n_out;
}

// ---------------------------------------------------------------------------------------------
// control kernels (a few threads; they keep the stop/continue decisions on the device)
// ---------------------------------------------------------------------------------------------
// end of one (level, warp) stage: executed iteration counts -> stats; advance the ping-pong bases
__global__ void k_stage_end(const u64* __restrict__ err, int errstride, PairCtl* ctl, int* iters, int B,
                            int total, int inner, int median_on, double thr_q, int level, int warp, int nlev, int warps)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const u64* e = err + (size_t)b * errstride;
    int n_it = total;
    for (int j = 0; j < total; ++j)
        if (!((double)e[j] > thr_q)) { n_it = j + 1; break; }
    const int n_out = n_it > 0 ? (n_it - 1) / inner + 1 : 0;
    PairCtl c = ctl[b];
    c.ubase = (c.ubase + n_it + (median_on ? n_out : 0)) & 1;
    c.pbase = (c.pbase + n_it) & 1;
    ctl[b] = c;
    int* o = iters + (((size_t)b * nlev + level) * warps + warp) * 2;
    o[0] = n_it; o[1] = n_out;
}

// mode 0: reset (coarsest level start); mode 1: after k_flow_up (flow moved to the other buffer, p restarts)
__global__ void k_ctl_set(PairCtl* ctl, int B, int mode)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mode == 0) { ctl[b].ubase = 0; ctl[b].pbase = 0; }
    else { ctl[b].ubase ^= 1; ctl[b].pbase = 0; }
}

// merge(u1,u2) -> interleaved [B][H][W][2], times the caller's unit scale (reference :600)
__global__ __launch_bounds__(256) void k_output(StateBufs sb, const PairCtl* __restrict__ ctl, Geom g, float scale,
                                                float* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const int uc = ctl[b].ubase & 1;
    const size_t i = (size_t)b * g.splane + (size_t)y * g.pitch + x;
    float2 v = make_float2(sb.u1[uc][i] * scale, sb.u2[uc][i] * scale);
    reinterpret_cast<float2*>(out)[((size_t)b * g.h + y) * g.w + x] = v;
}

// ---------------------------------------------------------------------------------------------
// Frame conditioning on the device: img2uint8(rgb2gray(frame)) of the reference
// (/root/reference/optical_flow/calculate_optical_flow.py:588, optical_flow_utils.py:30-31), per frame:
//   g = (R/255)*0.2125 + (G/255)*0.7154 + (B/255)*0.0721   (float64, skimage.color.rgb2gray)
//   u8 = rint(((g - min g) / max g) * 255)                  (the reference divides by max, NOT max - min)
// Pass 1 reduces per-frame min / max of g (non-negative doubles order like their bit patterns, so integer atomics do);
// pass 2 recomputes g and writes the byte.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double luma_f64(const uint8_t* p)
{
    return ((double)p[0] / 255.0) * 0.2125 + ((double)p[1] / 255.0) * 0.7154 + ((double)p[2] / 255.0) * 0.0721;
}

__global__ __launch_bounds__(256) void k_cond_minmax(const uint8_t* __restrict__ rgb, size_t npx, u64* __restrict__ mm /* [F][2] */)
{
    __shared__ u64 smin[4], smax[4];
    const int f = blockIdx.y;
    const uint8_t* src = rgb + (size_t)f * npx * 3;
    u64 lo = ~0ull, hi = 0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const u64 b = (u64)__double_as_longlong(luma_f64(src + i * 3));
        lo = b < lo ? b : lo; hi = b > hi ? b : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = smin[w] < lo ? smin[w] : lo; hi = smax[w] > hi ? smax[w] : hi; }
        lo = smin[0] < lo ? smin[0] : lo; hi = smax[0] > hi ? smax[0] : hi;
        atomicMin(&mm[2 * f], lo);
        atomicMax(&mm[2 * f + 1], hi);
    }
}

__global__ __launch_bounds__(256) void k_cond_norm(const uint8_t* __restrict__ rgb, size_t npx, const u64* __restrict__ mm, uint8_t* __restrict__ out)
{
    const int f = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const double mn = __longlong_as_double((long long)mm[2 * f]), mx = __longlong_as_double((long long)mm[2 * f + 1]);
    const double g = luma_f64(rgb + ((size_t)f * npx + i) * 3);
    double v = rint(((g - mn) / mx) * 255.0);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);          // NaN (all-black frame: 0/0) falls through to 0 below
    out[(size_t)f * npx + i] = (uint8_t)(v == v ? (int)v : 0);
}
