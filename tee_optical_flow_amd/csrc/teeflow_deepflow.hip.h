// teeflow_deepflow.hip.h -- gfx950 kernels of the DeepFlow path (BASELINE config 4).
//
// What the reference runs: cv2.optflow.createOptFlow_DeepFlow().calc(I0, I1, None)
// (/root/reference/optical_flow/calculate_optical_flow.py:568, 631) = Gaussian pre-blur, a x0.95 pyramid and, per level,
// cv::VariationalRefinement (robust colour+gradient-constancy data term, robust isotropic smoothness, red-black SOR).
// The arithmetic contract is the same as for DualTVL1: IEEE single precision, expressions evaluated as written, no FMA
// contraction -> bit-identical to oracle/deepflow_oracle.c (which states what is restated from memory and is UNPINNED
// against real OpenCV).
//
// A batch of B pairs runs in lock-step; iteration counts are fixed (5 fixed-point x 25 SOR sweeps per level), so there is
// no device-side control flow here.  All kernels are bandwidth-class stencils (no MFMA).
#pragma once
#include "teeflow_kernels.hip.h"

struct DfBufs {
    float *avg, *Iz, *Ix, *Iy, *Ixx, *Ixy, *Iyy, *Ixz, *Iyz;   // image derivative planes of the current level
    float *A11, *A12, *A22, *b1, *b2, *wg;                      // linear system of the current fixed-point iteration
    float *du, *dv;                                             // flow increment
    float *du2, *dv2;                                           // second copy: the fused SOR kernel reads (du,dv), writes (du2,dv2)
    float *Wu[2], *Wv[2];                                       // flow entering the level (ping-pong across levels)
};

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

#define DF_XY()                                                                                             \
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z; \
    if (x >= g.w || y >= g.h) return;                                                                       \
    const int W = g.w, H = g.h, pitch = g.pitch; (void)W; (void)H;                                          \
    const size_t j = (size_t)y * pitch + x

// GaussianBlur 3x3 (separable, BORDER_REFLECT_101): row pass of the three rows, then the column pass
__global__ __launch_bounds__(256) void k_df_blur(const float* __restrict__ src, float* __restrict__ dst, Geom g, float k0, float k1)
{
    DF_XY();
    const float* S = src + (size_t)b * g.plane;
    const int xl = reflect101(x - 1, W), xr = reflect101(x + 1, W);
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* R = S + (size_t)reflect101(y - 1 + r, H) * pitch;
        t[r] = R[x] * k0 + (R[xl] + R[xr]) * k1;
    }
    dst[(size_t)b * g.plane + j] = t[1] * k0 + (t[0] + t[2]) * k1;
}

// prepareBuffers, part 1: warp I1 by W (cv::remap INTER_LINEAR, 1/32-px fixed point, BORDER_CONSTANT 0);
// Iavg = (I0 + Iw)/2, Iz = Iw - I0; the level's flow increment starts at zero
__global__ __launch_bounds__(256) void k_df_warp(const float* __restrict__ pyr, int off0, int off1, DfBufs d, int cur, Geom g)
{
    DF_XY();
    const size_t po = (size_t)b * g.splane;
    const float* __restrict__ I0 = pyr + (size_t)(off0 + b) * g.plane;
    const float* __restrict__ I1 = pyr + (size_t)(off1 + b) * g.plane;
    const float u = d.Wu[cur][po + j], v = d.Wv[cur][po + j];
    const float mx = (float)x + u, my = (float)y + v;
    const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
    const float fx = (float)(sx & 31) * (1.f / 32), fy = (float)(sy & 31) * (1.f / 32);
    const float wx0 = 1.f - fx, wx1 = fx, wy0 = 1.f - fy, wy1 = fy;
    const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
    const int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    float r;
    if ((unsigned)ix < (unsigned)(W - 1) && (unsigned)iy < (unsigned)(H - 1)) {
        const float* S = I1 + (size_t)iy * pitch + ix;
        r = S[0] * w0 + S[1] * w1 + S[pitch] * w2 + S[pitch + 1] * w3;
    } else if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) {
        r = 0.f;
    } else {
        const bool x0ok = ix >= 0 && ix < W, x1ok = ix + 1 >= 0 && ix + 1 < W, y0ok = iy >= 0 && iy < H, y1ok = iy + 1 >= 0 && iy + 1 < H;
        const float v0 = x0ok && y0ok ? I1[(size_t)iy * pitch + ix] : 0.f;
        const float v1 = x1ok && y0ok ? I1[(size_t)iy * pitch + ix + 1] : 0.f;
        const float v2 = x0ok && y1ok ? I1[(size_t)(iy + 1) * pitch + ix] : 0.f;
        const float v3 = x1ok && y1ok ? I1[(size_t)(iy + 1) * pitch + ix + 1] : 0.f;
        r = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
    }
    const float i0 = I0[j];
    d.avg[po + j] = (i0 + r) * 0.5f;
    d.Iz[po + j] = r - i0;
    d.du[po + j] = 0.f;
    d.dv[po + j] = 0.f;
}

// Sobel ksize 1 ([-1 0 1], BORDER_REPLICATE): first derivatives of Iavg and Iz
__global__ __launch_bounds__(256) void k_df_grad1(DfBufs d, Geom g)
{
    DF_XY();
    const size_t po = (size_t)b * g.splane;
    const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x + 1 : W - 1, yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;
    const float* A = d.avg + po;
    const float* Z = d.Iz + po;
    d.Ix[po + j] = A[(size_t)y * pitch + xr] - A[(size_t)y * pitch + xl];
    d.Iy[po + j] = A[(size_t)yd * pitch + x] - A[(size_t)yu * pitch + x];
    d.Ixz[po + j] = Z[(size_t)y * pitch + xr] - Z[(size_t)y * pitch + xl];
    d.Iyz[po + j] = Z[(size_t)yd * pitch + x] - Z[(size_t)yu * pitch + x];
}

// second derivatives: Ixx, Ixy from Ix; Iyy from Iy
__global__ __launch_bounds__(256) void k_df_grad2(DfBufs d, Geom g)
{
    DF_XY();
    const size_t po = (size_t)b * g.splane;
    const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x + 1 : W - 1, yu = y > 0 ? y - 1 : 0, yd = y < H - 1 ? y + 1 : H - 1;
    const float* X = d.Ix + po;
    const float* Y = d.Iy + po;
    d.Ixx[po + j] = X[(size_t)y * pitch + xr] - X[(size_t)y * pitch + xl];
    d.Ixy[po + j] = X[(size_t)yd * pitch + x] - X[(size_t)yu * pitch + x];
    d.Iyy[po + j] = Y[(size_t)yd * pitch + x] - Y[(size_t)yu * pitch + x];
}

struct DfConst { float zeta2, eps2, delta2, gamma2, alpha2, omega; };

// robust data term (colour + gradient constancy) -> A11, A12, A22, b1, b2; smoothness weight of the current flow W + dW
__global__ __launch_bounds__(256) void k_df_data(DfBufs d, int cur, Geom g, DfConst c)
{
    DF_XY();
    const size_t i = (size_t)b * g.splane + j;
    const float Ix = d.Ix[i], Iy = d.Iy[i], Iz = d.Iz[i], Ixx = d.Ixx[i], Ixy = d.Ixy[i], Iyy = d.Iyy[i], Ixz = d.Ixz[i], Iyz = d.Iyz[i];
    const float du = d.du[i], dv = d.dv[i];
    float derivNorm = Ix * Ix + Iy * Iy + c.zeta2;
    const float Ik1z = Iz + Ix * du + Iy * dv;
    float weight = (c.delta2 / sqrtf(Ik1z * Ik1z / derivNorm + c.eps2)) / derivNorm;
    float a11 = weight * (Ix * Ix) + c.zeta2;
    float a12 = weight * (Ix * Iy);
    float a22 = weight * (Iy * Iy) + c.zeta2;
    float b1 = -weight * (Iz * Ix);
    float b2 = -weight * (Iz * Iy);
    derivNorm = Ixx * Ixx + Ixy * Ixy + c.zeta2;
    const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + c.zeta2;
    const float Ik1zx = Ixz + Ixx * du + Ixy * dv;
    const float Ik1zy = Iyz + Ixy * du + Iyy * dv;
    weight = c.gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + c.eps2);
    a11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
    a12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
    a22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
    b1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
    b2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
    d.A11[i] = a11; d.A12[i] = a12; d.A22[i] = a22; d.b1[i] = b1; d.b2[i] = b2;
    // smoothness weight: forward differences of W + dW (0 across the border)
    const float* Wu = d.Wu[cur];
    const float* Wv = d.Wv[cur];
    const float cu = Wu[i] + du, cv = Wv[i] + dv;
    const float ux = x < W - 1 ? (Wu[i + 1] + d.du[i + 1]) - cu : 0.f;
    const float vx = x < W - 1 ? (Wv[i + 1] + d.dv[i + 1]) - cv : 0.f;
    const float uy = y < H - 1 ? (Wu[i + pitch] + d.du[i + pitch]) - cu : 0.f;
    const float vy = y < H - 1 ? (Wv[i + pitch] + d.dv[i + pitch]) - cv : 0.f;
    d.wg[i] = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
}

// smoothness contributions gathered per pixel in upstream's scatter order (red pass before black pass, horizontal before vertical)
__global__ __launch_bounds__(256) void k_df_smooth(DfBufs d, int cur, Geom g)
{
    DF_XY();
    const size_t i = (size_t)b * g.splane + j;
    const float* Wu = d.Wu[cur];
    const float* Wv = d.Wv[cur];
    float a11 = d.A11[i], a22 = d.A22[i], b1 = d.b1[i], b2 = d.b2[i];
    const bool red = ((x + y) & 1) == 0;
    const float wu = Wu[i], wv = Wv[i], wself = d.wg[i];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool own = red ? (k == 0) : (k == 1);
        if (own) { if (x < W - 1) { b1 += wself * (Wu[i + 1] - wu); a11 += wself; b2 += wself * (Wv[i + 1] - wv); a22 += wself; } }
        else if (x > 0) { const float ww = d.wg[i - 1]; b1 -= ww * (wu - Wu[i - 1]); a11 += ww; b2 -= ww * (wv - Wv[i - 1]); a22 += ww; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool own = red ? (k == 0) : (k == 1);
        if (own) { if (y < H - 1) { b1 += wself * (Wu[i + pitch] - wu); a11 += wself; b2 += wself * (Wv[i + pitch] - wv); a22 += wself; } }
        else if (y > 0) { const float ww = d.wg[i - pitch]; b1 -= ww * (wu - Wu[i - pitch]); a11 += ww; b2 -= ww * (wv - Wv[i - pitch]); a22 += ww; }
    }
    d.A11[i] = a11; d.A22[i] = a22; d.b1[i] = b1; d.b2[i] = b2;
}

// k_df_data + k_df_smooth in one pass: the data term stays in registers, and the smoothness weights of the pixel's left and
// upper neighbours are recomputed here (same expression on the same operands -> same bits) instead of being read back, so
// A11 / A22 / b1 / b2 are written once and never re-read: 72 instead of 116 B/px per fixed-point iteration.
__global__ __launch_bounds__(256) void k_df_data_smooth(DfBufs d, int cur, Geom g, DfConst c)
{
    DF_XY();
    const size_t i = (size_t)b * g.splane + j;
    const float Ix = d.Ix[i], Iy = d.Iy[i], Iz = d.Iz[i], Ixx = d.Ixx[i], Ixy = d.Ixy[i], Iyy = d.Iyy[i], Ixz = d.Ixz[i], Iyz = d.Iyz[i];
    const float du = d.du[i], dv = d.dv[i];
    float derivNorm = Ix * Ix + Iy * Iy + c.zeta2;
    const float Ik1z = Iz + Ix * du + Iy * dv;
    float weight = (c.delta2 / sqrtf(Ik1z * Ik1z / derivNorm + c.eps2)) / derivNorm;
    float a11 = weight * (Ix * Ix) + c.zeta2;
    float a12 = weight * (Ix * Iy);
    float a22 = weight * (Iy * Iy) + c.zeta2;
    float b1 = -weight * (Iz * Ix);
    float b2 = -weight * (Iz * Iy);
    derivNorm = Ixx * Ixx + Ixy * Ixy + c.zeta2;
    const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + c.zeta2;
    const float Ik1zx = Ixz + Ixx * du + Ixy * dv;
    const float Ik1zy = Iyz + Ixy * du + Iyy * dv;
    weight = c.gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + c.eps2);
    a11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
    a12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
    a22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
    b1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
    b2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
    // smoothness weights (forward differences of W + dW, 0 across the border) of this pixel, its left and its upper neighbour
    const float* Wu = d.Wu[cur];
    const float* Wv = d.Wv[cur];
    const bool hl = x > 0, hr = x < W - 1, hu = y > 0, hd = y < H - 1;
    const float wu = Wu[i], wv = Wv[i];
    const float cu = wu + du, cv = wv + dv;
    const float wuR = hr ? Wu[i + 1] : 0.f, wvR = hr ? Wv[i + 1] : 0.f, wuD = hd ? Wu[i + pitch] : 0.f, wvD = hd ? Wv[i + pitch] : 0.f;
    const float wuL = hl ? Wu[i - 1] : 0.f, wvL = hl ? Wv[i - 1] : 0.f, wuU = hu ? Wu[i - pitch] : 0.f, wvU = hu ? Wv[i - pitch] : 0.f;
    float wself, wleft = 0.f, wup = 0.f;
    {
        const float ux = hr ? (wuR + d.du[i + 1]) - cu : 0.f, vx = hr ? (wvR + d.dv[i + 1]) - cv : 0.f;
        const float uy = hd ? (wuD + d.du[i + pitch]) - cu : 0.f, vy = hd ? (wvD + d.dv[i + pitch]) - cv : 0.f;
        wself = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
    }
    if (hl) {
        const float cuL = wuL + d.du[i - 1], cvL = wvL + d.dv[i - 1];
        const float ux = cu - cuL, vx = cv - cvL;                               // the left pixel always has a right neighbour: this one
        const float uy = hd ? (Wu[i - 1 + pitch] + d.du[i - 1 + pitch]) - cuL : 0.f, vy = hd ? (Wv[i - 1 + pitch] + d.dv[i - 1 + pitch]) - cvL : 0.f;
        wleft = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
    }
    if (hu) {
        const float cuU = wuU + d.du[i - pitch], cvU = wvU + d.dv[i - pitch];
        const float ux = hr ? (Wu[i - pitch + 1] + d.du[i - pitch + 1]) - cuU : 0.f, vx = hr ? (Wv[i - pitch + 1] + d.dv[i - pitch + 1]) - cvU : 0.f;
        const float uy = cu - cuU, vy = cv - cvU;                               // the upper pixel always has a lower neighbour: this one
        wup = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
    }
    // smoothness contributions in upstream's scatter order (k_df_smooth): red pass before black pass, horizontal before vertical
    const bool red = ((x + y) & 1) == 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool own = red ? (k == 0) : (k == 1);
        if (own) { if (hr) { b1 += wself * (wuR - wu); a11 += wself; b2 += wself * (wvR - wv); a22 += wself; } }
        else if (hl) { b1 -= wleft * (wu - wuL); a11 += wleft; b2 -= wleft * (wv - wvL); a22 += wleft; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool own = red ? (k == 0) : (k == 1);
        if (own) { if (hd) { b1 += wself * (wuD - wu); a11 += wself; b2 += wself * (wvD - wv); a22 += wself; } }
        else if (hu) { b1 -= wup * (wu - wuU); a11 += wup; b2 -= wup * (wv - wvU); a22 += wup; }
    }
    d.A11[i] = a11; d.A12[i] = a12; d.A22[i] = a22; d.b1[i] = b1; d.b2[i] = b2; d.wg[i] = wself;
}

// one colour of a red-black SOR sweep on dW (thread = one pixel of that colour)
__global__ __launch_bounds__(256) void k_df_sor(DfBufs d, Geom g, int color, float omega)
{
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (y >= g.h) return;
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)) + ((y + color) & 1);
    if (x >= g.w) return;
    const int W = g.w, H = g.h, pitch = g.pitch;
    const size_t i = (size_t)b * g.splane + (size_t)y * pitch + x;
    const float wl = x > 0 ? d.wg[i - 1] : 0.f, wu_ = y > 0 ? d.wg[i - pitch] : 0.f;
    const float ws = d.wg[i];
    const float wr = x < W - 1 ? ws : 0.f, wd = y < H - 1 ? ws : 0.f;
    const float dul = x > 0 ? d.du[i - 1] : 0.f, dur = x < W - 1 ? d.du[i + 1] : 0.f;
    const float duu = y > 0 ? d.du[i - pitch] : 0.f, dud = y < H - 1 ? d.du[i + pitch] : 0.f;
    const float dvl = x > 0 ? d.dv[i - 1] : 0.f, dvr = x < W - 1 ? d.dv[i + 1] : 0.f;
    const float dvu = y > 0 ? d.dv[i - pitch] : 0.f, dvd = y < H - 1 ? d.dv[i + pitch] : 0.f;
    const float sigmaU = wl * dul + wr * dur + wu_ * duu + wd * dud;
    const float sigmaV = wl * dvl + wr * dvr + wu_ * dvu + wd * dvd;
    const float a12 = d.A12[i];
    float du = d.du[i], dv = d.dv[i];
    du += omega * ((sigmaU + d.b1[i] - dv * a12) / d.A11[i] - du);
    dv += omega * ((sigmaV + d.b2[i] - du * a12) / d.A22[i] - dv);
    d.du[i] = du; d.dv[i] = dv;
}

// S complete red-black sweeps per launch on a 64x32 tile with a halo of 2S pixels kept in LDS (du, dv and the weights;
// the five system coefficients of every pixel a thread owns stay in its registers).  A half-sweep only moves information one
// pixel, so after 2S half-sweeps everything farther than 2S-1 pixels from the staged region's edge is exactly what the
// one-colour-per-launch form (k_df_sor) computes; only the tile is written back -- to the second (du2, dv2) copy, because
// neighbouring tiles still read this tile's (du, dv) as their halo.  HBM traffic per sweep drops ~3x at S = 2.
// Each thread owns NS "slots" (row, column pair); a slot holds one red and one black pixel, so in every half-sweep all
// lanes have work (no idle colour), and the slot's geometry flags are computed once.
// LDS holds even-x and odd-x pixels in separate arrays (same-colour pixels of a row are then contiguous: no 2-way bank
// conflicts of a stride-2 access), and a slot's two pixels are fetched / written back with one 8-byte access per plane.
// S > 0: S sweeps on TW x TH tiles with a 2S halo.  S == 0: the region IS the image (W <= TW, H <= TH; one block per pair,
// no halo, nothing recomputed) and `nsw` sweeps -- a whole fixed-point iteration's SOR -- run in one launch.
// DIET = 1 (knob sor_diet, off by default -- see the measurement at its declaration in teeflow.hip): what never changes during the sweeps is hoisted out of them -- per (colour, slot) the four neighbour
// weights already masked by the image border, the LDS addresses of the pixel and its four neighbours (a missing neighbour
// points at the pixel itself and carries weight 0), and the two diagonals pre-scaled with their refined reciprocals, so
// that each of the two IEEE divisions of an update is the 7-instruction tail of the correctly rounded sequence (div1s)
// instead of the full one.  Same products, same sums, same quotients -> same bits (a block whose diagonals leave the
// range div1s is exact for falls back to the plain division).  The SOR kernel is VALU-bound (profiles/r02_sq_counters.json:
// the vector ALU is busy ~90-100 % of a launch), so instructions are what it pays for: 12 selects, 3 weight loads, the
// address arithmetic and 2 reciprocal set-ups less per update.
__device__ __forceinline__ float rcp1s(float bs)
{
    const float r = __builtin_amdgcn_rcpf(bs);
    const float e = __builtin_fmaf(-bs, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div1s(float a, float b, float bs, float rs)     // a / b, see div2s in teeflow_kernels.hip.h
{
    const float as = a * 0x1p64f;
    float q = as * rs;
    float e = __builtin_fmaf(-bs, q, as);
    q = __builtin_fmaf(e, rs, q);
    e = __builtin_fmaf(-bs, q, as);
    q = __builtin_fmaf(e, rs, q);
    return __builtin_amdgcn_div_fixupf(q, b, a);
}

// MW = 1 (tile form only, S > 0): the staged weight plane is replaced by two EDGE-weight planes, WX[p] = w[p] if the pixel has a
// right neighbour in the image else 0 and WY[p] likewise for the pixel below (0 outside the image).  The four weights of an
// update are then WX[self], WX[left], WY[self], WY[up] and all of its loads are unconditional: a neighbour that does not
// exist lies outside the image, where du, dv and the edge weights are staged as 0, so its product is the same 0 * 0 the
// flag-guarded form computes.  Same operations in the same order -> same bits; what goes away is the exec-mask juggling the
// compiler makes of the twelve guarded loads (~90 SALU instructions per six updates).  LDS grows from 6 to 8 planes (61 KB
// at S = 4): still two 1024-thread blocks per CU.
template <int S, int TW, int TH, int NT, int DIET = 0, int MW = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 1024 && S > 0 && !DIET ? 8 : 1, NT == 1024 && S > 0 && !DIET ? 8 : 8))) void k_df_sor_fused(DfBufs d, Geom g, float omega, int nsw)
{
    static_assert(!MW || (S > 0 && !DIET), "edge-weight planes: tile form only");
    constexpr int HL = 2 * S, RW = TW + 2 * HL, RH = TH + 2 * HL, HW = RW / 2, NSLOT = HW * RH, NS = (NSLOT + NT - 1) / NT;
    constexpr int PAD = S == 0 ? HW + 4 : 0;     // whole-image form: row 0 is updated, its (unused) "row above" address must stay inside LDS
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS planes [x parity][ry * HW + rx / 2]: du at (0 + parity) * NSLOT, dv at (2 + parity) * NSLOT, weights at (4 + parity) * NSLOT
    // (addressed by offset: pointer arrays initialised from the LDS base trip a static-initializer limitation of the compiler)
    constexpr int ODU = PAD, ODV = PAD + 2 * NSLOT, OWG = PAD + 4 * NSLOT;
    const int b = blockIdx.z, x0 = blockIdx.x * TW - HL, y0 = blockIdx.y * TH - HL;   // x0 is even
    const int W = g.w, H = g.h, pitch = g.pitch;
    const size_t po = (size_t)b * g.splane;
    // per slot and colour c (0 = (x+y) even): flags and coefficients of the slot's pixel of that colour
    unsigned flg[2][NS];   // bit0 update allowed, bit1 has left, bit2 has right, bit3 has up, bit4 has down, bit5 write back, bit6 x parity
    float a11[2][NS], a12[2][NS], a22[2][NS], b1[2][NS], b2[2][NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int q = threadIdx.x + k * NT;
        const int ry = q / HW, qx = q - ry * HW;
        const int gy = y0 + ry, gxe = x0 + 2 * qx;
        float2 vdu = make_float2(0, 0), vdv = vdu, vw = vdu, v11 = make_float2(1, 1), v12 = vdu, v22 = v11, vb1 = vdu, vb2 = vdu;
        const bool rowok = q < NSLOT && gy >= 0 && gy < H;
        if (rowok && gxe >= 0 && gxe + 1 < W) {
            const size_t i = po + (size_t)gy * pitch + gxe;
            vdu = *reinterpret_cast<const float2*>(d.du + i); vdv = *reinterpret_cast<const float2*>(d.dv + i);
            vw = *reinterpret_cast<const float2*>(d.wg + i);
            v11 = *reinterpret_cast<const float2*>(d.A11 + i); v12 = *reinterpret_cast<const float2*>(d.A12 + i);
            v22 = *reinterpret_cast<const float2*>(d.A22 + i);
            vb1 = *reinterpret_cast<const float2*>(d.b1 + i); vb2 = *reinterpret_cast<const float2*>(d.b2 + i);
        } else if (rowok) {
            if (gxe >= 0 && gxe < W) {
                const size_t i = po + (size_t)gy * pitch + gxe;
                vdu.x = d.du[i]; vdv.x = d.dv[i]; vw.x = d.wg[i]; v11.x = d.A11[i]; v12.x = d.A12[i]; v22.x = d.A22[i]; vb1.x = d.b1[i]; vb2.x = d.b2[i];
            }
            if (gxe + 1 >= 0 && gxe + 1 < W) {
                const size_t i = po + (size_t)gy * pitch + gxe + 1;
                vdu.y = d.du[i]; vdv.y = d.dv[i]; vw.y = d.wg[i]; v11.y = d.A11[i]; v12.y = d.A12[i]; v22.y = d.A22[i]; vb1.y = d.b1[i]; vb2.y = d.b2[i];
            }
        }
        if (q < NSLOT) {
            smem[ODU + q] = vdu.x; smem[ODU + NSLOT + q] = vdu.y; smem[ODV + q] = vdv.x; smem[ODV + NSLOT + q] = vdv.y;
            if constexpr (MW) {
                // WX at OWG (+ parity * NSLOT), WY two planes further; vw is already 0 outside the image
                const bool dn = gy < H - 1;
                smem[OWG + q] = gxe < W - 1 ? vw.x : 0.f; smem[OWG + NSLOT + q] = gxe + 1 < W - 1 ? vw.y : 0.f;
                smem[OWG + 2 * NSLOT + q] = dn ? vw.x : 0.f; smem[OWG + 3 * NSLOT + q] = dn ? vw.y : 0.f;
            } else {
                smem[OWG + q] = vw.x; smem[OWG + NSLOT + q] = vw.y;
            }
        }
        const int ce = (x0 + y0 + ry) & 1;                        // colour of the slot's even-x pixel
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool odd = c != ce;                             // the pixel of colour c is the odd-x one
            a11[c][k] = odd ? v11.y : v11.x; a12[c][k] = odd ? v12.y : v12.x; a22[c][k] = odd ? v22.y : v22.x;
            b1[c][k] = odd ? vb1.y : vb1.x; b2[c][k] = odd ? vb2.y : vb2.x;
            const int rx = 2 * qx + (odd ? 1 : 0), gx = x0 + rx;
            const bool inimg = rowok && gx >= 0 && gx < W;
            // a pixel can be updated when each neighbour is either outside the image (no edge) or inside the staged region
            const bool ok = inimg && (gx == 0 || rx > 0) && (gx == W - 1 || rx < RW - 1) && (gy == 0 || ry > 0) && (gy == H - 1 || ry < RH - 1);
            const bool wb = inimg && rx >= HL && rx < HL + TW && ry >= HL && ry < HL + TH;
            flg[c][k] = (ok ? 1u : 0u) | (gx > 0 ? 2u : 0u) | (gx < W - 1 ? 4u : 0u) | (gy > 0 ? 8u : 0u) | (gy < H - 1 ? 16u : 0u) |
                        (wb ? 32u : 0u) | (odd ? 64u : 0u);
        }
    }
    __syncthreads();
    const int sweeps = S > 0 ? S : nsw;
    bool generic = !DIET;
    if constexpr (DIET) {
        float wL[2][NS], wR[2][NS], wU[2][NS], wD[2][NS], s11[2][NS], r11[2][NS], s22[2][NS], r22[2][NS];
        int oS[2][NS], oL[2][NS], oR[2][NS], oU[2][NS], oD[2][NS];     // float offsets into smem of du at the pixel / its neighbours
        int bad = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const unsigned f = flg[c][k];
                const int q = threadIdx.x + k * NT, par = (f >> 6) & 1;
                const int A = ODU + par * NSLOT, Bo = ODU + (par ^ 1) * NSLOT;
                const int li = par ? q : q - 1, ri = par ? q + 1 : q;
                wL[c][k] = wR[c][k] = wU[c][k] = wD[c][k] = 0.f;
                oS[c][k] = oL[c][k] = oR[c][k] = oU[c][k] = oD[c][k] = A + (q < NSLOT ? q : 0);
                s11[c][k] = s22[c][k] = 0x1p64f; r11[c][k] = r22[c][k] = 0x1p-64f;
                if (f & 1u) {
                    const float ws = smem[OWG - ODU + A + q];
                    if (f & 2u) { wL[c][k] = smem[OWG - ODU + Bo + li]; oL[c][k] = Bo + li; }
                    if (f & 4u) { wR[c][k] = ws; oR[c][k] = Bo + ri; }
                    if (f & 8u) { wU[c][k] = smem[OWG - ODU + A + q - HW]; oU[c][k] = A + q - HW; }
                    if (f & 16u) { wD[c][k] = ws; oD[c][k] = A + q + HW; }
                    s11[c][k] = a11[c][k] * 0x1p64f; r11[c][k] = rcp1s(s11[c][k]);
                    s22[c][k] = a22[c][k] * 0x1p64f; r22[c][k] = rcp1s(s22[c][k]);
                    // div1s is the exact quotient for 2^-24 < |b| < 2^60 (and |a| < 2^60): keep well inside
                    bad |= !(fabsf(a11[c][k]) > 0x1p-20f && fabsf(a11[c][k]) < 0x1p50f && fabsf(a22[c][k]) > 0x1p-20f && fabsf(a22[c][k]) < 0x1p50f);
                }
            }
        generic = __builtin_amdgcn_readfirstlane(__syncthreads_or(bad)) != 0;     // block-uniform: such a block takes the plain form below
#pragma unroll 1
        for (int sw = 0; sw < (generic ? 0 : sweeps); ++sw) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    if (flg[c][k] & 1u) {
                        const float* P = smem;
                        constexpr int DV = 2 * NSLOT;                    // dv planes sit 2 * NSLOT floats behind the du planes
                        const float sigmaU = wL[c][k] * P[oL[c][k]] + wR[c][k] * P[oR[c][k]] + wU[c][k] * P[oU[c][k]] + wD[c][k] * P[oD[c][k]];
                        const float sigmaV = wL[c][k] * P[oL[c][k] + DV] + wR[c][k] * P[oR[c][k] + DV] + wU[c][k] * P[oU[c][k] + DV] + wD[c][k] * P[oD[c][k] + DV];
                        float du = P[oS[c][k]], dv = P[oS[c][k] + DV];
                        const float n1 = sigmaU + b1[c][k] - dv * a12[c][k];
                        du += omega * (div1s(n1, a11[c][k], s11[c][k], r11[c][k]) - du);
                        const float n2 = sigmaV + b2[c][k] - du * a12[c][k];
                        dv += omega * (div1s(n2, a22[c][k], s22[c][k], r22[c][k]) - dv);
                        smem[oS[c][k]] = du; smem[oS[c][k] + DV] = dv;
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll 1
    for (int sw = 0; sw < (generic ? sweeps : 0); ++sw) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const unsigned f = flg[c][k];
                if constexpr (MW) {
                    if (f & 1u) {
                        const int q = threadIdx.x + k * NT;
                        const int par = (f >> 6) & 1;
                        const int A = par * NSLOT + q, Bl = (par ^ 1) * NSLOT + (par ? q : q - 1);     // self / left slot; right = left + 1
                        const float* P = smem;
                        const float wr = P[OWG + A], wd = P[OWG + 2 * NSLOT + A], wl = P[OWG + Bl], wu_ = P[OWG + 2 * NSLOT + A - HW];
                        const float sigmaU = wl * P[ODU + Bl] + wr * P[ODU + Bl + 1] + wu_ * P[ODU + A - HW] + wd * P[ODU + A + HW];
                        const float sigmaV = wl * P[ODV + Bl] + wr * P[ODV + Bl + 1] + wu_ * P[ODV + A - HW] + wd * P[ODV + A + HW];
                        float du = P[ODU + A], dv = P[ODV + A];
                        du += omega * ((sigmaU + b1[c][k] - dv * a12[c][k]) / a11[c][k] - du);
                        dv += omega * ((sigmaV + b2[c][k] - du * a12[c][k]) / a22[c][k] - dv);
                        smem[ODU + A] = du; smem[ODV + A] = dv;
                    }
                } else
                if (f & 1u) {
                    const int q = threadIdx.x + k * NT;
                    const int par = (f >> 6) & 1;
                    // own-parity arrays hold the pixel and its vertical neighbours; the other parity holds left / right
                    float* duA = smem + ODU + par * NSLOT; float* dvA = smem + ODV + par * NSLOT;
                    const float* wA = smem + OWG + par * NSLOT;
                    const float* duB = smem + ODU + (par ^ 1) * NSLOT; const float* dvB = smem + ODV + (par ^ 1) * NSLOT;
                    const float* wB = smem + OWG + (par ^ 1) * NSLOT;
                    const int li = par ? q : q - 1, ri = par ? q + 1 : q;
                    const float ws = wA[q];
                    const float wl = (f & 2u) ? wB[li] : 0.f, wu_ = (f & 8u) ? wA[q - HW] : 0.f;
                    const float wr = (f & 4u) ? ws : 0.f, wd = (f & 16u) ? ws : 0.f;
                    const float dul = (f & 2u) ? duB[li] : 0.f, dur = (f & 4u) ? duB[ri] : 0.f;
                    const float duu = (f & 8u) ? duA[q - HW] : 0.f, dud = (f & 16u) ? duA[q + HW] : 0.f;
                    const float dvl = (f & 2u) ? dvB[li] : 0.f, dvr = (f & 4u) ? dvB[ri] : 0.f;
                    const float dvu = (f & 8u) ? dvA[q - HW] : 0.f, dvd = (f & 16u) ? dvA[q + HW] : 0.f;
                    const float sigmaU = wl * dul + wr * dur + wu_ * duu + wd * dud;
                    const float sigmaV = wl * dvl + wr * dvr + wu_ * dvu + wd * dvd;
                    float du = duA[q], dv = dvA[q];
                    du += omega * ((sigmaU + b1[c][k] - dv * a12[c][k]) / a11[c][k] - du);
                    dv += omega * ((sigmaV + b2[c][k] - du * a12[c][k]) / a22[c][k] - dv);
                    duA[q] = du; dvA[q] = dv;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const unsigned f0 = flg[0][k], f1 = flg[1][k];
        if ((f0 | f1) & 32u) {
            const int q = threadIdx.x + k * NT;
            const int ry = q / HW, qx = q - ry * HW;
            const size_t i = po + (size_t)(y0 + ry) * pitch + (x0 + 2 * qx);   // other tiles still read (du, dv) of this tile as halo
            if ((f0 & f1) & 32u) {
                *reinterpret_cast<float2*>(d.du2 + i) = make_float2(smem[ODU + q], smem[ODU + NSLOT + q]);
                *reinterpret_cast<float2*>(d.dv2 + i) = make_float2(smem[ODV + q], smem[ODV + NSLOT + q]);
            } else {
                const int odd = ((f0 & 32u) ? f0 : f1) >> 6 & 1;
                d.du2[i + odd] = smem[ODU + odd * NSLOT + q]; d.dv2[i + odd] = smem[ODV + odd * NSLOT + q];
            }
        }
    }
}

// W + dW of this level -> flow of the next finer level (resize INTER_LINEAR to its size, times 1/downscaleFactor)
__global__ __launch_bounds__(256) void k_df_up(DfBufs d, int cur, Geom gs, Geom gd, double scale_x, double scale_y, float mul)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (dx >= gd.w || dy >= gd.h) return;
    // the sum W + dW is formed first (upstream merges it into W before resizing); avg / Iz are free at this point
    const size_t di = (size_t)b * gd.splane + (size_t)dy * gd.pitch + dx;
    d.Wu[cur ^ 1][di] = resize_px(d.avg + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, scale_x, scale_y) * mul;
    d.Wv[cur ^ 1][di] = resize_px(d.Iz + (size_t)b * gs.splane, gs.w, gs.h, gs.pitch, dx, dy, scale_x, scale_y) * mul;
}

// avg <- Wu + du, Iz <- Wv + dv (scratch planes reused as the level's final flow)
__global__ __launch_bounds__(256) void k_df_sum(DfBufs d, int cur, Geom g)
{
    DF_XY();
    const size_t i = (size_t)b * g.splane + j;
    d.avg[i] = d.Wu[cur][i] + d.du[i];
    d.Iz[i] = d.Wv[cur][i] + d.dv[i];
}

__global__ __launch_bounds__(256) void k_df_out(DfBufs d, Geom g, float scale, float* __restrict__ out)
{
    DF_XY();
    const size_t i = (size_t)b * g.splane + j;
    reinterpret_cast<float2*>(out)[((size_t)b * g.h + y) * g.w + x] = make_float2(d.avg[i] * scale, d.Iz[i] * scale);
}
