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
    float *du2, *dv2;                                           // second copy: the register-tile SOR kernel reads (du,dv), writes (du2,dv2)
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

// k_df_data + k_df_smooth in one pass, FOUR pixels per thread (k_df_data_smooth4): the data term stays in registers, and the smoothness
// weights of a pixel's left and upper neighbours are recomputed (same expression on the same operands -> same bits) instead of being read
// back, so A11 / A22 / b1 / b2 are written once and never re-read: 72 instead of 116 B/px per fixed-point iteration.  (The one-pixel-per-
// thread form of the same fusion was removed in round 4; the two-kernel form above stays as the plain reference the fused one is tested against.)
// Every plane is read with 16-byte loads (the 4-byte loads of the one-pixel form
// reach 3.4 TB/s on its 72 B per pixel; this kernel is 14 % of a DeepFlow solve), the weight of a pixel's left neighbour is the
// previous pixel's own weight (same expression on the same operands: computed once), only the quad's first pixel recomputes it.
// Same operations in the same order per pixel -> bit-identical to k_df_data + k_df_smooth.
__global__ __launch_bounds__(256) void k_df_data_smooth4(DfBufs d, int cur, Geom g, DfConst c)
{
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    const int W = g.w, H = g.h, pitch = g.pitch;
    if (x >= W || y >= H) return;
    const size_t i0 = (size_t)b * g.splane + (size_t)y * pitch + x;
    const float* Wu = d.Wu[cur];
    const float* Wv = d.Wv[cur];
    const bool hu = y > 0, hd = y < H - 1;
    // W and W + dW on rows y-1, y, y+1, columns x-1 .. x+4 (index 0 = column x-1); rows / columns outside the image stay 0 and are never used
    float wu[3][6], wv[3][6], su[3][6], sv[3][6], du4[4], dv4[4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const bool rok = r == 1 || (r == 0 ? hu : hd);
        float4 a = make_float4(0, 0, 0, 0), bq = a, cq = a, dq = a;
        float al = 0.f, bl = 0.f, cl = 0.f, dl = 0.f, ar = 0.f, br = 0.f, cr = 0.f, dr = 0.f;
        if (rok) {
            const size_t i = (size_t)((long long)i0 + (long long)(r - 1) * pitch);
            a = ld4(Wu + i); bq = ld4(Wv + i); cq = ld4(d.du + i); dq = ld4(d.dv + i);
            if (x > 0) { al = Wu[i - 1]; bl = Wv[i - 1]; cl = d.du[i - 1]; dl = d.dv[i - 1]; }
            if (x + 4 < W) { ar = Wu[i + 4]; br = Wv[i + 4]; cr = d.du[i + 4]; dr = d.dv[i + 4]; }
        }
        const float A[6] = {al, a.x, a.y, a.z, a.w, ar}, Bv[6] = {bl, bq.x, bq.y, bq.z, bq.w, br};
        const float Cc[6] = {cl, cq.x, cq.y, cq.z, cq.w, cr}, Dd[6] = {dl, dq.x, dq.y, dq.z, dq.w, dr};
#pragma unroll
        for (int j = 0; j < 6; ++j) { wu[r][j] = A[j]; wv[r][j] = Bv[j]; su[r][j] = A[j] + Cc[j]; sv[r][j] = Bv[j] + Dd[j]; }
        if (r == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { du4[k] = Cc[k + 1]; dv4[k] = Dd[k + 1]; }
        }
    }
    float Ix[4], Iy[4], Iz[4], Ixx[4], Ixy[4], Iyy[4], Ixz[4], Iyz[4];
    { float4 v;
      v = ld4(d.Ix + i0); UNPACK4(Ix, v) v = ld4(d.Iy + i0); UNPACK4(Iy, v) v = ld4(d.Iz + i0); UNPACK4(Iz, v)
      v = ld4(d.Ixx + i0); UNPACK4(Ixx, v) v = ld4(d.Ixy + i0); UNPACK4(Ixy, v) v = ld4(d.Iyy + i0); UNPACK4(Iyy, v)
      v = ld4(d.Ixz + i0); UNPACK4(Ixz, v) v = ld4(d.Iyz + i0); UNPACK4(Iyz, v) }
    float o11[4], o12[4], o22[4], ob1[4], ob2[4], owg[4];
    float wprev = 0.f;                  // weight of the pixel left of the current one
    if (x > 0) {                        // ... of the quad's first pixel: column x-1 always has a right neighbour (this one)
        const float ux = su[1][1] - su[1][0], vx = sv[1][1] - sv[1][0];
        const float uy = hd ? su[2][0] - su[1][0] : 0.f, vy = hd ? sv[2][0] - sv[1][0] : 0.f;
        wprev = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = k + 1;            // column index into the 6-wide rows
        const bool hl = x + k > 0, hr = x + k < W - 1;
        // data term (k_df_data)
        const float ddu = du4[k], ddv = dv4[k];
        float derivNorm = Ix[k] * Ix[k] + Iy[k] * Iy[k] + c.zeta2;
        const float Ik1z = Iz[k] + Ix[k] * ddu + Iy[k] * ddv;
        float weight = (c.delta2 / sqrtf(Ik1z * Ik1z / derivNorm + c.eps2)) / derivNorm;
        float a11 = weight * (Ix[k] * Ix[k]) + c.zeta2;
        float a12 = weight * (Ix[k] * Iy[k]);
        float a22 = weight * (Iy[k] * Iy[k]) + c.zeta2;
        float b1 = -weight * (Iz[k] * Ix[k]);
        float b2 = -weight * (Iz[k] * Iy[k]);
        derivNorm = Ixx[k] * Ixx[k] + Ixy[k] * Ixy[k] + c.zeta2;
        const float derivNorm2 = Iyy[k] * Iyy[k] + Ixy[k] * Ixy[k] + c.zeta2;
        const float Ik1zx = Ixz[k] + Ixx[k] * ddu + Ixy[k] * ddv;
        const float Ik1zy = Iyz[k] + Ixy[k] * ddu + Iyy[k] * ddv;
        weight = c.gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + c.eps2);
        a11 += weight * (Ixx[k] * Ixx[k] / derivNorm + Ixy[k] * Ixy[k] / derivNorm2);
        a12 += weight * (Ixx[k] * Ixy[k] / derivNorm + Ixy[k] * Iyy[k] / derivNorm2);
        a22 += weight * (Ixy[k] * Ixy[k] / derivNorm + Iyy[k] * Iyy[k] / derivNorm2);
        b1 += -weight * (Ixx[k] * Ixz[k] / derivNorm + Ixy[k] * Iyz[k] / derivNorm2);
        b2 += -weight * (Ixy[k] * Ixz[k] / derivNorm + Iyy[k] * Iyz[k] / derivNorm2);
        // smoothness weights of this pixel and of the one above it (forward differences of W + dW, 0 across the border)
        const float cu = su[1][j], cv = sv[1][j];
        float wself, wup = 0.f;
        {
            const float ux = hr ? su[1][j + 1] - cu : 0.f, vx = hr ? sv[1][j + 1] - cv : 0.f;
            const float uy = hd ? su[2][j] - cu : 0.f, vy = hd ? sv[2][j] - cv : 0.f;
            wself = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
        }
        if (hu) {
            const float cuU = su[0][j], cvU = sv[0][j];
            const float ux = hr ? su[0][j + 1] - cuU : 0.f, vx = hr ? sv[0][j + 1] - cvU : 0.f;
            const float uy = cu - cuU, vy = cv - cvU;
            wup = c.alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + c.eps2);
        }
        const float wleft = wprev;
        // smoothness contributions in upstream's scatter order: red pass before black pass, horizontal before vertical
        const float pu = wu[1][j], pv = wv[1][j];
        const bool red = ((x + k + y) & 1) == 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool own = red ? (q == 0) : (q == 1);
            if (own) { if (hr) { b1 += wself * (wu[1][j + 1] - pu); a11 += wself; b2 += wself * (wv[1][j + 1] - pv); a22 += wself; } }
            else if (hl) { b1 -= wleft * (pu - wu[1][j - 1]); a11 += wleft; b2 -= wleft * (pv - wv[1][j - 1]); a22 += wleft; }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool own = red ? (q == 0) : (q == 1);
            if (own) { if (hd) { b1 += wself * (wu[2][j] - pu); a11 += wself; b2 += wself * (wv[2][j] - pv); a22 += wself; } }
            else if (hu) { b1 -= wup * (pu - wu[0][j]); a11 += wup; b2 -= wup * (pv - wv[0][j]); a22 += wup; }
        }
        o11[k] = a11; o12[k] = a12; o22[k] = a22; ob1[k] = b1; ob2[k] = b2; owg[k] = wself;
        wprev = wself;
    }
    if (x + 3 < W) {
        st4(d.A11 + i0, PACK4(o11)); st4(d.A12 + i0, PACK4(o12)); st4(d.A22 + i0, PACK4(o22));
        st4(d.b1 + i0, PACK4(ob1)); st4(d.b2 + i0, PACK4(ob2)); st4(d.wg + i0, PACK4(owg));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x + k < W) { d.A11[i0 + k] = o11[k]; d.A12[i0 + k] = o12[k]; d.A22[i0 + k] = o22[k]; d.b1[i0 + k] = ob1[k]; d.b2[i0 + k] = ob2[k]; d.wg[i0 + k] = owg[k]; }
    }
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

// refined reciprocal: v_rcp_f32 + one Newton step (what the hardware's own division sequence starts with)
__device__ __forceinline__ float rcp1s(float bs)
{
    const float r = __builtin_amdgcn_rcpf(bs);
    const float e = __builtin_fmaf(-bs, r, 1.0f);
    return __builtin_fmaf(e, r, r);
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
