/*
 * oracle/deepflow_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement of what the reference runs for OF_algo == 'deepflow':
 *   /root/reference/optical_flow/calculate_optical_flow.py:568   cv2.optflow.createOptFlow_DeepFlow()
 *   /root/reference/optical_flow/calculate_optical_flow.py:631   OF_model.calc(saliency_1, saliency_2, None)
 * (the reference's own CLI hard-codes this algorithm, calculate_optical_flow.py:735-739).
 *
 * The arithmetic lives in opencv-contrib (optflow/src/deepflow.cpp) and opencv (video/src/variational_refinement.cpp,
 * imgproc GaussianBlur / resize / remap / Sobel), none of which is in /root/reference or installable here.  This file
 * restates the published algorithm from SURVEY.md Appendix B and from the upstream code as the author remembers it:
 *
 *        >>>>>>>>  PARITY UNPINNED (vs real OpenCV), and LOWER CONFIDENCE than the DualTVL1 oracle  <<<<<<<<
 *
 * What is restated:
 *   OpticalFlowDeepFlow::calc : convertTo(CV_32F) (values stay 0..255); GaussianBlur(3x3, sigma 0.6, REFLECT_101);
 *       pyramid size_{l+1} = (int)(size_l*0.95f + 0.5f) while both sides > 25 (resize INTER_LINEAR to that size);
 *       W = 0 at the coarsest level; per level VariationalRefinement with alpha 4*1, delta 0.5/3, gamma 5/3,
 *       5 fixed-point iterations x 25 SOR sweeps, omega 1.6; W = resize(W, next size) * (1/0.95f).
 *   VariationalRefinement::calcUV : warp I1 by W (remap INTER_LINEAR, 1/32-px fixed point, BORDER_CONSTANT 0);
 *       Iavg = (I0 + Iw)/2, Iz = Iw - I0; derivatives by Sobel ksize 1 ([-1 0 1], no 1/2, BORDER_REPLICATE):
 *       Ix, Iy of Iavg; Ixx, Ixy of Ix; Iyy of Iy; Ixz, Iyz of Iz;  dW = 0;
 *       fixed-point loop { robust data term (colour + gradient constancy, zeta 0.1, eps 1e-3) -> A11,A12,A22,b1,b2;
 *       smoothness weight w = (alpha/2)/sqrt(|grad(W+dW)|^2 + eps^2) from forward differences; every in-image
 *       edge (p,q) to the right / below p carries w(p): A(p)+=w, A(q)+=w, b(p)+=w(W(q)-W(p)), b(q)-=w(W(q)-W(p));
 *       25 x red-black SOR on dW (u then v per pixel, v sees the new u) }.
 * The red/black split buffers of upstream are a memory layout, not arithmetic: they are not reproduced.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Float expressions are evaluated as written, left to right, without FMA contraction (-ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    float sigma; int min_size; float downscale_factor; int fixed_point_iterations; int sor_iterations;
    float alpha, delta, gamma, omega;
    float zeta, epsilon;      /* VariationalRefinement constants */
} dfo_params;

ORC_API void dfo_default_params(dfo_params* p)
{
    p->sigma = 0.6f; p->min_size = 25; p->downscale_factor = 0.95f; p->fixed_point_iterations = 5; p->sor_iterations = 25;
    p->alpha = 1.0f; p->delta = 0.5f; p->gamma = 5.0f; p->omega = 1.6f; p->zeta = 0.1f; p->epsilon = 0.001f;
}

static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

/* cv::resize INTER_LINEAR, CV_32FC1, explicit dsize (same generic path as the DualTVL1 oracle) */
ORC_API void dfo_resize_linear(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    float* a1s = (float*)malloc(sizeof(float) * (size_t)dw);
    unsigned char* tail = (unsigned char*)malloc((size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        tail[dx] = 0;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { tail[dx] = 1; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx; a1s[dx] = fx;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        const float b0 = 1.f - fy, b1 = fy;
        const float* S0 = src + (size_t)clipi(sy, 0, sh) * sw;
        const float* S1 = src + (size_t)clipi(sy + 1, 0, sh) * sw;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            float t0, t1;
            if (tail[dx]) { t0 = S0[sx]; t1 = S1[sx]; }
            else { const float a1 = a1s[dx], a0 = 1.f - a1; t0 = S0[sx] * a0 + S0[sx + 1] * a1; t1 = S1[sx] * a0 + S1[sx + 1] * a1; }
            dst[(size_t)dy * dw + dx] = t0 * b0 + t1 * b1;
        }
    }
    free(xofs); free(a1s); free(tail);
}

/* getGaussianKernel(3, sigma, CV_32F): normalised in double, cast to float */
ORC_API void dfo_gauss3(float sigma, float* k /* [2]: centre, side */)
{
    const double s2 = -0.5 / ((double)sigma * (double)sigma);
    const double t0 = exp(s2 * 1.0), t1 = exp(0.0);
    const double inv = 1.0 / (t0 + t1 + t0);
    k[0] = (float)(t1 * inv); k[1] = (float)(t0 * inv);
}

/* GaussianBlur(src, dst, Size(3,3), sigma), BORDER_REFLECT_101: row pass then column pass, symmetric small-kernel form */
ORC_API void dfo_gauss_blur3(const float* src, int w, int h, float sigma, float* dst)
{
    float k[2];
    dfo_gauss3(sigma, k);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* S = src + (size_t)y * w;
            tmp[(size_t)y * w + x] = S[x] * k[0] + (S[reflect101(x - 1, w)] + S[reflect101(x + 1, w)]) * k[1];
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* S0 = tmp + (size_t)reflect101(y - 1, h) * w;
        const float* S1 = tmp + (size_t)y * w;
        const float* S2 = tmp + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x) dst[(size_t)y * w + x] = S1[x] * k[0] + (S0[x] + S2[x]) * k[1];
    }
    free(tmp);
}

/* cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on CV_32FC1 with float maps (x+u, y+v) */
ORC_API void dfo_warp_linear(const float* I1, int w, int h, const float* u, const float* v, float* dst)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float mx = x + u[i], my = y + v[i];
            const int sx = cv_round_f(mx * 32), sy = cv_round_f(my * 32);
            const float fx = (sx & 31) * (1.f / 32), fy = (sy & 31) * (1.f / 32);
            const float wx0 = 1.f - fx, wx1 = fx, wy0 = 1.f - fy, wy1 = fy;
            const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
            const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
            float r;
            if ((unsigned)ix < (unsigned)(w - 1) && (unsigned)iy < (unsigned)(h - 1)) {
                const float* S = I1 + (size_t)iy * w + ix;
                r = S[0] * w0 + S[1] * w1 + S[w] * w2 + S[w + 1] * w3;
            } else if (ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0) {
                r = 0.f;
            } else {
                const int x0ok = ix >= 0 && ix < w, x1ok = ix + 1 >= 0 && ix + 1 < w, y0ok = iy >= 0 && iy < h, y1ok = iy + 1 >= 0 && iy + 1 < h;
                const float v0 = x0ok && y0ok ? I1[(size_t)iy * w + ix] : 0.f;
                const float v1 = x1ok && y0ok ? I1[(size_t)iy * w + ix + 1] : 0.f;
                const float v2 = x0ok && y1ok ? I1[(size_t)(iy + 1) * w + ix] : 0.f;
                const float v3 = x1ok && y1ok ? I1[(size_t)(iy + 1) * w + ix + 1] : 0.f;
                r = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
            }
            dst[i] = r;
        }
}

/* Sobel ksize 1: d/dx = S[x+1] - S[x-1], d/dy likewise, BORDER_REPLICATE, no scaling */
static void grad_x(const float* s, int w, int h, float* d)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int xl = x > 0 ? x - 1 : 0, xr = x < w - 1 ? x + 1 : w - 1;
            d[(size_t)y * w + x] = s[(size_t)y * w + xr] - s[(size_t)y * w + xl];
        }
}
static void grad_y(const float* s, int w, int h, float* d)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const int yu = y > 0 ? y - 1 : 0, yd = y < h - 1 ? y + 1 : h - 1;
        for (int x = 0; x < w; ++x) d[(size_t)y * w + x] = s[(size_t)yd * w + x] - s[(size_t)yu * w + x];
    }
}

typedef struct {
    float *Iw, *Ix, *Iy, *Iz, *Ixx, *Ixy, *Iyy, *Ixz, *Iyz, *A11, *A12, *A22, *b1, *b2, *wgt, *tu, *tv, *du, *dv, *tmp;
} vr_bufs;

/* derivative planes of one VariationalRefinement call (prepareBuffers) */
ORC_API void dfo_derivatives(const float* I0, const float* I1, int w, int h, const float* u, const float* v,
                             float* Ix, float* Iy, float* Iz, float* Ixx, float* Ixy, float* Iyy, float* Ixz, float* Iyz)
{
    const size_t n = (size_t)w * h;
    float* Iw = (float*)malloc(n * 4);
    float* avg = (float*)malloc(n * 4);
    dfo_warp_linear(I1, w, h, u, v, Iw);
    for (size_t i = 0; i < n; ++i) { avg[i] = (I0[i] + Iw[i]) * 0.5f; Iz[i] = Iw[i] - I0[i]; }
    grad_x(avg, w, h, Ix); grad_y(avg, w, h, Iy);
    grad_x(Iz, w, h, Ixz); grad_y(Iz, w, h, Iyz);
    grad_x(Ix, w, h, Ixx); grad_y(Ix, w, h, Ixy); grad_y(Iy, w, h, Iyy);
    free(Iw); free(avg);
}

/* one fixed-point iteration: linear system from (W, dW), then `sor` red-black sweeps on dW */
static void fixed_point_iteration(const dfo_params* P, float alpha, float delta, float gamma, int w, int h,
                                  const float* Ix, const float* Iy, const float* Iz, const float* Ixx, const float* Ixy,
                                  const float* Iyy, const float* Ixz, const float* Iyz, const float* Wu, const float* Wv,
                                  float* du, float* dv, float* A11, float* A12, float* A22, float* b1, float* b2, float* wg)
{
    const float zeta2 = P->zeta * P->zeta, eps2 = P->epsilon * P->epsilon;
    const float delta2 = delta / 2, gamma2 = gamma / 2, alpha2 = alpha / 2;
    /* data term */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t j = (size_t)y * w + x;
            float derivNorm = Ix[j] * Ix[j] + Iy[j] * Iy[j] + zeta2;
            const float Ik1z = Iz[j] + Ix[j] * du[j] + Iy[j] * dv[j];
            float weight = (delta2 / sqrtf(Ik1z * Ik1z / derivNorm + eps2)) / derivNorm;
            float a11 = weight * (Ix[j] * Ix[j]) + zeta2;
            float a12 = weight * (Ix[j] * Iy[j]);
            float a22 = weight * (Iy[j] * Iy[j]) + zeta2;
            float bb1 = -weight * (Iz[j] * Ix[j]);
            float bb2 = -weight * (Iz[j] * Iy[j]);
            derivNorm = Ixx[j] * Ixx[j] + Ixy[j] * Ixy[j] + zeta2;
            const float derivNorm2 = Iyy[j] * Iyy[j] + Ixy[j] * Ixy[j] + zeta2;
            const float Ik1zx = Ixz[j] + Ixx[j] * du[j] + Ixy[j] * dv[j];
            const float Ik1zy = Iyz[j] + Ixy[j] * du[j] + Iyy[j] * dv[j];
            weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
            a11 += weight * (Ixx[j] * Ixx[j] / derivNorm + Ixy[j] * Ixy[j] / derivNorm2);
            a12 += weight * (Ixx[j] * Ixy[j] / derivNorm + Ixy[j] * Iyy[j] / derivNorm2);
            a22 += weight * (Ixy[j] * Ixy[j] / derivNorm + Iyy[j] * Iyy[j] / derivNorm2);
            bb1 += -weight * (Ixx[j] * Ixz[j] / derivNorm + Ixy[j] * Iyz[j] / derivNorm2);
            bb2 += -weight * (Ixy[j] * Ixz[j] / derivNorm + Iyy[j] * Iyz[j] / derivNorm2);
            A11[j] = a11; A12[j] = a12; A22[j] = a22; b1[j] = bb1; b2[j] = bb2;
        }
    /* smoothness weights from the current flow W + dW (forward differences, 0 across the border) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t j = (size_t)y * w + x;
            const float cu = Wu[j] + du[j], cv = Wv[j] + dv[j];
            const float ux = x < w - 1 ? (Wu[j + 1] + du[j + 1]) - cu : 0.f;
            const float vx = x < w - 1 ? (Wv[j + 1] + dv[j + 1]) - cv : 0.f;
            const float uy = y < h - 1 ? (Wu[j + w] + du[j + w]) - cu : 0.f;
            const float vy = y < h - 1 ? (Wv[j + w] + dv[j + w]) - cv : 0.f;
            wg[j] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
        }
    /* smoothness contributions, gathered per pixel in upstream's pass order: horizontal pass (own right edge, then the
     * left neighbour's edge), then vertical pass (own lower edge, then the upper neighbour's edge) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t j = (size_t)y * w + x;
            float a11 = A11[j], a22 = A22[j], bb1 = b1[j], bb2 = b2[j];
            /* upstream scatters edge contributions in a red pass then a black pass, so a red pixel ((x+y) even) receives its
             * own edge before its neighbour's and a black pixel the other way round; horizontal passes precede vertical ones */
            const int red = ((x + y) & 1) == 0;
            for (int k = 0; k < 2; ++k) {
                const int own = red ? (k == 0) : (k == 1);
                if (own) { if (x < w - 1) { const float ww = wg[j]; bb1 += ww * (Wu[j + 1] - Wu[j]); a11 += ww; bb2 += ww * (Wv[j + 1] - Wv[j]); a22 += ww; } }
                else if (x > 0) { const float ww = wg[j - 1]; bb1 -= ww * (Wu[j] - Wu[j - 1]); a11 += ww; bb2 -= ww * (Wv[j] - Wv[j - 1]); a22 += ww; }
            }
            for (int k = 0; k < 2; ++k) {
                const int own = red ? (k == 0) : (k == 1);
                if (own) { if (y < h - 1) { const float ww = wg[j]; bb1 += ww * (Wu[j + w] - Wu[j]); a11 += ww; bb2 += ww * (Wv[j + w] - Wv[j]); a22 += ww; } }
                else if (y > 0) { const float ww = wg[j - w]; bb1 -= ww * (Wu[j] - Wu[j - w]); a11 += ww; bb2 -= ww * (Wv[j] - Wv[j - w]); a22 += ww; }
            }
            A11[j] = a11; A22[j] = a22; b1[j] = bb1; b2[j] = bb2;
        }
    /* red-black SOR on dW */
    for (int it = 0; it < P->sor_iterations; ++it)
        for (int color = 0; color < 2; ++color) {
#pragma omp parallel for schedule(static)
            for (int y = 0; y < h; ++y)
                for (int x = (y + color) & 1; x < w; x += 2) {
                    const size_t j = (size_t)y * w + x;
                    const float wl = x > 0 ? wg[j - 1] : 0.f, wu_ = y > 0 ? wg[j - w] : 0.f;
                    const float wr = x < w - 1 ? wg[j] : 0.f, wd = y < h - 1 ? wg[j] : 0.f;
                    const float dul = x > 0 ? du[j - 1] : 0.f, dur = x < w - 1 ? du[j + 1] : 0.f;
                    const float duu = y > 0 ? du[j - w] : 0.f, dud = y < h - 1 ? du[j + w] : 0.f;
                    const float dvl = x > 0 ? dv[j - 1] : 0.f, dvr = x < w - 1 ? dv[j + 1] : 0.f;
                    const float dvu = y > 0 ? dv[j - w] : 0.f, dvd = y < h - 1 ? dv[j + w] : 0.f;
                    const float sigmaU = wl * dul + wr * dur + wu_ * duu + wd * dud;
                    const float sigmaV = wl * dvl + wr * dvr + wu_ * dvu + wd * dvd;
                    du[j] += P->omega * ((sigmaU + b1[j] - dv[j] * A12[j]) / A11[j] - du[j]);
                    dv[j] += P->omega * ((sigmaV + b2[j] - du[j] * A12[j]) / A22[j] - dv[j]);
                }
        }
}

/* VariationalRefinement::calcUV with explicit parameters; u, v updated in place */
ORC_API void dfo_variational_refine(const dfo_params* P, float alpha, float delta, float gamma, const float* I0, const float* I1,
                                    int w, int h, float* u, float* v)
{
    const size_t n = (size_t)w * h;
    float* buf = (float*)malloc(n * 4 * 16);
    float *Ix = buf, *Iy = buf + n, *Iz = buf + 2 * n, *Ixx = buf + 3 * n, *Ixy = buf + 4 * n, *Iyy = buf + 5 * n, *Ixz = buf + 6 * n,
          *Iyz = buf + 7 * n, *A11 = buf + 8 * n, *A12 = buf + 9 * n, *A22 = buf + 10 * n, *b1 = buf + 11 * n, *b2 = buf + 12 * n,
          *wg = buf + 13 * n, *du = buf + 14 * n, *dv = buf + 15 * n;
    dfo_derivatives(I0, I1, w, h, u, v, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz);
    memset(du, 0, n * 4); memset(dv, 0, n * 4);
    for (int i = 0; i < P->fixed_point_iterations; ++i)
        fixed_point_iteration(P, alpha, delta, gamma, w, h, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz, u, v, du, dv, A11, A12, A22, b1, b2, wg);
    for (size_t i = 0; i < n; ++i) { u[i] = u[i] + du[i]; v[i] = v[i] + dv[i]; }
    free(buf);
}

/* pyramid sizes of OpticalFlowDeepFlow::buildPyramid; returns the number of levels (<= cap) */
ORC_API int dfo_pyramid_sizes(const dfo_params* P, int W, int H, int* ws, int* hs, int cap)
{
    int n = 1;
    ws[0] = W; hs[0] = H;
    while (n < cap) {
        const int nw = (int)(ws[n - 1] * P->downscale_factor + 0.5f), nh = (int)(hs[n - 1] * P->downscale_factor + 0.5f);
        if (nh <= P->min_size || nw <= P->min_size) break;
        ws[n] = nw; hs[n] = nh; ++n;
    }
    return n;
}

/* OpticalFlowDeepFlow::calc on frames already converted to CV_32F (`convertTo(CV_32F)` without a factor: uint8 frames keep their 0..255
 * values, float frames are taken as they are -- a saliency map in [0,1] stays in [0,1], with zeta and epsilon unchanged).
 * flow float32 [H][W][2].  Returns the number of pyramid levels. */
static int deepflow_core(const dfo_params* P, const float* f0, const float* f1, int H, int W, float* flow)
{
    enum { CAP = 256 };
    int ws[CAP], hs[CAP];
    const int L = dfo_pyramid_sizes(P, W, H, ws, hs, CAP);
    float** p0 = (float**)calloc((size_t)L, sizeof(float*));
    float** p1 = (float**)calloc((size_t)L, sizeof(float*));
    const size_t n0 = (size_t)W * H;
    p0[0] = (float*)malloc(n0 * 4); p1[0] = (float*)malloc(n0 * 4);
    dfo_gauss_blur3(f0, W, H, P->sigma, p0[0]);
    dfo_gauss_blur3(f1, W, H, P->sigma, p1[0]);
    for (int l = 1; l < L; ++l) {
        p0[l] = (float*)malloc((size_t)ws[l] * hs[l] * 4); p1[l] = (float*)malloc((size_t)ws[l] * hs[l] * 4);
        dfo_resize_linear(p0[l - 1], ws[l - 1], hs[l - 1], p0[l], ws[l], hs[l]);
        dfo_resize_linear(p1[l - 1], ws[l - 1], hs[l - 1], p1[l], ws[l], hs[l]);
    }
    float* u = (float*)calloc(n0, 4);
    float* v = (float*)calloc(n0, 4);
    float* u2 = (float*)malloc(n0 * 4);
    float* v2 = (float*)malloc(n0 * 4);
    const float mul = 1.0f / P->downscale_factor;
    for (int l = L - 1; l >= 0; --l) {
        dfo_variational_refine(P, 4 * P->alpha, P->delta / 3, P->gamma / 3, p0[l], p1[l], ws[l], hs[l], u, v);
        if (l > 0) {
            dfo_resize_linear(u, ws[l], hs[l], u2, ws[l - 1], hs[l - 1]);
            dfo_resize_linear(v, ws[l], hs[l], v2, ws[l - 1], hs[l - 1]);
            const size_t n = (size_t)ws[l - 1] * hs[l - 1];
            for (size_t i = 0; i < n; ++i) { u[i] = u2[i] * mul; v[i] = v2[i] * mul; }
        }
    }
    for (size_t i = 0; i < n0; ++i) { flow[2 * i] = u[i]; flow[2 * i + 1] = v[i]; }
    for (int l = 0; l < L; ++l) { free(p0[l]); free(p1[l]); }
    free(p0); free(p1); free(u); free(v); free(u2); free(v2);
    return L;
}

/* I0/I1 uint8 [H][W] (what the reference hands over with no_saliency=True, calculate_optical_flow.py:588, 631) */
ORC_API int dfo_deepflow_calc(const dfo_params* P, const uint8_t* I0u8, const uint8_t* I1u8, int H, int W, float* flow)
{
    if (!P || !I0u8 || !I1u8 || !flow || H < 1 || W < 1) return -1;
    const size_t n0 = (size_t)W * H;
    float* a = (float*)malloc(n0 * 4); float* b = (float*)malloc(n0 * 4);
    for (size_t i = 0; i < n0; ++i) { a[i] = (float)I0u8[i]; b[i] = (float)I1u8[i]; }
    const int L = deepflow_core(P, a, b, H, W, flow);
    free(a); free(b);
    return L;
}

/* I0/I1 float32 [H][W], used as they are (what reaches OF_model.calc when the frames are computeSaliency()'s CV_32F maps, :586, :631) */
ORC_API int dfo_deepflow_calc_f32(const dfo_params* P, const float* I0, const float* I1, int H, int W, float* flow)
{
    if (!P || !I0 || !I1 || !flow || H < 1 || W < 1) return -1;
    return deepflow_core(P, I0, I1, H, W, flow);
}
