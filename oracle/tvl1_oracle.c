/*
 * oracle/tvl1_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement of the dense-flow solver the reference delegates to at
 *   /root/reference/optical_flow/calculate_optical_flow.py:577-578  (createOptFlow_DualTVL1 + setLambda)
 *   /root/reference/optical_flow/calculate_optical_flow.py:642      (OF_model.calc(I0, I1, None))
 * i.e. cv2.optflow.DualTVL1OpticalFlow::calc on the CPU branch.
 *
 * The arithmetic lives in a third-party dependency that is ABSENT from /root/reference and from this
 * image: opencv-contrib-python >= 4.5.0 (requirements.txt:6-7, unpinned).  This file restates the
 * published algorithm of opencv_contrib/modules/optflow/src/tvl1flow.cpp together with the pieces of
 * opencv/modules/imgproc it calls (resize INTER_LINEAR, remap INTER_CUBIC/BORDER_CONSTANT,
 * medianBlur) from the description in SURVEY.md Appendix A and from the upstream algorithm as the
 * author knows it.  The reference holds NO tests, golden vectors or fixtures for this path and cv2
 * cannot be imported here:
 *
 *        >>>>>>>>  PARITY UNPINNED (vs real OpenCV)  <<<<<<<<
 *
 * What IS pinned: operator known-answer tests (tests/test_oracle_kat.py) and the behaviour-level
 * KATs of SURVEY.md section 8c (zero flow on identical frames, recovery of a known translation,
 * transpose/flip symmetries).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Deliberate, documented deviations from the upstream CPU code (each smaller than upstream's own
 * float-accumulation noise, each made so that results do not depend on thread count / summation order):
 *   D1. estimateU's convergence sum `error += du1^2 + du2^2` is accumulated upstream in one float,
 *       serially in raster order.  Here (err_mode 0, default) each per-pixel float term t is mapped
 *       to the integer rint(min(t,4096) * 2^30) and summed in uint64 -- exact, order independent.
 *       err_mode 1 keeps the literal float raster-order sum for comparison.
 *   D2. hypot(a,b) of two floats (upstream: libm double hypot, result cast to float) is restated as
 *       (float)sqrt((double)a*a + (double)b*b): every step IEEE-754 correctly rounded, so any
 *       conforming CPU/GPU reproduces it bit for bit; differs from libm's correctly rounded hypot
 *       only in double-rounding corner cases (~1e-9 of calls, 1 float ulp).
 *   D3. float expressions are evaluated exactly as written, left to right, WITHOUT fused
 *       multiply-add (-ffp-contract=off).  Upstream's SIMD paths may fuse a*b+c on FMA builds.
 *
 * To verify first when cv2 is ever available (tests/test_cv2_probe.py): the first-row / first-column / corner forms of
 * the divergence written here are the author's recollection of tvl1flow.cpp's divergence(); an advisor recalled a
 * version whose parallel body covers y >= 1, x >= 1 only -- upstream then fills row 0, column 0 and the corner in three
 * serial loops after the parallel_for_, which is what is restated here.
 *
 * variant 1: what a CUDA box runs for OF_algo='TVL1' in the reference --
 * cv2.cuda.OpticalFlowDual_TVL1 (calculate_optical_flow.py:572-575, 633-639; upstream
 * opencv_contrib/modules/cudaoptflow/src/tvl1flow.cpp + cuda/tvl1flow.cu, [UPSTREAM-FROM-MEMORY], lower confidence than
 * the CPU variant).  The four differences SURVEY.md Appendix A lists are restated: one loop of `iterations` =
 * inner*outer (300) per warp, no median filtering, the convergence sum evaluated only on odd iterations and only once
 * the running `prevError` has dropped below the threshold, and a warp that samples I1, I1x, I1y with a weight-normalised
 * Catmull-Rom (A = -0.5) bicubic over ceil(w-2)..floor(w+2) taps with clamp addressing instead of cv::remap.
 * Round 4: the pyramid and the flow upsampling of variant 1 sample as cuda::resize's INTER_LINEAR kernel does (orc_resize_cuda: no
 * half-pixel shift, replicate at the far edges), also [UPSTREAM-FROM-MEMORY].  Not modelled: nvcc's default FMA contraction,
 * cuda::sum's reduction order (D1's exact sum is used), the texture path cuda::resize may take for some sizes.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    double tau, lambda, theta, epsilon, scale_step, gamma;
    int nscales, warps, inner_iterations, outer_iterations, median_filtering, use_initial_flow;
    int err_mode;  /* 0 = exact fixed-point sum (D1), 1 = upstream's float raster sum */
    int variant;   /* 0 = cv2.optflow CPU DualTVL1 (the parity target), 1 = cv2.cuda.OpticalFlowDual_TVL1 semantics (row a5, below) */
} orc_params;

/* ------------------------------------------------------------------------------------------- */
/* OpenCV rounding helpers (cvFloor / cvRound / saturate_cast<short>)                           */
/* ------------------------------------------------------------------------------------------- */
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* nearest-even, as cvtss2si */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

ORC_API void orc_default_params(orc_params* p)
{
    /* cv2.optflow.createOptFlow_DualTVL1() defaults */
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->nscales = 5; p->warps = 5;
    p->epsilon = 0.01; p->inner_iterations = 30; p->outer_iterations = 10; p->scale_step = 0.8;
    p->gamma = 0.0; p->median_filtering = 5; p->use_initial_flow = 0; p->err_mode = 0; p->variant = 0;
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* cv::resize(..., INTER_LINEAR) on CV_32FC1 (imgproc/resize.cpp generic path):                  */
/*   fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx;                           */
/*   x: sx<0 -> (sx,fx)=(0,0);  sx>=W-1 -> value = S[W-1] (the dx>=xmax tail of HResizeLinear)   */
/*   y: rows clip(sy), clip(sy+1), weights (1-fy, fy) NOT zeroed at the borders                  */
/*   horizontal pass first (t = S[sx]*a0 + S[sx+1]*a1), then vertical (t0*b0 + t1*b1).           */
/* ------------------------------------------------------------------------------------------- */
ORC_API void orc_resize_linear(const float* src, int sw, int sh, float* dst, int dw, int dh,
                               double inv_scale_x, double inv_scale_y)
{
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
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
        float* D = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            float t0, t1;
            if (tail[dx]) { t0 = S0[sx]; t1 = S1[sx]; }
            else {
                const float a1 = a1s[dx], a0 = 1.f - a1;
                t0 = S0[sx] * a0 + S0[sx + 1] * a1;
                t1 = S1[sx] * a0 + S1[sx + 1] * a1;
            }
            D[dx] = t0 * b0 + t1 * b1;
        }
    }
    free(xofs); free(a1s); free(tail);
}

/* ------------------------------------------------------------------------------------------- */
/* cv::cuda::resize(..., INTER_LINEAR) on CV_32FC1 as cudawarping's resize_linear kernel samples   */
/* ([UPSTREAM-FROM-MEMORY], used by variant 1 only): NO half-pixel shift, replicate at the right /  */
/* bottom edge, four weighted taps accumulated in float in this order:                             */
/*   src_x = dx * (float)(1/fx);  x1 = floor(src_x), x2 = x1 + 1, x2r = min(x2, W-1)  (same in y)   */
/*   out = S(y1,x1)*((x2-src_x)*(y2-src_y)) + S(y1,x2r)*((src_x-x1)*(y2-src_y))                     */
/*       + S(y2r,x1)*((x2-src_x)*(src_y-y1)) + S(y2r,x2r)*((src_x-x1)*(src_y-y1))                   */
/* (nvcc would contract the multiply-adds into FMAs; not modelled, deviation D3 as everywhere.)      */
/* ------------------------------------------------------------------------------------------- */
ORC_API void orc_resize_cuda(const float* src, int sw, int sh, float* dst, int dw, int dh, double inv_scale_x, double inv_scale_y)
{
    const float scale_x = (float)(1.0 / inv_scale_x), scale_y = (float)(1.0 / inv_scale_y);
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        const float src_y = (float)dy * scale_y;
        int y1 = (int)floorf(src_y);
        if (y1 > sh - 1) y1 = sh - 1;
        const int y2 = y1 + 1, y2r = y2 < sh - 1 ? y2 : sh - 1;
        for (int dx = 0; dx < dw; ++dx) {
            const float src_x = (float)dx * scale_x;
            int x1 = (int)floorf(src_x);
            if (x1 > sw - 1) x1 = sw - 1;
            const int x2 = x1 + 1, x2r = x2 < sw - 1 ? x2 : sw - 1;
            float out = 0.f;
            out = out + src[(size_t)y1 * sw + x1] * (((float)x2 - src_x) * ((float)y2 - src_y));
            out = out + src[(size_t)y1 * sw + x2r] * ((src_x - (float)x1) * ((float)y2 - src_y));
            out = out + src[(size_t)y2r * sw + x1] * (((float)x2 - src_x) * (src_y - (float)y1));
            out = out + src[(size_t)y2r * sw + x2r] * ((src_x - (float)x1) * (src_y - (float)y1));
            dst[(size_t)dy * dw + dx] = out;
        }
    }
}

/* dsize for resize(src, Size(), f, f): saturate_cast<int>(ssize*f) == cvRound (half-to-even) */
ORC_API int orc_scaled_size(int s, double f) { return cv_round_d(s * f); }

/* ------------------------------------------------------------------------------------------- */
/* tvl1flow.cpp: centeredGradient -- 0.5*(next - prev), missing neighbour replaced by the pixel  */
/* ------------------------------------------------------------------------------------------- */
ORC_API void orc_centered_gradient(const float* src, int w, int h, float* dx, float* dy)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* cur = src + (size_t)y * w;
        const float* prev = src + (size_t)(y > 0 ? y - 1 : 0) * w;
        const float* next = src + (size_t)(y < h - 1 ? y + 1 : h - 1) * w;
        for (int x = 0; x < w; ++x) {
            const int xl = x > 0 ? x - 1 : 0, xr = x < w - 1 ? x + 1 : w - 1;
            dx[(size_t)y * w + x] = 0.5f * (cur[xr] - cur[xl]);
            dy[(size_t)y * w + x] = 0.5f * (next[x] - prev[x]);
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* imgproc: bicubic coefficient table (interpolateCubic, A=-0.75, 32 sub-pixel positions)        */
/* ------------------------------------------------------------------------------------------- */
ORC_API void orc_bicubic_tab(float* tab /* [32][4] */)
{
    const float A = -0.75f;
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; ++i) {
        float x = i * scale;
        float* c = tab + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

/* cv::remap(src, dst, mapx, mapy, INTER_CUBIC, BORDER_CONSTANT, 0) for CV_32FC1 with two CV_32FC1
 * maps: coordinates quantised to 1/32 px (cvRound(m*32)), integer part saturate_cast<short>,
 * 2-D weight = wy[k1]*wx[k2]; interior taps summed row by row, border taps strictly sequentially. */
static inline float remap_bicubic_px(const float* src, int W, int H, float mx, float my, const float* tab)
{
    const int sx = cv_round_f(mx * 32), sy = cv_round_f(my * 32);
    const float* wx = tab + (sx & 31) * 4;
    const float* wy = tab + (sy & 31) * 4;
    const int ix = sat_short(sx >> 5) - 1, iy = sat_short(sy >> 5) - 1;
    const unsigned width1 = (unsigned)(W - 3 > 0 ? W - 3 : 0), height1 = (unsigned)(H - 3 > 0 ? H - 3 : 0);
    float w[16];
    for (int k1 = 0; k1 < 4; ++k1)
        for (int k2 = 0; k2 < 4; ++k2) w[k1 * 4 + k2] = wy[k1] * wx[k2];
    if ((unsigned)ix < width1 && (unsigned)iy < height1) {
        const float* S = src + (size_t)iy * W + ix;
        float sum = S[0] * w[0] + S[1] * w[1] + S[2] * w[2] + S[3] * w[3];
        S += W;
        sum += S[0] * w[4] + S[1] * w[5] + S[2] * w[6] + S[3] * w[7];
        S += W;
        sum += S[0] * w[8] + S[1] * w[9] + S[2] * w[10] + S[3] * w[11];
        S += W;
        sum += S[0] * w[12] + S[1] * w[13] + S[2] * w[14] + S[3] * w[15];
        return sum;
    }
    if (ix >= W || ix + 4 <= 0 || iy >= H || iy + 4 <= 0) return 0.f;
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) {
        const int yi = iy + i;
        if (yi < 0 || yi >= H) continue;
        const float* S = src + (size_t)yi * W;
        for (int j = 0; j < 4; ++j) {
            const int xj = ix + j;
            if (xj >= 0 && xj < W) sum += (S[xj] - 0.f) * w[i * 4 + j];
        }
    }
    return sum;
}

/* tvl1flow.cpp procOneScale, one warp: buildFlowMap + 3x remap + calcGradRho */
ORC_API void orc_warp(const float* I0, const float* I1, const float* I1x, const float* I1y,
                      const float* u1, const float* u2, int w, int h,
                      float* I1wx, float* I1wy, float* grad, float* rho_c)
{
    float tab[128];
    orc_bicubic_tab(tab);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float mx = x + u1[i], my = y + u2[i];
            const float I1w = remap_bicubic_px(I1, w, h, mx, my, tab);
            const float wx = remap_bicubic_px(I1x, w, h, mx, my, tab);
            const float wy = remap_bicubic_px(I1y, w, h, mx, my, tab);
            const float Ix2 = wx * wx, Iy2 = wy * wy;
            I1wx[i] = wx; I1wy[i] = wy;
            grad[i] = Ix2 + Iy2;
            rho_c[i] = (I1w - wx * u1[i] - wy * u2[i] - I0[i]);
        }
}

/* cuda/tvl1flow.cu: bicubicCoeff + warpBackwardKernel (texture fetches: point sampling, clamp addressing) */
static inline float cuda_bicubic_coeff(float x_)
{
    const float x = fabsf(x_);
    if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

ORC_API void orc_warp_cuda(const float* I0, const float* I1, const float* I1x, const float* I1y,
                           const float* u1, const float* u2, int w, int h,
                           float* I1wx, float* I1wy, float* grad, float* rho_c)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float u1v = u1[i], u2v = u2[i];
            const float wx = (float)x + u1v, wy = (float)y + u2v;
            const int xmin = (int)ceilf(wx - 2.0f), xmax = (int)floorf(wx + 2.0f);
            const int ymin = (int)ceilf(wy - 2.0f), ymax = (int)floorf(wy + 2.0f);
            float sum = 0.0f, sumx = 0.0f, sumy = 0.0f, wsum = 0.0f;
            for (int cy = ymin; cy <= ymax; ++cy)
                for (int cx = xmin; cx <= xmax; ++cx) {
                    const float wt = cuda_bicubic_coeff(wx - (float)cx) * cuda_bicubic_coeff(wy - (float)cy);
                    const size_t j = (size_t)clipi(cy, 0, h) * w + clipi(cx, 0, w);
                    sum += wt * I1[j];
                    sumx += wt * I1x[j];
                    sumy += wt * I1y[j];
                    wsum += wt;
                }
            const float coeff = 1.0f / wsum;
            const float I1w = sum * coeff, gx = sumx * coeff, gy = sumy * coeff;
            const float Ix2 = gx * gx, Iy2 = gy * gy;
            I1wx[i] = gx; I1wy[i] = gy;
            grad[i] = Ix2 + Iy2;
            rho_c[i] = (I1w - gx * u1v - gy * u2v - I0[i]);
        }
}

ORC_API void orc_remap_bicubic(const float* src, int w, int h, const float* mapx, const float* mapy, float* dst)
{
    float tab[128];
    orc_bicubic_tab(tab);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            dst[i] = remap_bicubic_px(src, w, h, mapx[i], mapy[i], tab);
        }
}

/* ------------------------------------------------------------------------------------------- */
/* cv::medianBlur(ksize 3|5) on CV_32FC1: exact median of the window, BORDER_REPLICATE.          */
/* (The value is unique, so any correct selection reproduces upstream's sorting network.)        */
/* ------------------------------------------------------------------------------------------- */
ORC_API void orc_median_blur(const float* src, int w, int h, int ksize, float* dst)
{
    const int r = ksize / 2, n = ksize * ksize;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float p[25];
            int k = 0;
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = y + dy < 0 ? 0 : (y + dy > h - 1 ? h - 1 : y + dy);
                for (int dx = -r; dx <= r; ++dx) {
                    const int xx = x + dx < 0 ? 0 : (x + dx > w - 1 ? w - 1 : x + dx);
                    p[k++] = src[(size_t)yy * w + xx];
                }
            }
            for (int i = 1; i < n; ++i) {  /* insertion sort: simple and independent of any network */
                float v = p[i]; int j = i - 1;
                while (j >= 0 && p[j] > v) { p[j + 1] = p[j]; --j; }
                p[j + 1] = v;
            }
            dst[(size_t)y * w + x] = p[n / 2];
        }
}

/* ------------------------------------------------------------------------------------------- */
/* One inner iteration: estimateV + divergence + estimateU + forwardGradient + estimateDualVars  */
/* ------------------------------------------------------------------------------------------- */
#define ERR_SCALE 1073741824.0f /* 2^30 */
#define ERR_CAP 4096.0f

static inline uint64_t err_quant(float t)
{
    t = fminf(t, ERR_CAP);             /* also maps NaN -> cap */
    return (uint64_t)llrintf(t * ERR_SCALE);
}

/* returns the convergence `error` of this iteration as double (err_mode 0: q/2^30) */
static double iterate_once(const float* I1wx, const float* I1wy, const float* grad, const float* rho_c,
                           float* u1, float* u2, float* u3, float* p11, float* p12, float* p21, float* p22,
                           float* p31, float* p32, float* v1, float* v2, float* v3,
                           float* div1, float* div2, float* div3, int w, int h,
                           float l_t, float theta, float taut, float gamma, int err_mode, uint64_t* q_out)
{
    const int use_gamma = gamma != 0.f;
    /* estimateV (thresholding operator TH) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float rho = use_gamma ? rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]) + gamma * u3[i]
                                        : rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (rho < -l_t * grad[i]) {
                d1 = l_t * I1wx[i]; d2 = l_t * I1wy[i];
                if (use_gamma) d3 = l_t * gamma;
            } else if (rho > l_t * grad[i]) {
                d1 = -l_t * I1wx[i]; d2 = -l_t * I1wy[i];
                if (use_gamma) d3 = -l_t * gamma;
            } else if (grad[i] > FLT_EPSILON) {
                const float fi = -rho / grad[i];
                d1 = fi * I1wx[i]; d2 = fi * I1wy[i];
                if (use_gamma) d3 = fi * gamma;
            }
            v1[i] = u1[i] + d1; v2[i] = u2[i] + d2;
            if (use_gamma) v3[i] = u3[i] + d3;
        }
    /* divergence of (p11,p12), (p21,p22), (p31,p32): backward differences, upstream's border forms */
    for (int c = 0; c < (use_gamma ? 3 : 2); ++c) {
        const float* a = c == 0 ? p11 : (c == 1 ? p21 : p31);
        const float* b = c == 0 ? p12 : (c == 1 ? p22 : p32);
        float* d = c == 0 ? div1 : (c == 1 ? div2 : div3);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                if (y > 0 && x > 0) {
                    const float v1x = a[i] - a[i - 1];
                    const float v2y = b[i] - b[i - w];
                    d[i] = v1x + v2y;
                } else if (y == 0 && x > 0) d[i] = a[i] - a[i - 1] + b[i];
                else if (x == 0 && y > 0) d[i] = a[i] + b[i] - b[i - w];
                else d[i] = a[i] + b[i];
            }
    }
    /* estimateU */
    double error;
    uint64_t q = 0;
    if (err_mode == 1) {
        float ferr = 0.f;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                const float u1k = u1[i], u2k = u2[i], u3k = use_gamma ? u3[i] : 0;
                u1[i] = v1[i] + theta * div1[i];
                u2[i] = v2[i] + theta * div2[i];
                if (use_gamma) u3[i] = v3[i] + theta * div3[i];
                ferr += use_gamma ? (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k) + (u3[i] - u3k) * (u3[i] - u3k)
                                  : (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
            }
        error = ferr;
    } else {
#pragma omp parallel for schedule(static) reduction(+ : q)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                const float u1k = u1[i], u2k = u2[i], u3k = use_gamma ? u3[i] : 0;
                u1[i] = v1[i] + theta * div1[i];
                u2[i] = v2[i] + theta * div2[i];
                if (use_gamma) u3[i] = v3[i] + theta * div3[i];
                const float t = use_gamma ? (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k) + (u3[i] - u3k) * (u3[i] - u3k)
                                          : (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
                q += err_quant(t);
            }
        error = (double)q * (1.0 / 1073741824.0);
    }
    if (q_out) *q_out = q;
    /* forwardGradient + estimateDualVariables (fused per pixel; values identical to separate passes) */
    for (int c = 0; c < (use_gamma ? 3 : 2); ++c) {
        const float* u = c == 0 ? u1 : (c == 1 ? u2 : u3);
        float* pa = c == 0 ? p11 : (c == 1 ? p21 : p31);
        float* pb = c == 0 ? p12 : (c == 1 ? p22 : p32);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                const float ux = x < w - 1 ? u[i + 1] - u[i] : 0.0f;
                const float uy = y < h - 1 ? u[i + w] - u[i] : 0.0f;
                const float g = (float)sqrt((double)ux * (double)ux + (double)uy * (double)uy); /* D2 */
                const float ng = 1.0f + taut * g;
                pa[i] = (pa[i] + taut * ux) / ng;
                pb[i] = (pb[i] + taut * uy) / ng;
            }
    }
    return error;
}

/* Exposed for kernel-level parity tests: run `nsteps` inner iterations (no median, no stop test) on
 * caller state; err_q[k] receives the exact error sum (2^-30 units) of step k. gamma == 0 only. */
ORC_API void orc_iterate(const float* I1wx, const float* I1wy, const float* grad, const float* rho_c,
                         float* u1, float* u2, float* p11, float* p12, float* p21, float* p22,
                         int w, int h, double lambda, double theta, double tau, int nsteps, uint64_t* err_q)
{
    const size_t n = (size_t)w * h;
    float* tmp = (float*)malloc(sizeof(float) * n * 4);
    const float l_t = (float)(lambda * theta), taut = (float)(tau / theta);
    for (int k = 0; k < nsteps; ++k)
        iterate_once(I1wx, I1wy, grad, rho_c, u1, u2, NULL, p11, p12, p21, p22, NULL, NULL,
                     tmp, tmp + n, NULL, tmp + 2 * n, tmp + 3 * n, NULL, w, h, l_t, (float)theta, taut, 0.f, 0,
                     err_q ? err_q + k : NULL);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------- */
/* procOneScale + calc                                                                           */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    float *I1x, *I1y, *I1wx, *I1wy, *grad, *rho_c, *v1, *v2, *v3, *div1, *div2, *div3;
    float *p11, *p12, *p21, *p22, *p31, *p32, *med;
} orc_bufs;

static void proc_one_scale(const orc_params* P, const float* I0, const float* I1, float* u1, float* u2, float* u3,
                           int w, int h, orc_bufs* B, int* n_inner_out, int* n_outer_out)
{
    const size_t n = (size_t)w * h;
    const float scaledEpsilon = (float)(P->epsilon * P->epsilon * (double)(w * h));
    const int use_gamma = P->gamma != 0.;
    orc_centered_gradient(I1, w, h, B->I1x, B->I1y);
    memset(B->p11, 0, n * 4); memset(B->p12, 0, n * 4); memset(B->p21, 0, n * 4); memset(B->p22, 0, n * 4);
    if (use_gamma) { memset(B->p31, 0, n * 4); memset(B->p32, 0, n * 4); }
    const float l_t = (float)(P->lambda * P->theta);
    const float taut = (float)(P->tau / P->theta);
    for (int wi = 0; wi < P->warps && P->variant == 1; ++wi) {
        /* cudaoptflow tvl1flow.cpp procOneScale: one loop, error only on odd iterations once prevError < threshold */
        orc_warp_cuda(I0, I1, B->I1x, B->I1y, u1, u2, w, h, B->I1wx, B->I1wy, B->grad, B->rho_c);
        const int iterations = P->inner_iterations * P->outer_iterations;
        /* scaledEpsilon, error and prevError are doubles in the CUDA class (the CPU class keeps a float scaledEpsilon) */
        const double scaledEps = P->epsilon * P->epsilon * (double)(w * h);
        double error = DBL_MAX, prevError = 0.0;
        int n = 0;
        for (; error > scaledEps && n < iterations; ++n) {
            const int calcError = P->epsilon > 0 && (n & 1) && prevError < scaledEps;
            const double e = iterate_once(B->I1wx, B->I1wy, B->grad, B->rho_c, u1, u2, u3, B->p11, B->p12, B->p21, B->p22,
                                          B->p31, B->p32, B->v1, B->v2, B->v3, B->div1, B->div2, B->div3, w, h,
                                          l_t, (float)P->theta, taut, (float)P->gamma, P->err_mode, NULL);
            if (calcError) { error = e; prevError = error; }
            else { error = DBL_MAX; prevError -= scaledEps; }
        }
        if (n_inner_out) n_inner_out[wi] = n;
        if (n_outer_out) n_outer_out[wi] = 0;
    }
    for (int wi = 0; wi < P->warps && P->variant != 1; ++wi) {
        orc_warp(I0, I1, B->I1x, B->I1y, u1, u2, w, h, B->I1wx, B->I1wy, B->grad, B->rho_c);
        double error = (double)FLT_MAX;
        int n_in = 0, n_out = 0;
        for (int no = 0; error > (double)scaledEpsilon && no < P->outer_iterations; ++no) {
            if (P->median_filtering > 1) {
                orc_median_blur(u1, w, h, P->median_filtering, B->med); memcpy(u1, B->med, n * 4);
                orc_median_blur(u2, w, h, P->median_filtering, B->med); memcpy(u2, B->med, n * 4);
            }
            ++n_out;
            for (int ni = 0; error > (double)scaledEpsilon && ni < P->inner_iterations; ++ni) {
                error = iterate_once(B->I1wx, B->I1wy, B->grad, B->rho_c, u1, u2, u3, B->p11, B->p12, B->p21, B->p22,
                                     B->p31, B->p32, B->v1, B->v2, B->v3, B->div1, B->div2, B->div3, w, h,
                                     l_t, (float)P->theta, taut, (float)P->gamma, P->err_mode, NULL);
                ++n_in;
            }
        }
        if (n_inner_out) n_inner_out[wi] = n_in;
        if (n_outer_out) n_outer_out[wi] = n_out;
    }
}

/* Entry point.  I0/I1: uint8 [H][W]; flow: float32 [H][W][2] (x then y displacement, pixels/frame).
 * iters (optional): int32 [nscales][warps][2] = (inner iterations, outer iterations) executed, level 0 first.
 * returns the number of pyramid levels actually used (>0), or a negative error code. */
static int tvl1_core(const orc_params* P, const uint8_t* I0u8, const uint8_t* I1u8, const float* I0f, const float* I1f,
                     int H, int W, float* flow, int* iters)
{
    if (!P || !(I0u8 || I0f) || !(I1u8 || I1f) || !flow || H <= 0 || W <= 0) return -1;
    if (P->nscales < 1 || P->nscales > 64 || P->warps < 0 || P->use_initial_flow) return -2;
    if (P->median_filtering != 1 && P->median_filtering != 3 && P->median_filtering != 5) return -3;
    int nscales = P->nscales;
    const int use_gamma = P->gamma != 0.;
    int ws[64], hs[64];
    float *I0s[64], *I1s[64], *u1s[64], *u2s[64], *u3s[64];
    memset(I0s, 0, sizeof I0s); memset(I1s, 0, sizeof I1s); memset(u1s, 0, sizeof u1s);
    memset(u2s, 0, sizeof u2s); memset(u3s, 0, sizeof u3s);
    const size_t n0 = (size_t)W * H;
    ws[0] = W; hs[0] = H;
    I0s[0] = (float*)malloc(n0 * 4); I1s[0] = (float*)malloc(n0 * 4);
    if (I0u8) for (size_t i = 0; i < n0; ++i) { I0s[0][i] = (float)I0u8[i]; I1s[0][i] = (float)I1u8[i]; } /* convertTo(.., 1.0) */
    /* CV_32F input: convertTo(.., 255.0) = src * 255 + 0 in fp32 (an fma with a zero addend rounds like the multiply) */
    else for (size_t i = 0; i < n0; ++i) { I0s[0][i] = I0f[i] * 255.0f; I1s[0][i] = I1f[i] * 255.0f; }
    u1s[0] = (float*)malloc(n0 * 4); u2s[0] = (float*)malloc(n0 * 4);
    if (use_gamma) u3s[0] = (float*)malloc(n0 * 4);
    for (int s = 1; s < nscales; ++s) {
        ws[s] = orc_scaled_size(ws[s - 1], P->scale_step);
        hs[s] = orc_scaled_size(hs[s - 1], P->scale_step);
        if (ws[s] < 1 || hs[s] < 1) { nscales = s; break; }
        const size_t n = (size_t)ws[s] * hs[s];
        I0s[s] = (float*)malloc(n * 4); I1s[s] = (float*)malloc(n * 4);
        /* variant 1: the CUDA class builds its pyramid with cuda::resize(.., Size(), scaleStep, scaleStep) */
        (P->variant == 1 ? orc_resize_cuda : orc_resize_linear)(I0s[s - 1], ws[s - 1], hs[s - 1], I0s[s], ws[s], hs[s], P->scale_step, P->scale_step);
        (P->variant == 1 ? orc_resize_cuda : orc_resize_linear)(I1s[s - 1], ws[s - 1], hs[s - 1], I1s[s], ws[s], hs[s], P->scale_step, P->scale_step);
        if (ws[s] < 16 || hs[s] < 16) { nscales = s; break; }
        u1s[s] = (float*)malloc(n * 4); u2s[s] = (float*)malloc(n * 4);
        if (use_gamma) u3s[s] = (float*)malloc(n * 4);
    }
    {
        const size_t n = (size_t)ws[nscales - 1] * hs[nscales - 1];
        memset(u1s[nscales - 1], 0, n * 4); memset(u2s[nscales - 1], 0, n * 4);
        if (use_gamma) memset(u3s[nscales - 1], 0, n * 4);
    }
    orc_bufs B;
    float** bp = (float**)&B;
    for (size_t k = 0; k < sizeof(B) / sizeof(float*); ++k) bp[k] = (float*)malloc(n0 * 4);
    if (iters) memset(iters, 0, sizeof(int) * (size_t)P->nscales * (size_t)P->warps * 2);
    int* n_in = (int*)calloc((size_t)(P->warps > 0 ? P->warps : 1), sizeof(int));
    int* n_out = (int*)calloc((size_t)(P->warps > 0 ? P->warps : 1), sizeof(int));
    for (int s = nscales - 1; s >= 0; --s) {
        proc_one_scale(P, I0s[s], I1s[s], u1s[s], u2s[s], u3s[s], ws[s], hs[s], &B, n_in, n_out);
        if (iters)
            for (int wi = 0; wi < P->warps; ++wi) {
                iters[((size_t)s * P->warps + wi) * 2 + 0] = n_in[wi];
                iters[((size_t)s * P->warps + wi) * 2 + 1] = n_out[wi];
            }
        if (s == 0) break;
        const double isx = (double)ws[s - 1] / ws[s], isy = (double)hs[s - 1] / hs[s];
        /* variant 1: cuda::resize(u, u_finer, size(finer level)) -- fx = dsize / ssize, then the same sampling rule */
        (P->variant == 1 ? orc_resize_cuda : orc_resize_linear)(u1s[s], ws[s], hs[s], u1s[s - 1], ws[s - 1], hs[s - 1], isx, isy);
        (P->variant == 1 ? orc_resize_cuda : orc_resize_linear)(u2s[s], ws[s], hs[s], u2s[s - 1], ws[s - 1], hs[s - 1], isx, isy);
        if (use_gamma) orc_resize_linear(u3s[s], ws[s], hs[s], u3s[s - 1], ws[s - 1], hs[s - 1], isx, isy);
        const float mul = (float)(1 / P->scale_step);       /* multiply(u, Scalar::all(1/scaleStep), u) */
        const size_t n = (size_t)ws[s - 1] * hs[s - 1];
        for (size_t i = 0; i < n; ++i) { u1s[s - 1][i] *= mul; u2s[s - 1][i] *= mul; }
    }
    for (size_t i = 0; i < n0; ++i) { flow[2 * i] = u1s[0][i]; flow[2 * i + 1] = u2s[0][i]; }  /* merge */
    for (size_t k = 0; k < sizeof(B) / sizeof(float*); ++k) free(bp[k]);
    for (int s = 0; s < 64; ++s) { free(I0s[s]); free(I1s[s]); free(u1s[s]); free(u2s[s]); free(u3s[s]); }
    free(n_in); free(n_out);
    return nscales;
}

ORC_API int orc_tvl1_calc(const orc_params* P, const uint8_t* I0u8, const uint8_t* I1u8, int H, int W,
                          float* flow, int* iters)
{
    return tvl1_core(P, I0u8, I1u8, NULL, NULL, H, W, flow, iters);
}

/* Same for CV_32FC1 frames (cv2 accepts them with values in [0,1]; the reference itself always hands over uint8,
 * calculate_optical_flow.py:588). */
ORC_API int orc_tvl1_calc_f32(const orc_params* P, const float* I0, const float* I1, int H, int W, float* flow, int* iters)
{
    return tvl1_core(P, NULL, NULL, I0, I1, H, W, flow, iters);
}

/* Pyramid only (for kernel-level parity tests): writes level s (>=1) of the x0.8 pyramid of a u8 image */
ORC_API int orc_pyramid_level(const uint8_t* img, int H, int W, double scale_step, int level, float* out, int* ow, int* oh)
{
    int w = W, h = H;
    float* cur = (float*)malloc((size_t)W * H * 4);
    for (size_t i = 0; i < (size_t)W * H; ++i) cur[i] = (float)img[i];
    for (int s = 1; s <= level; ++s) {
        const int nw = orc_scaled_size(w, scale_step), nh = orc_scaled_size(h, scale_step);
        float* nxt = (float*)malloc((size_t)nw * nh * 4);
        orc_resize_linear(cur, w, h, nxt, nw, nh, scale_step, scale_step);
        free(cur); cur = nxt; w = nw; h = nh;
    }
    if (out) memcpy(out, cur, (size_t)w * h * 4);
    free(cur);
    *ow = w; *oh = h;
    return 0;
}
