// In-context cost of the pieces of the tvl1_iter arithmetic: the band_probe loop (two fused iterations, no LDS, no
// barriers, 3 waves per SIMD) with one piece at a time replaced by a cheap stand-in or by a candidate exact form.
//   HYP 0 = shipped exact hypot (f64 Goldschmidt)          1 = v_sqrt_f32(fma(a,a,b*b))      (inexact: cost share only)
//       2 = f32 rsq seed + ONE f64 correction + tie-zone test, exact fallback  (candidate, same bits)
//       3 = f64 rsq seed + one Goldschmidt step, no d-corrections + tie-zone test, exact fallback (candidate, same bits)
//   DIV 0 = shipped exact division    1 = a * rcp(b)       (inexact: cost share only)
//   ERR 0 = shipped exact error term  1 = none
// Also checks candidates 2 and 3 against the shipped hypot on random operands (bit compare) and counts how often the
// tie-zone test sends a wave to the fallback.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize ablate_probe.hip -o ablate_probe
#include "../../tee_optical_flow_amd/csrc/teeflow_kernels.hip.h"
#include <cstdio>
#include <vector>

// ---- hypot candidates ------------------------------------------------------------------------------------------------
// g within 2^-42.8 (relative) of sqrt(a^2+b^2); (float)g equals the oracle's (float)sqrt((double)a*a+(double)b*b) unless g
// lies within 2^13 double-ulps of a float rounding boundary -> `zone` is set and the caller recomputes with the exact form.
__device__ __forceinline__ float hypot_seed32(float a, float b, unsigned& zone)
{
    const float xs = fmaxf(__builtin_fmaf(a, a, b * b), 0x1p-100f);
    const float rs = __builtin_amdgcn_rsqf(xs);
    const float y = xs * rs, h = 0.5f * rs;
    const double ad = (double)a, bd = (double)b;
    const double x = __builtin_fma(ad, ad, bd * bd);
    const double yd = (double)y, hd = (double)h;
    const double r = __builtin_fma(-yd, yd, x);
    const double g = __builtin_fma(r, hd, yd);
    const unsigned lo = (unsigned)__double_as_longlong(g);
    zone |= ((lo - (0x10000000u - 0x2000u)) & 0x1FFFFFFFu) < 0x4000u ? 1u : 0u;
    return (float)g;
}
__device__ __forceinline__ float hypot_short64(float a, float b, unsigned& zone)
{
    const double ad = (double)a, bd = (double)b;
    const double x = __builtin_fmax(__builtin_fma(ad, ad, bd * bd), 0x1p-400);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    const unsigned lo = (unsigned)__double_as_longlong(g);
    zone |= ((lo - (0x10000000u - 0x20000u)) & 0x1FFFFFFFu) < 0x40000u ? 1u : 0u;
    return (float)g;
}

template <int HYP>
__device__ __forceinline__ void hyp4(f2 u1x, f2 u1y, f2 u2x, f2 u2y, f2& g1, f2& g2)
{
    if (HYP == 0) {
        g1 = mk2(hypot_exact2(u1x.x, u1y.x), hypot_exact2(u1x.y, u1y.y));
        g2 = mk2(hypot_exact2(u2x.x, u2y.x), hypot_exact2(u2x.y, u2y.y));
    } else if (HYP == 1) {
        g1 = mk2(__builtin_amdgcn_sqrtf(__builtin_fmaf(u1x.x, u1x.x, u1y.x * u1y.x)), __builtin_amdgcn_sqrtf(__builtin_fmaf(u1x.y, u1x.y, u1y.y * u1y.y)));
        g2 = mk2(__builtin_amdgcn_sqrtf(__builtin_fmaf(u2x.x, u2x.x, u2y.x * u2y.x)), __builtin_amdgcn_sqrtf(__builtin_fmaf(u2x.y, u2x.y, u2y.y * u2y.y)));
    } else {
        unsigned zone = 0;
        if (HYP == 2) {
            g1 = mk2(hypot_seed32(u1x.x, u1y.x, zone), hypot_seed32(u1x.y, u1y.y, zone));
            g2 = mk2(hypot_seed32(u2x.x, u2y.x, zone), hypot_seed32(u2x.y, u2y.y, zone));
        } else {
            g1 = mk2(hypot_short64(u1x.x, u1y.x, zone), hypot_short64(u1x.y, u1y.y, zone));
            g2 = mk2(hypot_short64(u2x.x, u2y.x, zone), hypot_short64(u2x.y, u2y.y, zone));
        }
        if (__builtin_expect(__any((int)zone), 0)) {
            g1 = mk2(hypot_exact2(u1x.x, u1y.x), hypot_exact2(u1x.y, u1y.y));
            g2 = mk2(hypot_exact2(u2x.x, u2y.x), hypot_exact2(u2x.y, u2y.y));
        }
    }
}

template <int DIV>
__device__ __forceinline__ f2 divv(f2 a, f2 b, f2 bs, f2 rs)
{
    if (DIV == 0) return div2s(a, b, bs, rs);
    return a * mk2(__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y));
}

template <int HYP, int DIV>
__device__ __forceinline__ void p_pair_v(float taut, f2 u1x, f2 u1y, f2 u2x, f2 u2y, f2 p11, f2 p12, f2 p21, f2 p22,
                                         f2& o11, f2& o12, f2& o21, f2& o22)
{
    f2 g1, g2;
    hyp4<HYP>(u1x, u1y, u2x, u2y, g1, g2);
    const f2 ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
    if (DIV == 0) {
        const f2 ns1 = ng1 * 0x1p64f, ns2 = ng2 * 0x1p64f;
        const f2 r1 = rcp2s(ns1), r2 = rcp2s(ns2);
        o11 = div2s(p11 + taut * u1x, ng1, ns1, r1); o12 = div2s(p12 + taut * u1y, ng1, ns1, r1);
        o21 = div2s(p21 + taut * u2x, ng2, ns2, r2); o22 = div2s(p22 + taut * u2y, ng2, ns2, r2);
    } else {
        const f2 r1 = mk2(__builtin_amdgcn_rcpf(ng1.x), __builtin_amdgcn_rcpf(ng1.y)), r2 = mk2(__builtin_amdgcn_rcpf(ng2.x), __builtin_amdgcn_rcpf(ng2.y));
        o11 = (p11 + taut * u1x) * r1; o12 = (p12 + taut * u1y) * r1;
        o21 = (p21 + taut * u2x) * r2; o22 = (p22 + taut * u2y) * r2;
    }
}

template <int HYP, int DIV>
__device__ __forceinline__ void p_quad_v(float taut, const float* u1x, const float* u1y, const float* u2x, const float* u2y,
                                         const float* p11, const float* p12, const float* p21, const float* p22,
                                         float* o11, float* o12, float* o21, float* o22)
{
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        f2 a, b, c, d;
        p_pair_v<HYP, DIV>(taut, mk2(u1x[h], u1x[h + 1]), mk2(u1y[h], u1y[h + 1]), mk2(u2x[h], u2x[h + 1]), mk2(u2y[h], u2y[h + 1]),
                           mk2(p11[h], p11[h + 1]), mk2(p12[h], p12[h + 1]), mk2(p21[h], p21[h + 1]), mk2(p22[h], p22[h + 1]), a, b, c, d);
        o11[h] = a.x; o11[h + 1] = a.y; o12[h] = b.x; o12[h + 1] = b.y;
        o21[h] = c.x; o21[h + 1] = c.y; o22[h] = d.x; o22[h + 1] = d.y;
    }
}

template <int DIV>
__device__ __forceinline__ void u_pair_v(float l_t, float theta, f2 u1k, f2 u2k, f2 wx, f2 wy, f2 rc, f2 p11, f2 p12, f2 p21,
                                         f2 p22, f2 p12u, f2 p22u, float l11, float l21, bool ytop, bool x0, f2& u1n, f2& u2n)
{
    const f2 Ix2 = wx * wx, Iy2 = wy * wy;
    const f2 grad = Ix2 + Iy2;
    const f2 rho = rc + (wx * u1k + wy * u2k);
    const f2 lg = l_t * grad;
    f2 fi;
    if (DIV == 0) { const f2 grads = grad * 0x1p64f; fi = div2s(-rho, grad, grads, rcp2s(grads)); }
    else fi = -rho * mk2(__builtin_amdgcn_rcpf(grad.x), __builtin_amdgcn_rcpf(grad.y));
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

template <int DIV>
__device__ __forceinline__ void u_quad_v(float l_t, float theta, const QuadU& q, bool ytop, bool x0, float* u1n, float* u2n)
{
    f2 a1, a2, b1, b2;
    u_pair_v<DIV>(l_t, theta, mk2(q.u1k[0], q.u1k[1]), mk2(q.u2k[0], q.u2k[1]), mk2(q.wx[0], q.wx[1]), mk2(q.wy[0], q.wy[1]),
                  mk2(q.r[0], q.r[1]), mk2(q.p11[0], q.p11[1]), mk2(q.p12[0], q.p12[1]), mk2(q.p21[0], q.p21[1]),
                  mk2(q.p22[0], q.p22[1]), mk2(q.p12u[0], q.p12u[1]), mk2(q.p22u[0], q.p22u[1]), q.l11, q.l21, ytop, x0, a1, a2);
    u_pair_v<DIV>(l_t, theta, mk2(q.u1k[2], q.u1k[3]), mk2(q.u2k[2], q.u2k[3]), mk2(q.wx[2], q.wx[3]), mk2(q.wy[2], q.wy[3]),
                  mk2(q.r[2], q.r[3]), mk2(q.p11[2], q.p11[3]), mk2(q.p12[2], q.p12[3]), mk2(q.p21[2], q.p21[3]),
                  mk2(q.p22[2], q.p22[3]), mk2(q.p12u[2], q.p12u[3]), mk2(q.p22u[2], q.p22u[3]), q.p11[1], q.p21[1], ytop, false, b1, b2);
    u1n[0] = a1.x; u1n[1] = a1.y; u1n[2] = b1.x; u1n[3] = b1.y;
    u2n[0] = a2.x; u2n[1] = a2.y; u2n[2] = b2.x; u2n[3] = b2.y;
}

template <int ERR>
__device__ __forceinline__ double err_v(const float* u1n, const float* u1k, const float* u2n, const float* u2k, const unsigned* keep)
{
    if (ERR == 0) return tv_err_quad_pk(u1n, u1k, u2n, u2k, keep);
    return 0.0;
}

// LAYOUT 0: 15 separate planes (what the engine has).  LAYOUT 1: the planes of a group share a row -- constants {wx, wy, rho},
// flow {u1, u2}, dual {p11..p22} on the read side, flow / dual on the write side: 5 streams instead of 15.  LAYOUT 2: all 9 input
// planes of a row contiguous, all 6 output planes contiguous: 2 streams.  Same bytes, same arithmetic.
template <int LAYOUT>
__device__ __forceinline__ size_t in_off(int k, size_t rowidx, int pitch, size_t plane, int x)
{
    // k: 0 u1, 1 u2, 2 wx, 3 wy, 4 rho, 5..8 p
    if (LAYOUT == 0) return (size_t)k * plane + rowidx * pitch + x;
    if (LAYOUT == 2) return (rowidx * 9 + k) * pitch + x;
    if (k < 2) return (rowidx * 2 + k) * pitch + x;                                   // flow group at plane 0..1
    if (k < 5) return 2 * plane + (rowidx * 3 + (k - 2)) * pitch + x;                 // constants at plane 2..4
    return 5 * plane + (rowidx * 4 + (k - 5)) * pitch + x;                            // dual at plane 5..8
}
template <int LAYOUT>
__device__ __forceinline__ size_t out_off(int k, size_t rowidx, int pitch, size_t plane, int x)
{
    // k: 0 u1, 1 u2, 2..5 p
    if (LAYOUT == 0) return (size_t)k * plane + rowidx * pitch + x;
    if (LAYOUT == 2) return (rowidx * 6 + k) * pitch + x;
    if (k < 2) return (rowidx * 2 + k) * pitch + x;
    return 2 * plane + (rowidx * 4 + (k - 2)) * pitch + x;
}

template <int HYP, int DIV, int ERR, int WAVES, int LAYOUT = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_ab(const float* __restrict__ in, float* __restrict__ out, int rows, int pitch, size_t plane, float l_t, float theta, float taut,
          unsigned long long* __restrict__ err, int wrap)
{
    const int lane = threadIdx.x, x = lane * 4;
    const size_t base = (size_t)blockIdx.x * rows;
    float pu1[4] = {0, 0, 0, 0}, pu2[4] = {0, 0, 0, 0}, pp12[4] = {0, 0, 0, 0}, pp22[4] = {0, 0, 0, 0};
    double accA = 0.0, accB = 0.0;
    unsigned inw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) inw[j] = opaque_u(~0u);
    const unsigned keep[4] = {inw[0], inw[1], inw[2], inw[3]};
    for (int r = 0; r < rows; ++r) {
        const size_t ri = base + (size_t)(r & wrap);
        QuadU qu;
        float4 v;
        v = ld4(in + in_off<LAYOUT>(0, ri, pitch, plane, x)); UNPACK4(qu.u1k, v)
        v = ld4(in + in_off<LAYOUT>(1, ri, pitch, plane, x)); UNPACK4(qu.u2k, v)
        v = ld4(in + in_off<LAYOUT>(2, ri, pitch, plane, x)); UNPACK4(qu.wx, v)
        v = ld4(in + in_off<LAYOUT>(3, ri, pitch, plane, x)); UNPACK4(qu.wy, v)
        v = ld4(in + in_off<LAYOUT>(4, ri, pitch, plane, x)); UNPACK4(qu.r, v)
        v = ld4(in + in_off<LAYOUT>(5, ri, pitch, plane, x)); UNPACK4(qu.p11, v)
        v = ld4(in + in_off<LAYOUT>(6, ri, pitch, plane, x)); UNPACK4(qu.p12, v)
        v = ld4(in + in_off<LAYOUT>(7, ri, pitch, plane, x)); UNPACK4(qu.p21, v)
        v = ld4(in + in_off<LAYOUT>(8, ri, pitch, plane, x)); UNPACK4(qu.p22, v)
#pragma unroll
        for (int i = 0; i < 4; ++i) { qu.p12u[i] = pp12[i]; qu.p22u[i] = pp22[i]; }
        qu.l11 = __shfl_up(qu.p11[3], 1, 64); qu.l21 = __shfl_up(qu.p21[3], 1, 64);
        float u1a[4], u2a[4], u1b[4], u2b[4], q11[4], q12[4], q21[4], q22[4], s11[4], s12[4], s21[4], s22[4];
        u_quad_v<DIV>(l_t, theta, qu, r == 0, lane == 0, u1a, u2a);
        accA += err_v<ERR>(u1a, qu.u1k, u2a, qu.u2k, keep);
        float ux1[4], uy1[4], ux2[4], uy2[4];
        const float rr1 = __shfl_down(u1a[0], 1, 64), rr2 = __shfl_down(u2a[0], 1, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u1a[i + 1] : rr1, e2 = i < 3 ? u2a[i + 1] : rr2;
            ux1[i] = mask_f(e1 - u1a[i], inw[i + 1]); ux2[i] = mask_f(e2 - u2a[i], inw[i + 1]);
            uy1[i] = mask_f(u1a[i] - pu1[i], inw[0]); uy2[i] = mask_f(u2a[i] - pu2[i], inw[0]);
        }
        p_quad_v<HYP, DIV>(taut, ux1, uy1, ux2, uy2, qu.p11, qu.p12, qu.p21, qu.p22, q11, q12, q21, q22);
        QuadU q2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q2.u1k[i] = u1a[i]; q2.u2k[i] = u2a[i]; q2.wx[i] = qu.wx[i]; q2.wy[i] = qu.wy[i]; q2.r[i] = qu.r[i];
            q2.p11[i] = q11[i]; q2.p12[i] = q12[i]; q2.p21[i] = q21[i]; q2.p22[i] = q22[i]; q2.p12u[i] = pp12[i]; q2.p22u[i] = pp22[i];
        }
        q2.l11 = __shfl_up(q11[3], 1, 64); q2.l21 = __shfl_up(q21[3], 1, 64);
        u_quad_v<DIV>(l_t, theta, q2, r == 0, lane == 0, u1b, u2b);
        accB += err_v<ERR>(u1b, u1a, u2b, u2a, keep);
        const float t1 = __shfl_down(u1b[0], 1, 64), t2 = __shfl_down(u2b[0], 1, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? u1b[i + 1] : t1, e2 = i < 3 ? u2b[i + 1] : t2;
            ux1[i] = mask_f(e1 - u1b[i], inw[i + 1]); ux2[i] = mask_f(e2 - u2b[i], inw[i + 1]);
            uy1[i] = mask_f(u1b[i] - pu1[i], inw[0]); uy2[i] = mask_f(u2b[i] - pu2[i], inw[0]);
        }
        p_quad_v<HYP, DIV>(taut, ux1, uy1, ux2, uy2, q11, q12, q21, q22, s11, s12, s21, s22);
        st4(out + out_off<LAYOUT>(0, ri, pitch, plane, x), PACK4(u1b)); st4(out + out_off<LAYOUT>(1, ri, pitch, plane, x), PACK4(u2b));
        st4(out + out_off<LAYOUT>(2, ri, pitch, plane, x), PACK4(s11)); st4(out + out_off<LAYOUT>(3, ri, pitch, plane, x), PACK4(s12));
        st4(out + out_off<LAYOUT>(4, ri, pitch, plane, x), PACK4(s21)); st4(out + out_off<LAYOUT>(5, ri, pitch, plane, x), PACK4(s22));
#pragma unroll
        for (int i = 0; i < 4; ++i) { pu1[i] = u1b[i]; pu2[i] = u2b[i]; pp12[i] = s12[i]; pp22[i] = s22[i]; }
    }
    const unsigned long long a = (unsigned long long)accA + (unsigned long long)accB;
    if (a == 0x123456789ull) err[0] = a;
}

__global__ void k_fill(float* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761ull) % 2001ull) * 1e-3f - 1.0f;
}

// ---- exactness of the candidates: random operands over the magnitudes a flow gradient takes (and far beyond) -----------
__device__ __forceinline__ unsigned rng32(unsigned long long& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(s >> 32);
}
__global__ void k_check(unsigned long long* out /* [6]: n, zone2, bad2, zone3, bad3, wavezone2 */, int per_thread, int mode)
{
    unsigned long long s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1) + mode;
    unsigned long long n = 0, z2 = 0, b2 = 0, z3 = 0, b3 = 0, wz2 = 0;
    for (int i = 0; i < per_thread; ++i) {
        float a, b;
        if (mode == 0) {            // uniform mantissas, exponents spread over 2^-40 .. 2^8, random signs, some zeros
            const unsigned ra = rng32(s), rb = rng32(s), re = rng32(s);
            const int ea = 127 - 40 + (int)(re % 49), eb = 127 - 40 + (int)((re >> 8) % 49);
            a = __uint_as_float((ra & 0x807FFFFFu) | ((unsigned)ea << 23));
            b = __uint_as_float((rb & 0x807FFFFFu) | ((unsigned)eb << 23));
            if ((re >> 20) % 37 == 0) b = 0.f;
            if ((re >> 26) % 41 == 0) a = 0.f;
        } else if (mode == 1) {     // similar magnitudes (the hard case for cancellation-free sums), typical gradient sizes
            const unsigned ra = rng32(s), rb = rng32(s), re = rng32(s);
            const int e = 127 - 12 + (int)(re % 14);
            a = __uint_as_float((ra & 0x807FFFFFu) | ((unsigned)e << 23));
            b = __uint_as_float((rb & 0x807FFFFFu) | ((unsigned)(e - (int)((re >> 8) % 3)) << 23));
        } else {                    // small integers / dyadic values: exact squares and exact ties do occur here
            const unsigned ra = rng32(s), rb = rng32(s);
            a = (float)(int)(ra % 4096) * 0x1p-6f;
            b = (float)(int)(rb % 4096) * 0x1p-6f;
        }
        const float ref = hypot_exact2(a, b);
        unsigned zn2 = 0, zn3 = 0;
        const float c2 = hypot_seed32(a, b, zn2), c3 = hypot_short64(a, b, zn3);
        // what matters downstream is ng = 1 + taut*g; a result below 2^-60 cannot change it, so such differences do not count
        const bool d2 = !zn2 && c2 != ref && !(ref < 0x1p-60f && c2 < 0x1p-60f);
        const bool d3 = !zn3 && c3 != ref && !(ref < 0x1p-60f && c3 < 0x1p-60f);
        ++n; z2 += zn2; z3 += zn3; b2 += d2; b3 += d3;
        wz2 += (__any((int)zn2) && (threadIdx.x & 63) == 0) ? 1 : 0;
    }
    atomicAdd(&out[0], n); atomicAdd(&out[1], z2); atomicAdd(&out[2], b2); atomicAdd(&out[3], z3); atomicAdd(&out[4], b3); atomicAdd(&out[5], wz2);
}

template <int HYP, int DIV, int ERR, int WAVES, int LAYOUT = 0>
static double run(const char* name, const float* in, float* out, int rows, int pitch, size_t plane, unsigned long long* err, int cus, int wrap)
{
    const int waves_total = cus * 4 * WAVES;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_ab<HYP, DIV, ERR, WAVES, LAYOUT>), dim3(waves_total), dim3(64), 0, 0, in, out, 4, pitch, plane, 0.045f, 0.3f, 0.8333f, err, wrap);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_ab<HYP, DIV, ERR, WAVES, LAYOUT>), dim3(waves_total), dim3(64), 0, 0, in, out, rows, pitch, plane, 0.045f, 0.3f, 0.8333f, err, wrap);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double px_it = (double)waves_total * rows * 256 * 2;
    const double cyc = best * 1e-3 * 2.4e9 / rows / WAVES;      // SIMD cycles (2.4 GHz nominal) per wave-row of 512 px-iterations
    printf("%-46s w/SIMD %d %s: %7.3f ms -> %6.1f Gpx-it/s, %6.0f SIMD-cycles per wave-row, %5.2f per px-it\n", name, WAVES,
           wrap == 7 ? "cached" : "stream", best, px_it / best * 1e-6, cyc, cyc / 512);
    return cyc;
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, pitch = 256, rows = 512;
    const size_t waves_max = (size_t)cus * 4 * 3, plane = waves_max * rows * pitch;
    float *in, *out; unsigned long long* err;
    (void)hipMalloc(&in, plane * 9 * sizeof(float)); (void)hipMalloc(&out, plane * 6 * sizeof(float)); (void)hipMalloc(&err, 64);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, plane * 9);
    (void)hipDeviceSynchronize();
    printf("device %s, %d CUs, clock %d kHz\n", pr.name, cus, pr.clockRate);
    for (int wrap : {0x7fffffff, 7}) {
        const double base = run<0, 0, 0, 3>("shipped arithmetic", in, out, rows, pitch, plane, err, cus, wrap);
        const double nh = run<1, 0, 0, 3>("hypot -> v_sqrt_f32 (inexact)", in, out, rows, pitch, plane, err, cus, wrap);
        const double nd = run<0, 1, 0, 3>("divisions -> a*rcp(b) (inexact)", in, out, rows, pitch, plane, err, cus, wrap);
        const double ne = run<0, 0, 1, 3>("no error term", in, out, rows, pitch, plane, err, cus, wrap);
        const double fl = run<1, 1, 1, 3>("all three cheap (floor of this structure)", in, out, rows, pitch, plane, err, cus, wrap);
        const double c2 = run<2, 0, 0, 3>("CANDIDATE hypot: f32 seed + 1 f64 step + zone", in, out, rows, pitch, plane, err, cus, wrap);
        const double c3 = run<3, 0, 0, 3>("CANDIDATE hypot: short f64 + zone", in, out, rows, pitch, plane, err, cus, wrap);
        printf("  in-context cost per px-iteration (SIMD cycles): 2 hypots %.1f, 5 divisions %.1f, error term %.1f, rest %.1f; candidates save %.1f / %.1f\n",
               (base - nh) / 512, (base - nd) / 512, (base - ne) / 512, fl / 512, (base - c2) / 512, (base - c3) / 512);
    }
    // memory layout of the 15 planes, same bytes and arithmetic (streaming): does the number of concurrent streams limit the HBM rate?
    run<0, 0, 0, 3, 0>("LAYOUT 15 separate planes, shipped arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<0, 0, 0, 3, 1>("LAYOUT 5 row-interleaved groups", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<0, 0, 0, 3, 2>("LAYOUT 2 streams (9 in / 6 out interleaved)", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<1, 1, 1, 3, 0>("LAYOUT 15 planes, cheap arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<1, 1, 1, 3, 1>("LAYOUT 5 groups, cheap arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<1, 1, 1, 3, 2>("LAYOUT 2 streams, cheap arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<0, 0, 0, 2>("shipped arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<2, 0, 0, 2>("CANDIDATE f32 seed", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    run<0, 0, 0, 1>("shipped arithmetic", in, out, rows, pitch, plane, err, cus, 0x7fffffff);
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(err, 0, 64);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, err, 2048, mode);
        unsigned long long h[6];
        (void)hipMemcpy(h, err, sizeof h, hipMemcpyDeviceToHost);
        printf("exactness mode %d: %llu operand pairs | f32-seed: %llu in zone (%.2e), %llu WRONG outside zone, waves with a zone lane %.3e | short-f64: %llu in zone (%.2e), %llu WRONG\n",
               mode, h[0], h[1], (double)h[1] / h[0], h[2], (double)h[5] / ((double)h[0] / 64), h[3], (double)h[3] / h[0], h[4]);
    }
    return 0;
}
