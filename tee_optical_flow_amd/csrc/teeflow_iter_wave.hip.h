// teeflow_iter_wave.hip.h -- tvl1_iter, two iterations per launch, ONE WAVE PER STRIP, the row pipeline in registers (gfx950).
//
// What it replaces: the inner loop of OpenCV's DualTVL1 procOneScale as the reference reaches it through
// /root/reference/optical_flow/calculate_optical_flow.py:577-578, 642 (restated in oracle/tvl1_oracle.c:430-515).  Same
// schedule, modes (NORMAL / REPLAY / EXIT), strip rule and arithmetic helpers as k_iter2_rows -- hence the same bits -- in
// a different shape:
//
//   k_iter2_rows   a 256-thread block marches RY rows per step; the iterates of the rows in flight live in LDS, three
//                  barriers per step, the fill of the three-stage pipeline costs 2 RY rows per strip.
//   k_iter2_wave   a wave owns a full-width strip: lane l holds PX consecutive pixels of a row (PX = 4 / 6 / 8: levels up to
//                  256 / 384 / 512 px wide).  Of the three rows in flight -- row y (first primal update), row y-1 (first dual +
//                  second primal update), row y-2 (second dual update, store) -- rows y and y-1 live in two register sets that
//                  swap roles (interior loop unrolled twice: no copies), row y-2 and the warp constants between a row's two primal
//                  updates wait in lane-private LDS slots (WvPark: no barrier, a wave's LDS operations execute in order); the
//                  horizontal neighbours of a lane's first / last pixel come from the adjacent lane through DPP (wave_shr /
//                  wave_shl).  Every row predicate is wave-uniform, so the halo rows of a strip execute only the stages they need
//                  (a strip of R rows costs R + ~5 short row-steps instead of R + 3 + 2 RY full ones), and the interior of a strip
//                  (wv_step_full) has no predicate at all.
//
// Two launch forms (template parameter PF): two or three waves per SIMD that hide each other's load latency (PF = false), or ONE
// wave per SIMD that hides its own: the loads of row y+1 are in flight while row y is worked on, in accumulation registers the compiler
// is not told about (the landing zone, below).  k_iter3_wave (teeflow_iter3_wave.hip.h) builds on the second form.  All of them are
// OPTIONS (tf_set_tuning "iter_variant" 4 / 5 / 6): bit-identical to k_iter2_rows and, on the whole step, no faster -- DESIGN.md
// section 4c has the measurements and what they say.
//
// The launch is one round of resident waves (strip_rule_min): slots = CUs x resident waves per CU.
#pragma once
#include "teeflow_kernels.hip.h"
#include <type_traits>

__device__ __forceinline__ float wv_from_left(float v)      // lane l receives lane l-1's value (lane 0: 0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float wv_from_right(float v)     // lane l receives lane l+1's value (lane 63: 0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

template <int PX>
__device__ __forceinline__ void wv_ld(f2 (&d)[PX / 2], const float* __restrict__ p, unsigned off)
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + (size_t)(off * 4u) + 16 * q);
            d[2 * q] = mk2(v.x, v.y); d[2 * q + 1] = mk2(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) {
            const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p) + (size_t)(off * 4u) + 8 * h);
            d[h] = mk2(v.x, v.y);
        }
    }
}
template <int PX>
__device__ __forceinline__ void wv_st(float* __restrict__ p, unsigned off, const f2 (&d)[PX / 2])
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q)
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(p) + (size_t)(off * 4u) + 16 * q) = make_float4(d[2 * q].x, d[2 * q].y, d[2 * q + 1].x, d[2 * q + 1].y);
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) *reinterpret_cast<float2*>(reinterpret_cast<char*>(p) + (size_t)(off * 4u) + 8 * h) = make_float2(d[h].x, d[h].y);
    }
}
template <int PP>
__device__ __forceinline__ void wv_zero(f2 (&d)[PP])
{
#pragma unroll
    for (int h = 0; h < PP; ++h) d[h] = mk2(0.f, 0.f);
}

// convergence terms of two neighbouring pixels (same float operations as tv_err_quad_pk; the terms are integers < 2^43, so the
// double sum is exact whatever its order)
__device__ __forceinline__ double wv_err_pair(f2 un1, f2 uk1, f2 un2, f2 uk2, unsigned keep0, unsigned keep1)
{
    const f2 e1 = un1 - uk1, e2 = un2 - uk2;
    const f2 t = e1 * e1 + e2 * e2;
    const float v0 = __builtin_rintf(fminf(t.x, ERR_CAP_F) * ERR_SCALE_F), v1 = __builtin_rintf(fminf(t.y, ERR_CAP_F) * ERR_SCALE_F);
    return (double)mask_f(v0, keep0) + (double)mask_f(v1, keep1);
}

// first dual update of a row (or the second: same form): forward differences of the iterate `u` of the row (right neighbour of
// the lane's last pixel through DPP, the row below in `dn`), then estimateDualVariables on p -> o.  LASTROW_POSSIBLE = false: the
// caller knows the row is not the image's last one (no mask on the vertical difference).
template <int PP, bool LASTROW_POSSIBLE>
__device__ __forceinline__ void wv_dual_row(float taut, const f2 (&u1)[PP], const f2 (&u2)[PP], const f2 (&dn1)[PP], const f2 (&dn2)[PP],
                                            const f2 (&p11)[PP], const f2 (&p12)[PP], const f2 (&p21)[PP], const f2 (&p22)[PP],
                                            const unsigned* inw, unsigned mnl,
                                            f2 (&o11)[PP], f2 (&o12)[PP], f2 (&o21)[PP], f2 (&o22)[PP])
{
    const float rr1 = wv_from_right(u1[0].x), rr2 = wv_from_right(u2[0].x);
#pragma unroll
    for (int h = 0; h < PP; ++h) {
        const float n1 = h < PP - 1 ? u1[h < PP - 1 ? h + 1 : h].x : rr1, n2 = h < PP - 1 ? u2[h < PP - 1 ? h + 1 : h].x : rr2;
        const f2 u1x = mk2(mask_f(u1[h].y - u1[h].x, inw[2 * h + 1]), mask_f(n1 - u1[h].y, inw[2 * h + 2]));
        const f2 u2x = mk2(mask_f(u2[h].y - u2[h].x, inw[2 * h + 1]), mask_f(n2 - u2[h].y, inw[2 * h + 2]));
        const f2 d1 = dn1[h] - u1[h], d2 = dn2[h] - u2[h];
        f2 u1y = d1, u2y = d2;
        if constexpr (LASTROW_POSSIBLE) { u1y = mk2(mask_f(d1.x, mnl), mask_f(d1.y, mnl)); u2y = mk2(mask_f(d2.x, mnl), mask_f(d2.y, mnl)); }
        tv_p_pair(taut, u1x, u1y, u2x, u2y, p11[h], p12[h], p21[h], p22[h], o11[h], o12[h], o21[h], o22[h]);
    }
}

// primal update of a row: estimateV + divergence + estimateU (left neighbour of the lane's first pixel through DPP)
template <int PP>
__device__ __forceinline__ void wv_primal_row(float l_t, float theta, const f2 (&u1)[PP], const f2 (&u2)[PP], const f2 (&wx)[PP], const f2 (&wy)[PP],
                                              const f2 (&rc)[PP], const f2 (&p11)[PP], const f2 (&p12)[PP], const f2 (&p21)[PP], const f2 (&p22)[PP],
                                              const f2 (&up12)[PP], const f2 (&up22)[PP], bool ytop, bool lane0, f2 (&n1)[PP], f2 (&n2)[PP])
{
    const float l11 = wv_from_left(p11[PP - 1].y), l21 = wv_from_left(p21[PP - 1].y);
#pragma unroll
    for (int h = 0; h < PP; ++h) {
        const float a = h == 0 ? l11 : p11[h > 0 ? h - 1 : 0].y, b = h == 0 ? l21 : p21[h > 0 ? h - 1 : 0].y;
        tv_u_pair(l_t, theta, u1[h], u2[h], wx[h], wy[h], rc[h], p11[h], p12[h], p21[h], p22[h], up12[h], up22[h], a, b, ytop,
                  h == 0 ? lane0 : false, n1[h], n2[h]);
    }
}

// The planes a strip reads and the planes it writes are different buffers (ping-pong halves).  They arrive as __restrict__
// PARAMETERS of iter2_wave_rows: only that form tells the compiler that a row's loads do not alias the previous row's stores --
// with restrict-qualified locals it drains vmcnt(0) (every store acknowledged) before it issues the next row's loads.
struct WvGeom { int W, H, pitch, y0, R, x; bool replay, lane0; float l_t, theta, taut; };
struct WvPlanes {
    const float* __restrict__ gu1; const float* __restrict__ gu2; const float* __restrict__ g11; const float* __restrict__ g12;
    const float* __restrict__ g21; const float* __restrict__ g22; const float* __restrict__ gwx; const float* __restrict__ gwy;
    const float* __restrict__ grh;
    float* __restrict__ ou1; float* __restrict__ ou2; float* __restrict__ o11; float* __restrict__ o12; float* __restrict__ o21; float* __restrict__ o22;
};

// Where the three rows in flight live.  Row y (C) and row y-1 (A) are register sets that swap roles from step to step (the interior
// loop is unrolled twice: no copies).  Row y-2 -- second iterate + the dual variable after the first dual update, six values per
// pixel -- waits in LDS: every lane parks its own pixels there at the end of stage 2 and takes them back in the next step's stage 3
// (lane-private slots, a wave's LDS operations execute in order: no barrier).  That keeps a 512-px row at 8 px per lane inside 256
// VGPRs = two waves per SIMD, which is what hides a row's load latency (the other wave computes meanwhile).
template <int PP>
struct WvSet { f2 u1[PP], u2[PP], wx[PP], wy[PP], r[PP], p11[PP], p12[PP], p21[PP], p22[PP]; };

template <int PX> struct WvPark {
    static constexpr int NV = PX % 4 == 0 ? PX / 4 : PX / 2;                     // vectors per lane and plane
    typedef typename std::conditional<PX % 4 == 0, float4, float2>::type vec;
    vec v[9][NV][64];                                                            // [u1 u2 p11 p12 p21 p22 | wx wy r][vector][lane]: conflict-free
};
template <int PX>
__device__ __forceinline__ void wv_park(WvPark<PX>& s, int plane, int lane, const f2 (&d)[PX / 2])
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) s.v[plane][q][lane] = make_float4(d[2 * q].x, d[2 * q].y, d[2 * q + 1].x, d[2 * q + 1].y);
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) s.v[plane][h][lane] = make_float2(d[h].x, d[h].y);
    }
}
template <int PX>
__device__ __forceinline__ void wv_unpark(const WvPark<PX>& s, int plane, int lane, f2 (&d)[PX / 2])
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) { const float4 v = s.v[plane][q][lane]; d[2 * q] = mk2(v.x, v.y); d[2 * q + 1] = mk2(v.z, v.w); }
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) { const float2 v = s.v[plane][h][lane]; d[h] = mk2(v.x, v.y); }
    }
}

struct WvRows { int y0, yout_hi, s1_lo, s1_hi, p1_hi, H, pitch, x, lane; bool replay, lane0; float l_t, theta, taut; };

// the loads of row y into a set.  At the first launch of a level the dual variable is zero by definition: the host points
// g11..g22 at a plane of zeros, so the number of loads never depends on it.
template <int PX>
__device__ __forceinline__ void wv_issue(const WvRows& k, const WvPlanes& g, int y, WvSet<PX / 2>& N)
{
    const unsigned row = (unsigned)(y * k.pitch + k.x);
    wv_ld<PX>(N.u1, g.gu1, row); wv_ld<PX>(N.u2, g.gu2, row);
    wv_ld<PX>(N.wx, g.gwx, row); wv_ld<PX>(N.wy, g.gwy, row); wv_ld<PX>(N.r, g.grh, row);
    wv_ld<PX>(N.p11, g.g11, row); wv_ld<PX>(N.p12, g.g12, row); wv_ld<PX>(N.p21, g.g21, row); wv_ld<PX>(N.p22, g.g22, row);
}

// ---- the prefetch form's landing zone -------------------------------------------------------------------------------------------
// One wave per SIMD owns 512 registers, but the VALU reads only the 256 architectural ones.  The other half takes the loads of the
// NEXT row: global_load writes accumulation registers directly, nothing waits for them while the current row is worked on, and
// when the row's turn comes they move to a register set (v_accvgpr_read, one per value).  The three functions are generated
// (teeflow_land_gen.hip.h, tools/gen_land_regs.py -- its header says why the registers are fixed and hidden from hipcc):
//   wv_land_issue<PX>(k, g, y)   the 9 planes of row y -> landing zone; nothing is waited for
//   wv_land_wait<PX>()           s_waitcnt vmcnt(0).  MEASURED ON gfx950: a wave's stores can retire before OLDER loads of the same wave
//                                have returned, so "vmcnt(<stores issued since>)" does NOT mean the loads are there (results differed on
//                                full-size batches).  Hence vmcnt(0), placed BEFORE the step's stores go out: all that is outstanding then
//                                is the landing zone itself and the previous step's stores, which have had a whole step to complete.
//   wv_land_copy<PX>(C)          landing zone -> register set C (after wv_land_wait)
template <int PX> __device__ __forceinline__ void wv_land_issue(const WvRows& k, const WvPlanes& g, int y);
template <int PX> __device__ __forceinline__ void wv_land_wait();
template <int PX> __device__ __forceinline__ void wv_land_copy(WvSet<PX / 2>& C);
// Two zones (the three-iteration kernel keeps rows y+1 AND y+2 in flight; a row's zone is its parity):
//   wv_land2_issue<PX, Z>(k, g, y)   row y -> zone Z
//   wv_land2_wait_older<PX>()        s_waitcnt vmcnt(<loads per row>): everything but the NEWEST row's loads has arrived.  Sound although
//                                    stores retire out of order: loads return in order among themselves, so an outstanding load of the
//                                    older row would imply that all loads of the newer row are outstanding too -- more than the count allows.
//                                    Only valid right after a newer row has been issued; otherwise wv_land_wait<PX>() (vmcnt(0)).
//   wv_land2_copy<PX, Z>(C)          zone Z -> register set C
template <int PX, int Z> __device__ __forceinline__ void wv_land2_issue(const WvRows& k, const WvPlanes& g, int y);
template <int PX> __device__ __forceinline__ void wv_land2_wait_older();
template <int PX, int Z> __device__ __forceinline__ void wv_land2_copy(WvSet<PX / 2>& C);
#include "teeflow_land_gen.hip.h"

// One row step in the INTERIOR of a strip: every stage runs on a row that exists and is neither the image's first nor its last
// (y0 + 2 <= y <= min(yout_hi, s1_hi)), so there is no predicate left.  C takes row y, A holds row y-1 after its first primal
// update (first iterate, warp constants, the dual variable it started from), the park holds row y-2.
// PREFETCH: row y is already in C; the loads of row y+1 go out first, into the landing zone, and move to A's registers (row y-1 is
// done with them after stage 2) at the end of the step: one wave per SIMD hides its own load latency and needs half as many strips.
template <int PX, bool PREFETCH>
__device__ __forceinline__ void wv_step_full(const WvRows& k, const WvPlanes& g, int y, const unsigned* inw, WvSet<PX / 2>& C, WvSet<PX / 2>& A,
                                             WvPark<PX>& park, double& accA, double& accB)
{
    constexpr int PP = PX / 2;
    if constexpr (PREFETCH) wv_land_issue<PX>(k, g, y + 1);
    else wv_issue<PX>(k, g, y, C);
    // row y-1's warp constants come back from the park (slots 6..8), row y's take their place after stage 1: between its two primal
    // updates a row's constants do not occupy registers
    wv_unpark<PX>(park, 6, k.lane, A.wx); wv_unpark<PX>(park, 7, k.lane, A.wy); wv_unpark<PX>(park, 8, k.lane, A.r);
    // stage 1: row y, iteration `it` primal
    {
        f2 n1[PP], n2[PP];
        wv_primal_row<PP>(k.l_t, k.theta, C.u1, C.u2, C.wx, C.wy, C.r, C.p11, C.p12, C.p21, C.p22, A.p12, A.p22, false, k.lane0, n1, n2);
#pragma unroll
        for (int h = 0; h < PP; ++h) accA += wv_err_pair(n1[h], C.u1[h], n2[h], C.u2[h], inw[2 * h], inw[2 * h + 1]);
#pragma unroll
        for (int h = 0; h < PP; ++h) { C.u1[h] = n1[h]; C.u2[h] = n2[h]; }
        wv_park<PX>(park, 6, k.lane, C.wx); wv_park<PX>(park, 7, k.lane, C.wy); wv_park<PX>(park, 8, k.lane, C.r);
    }
    // stage 2: row y-1: iteration `it` dual, then `it+1` primal
    f2 q11[PP], q12[PP], q21[PP], q22[PP], m1[PP], m2[PP];
    f2 b12[PP], b22[PP];
    wv_unpark<PX>(park, 3, k.lane, b12); wv_unpark<PX>(park, 5, k.lane, b22);
    wv_dual_row<PP, false>(k.taut, A.u1, A.u2, C.u1, C.u2, A.p11, A.p12, A.p21, A.p22, inw, ~0u, q11, q12, q21, q22);
    wv_primal_row<PP>(k.l_t, k.theta, A.u1, A.u2, A.wx, A.wy, A.r, q11, q12, q21, q22, b12, b22, false, k.lane0, m1, m2);
#pragma unroll
    for (int h = 0; h < PP; ++h) accB += wv_err_pair(m1[h], A.u1[h], m2[h], A.u2[h], inw[2 * h], inw[2 * h + 1]);
    // stage 3: row y-2: iteration `it+1` dual, store
    {
        f2 bu1[PP], bu2[PP], b11[PP], b21[PP], r11[PP], r12[PP], r21[PP], r22[PP];
        wv_unpark<PX>(park, 0, k.lane, bu1); wv_unpark<PX>(park, 1, k.lane, bu2); wv_unpark<PX>(park, 2, k.lane, b11); wv_unpark<PX>(park, 4, k.lane, b21);
        wv_dual_row<PP, false>(k.taut, bu1, bu2, m1, m2, b11, b12, b21, b22, inw, ~0u, r11, r12, r21, r22);
        if constexpr (PREFETCH) wv_land_wait<PX>();             // before this step's stores go out (see wv_land_wait)
        const unsigned prow = (unsigned)((y - 2) * k.pitch + k.x);
        wv_st<PX>(g.ou1, prow, bu1); wv_st<PX>(g.ou2, prow, bu2);
        wv_st<PX>(g.o11, prow, r11); wv_st<PX>(g.o12, prow, r12); wv_st<PX>(g.o21, prow, r21); wv_st<PX>(g.o22, prow, r22);
    }
    // row y-1 moves to the park
    wv_park<PX>(park, 0, k.lane, m1); wv_park<PX>(park, 1, k.lane, m2);
    wv_park<PX>(park, 2, k.lane, q11); wv_park<PX>(park, 3, k.lane, q12); wv_park<PX>(park, 4, k.lane, q21); wv_park<PX>(park, 5, k.lane, q22);
    if constexpr (PREFETCH) wv_land_copy<PX>(A);          // row y+1 moves into the registers row y-1 has just left
}

// One row step anywhere (the first and last rows of a strip, REPLAY strips): every stage under its row predicate; C moves to A by
// copies at the end.
template <int PX>
__device__ __forceinline__ void wv_step_any(const WvRows& k, const WvPlanes& g, int y, const unsigned* inw, WvSet<PX / 2>& C, WvSet<PX / 2>& A,
                                            WvPark<PX>& park, double& accA, double& accB, bool c_loaded = false)
{
    constexpr int PP = PX / 2;
    const bool replay = k.replay;
    // ================= stage 1: row y, iteration `it` primal =================
    wv_unpark<PX>(park, 6, k.lane, A.wx); wv_unpark<PX>(park, 7, k.lane, A.wy); wv_unpark<PX>(park, 8, k.lane, A.r);   // see wv_step_full
    if (y <= k.s1_hi) {
        f2 n1[PP], n2[PP];
        if (!c_loaded) wv_issue<PX>(k, g, y, C);
        wv_primal_row<PP>(k.l_t, k.theta, C.u1, C.u2, C.wx, C.wy, C.r, C.p11, C.p12, C.p21, C.p22, A.p12, A.p22, y == 0, k.lane0, n1, n2);
        if (!replay && y >= k.y0 && y <= k.yout_hi) {
#pragma unroll
            for (int h = 0; h < PP; ++h) accA += wv_err_pair(n1[h], C.u1[h], n2[h], C.u2[h], inw[2 * h], inw[2 * h + 1]);
        }
#pragma unroll
        for (int h = 0; h < PP; ++h) { C.u1[h] = n1[h]; C.u2[h] = n2[h]; }
        wv_park<PX>(park, 6, k.lane, C.wx); wv_park<PX>(park, 7, k.lane, C.wy); wv_park<PX>(park, 8, k.lane, C.r);
    }
    // ================= stage 2: row y-1: iteration `it` dual, then `it+1` primal =================
    const int yb = y - 1;
    f2 m1[PP], m2[PP];
    wv_zero(m1); wv_zero(m2);
    if (yb >= k.s1_lo && yb <= k.p1_hi) {
        const unsigned mnl = opaque_u(yb >= k.H - 1 ? 0u : ~0u);       // no row below the last one
        f2 q11[PP], q12[PP], q21[PP], q22[PP];
        wv_dual_row<PP, true>(k.taut, A.u1, A.u2, C.u1, C.u2, A.p11, A.p12, A.p21, A.p22, inw, mnl, q11, q12, q21, q22);
        if (replay) {
            if (yb >= k.y0) {        // yb <= yout_hi holds: p1_hi == yout_hi in REPLAY
                const unsigned prow = (unsigned)(yb * k.pitch + k.x);
                wv_st<PX>(g.ou1, prow, A.u1); wv_st<PX>(g.ou2, prow, A.u2);
                wv_st<PX>(g.o11, prow, q11); wv_st<PX>(g.o12, prow, q12); wv_st<PX>(g.o21, prow, q21); wv_st<PX>(g.o22, prow, q22);
            }
        } else {
            if (yb >= k.y0) {
                f2 b12[PP], b22[PP];
                wv_unpark<PX>(park, 3, k.lane, b12); wv_unpark<PX>(park, 5, k.lane, b22);
                wv_primal_row<PP>(k.l_t, k.theta, A.u1, A.u2, A.wx, A.wy, A.r, q11, q12, q21, q22, b12, b22, yb == 0, k.lane0, m1, m2);
                if (yb <= k.yout_hi) {
#pragma unroll
                    for (int h = 0; h < PP; ++h) accB += wv_err_pair(m1[h], A.u1[h], m2[h], A.u2[h], inw[2 * h], inw[2 * h + 1]);
                }
            }
            // ================= stage 3: row y-2: iteration `it+1` dual, store =================
            const int yc = y - 2;
            if (yc >= k.y0 && yc <= k.yout_hi) {
                const unsigned mnl3 = opaque_u(yc >= k.H - 1 ? 0u : ~0u);
                f2 bu1[PP], bu2[PP], b11[PP], b12[PP], b21[PP], b22[PP], r11[PP], r12[PP], r21[PP], r22[PP];
                wv_unpark<PX>(park, 0, k.lane, bu1); wv_unpark<PX>(park, 1, k.lane, bu2); wv_unpark<PX>(park, 2, k.lane, b11);
                wv_unpark<PX>(park, 3, k.lane, b12); wv_unpark<PX>(park, 4, k.lane, b21); wv_unpark<PX>(park, 5, k.lane, b22);
                wv_dual_row<PP, true>(k.taut, bu1, bu2, m1, m2, b11, b12, b21, b22, inw, mnl3, r11, r12, r21, r22);
                const unsigned prow = (unsigned)(yc * k.pitch + k.x);
                wv_st<PX>(g.ou1, prow, bu1); wv_st<PX>(g.ou2, prow, bu2);
                wv_st<PX>(g.o11, prow, r11); wv_st<PX>(g.o12, prow, r12); wv_st<PX>(g.o21, prow, r21); wv_st<PX>(g.o22, prow, r22);
            }
            wv_park<PX>(park, 0, k.lane, m1); wv_park<PX>(park, 1, k.lane, m2);
            wv_park<PX>(park, 2, k.lane, q11); wv_park<PX>(park, 3, k.lane, q12); wv_park<PX>(park, 4, k.lane, q21); wv_park<PX>(park, 5, k.lane, q22);
        }
    } else if (!replay) {
        // the strip's last step: row y-1 had no stage 2, row y-2 still needs its stage 3 (its lower neighbour is past the strip's rows:
        // y-2 == yout_hi == H-1, the vertical difference is masked)
        const int yc = y - 2;
        if (yc >= k.y0 && yc <= k.yout_hi) {
            const unsigned mnl3 = opaque_u(yc >= k.H - 1 ? 0u : ~0u);
            f2 bu1[PP], bu2[PP], b11[PP], b12[PP], b21[PP], b22[PP], r11[PP], r12[PP], r21[PP], r22[PP];
            wv_unpark<PX>(park, 0, k.lane, bu1); wv_unpark<PX>(park, 1, k.lane, bu2); wv_unpark<PX>(park, 2, k.lane, b11);
            wv_unpark<PX>(park, 3, k.lane, b12); wv_unpark<PX>(park, 4, k.lane, b21); wv_unpark<PX>(park, 5, k.lane, b22);
            wv_dual_row<PP, true>(k.taut, bu1, bu2, m1, m2, b11, b12, b21, b22, inw, mnl3, r11, r12, r21, r22);
            const unsigned prow = (unsigned)(yc * k.pitch + k.x);
            wv_st<PX>(g.ou1, prow, bu1); wv_st<PX>(g.ou2, prow, bu2);
            wv_st<PX>(g.o11, prow, r11); wv_st<PX>(g.o12, prow, r12); wv_st<PX>(g.o21, prow, r21); wv_st<PX>(g.o22, prow, r22);
        }
    }
    // A <- C
#pragma unroll
    for (int h = 0; h < PP; ++h) {
        A.u1[h] = C.u1[h]; A.u2[h] = C.u2[h];
        A.p11[h] = C.p11[h]; A.p12[h] = C.p12[h]; A.p21[h] = C.p21[h]; A.p22[h] = C.p22[h];
    }
}

template <int PP>
__device__ __forceinline__ void wv_set_zero(WvSet<PP>& s)
{
    wv_zero(s.u1); wv_zero(s.u2); wv_zero(s.wx); wv_zero(s.wy); wv_zero(s.r); wv_zero(s.p11); wv_zero(s.p12); wv_zero(s.p21); wv_zero(s.p22);
}

template <int PX, bool PF>
__device__ __forceinline__ void iter2_wave_rows(const WvGeom& kg, WvPark<PX>& park, const float* __restrict__ gu1, const float* __restrict__ gu2,
                                                const float* __restrict__ g11, const float* __restrict__ g12, const float* __restrict__ g21,
                                                const float* __restrict__ g22, const float* __restrict__ gwx, const float* __restrict__ gwy,
                                                const float* __restrict__ grh, float* __restrict__ ou1, float* __restrict__ ou2,
                                                float* __restrict__ o11, float* __restrict__ o12, float* __restrict__ o21, float* __restrict__ o22,
                                                u64* qa_out, u64* qb_out)
{
    constexpr int PP = PX / 2;
    const int W = kg.W, H = kg.H, x = kg.x;
    const bool replay = kg.replay;
    const int y0 = kg.y0, R = kg.R;
    const WvPlanes g = {gu1, gu2, g11, g12, g21, g22, gwx, gwy, grh, ou1, ou2, o11, o12, o21, o22};

    // row ranges (absolute rows, clipped to the image).  NORMAL: first primal update on y0-1 .. y0+R+1, first dual on .. y0+R,
    // second primal on y0 .. y0+R, second dual + store on y0 .. y0+R-1.  REPLAY: first primal on y0 .. y0+R, first dual + store on
    // y0 .. y0+R-1.
    WvRows k;
    k.y0 = y0; k.H = H; k.pitch = kg.pitch; k.x = x; k.lane = (int)(threadIdx.x & 63); k.replay = replay; k.lane0 = kg.lane0;
    k.l_t = kg.l_t; k.theta = kg.theta; k.taut = kg.taut;
    k.yout_hi = y0 + R - 1 < H - 1 ? y0 + R - 1 : H - 1;
    k.s1_lo = replay ? y0 : (y0 > 0 ? y0 - 1 : 0);
    k.s1_hi = (replay ? y0 + R : y0 + R + 1) < H - 1 ? (replay ? y0 + R : y0 + R + 1) : H - 1;
    k.p1_hi = replay ? k.yout_hi : (y0 + R < H - 1 ? y0 + R : H - 1);
    const int y_end = replay ? k.yout_hi + 1 : k.yout_hi + 2;
    // interior rows: all three stages active, nothing at an image border (wv_step_full)
    const int full_lo = y0 + 2, full_hi = replay ? -1 : (k.yout_hi < k.s1_hi ? k.yout_hi : k.s1_hi);

    u64 qA = 0, qB = 0;
    if (x < W) {
        unsigned inw[PX + 1];            // inw[j] = all-ones iff column x + j lies inside the image
#pragma unroll
        for (int j = 0; j <= PX; ++j) inw[j] = opaque_u(x + j < W ? ~0u : 0u);
        double accA = 0.0, accB = 0.0;
        WvSet<PP> S0, S1;                // between steps: C = S0 (free), A = S1
        wv_set_zero(S0); wv_set_zero(S1);
        if (k.s1_lo > 0) {               // the row above the strip's first row: only its p12 / p22 are needed (divergence)
            const unsigned up = (unsigned)((k.s1_lo - 1) * k.pitch + x);
            wv_ld<PX>(S1.p12, g12, up); wv_ld<PX>(S1.p22, g22, up);
        }
        int y = k.s1_lo;
        for (; y <= y_end && y < full_lo; ++y) wv_step_any<PX>(k, g, y, inw, S0, S1, park, accA, accB);
        int fold = 0;
        bool c_loaded = false;
        if constexpr (PF) {
            // one wave per SIMD: the loads of row y+1 are in flight while row y is worked on (needs y+1 <= s1_hi)
            const int pf_hi = full_hi < k.s1_hi - 1 ? full_hi : k.s1_hi - 1;
            if (y + 1 <= pf_hi) {
                wv_issue<PX>(k, g, y, S0);
                for (; y + 1 <= pf_hi; y += 2) {
                    wv_step_full<PX, true>(k, g, y, inw, S0, S1, park, accA, accB);
                    wv_step_full<PX, true>(k, g, y + 1, inw, S1, S0, park, accA, accB);
                    if (++fold == 60) { qA += (u64)accA; qB += (u64)accB; accA = accB = 0.0; fold = 0; }   // keep the double sums exact (< 2^53)
                }
                c_loaded = true;         // row y sits in S0
            }
        } else {
            for (; y + 1 <= full_hi; y += 2) {
                wv_step_full<PX, false>(k, g, y, inw, S0, S1, park, accA, accB);
                wv_step_full<PX, false>(k, g, y + 1, inw, S1, S0, park, accA, accB);
                if (++fold == 60) { qA += (u64)accA; qB += (u64)accB; accA = accB = 0.0; fold = 0; }
            }
        }
        qA += (u64)accA; qB += (u64)accB; accA = accB = 0.0;
        for (; y <= y_end; ++y) { wv_step_any<PX>(k, g, y, inw, S0, S1, park, accA, accB, c_loaded); c_loaded = false; }   // a few interior rows + the strip's last rows
        qA += (u64)accA; qB += (u64)accB;
    }
    *qa_out = qA; *qb_out = qB;
}

template <int PX, bool PF>
__device__ __forceinline__ void iter2_wave_body(const IterArgs& a, const Iter2Blk& k, WvPark<PX>& park, u64* qa_out, u64* qb_out)
{
    const int lane = (int)(threadIdx.x & 63);
    const size_t po = (size_t)k.b * (size_t)k.splane;
    const int uc = k.uc, pc = k.pc;
    WvGeom g;
    g.W = k.W; g.H = k.H; g.pitch = k.pitch; g.y0 = k.strip * k.R; g.R = k.R; g.x = lane * PX;
    g.replay = k.replay; g.lane0 = lane == 0; g.l_t = a.l_t; g.theta = a.theta; g.taut = a.taut;
    // first launch of a level: the dual variable is zero, whatever the buffers hold -- read it from the plane of zeros
    const float* z = a.zplane;
    iter2_wave_rows<PX, PF>(g, park, a.sb.u1[uc] + po, a.sb.u2[uc] + po, k.pzero ? z : a.sb.p11[pc] + po, k.pzero ? z : a.sb.p12[pc] + po,
                        k.pzero ? z : a.sb.p21[pc] + po, k.pzero ? z : a.sb.p22[pc] + po,
                        a.wx + po, a.wy + po, a.rho + po, a.sb.u1[uc ^ 1] + po, a.sb.u2[uc ^ 1] + po, a.sb.p11[pc ^ 1] + po,
                        a.sb.p12[pc ^ 1] + po, a.sb.p21[pc ^ 1] + po, a.sb.p22[pc ^ 1] + po, qa_out, qb_out);
}

// grid = (max work items, 1, 1), 64 threads: every wave finds its (pair, strip) among the pairs that still iterate, exactly as
// k_iter2_rows does with `slots` > 0 (strip_rule with RY = 1 and `minrows` rows per strip at least).
#ifdef TF_WAVE_TIMING
__device__ unsigned long long g_wave_t[4096][4];    // per wave of the last recorded launch: s_memrealtime (100 MHz) at entry / after the prologue / at the end, rows
#define WAVE_T(i, v) do { if (rec && ln == 0 && blockIdx.x < 4096) g_wave_t[blockIdx.x][(i)] = (v); } while (0)
#else
#define WAVE_T(i, v) do { } while (0)
#endif

template <int PX, bool PF>
__global__ __launch_bounds__(64, PF ? 1 : (PX == 4 ? 3 : 2)) void k_iter2_wave(Iter2Args A, int slots, int minrows)
{
    __shared__ u64 sred[16];
    __shared__ WvPark<PX> park;
    const IterArgs& a = A.a;
    publish_active_count2(A);
    const int nchunk = (a.B + 63) >> 6, ln = (int)(threadIdx.x & 63);
#ifdef TF_WAVE_TIMING
    const bool rec = a.it == 2 && a.g.w == 512;       // level 0, second launch of a stage: (nearly) every pair active
    WAVE_T(0, __builtin_amdgcn_s_memrealtime()); WAVE_T(1, 0); WAVE_T(2, 0); WAVE_T(3, 0);
#endif
    for (int c = 0; c < nchunk; ++c) {
        const int pb = c * 64 + ln;
        const bool on = pb < a.B && pair_mode(a.err + (size_t)pb * a.errstride, a.it, A.total, a.thr_q, a.variant, a.thr_d) != M_EXIT;
        const u64 m = __ballot(on);
        if (ln == 0) sred[c] = m;
    }
    __syncthreads();
    int nact = 0;
    for (int c = 0; c < nchunk; ++c) nact += __popcll(sred[c]);
    nact = __builtin_amdgcn_readfirstlane(nact);       // wave-uniform by construction: keep it (and what follows from it) in SGPRs
    int R, S;
    strip_rule_min(nact, a.g.h, minrows, slots, &R, &S);
    const int item = blockIdx.x;
    if (item >= nact * S) return;                      // wave-uniform
    int kk = item / S;
    const int strip = item - kk * S;
    int c = 0;
    u64 m = sred[0];
    while (kk >= __popcll(m)) { kk -= __popcll(m); m = sred[++c]; }
    for (; kk > 0; --kk) m &= m - 1;                    // drop the kk lowest set bits
    const int b = __builtin_amdgcn_readfirstlane(c * 64 + (__ffsll((long long)m) - 1));
    u64* errb = a.err + (size_t)b * a.errstride;
    const int mode = __builtin_amdgcn_readfirstlane(pair_mode(errb, a.it, A.total, a.thr_q, a.variant, a.thr_d));   // wave-uniform
    if (mode == M_EXIT) return;
    const bool replay = mode == M_REPLAY;
    const PairCtl pc = a.ctl[b];
    const int utog = replay ? A.utog_prev : a.utog, ptog = replay ? A.ptog_prev : a.ptog;
    Iter2Blk blk;
    blk.b = b; blk.strip = strip; blk.R = R; blk.QX = 0; blk.RY = 1; blk.replay = replay;
    blk.pzero = (replay ? A.pzero_prev : a.pzero) != 0;
    blk.uc = __builtin_amdgcn_readfirstlane((pc.ubase ^ utog) & 1); blk.pc = __builtin_amdgcn_readfirstlane((pc.pbase ^ ptog) & 1);
    blk.it = a.it; blk.errb = errb;
    blk.W = a.g.w; blk.H = a.g.h; blk.pitch = a.g.pitch; blk.splane = a.g.splane;
    u64 qA, qB;
    WAVE_T(1, __builtin_amdgcn_s_memrealtime());
    iter2_wave_body<PX, PF>(a, blk, park, &qA, &qB);
    WAVE_T(2, __builtin_amdgcn_s_memrealtime() + (qA & 0)); WAVE_T(3, (unsigned long long)R);
    if (!replay) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { qA += __shfl_down(qA, off, 64); qB += __shfl_down(qB, off, 64); }
        if (ln == 0) {
            atomicAdd(&errb[blk.it], qA);
            atomicAdd(&errb[blk.it + 1], qB);
        }
    }
}
