// teeflow_iter3_wave.hip.h -- tvl1_iter, THREE inner iterations per launch, one wave per strip, one wave per SIMD (gfx950).
//
// What it replaces: the same inner loop as k_iter2_wave (teeflow_iter_wave.hip.h; OpenCV DualTVL1 procOneScale as the reference
// reaches it through /root/reference/optical_flow/calculate_optical_flow.py:577-578, 642, restated in oracle/tvl1_oracle.c:430-515),
// same arithmetic helpers -> same bits.  Why a third iteration: the two-iteration forms stream 60 B per pixel and pass and run at
// the memory system's rate for that read/write mix on full launches, so the only thing that makes an
// iteration cheaper is fewer bytes: 60 B per THREE iterations (20 B per pixel-iteration instead of 30).  Round 2's
// k_iter3_rows had the same byte count and gained nothing because its LDS need left two blocks per CU and with them a third fewer
// loads in flight; here the loads in flight do not depend on occupancy -- each of the four waves of a CU keeps a whole row (9
// planes) in flight in the accumulation registers while it works on the previous one (the landing zone of teeflow_iter_wave.hip.h).
//
//   step y:  stage 1  row y    primal 1 (row from the landing zone)
//            stage 2  row y-1  dual 1, primal 2        [a pass of ONE iteration stores after dual 1]
//            stage 3  row y-2  dual 2, primal 3        [a pass of TWO iterations stores after dual 2]
//            stage 4  row y-3  dual 3, store
//
// Rows of a strip y0 .. y0+R-1 in a pass of n iterations: primal j on y0-(n-j) .. y0+R+(n-j), dual j on the same rows but the last;
// each primal update adds its convergence term for rows y0 .. y0+R-1 (n = 3 only).  A pair whose stop test fires after the first or
// second iteration of a pass repeats that pass with n = 1 or 2 from the pass's own source buffers (REPLAY, as in the two-iteration
// forms).  Where the rows live: row y and row y-1 in two register sets that swap roles (no copies), row y-2's second iterate in
// registers and its dual variable in LDS, row y-3 in LDS, the warp constants of rows y .. y-2 in an LDS ring (every lane parks and
// fetches only its own pixels: no barrier).  38 KB of LDS per wave at 8 px per lane -- four waves fill a CU's 160 KB.
#pragma once
#include "teeflow_iter_wave.hip.h"

// Mode of a pair at pass `it` (a multiple of 3): NORMAL, EXIT, or REPLAY of *nrep (1 or 2) iterations of the previous pass.
__device__ __forceinline__ int pair_mode3(const u64* e, int it, int total, double thr, int* nrep)
{
    *nrep = 0;
    const bool on1 = it >= 1 ? (double)e[it - 1] > thr : true;
    const bool on2 = it >= 2 ? (double)e[it - 2] > thr : true;
    const bool on3 = it >= 3 ? (double)e[it - 3] > thr : true;
    if (it < total && (it == 0 || (on3 && on2 && on1))) return M_NORMAL;
    if (it >= 3) {
        const int j = it - 3;                                   // first iteration of the previous pass
        const bool act = j == 0 || ((double)e[j - 3] > thr && (double)e[j - 2] > thr && (double)e[j - 1] > thr);
        if (act) {
            if (!on3) { *nrep = 1; return M_REPLAY; }           // stopped after the first iteration of that pass
            if (!on2) { *nrep = 2; return M_REPLAY; }           // ... after the second
        }
    }
    return M_EXIT;
}

#ifdef TF_WAVE_TIMING
__device__ unsigned long long g_step_t[2][96];     // s_memrealtime after every row step of two waves of a level-0 NORMAL strip
#endif
template <int PX> struct WvPark3 {
    static constexpr int NV = WvPark<PX>::NV;
    typedef typename WvPark<PX>::vec vec;
    vec w[3][3][NV][64];         // ring: warp constants (wx wy rho_c) of rows y, y-1, y-2; slot = row mod 3
    vec b[4][NV][64];            // row y-2: dual variable after its first dual update (p11 p12 p21 p22)
    vec e[6][NV][64];            // row y-3: third iterate (u1 u2) and the dual variable after its second dual update
};
template <int PX>
__device__ __forceinline__ void wv_put(typename WvPark<PX>::vec (&pl)[WvPark<PX>::NV][64], int lane, const f2 (&d)[PX / 2])
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) pl[q][lane] = make_float4(d[2 * q].x, d[2 * q].y, d[2 * q + 1].x, d[2 * q + 1].y);
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) pl[h][lane] = make_float2(d[h].x, d[h].y);
    }
}
template <int PX>
__device__ __forceinline__ void wv_get(const typename WvPark<PX>::vec (&pl)[WvPark<PX>::NV][64], int lane, f2 (&d)[PX / 2])
{
    if constexpr (PX % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) { const float4 v = pl[q][lane]; d[2 * q] = mk2(v.x, v.y); d[2 * q + 1] = mk2(v.z, v.w); }
    } else {
#pragma unroll
        for (int h = 0; h < PX / 2; ++h) { const float2 v = pl[h][lane]; d[h] = mk2(v.x, v.y); }
    }
}

struct Wv3Rows {
    int y0, yout_hi, H, pitch, x, lane, n;     // n = iterations this pass performs (3: NORMAL with convergence sums; 1, 2: REPLAY)
    int lo[4], hiP[4], hiD[4];                 // rows of primal j / dual j (j = 1..n), clipped to the image
    bool lane0; float l_t, theta, taut;
    __device__ __forceinline__ bool inP(int j, int r) const { return r >= lo[j] && r <= hiP[j]; }
    __device__ __forceinline__ bool inD(int j, int r) const { return r >= lo[j] && r <= hiD[j]; }
};

template <int PP>
__device__ __forceinline__ void wv_err_row(double& acc, const f2 (&n1)[PP], const f2 (&k1)[PP], const f2 (&n2)[PP], const f2 (&k2)[PP], const unsigned* inw)
{
#pragma unroll
    for (int h = 0; h < PP; ++h) acc += wv_err_pair(n1[h], k1[h], n2[h], k2[h], inw[2 * h], inw[2 * h + 1]);
}

// One row step in the INTERIOR of a NORMAL strip (y0 + 3 <= y <= min(yout_hi, last loaded row - 1)): all four stages, no predicate.
// C holds row y (taken from the landing zone at the end of the previous step), A row y-1 after primal 1, Bu the second iterate of
// row y-2; the loads of row y+1 go out first and arrive in A's registers at the end.  s0/s1/s2 = ring slots of rows y, y-1, y-2.
template <int PX, int PAR>
__device__ __forceinline__ void wv3_step_full(const Wv3Rows& k, const WvRows& kr, const WvPlanes& g, int y, int s0, int s1, int s2, const unsigned* inw,
                                              WvSet<PX / 2>& C, WvSet<PX / 2>& A, f2 (&Bu1)[PX / 2], f2 (&Bu2)[PX / 2], 
                                              WvPark3<PX>& park, double& accA, double& accB, double& accC)
{
    constexpr int PP = PX / 2;
    const int lane = k.lane;
    wv_land2_issue<PX, PAR>(kr, g, y + 2);       // PAR = parity of y = zone of row y+2; row y+1 has been in flight since the previous step
    // stage 1: row y, primal 1
    {
        f2 n1[PP], n2[PP];
        wv_primal_row<PP>(k.l_t, k.theta, C.u1, C.u2, C.wx, C.wy, C.r, C.p11, C.p12, C.p21, C.p22, A.p12, A.p22, false, k.lane0, n1, n2);
        wv_err_row<PP>(accA, n1, C.u1, n2, C.u2, inw);
#pragma unroll
        for (int h = 0; h < PP; ++h) { C.u1[h] = n1[h]; C.u2[h] = n2[h]; }
        wv_put<PX>(park.w[s0][0], lane, C.wx); wv_put<PX>(park.w[s0][1], lane, C.wy); wv_put<PX>(park.w[s0][2], lane, C.r);
    }
    // stage 2: row y-1: dual 1, primal 2
    f2 q11[PP], q12[PP], q21[PP], q22[PP], m1[PP], m2[PP], b12[PP], b22[PP];
    wv_dual_row<PP, false>(k.taut, A.u1, A.u2, C.u1, C.u2, A.p11, A.p12, A.p21, A.p22, inw, ~0u, q11, q12, q21, q22);
    wv_get<PX>(park.w[s1][0], lane, A.wx); wv_get<PX>(park.w[s1][1], lane, A.wy); wv_get<PX>(park.w[s1][2], lane, A.r);
    wv_get<PX>(park.b[1], lane, b12); wv_get<PX>(park.b[3], lane, b22);
    wv_primal_row<PP>(k.l_t, k.theta, A.u1, A.u2, A.wx, A.wy, A.r, q11, q12, q21, q22, b12, b22, false, k.lane0, m1, m2);
    wv_err_row<PP>(accB, m1, A.u1, m2, A.u2, inw);
    // stage 3: row y-2: dual 2, primal 3
    f2 r11[PP], r12[PP], r21[PP], r22[PP], t1[PP], t2[PP], e12[PP], e22[PP];
    {
        f2 b11[PP], b21[PP], wx2[PP], wy2[PP], rc2[PP];
        wv_get<PX>(park.b[0], lane, b11); wv_get<PX>(park.b[2], lane, b21);
        wv_put<PX>(park.b[0], lane, q11); wv_put<PX>(park.b[1], lane, q12); wv_put<PX>(park.b[2], lane, q21); wv_put<PX>(park.b[3], lane, q22);   // row y-1 takes the slot
        wv_dual_row<PP, false>(k.taut, Bu1, Bu2, m1, m2, b11, b12, b21, b22, inw, ~0u, r11, r12, r21, r22);
        wv_get<PX>(park.w[s2][0], lane, wx2); wv_get<PX>(park.w[s2][1], lane, wy2); wv_get<PX>(park.w[s2][2], lane, rc2);
        wv_get<PX>(park.e[3], lane, e12); wv_get<PX>(park.e[5], lane, e22);
        wv_primal_row<PP>(k.l_t, k.theta, Bu1, Bu2, wx2, wy2, rc2, r11, r12, r21, r22, e12, e22, false, k.lane0, t1, t2);
        wv_err_row<PP>(accC, t1, Bu1, t2, Bu2, inw);
    }
    // stage 4: row y-3: dual 3, store
    {
        f2 eu1[PP], eu2[PP], e11[PP], e21[PP], s11[PP], s12[PP], s21[PP], s22[PP];
        wv_get<PX>(park.e[0], lane, eu1); wv_get<PX>(park.e[1], lane, eu2); wv_get<PX>(park.e[2], lane, e11); wv_get<PX>(park.e[4], lane, e21);
        wv_put<PX>(park.e[0], lane, t1); wv_put<PX>(park.e[1], lane, t2);                                             // row y-2 takes the slot
        wv_put<PX>(park.e[2], lane, r11); wv_put<PX>(park.e[3], lane, r12); wv_put<PX>(park.e[4], lane, r21); wv_put<PX>(park.e[5], lane, r22);
        wv_dual_row<PP, false>(k.taut, eu1, eu2, t1, t2, e11, e12, e21, e22, inw, ~0u, s11, s12, s21, s22);
        const unsigned prow = (unsigned)((y - 3) * k.pitch + k.x);
        wv_st<PX>(g.ou1, prow, eu1); wv_st<PX>(g.ou2, prow, eu2);
        wv_st<PX>(g.o11, prow, s11); wv_st<PX>(g.o12, prow, s12); wv_st<PX>(g.o21, prow, s21); wv_st<PX>(g.o22, prow, s22);
    }
#pragma unroll
    for (int h = 0; h < PP; ++h) { Bu1[h] = m1[h]; Bu2[h] = m2[h]; }
    wv_land2_wait_older<PX>();                                // row y+1 is there (row y+2's loads are the newer ones)
    wv_land2_copy<PX, PAR ^ 1>(A);                            // ... and moves into the registers row y-1 has just left
}

// One row step anywhere, for a pass of k.n iterations (the first and last rows of a strip, REPLAY strips): every stage under its
// row predicate; C moves to A by copies at the end.
template <int PX>
__device__ __forceinline__ void wv3_step_any(const Wv3Rows& k, const WvRows& kr, const WvPlanes& g, int y, const unsigned* inw, WvSet<PX / 2>& C,
                                             WvSet<PX / 2>& A, f2 (&Bu1)[PX / 2], f2 (&Bu2)[PX / 2], WvPark3<PX>& park, double& accA,
                                             double& accB, double& accC)
{
    constexpr int PP = PX / 2;
    const int lane = k.lane, n = k.n;
    // same protocol as the interior step: row y is in C on entry, row y+1 in flight towards the zone of its parity; row y+2's loads go out
    // now (if primal 1 covers that row), row y+1 is waited for and moves to C at the end
    const bool have1 = k.inP(1, y + 1), have2 = k.inP(1, y + 2);
    if (have2) { if (y & 1) wv_land2_issue<PX, 1>(kr, g, y + 2); else wv_land2_issue<PX, 0>(kr, g, y + 2); }
    const int s0 = (y + 3) % 3, s1 = (y + 2) % 3, s2 = (y + 1) % 3;          // ring slots of rows y, y-1, y-2 (y >= -2)
    const bool sums = n == 3;
    // ================= stage 1: row y, primal 1 =================
    if (k.inP(1, y)) {
        f2 n1[PP], n2[PP];
        wv_primal_row<PP>(k.l_t, k.theta, C.u1, C.u2, C.wx, C.wy, C.r, C.p11, C.p12, C.p21, C.p22, A.p12, A.p22, y == 0, k.lane0, n1, n2);
        if (sums && y >= k.y0 && y <= k.yout_hi) wv_err_row<PP>(accA, n1, C.u1, n2, C.u2, inw);
#pragma unroll
        for (int h = 0; h < PP; ++h) { C.u1[h] = n1[h]; C.u2[h] = n2[h]; }
        wv_put<PX>(park.w[s0][0], lane, C.wx); wv_put<PX>(park.w[s0][1], lane, C.wy); wv_put<PX>(park.w[s0][2], lane, C.r);
    }
    // ================= stage 2: row y-1: dual 1 [store], primal 2 =================
    const int yb = y - 1;
    const bool st2 = k.inD(1, yb);
    f2 q11[PP], q12[PP], q21[PP], q22[PP], m1[PP], m2[PP];
    wv_zero(q11); wv_zero(q12); wv_zero(q21); wv_zero(q22); wv_zero(m1); wv_zero(m2);
    if (st2) {
        const unsigned mnl = opaque_u(yb >= k.H - 1 ? 0u : ~0u);       // no row below the last one
        wv_dual_row<PP, true>(k.taut, A.u1, A.u2, C.u1, C.u2, A.p11, A.p12, A.p21, A.p22, inw, mnl, q11, q12, q21, q22);
        if (n == 1) {                    // dual 1 rows of a one-iteration pass are exactly the strip's rows
            const unsigned prow = (unsigned)(yb * k.pitch + k.x);
            wv_st<PX>(g.ou1, prow, A.u1); wv_st<PX>(g.ou2, prow, A.u2);
            wv_st<PX>(g.o11, prow, q11); wv_st<PX>(g.o12, prow, q12); wv_st<PX>(g.o21, prow, q21); wv_st<PX>(g.o22, prow, q22);
        } else if (k.inP(2, yb)) {
            f2 b12[PP], b22[PP];
            wv_get<PX>(park.w[s1][0], lane, A.wx); wv_get<PX>(park.w[s1][1], lane, A.wy); wv_get<PX>(park.w[s1][2], lane, A.r);
            wv_get<PX>(park.b[1], lane, b12); wv_get<PX>(park.b[3], lane, b22);
            wv_primal_row<PP>(k.l_t, k.theta, A.u1, A.u2, A.wx, A.wy, A.r, q11, q12, q21, q22, b12, b22, yb == 0, k.lane0, m1, m2);
            if (sums && yb >= k.y0 && yb <= k.yout_hi) wv_err_row<PP>(accB, m1, A.u1, m2, A.u2, inw);
        }
    }
    // ================= stage 3: row y-2: dual 2 [store], primal 3 =================
    const int yc = y - 2;
    const bool st3 = n >= 2 && k.inD(2, yc);
    f2 r11[PP], r12[PP], r21[PP], r22[PP], t1[PP], t2[PP];
    wv_zero(r11); wv_zero(r12); wv_zero(r21); wv_zero(r22); wv_zero(t1); wv_zero(t2);
    if (st3) {
        const unsigned mnl = opaque_u(yc >= k.H - 1 ? 0u : ~0u);
        f2 b11[PP], b12[PP], b21[PP], b22[PP];
        wv_get<PX>(park.b[0], lane, b11); wv_get<PX>(park.b[1], lane, b12); wv_get<PX>(park.b[2], lane, b21); wv_get<PX>(park.b[3], lane, b22);
        wv_dual_row<PP, true>(k.taut, Bu1, Bu2, m1, m2, b11, b12, b21, b22, inw, mnl, r11, r12, r21, r22);
        if (n == 2) {                    // dual 2 rows of a two-iteration pass are exactly the strip's rows
            const unsigned prow = (unsigned)(yc * k.pitch + k.x);
            wv_st<PX>(g.ou1, prow, Bu1); wv_st<PX>(g.ou2, prow, Bu2);
            wv_st<PX>(g.o11, prow, r11); wv_st<PX>(g.o12, prow, r12); wv_st<PX>(g.o21, prow, r21); wv_st<PX>(g.o22, prow, r22);
        } else if (k.inP(3, yc)) {
            f2 wx2[PP], wy2[PP], rc2[PP], e12[PP], e22[PP];
            wv_get<PX>(park.w[s2][0], lane, wx2); wv_get<PX>(park.w[s2][1], lane, wy2); wv_get<PX>(park.w[s2][2], lane, rc2);
            wv_get<PX>(park.e[3], lane, e12); wv_get<PX>(park.e[5], lane, e22);
            wv_primal_row<PP>(k.l_t, k.theta, Bu1, Bu2, wx2, wy2, rc2, r11, r12, r21, r22, e12, e22, yc == 0, k.lane0, t1, t2);
            if (yc >= k.y0 && yc <= k.yout_hi) wv_err_row<PP>(accC, t1, Bu1, t2, Bu2, inw);
        }
    }
    // ================= stage 4: row y-3: dual 3, store =================
    const int yd = y - 3;
    if (n == 3 && k.inD(3, yd)) {
        const unsigned mnl = opaque_u(yd >= k.H - 1 ? 0u : ~0u);
        f2 eu1[PP], eu2[PP], e11[PP], e12[PP], e21[PP], e22[PP], s11[PP], s12[PP], s21[PP], s22[PP];
        wv_get<PX>(park.e[0], lane, eu1); wv_get<PX>(park.e[1], lane, eu2); wv_get<PX>(park.e[2], lane, e11); wv_get<PX>(park.e[3], lane, e12);
        wv_get<PX>(park.e[4], lane, e21); wv_get<PX>(park.e[5], lane, e22);
        wv_dual_row<PP, true>(k.taut, eu1, eu2, t1, t2, e11, e12, e21, e22, inw, mnl, s11, s12, s21, s22);
        const unsigned prow = (unsigned)(yd * k.pitch + k.x);
        wv_st<PX>(g.ou1, prow, eu1); wv_st<PX>(g.ou2, prow, eu2);
        wv_st<PX>(g.o11, prow, s11); wv_st<PX>(g.o12, prow, s12); wv_st<PX>(g.o21, prow, s21); wv_st<PX>(g.o22, prow, s22);
    }
    // ================= the rows move on =================
    if (st3 && n == 3) {
        wv_put<PX>(park.e[0], lane, t1); wv_put<PX>(park.e[1], lane, t2);
        wv_put<PX>(park.e[2], lane, r11); wv_put<PX>(park.e[3], lane, r12); wv_put<PX>(park.e[4], lane, r21); wv_put<PX>(park.e[5], lane, r22);
    }
    if (st2 && n >= 2) {
        wv_put<PX>(park.b[0], lane, q11); wv_put<PX>(park.b[1], lane, q12); wv_put<PX>(park.b[2], lane, q21); wv_put<PX>(park.b[3], lane, q22);
#pragma unroll
        for (int h = 0; h < PP; ++h) { Bu1[h] = m1[h]; Bu2[h] = m2[h]; }
    }
#pragma unroll
    for (int h = 0; h < PP; ++h) {
        A.u1[h] = C.u1[h]; A.u2[h] = C.u2[h];
        A.p11[h] = C.p11[h]; A.p12[h] = C.p12[h]; A.p21[h] = C.p21[h]; A.p22[h] = C.p22[h];
    }
    if (have1) {
        if (have2) wv_land2_wait_older<PX>(); else wv_land_wait<PX>();
        if (y & 1) wv_land2_copy<PX, 0>(C); else wv_land2_copy<PX, 1>(C);      // row y+1's zone
    }
}

// the strip of one wave; planes as __restrict__ parameters (see iter2_wave_rows)
template <int PX>
__device__ __forceinline__ void iter3_wave_rows(const WvGeom& kg, int n, WvPark3<PX>& park, const float* __restrict__ gu1, const float* __restrict__ gu2,
                                                const float* __restrict__ g11, const float* __restrict__ g12, const float* __restrict__ g21,
                                                const float* __restrict__ g22, const float* __restrict__ gwx, const float* __restrict__ gwy,
                                                const float* __restrict__ grh, float* __restrict__ ou1, float* __restrict__ ou2,
                                                float* __restrict__ o11, float* __restrict__ o12, float* __restrict__ o21, float* __restrict__ o22,
                                                u64* qa_out, u64* qb_out, u64* qc_out)
{
    constexpr int PP = PX / 2;
    const int W = kg.W, H = kg.H, x = kg.x;
    const int y0 = kg.y0, R = kg.R;
    const WvPlanes g = {gu1, gu2, g11, g12, g21, g22, gwx, gwy, grh, ou1, ou2, o11, o12, o21, o22};
    Wv3Rows k;
    k.y0 = y0; k.H = H; k.pitch = kg.pitch; k.x = x; k.lane = (int)(threadIdx.x & 63); k.n = n; k.lane0 = kg.lane0;
    k.l_t = kg.l_t; k.theta = kg.theta; k.taut = kg.taut;
    k.yout_hi = y0 + R - 1 < H - 1 ? y0 + R - 1 : H - 1;
#pragma unroll
    for (int j = 1; j <= 3; ++j) {
        const int lo = y0 - (n - j), hi = y0 + R + (n - j);
        k.lo[j] = j > n ? 1 : (lo > 0 ? lo : 0);
        k.hiP[j] = j > n ? 0 : (hi < H - 1 ? hi : H - 1);
        k.hiD[j] = j > n ? 0 : (hi - 1 < H - 1 ? hi - 1 : H - 1);
    }
    k.lo[0] = 1; k.hiP[0] = 0; k.hiD[0] = 0;
    WvRows kr;                                   // what wv_issue / wv_land_issue read
    kr.pitch = kg.pitch; kr.x = x;
    const int y_first = k.lo[1], y_end = k.yout_hi + n;
    // interior rows of a NORMAL strip: all four stages on rows that exist, none at an image border, and row y+1 is a row primal 1 covers
    const int full_lo = y0 + 3;
    const int full_hi = n == 3 ? (k.yout_hi < k.hiP[1] - 2 ? k.yout_hi : k.hiP[1] - 2) : -1;      // row y+2 must be a row primal 1 covers

    u64 qA = 0, qB = 0, qC = 0;
    if (x < W) {
        unsigned inw[PX + 1];
#pragma unroll
        for (int j = 0; j <= PX; ++j) inw[j] = opaque_u(x + j < W ? ~0u : 0u);
        double accA = 0.0, accB = 0.0, accC = 0.0;
        WvSet<PP> S0, S1;                // between steps: C = S0, A = S1
        f2 Bu1[PP], Bu2[PP];
        wv_set_zero(S0); wv_set_zero(S1); wv_zero(Bu1); wv_zero(Bu2);
        if (y_first > 0) {               // the row above the strip's first row: only its p12 / p22 are needed (divergence)
            const unsigned up = (unsigned)((y_first - 1) * k.pitch + x);
            wv_ld<PX>(S1.p12, g12, up); wv_ld<PX>(S1.p22, g22, up);
        }
        int y = y_first;
#ifdef TF_WAVE_TIMING
        int ts = 0;
        const bool trec = kg.W == 512 && n == 3 && (blockIdx.x == 5 || blockIdx.x == 517) && (threadIdx.x & 63) == 0;
#define STEP_T() do { if (trec && ts < 96) g_step_t[blockIdx.x == 5 ? 0 : 1][ts++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STEP_T() do { } while (0)
#endif
        wv_issue<PX>(kr, g, y, S0);      // the strip's first row; every later row arrives through the landing zones
        if (k.inP(1, y + 1)) { if ((y + 1) & 1) wv_land2_issue<PX, 1>(kr, g, y + 1); else wv_land2_issue<PX, 0>(kr, g, y + 1); }
        STEP_T();
        for (; y <= y_end && (y < full_lo || (y & 1)); ++y) { wv3_step_any<PX>(k, kr, g, y, inw, S0, S1, Bu1, Bu2, park, accA, accB, accC); STEP_T(); }
        if (y + 1 <= full_hi) {          // y is even here
            int s0 = y % 3, fold = 0;
            for (; y + 1 <= full_hi; y += 2) {
                const int sa = s0, sb = s0 == 0 ? 2 : s0 - 1, sc = sb == 0 ? 2 : sb - 1;      // slots of rows y, y-1, y-2
                wv3_step_full<PX, 0>(k, kr, g, y, sa, sb, sc, inw, S0, S1, Bu1, Bu2, park, accA, accB, accC);
                STEP_T();
                const int sd = sa == 2 ? 0 : sa + 1;                                           // slot of row y+1
                wv3_step_full<PX, 1>(k, kr, g, y + 1, sd, sa, sb, inw, S1, S0, Bu1, Bu2, park, accA, accB, accC);
                STEP_T();
                s0 = sd == 2 ? 0 : sd + 1;
                if (++fold == 60) { qA += (u64)accA; qB += (u64)accB; qC += (u64)accC; accA = accB = accC = 0.0; fold = 0; }   // exact (< 2^53)
            }
        }
        qA += (u64)accA; qB += (u64)accB; qC += (u64)accC; accA = accB = accC = 0.0;
        for (; y <= y_end; ++y) { wv3_step_any<PX>(k, kr, g, y, inw, S0, S1, Bu1, Bu2, park, accA, accB, accC); STEP_T(); }
        qA += (u64)accA; qB += (u64)accB; qC += (u64)accC;
    }
    *qa_out = qA; *qb_out = qB; *qc_out = qC;
}

__device__ __forceinline__ void publish_active_count3(const Iter2Args& A)
{
    const IterArgs& a = A.a;
    if (a.host_slot && blockIdx.x == 0 && threadIdx.x < 64) {
        int c = 0, nr;
        for (int b2 = threadIdx.x; b2 < a.B; b2 += 64)
            c += pair_mode3(a.err + (size_t)b2 * a.errstride, a.it, A.total, a.thr_q, &nr) != M_EXIT ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (threadIdx.x == 0) __hip_atomic_store(a.host_slot, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// grid = (max work items, 1, 1), 64 threads, one wave per SIMD.  Iter2Args carries the same fields for a pass of three (a.it is a
// multiple of 3; the *_prev toggles are those of the previous pass, for REPLAY strips).
template <int PX>
__global__ __launch_bounds__(64, 1) void k_iter3_wave(Iter2Args A, int slots, int minrows)
{
    __shared__ u64 sred[16];
    __shared__ WvPark3<PX> park;
    const IterArgs& a = A.a;
    publish_active_count3(A);
    const int nchunk = (a.B + 63) >> 6, ln = (int)(threadIdx.x & 63);
#ifdef TF_WAVE_TIMING
    const bool rec = a.it == 3 && a.g.w == 512;       // level 0, second pass of a stage: (nearly) every pair active
    WAVE_T(0, __builtin_amdgcn_s_memrealtime()); WAVE_T(1, 0); WAVE_T(2, 0); WAVE_T(3, 0);
#endif
    for (int c = 0; c < nchunk; ++c) {
        const int pb = c * 64 + ln;
        int nr;
        const bool on = pb < a.B && pair_mode3(a.err + (size_t)pb * a.errstride, a.it, A.total, a.thr_q, &nr) != M_EXIT;
        const u64 m = __ballot(on);
        if (ln == 0) sred[c] = m;
    }
    __syncthreads();
    int nact = 0;
    for (int c = 0; c < nchunk; ++c) nact += __popcll(sred[c]);
    nact = __builtin_amdgcn_readfirstlane(nact);
    int R, S;
    strip_rule_min(nact, a.g.h, minrows, slots, &R, &S);
    const int item = blockIdx.x;
    if (item >= nact * S) return;                      // wave-uniform
    int kk = item / S;
    const int strip = item - kk * S;
    int c = 0;
    u64 m = sred[0];
    while (kk >= __popcll(m)) { kk -= __popcll(m); m = sred[++c]; }
    for (; kk > 0; --kk) m &= m - 1;
    const int b = __builtin_amdgcn_readfirstlane(c * 64 + (__ffsll((long long)m) - 1));
    u64* errb = a.err + (size_t)b * a.errstride;
    int nrep_v;
    const int mode = __builtin_amdgcn_readfirstlane(pair_mode3(errb, a.it, A.total, a.thr_q, &nrep_v));
    if (mode == M_EXIT) return;
    const bool replay = mode == M_REPLAY;
    const int n = replay ? __builtin_amdgcn_readfirstlane(nrep_v) : 3;
    const PairCtl pc = a.ctl[b];
    const int utog = replay ? A.utog_prev : a.utog, ptog = replay ? A.ptog_prev : a.ptog;
    const bool pzero = (replay ? A.pzero_prev : a.pzero) != 0;
    const int uc = __builtin_amdgcn_readfirstlane((pc.ubase ^ utog) & 1), pcur = __builtin_amdgcn_readfirstlane((pc.pbase ^ ptog) & 1);
    const size_t po = (size_t)b * (size_t)a.g.splane;
    WvGeom g;
    g.W = a.g.w; g.H = a.g.h; g.pitch = a.g.pitch; g.y0 = strip * R; g.R = R; g.x = ln * PX;
    g.replay = replay; g.lane0 = ln == 0; g.l_t = a.l_t; g.theta = a.theta; g.taut = a.taut;
    const float* z = a.zplane;           // first pass of a level: the dual variable is zero, whatever the buffers hold
    u64 qA, qB, qC;
    WAVE_T(1, __builtin_amdgcn_s_memrealtime());
    iter3_wave_rows<PX>(g, n, park, a.sb.u1[uc] + po, a.sb.u2[uc] + po, pzero ? z : a.sb.p11[pcur] + po, pzero ? z : a.sb.p12[pcur] + po,
                        pzero ? z : a.sb.p21[pcur] + po, pzero ? z : a.sb.p22[pcur] + po, a.wx + po, a.wy + po, a.rho + po,
                        a.sb.u1[uc ^ 1] + po, a.sb.u2[uc ^ 1] + po, a.sb.p11[pcur ^ 1] + po, a.sb.p12[pcur ^ 1] + po, a.sb.p21[pcur ^ 1] + po,
                        a.sb.p22[pcur ^ 1] + po, &qA, &qB, &qC);
    WAVE_T(2, __builtin_amdgcn_s_memrealtime() + (qA & 0)); WAVE_T(3, (unsigned long long)R);
    if (!replay) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { qA += __shfl_down(qA, off, 64); qB += __shfl_down(qB, off, 64); qC += __shfl_down(qC, off, 64); }
        if (ln == 0) {
            atomicAdd(&errb[a.it], qA);
            atomicAdd(&errb[a.it + 1], qB);
            atomicAdd(&errb[a.it + 2], qC);
        }
    }
}

// median for the three-iterations-per-launch schedule: a pair takes part iff it is in NORMAL mode at `it`
template <int KS>
__global__ __launch_bounds__(256) void k_median3(MedArgs a, int total)
{
    constexpr int R = KS / 2, TWm = 64, THm = 16, LW = TWm + 2 * R, LH = THm + 2 * R;
    __shared__ __attribute__((aligned(16))) float t[LH][LW];
    const int b = blockIdx.z >> 1, plane = blockIdx.z & 1;
    int nr;
    if (pair_mode3(a.err + (size_t)b * a.errstride, a.it, total, a.thr_q, &nr) != M_NORMAL) return;
    const int uc = (a.ctl[b].ubase ^ a.utog) & 1;
    const size_t po = (size_t)b * a.g.splane;
    const float* __restrict__ src = (plane ? a.sb.u2[uc] : a.sb.u1[uc]) + po;
    float* __restrict__ dst = (plane ? a.sb.u2[uc ^ 1] : a.sb.u1[uc ^ 1]) + po;
    median_block<KS>(t, src, dst, blockIdx.x * TWm, blockIdx.y * THm, a.g.w, a.g.h, a.g.pitch);
}

// stage end for the three-iterations-per-launch schedule: a pair took part in ceil(n_it / 3) passes (a REPLAY pass writes the half
// its overshoot pass wrote, so it does not count)
__global__ void k_stage_end3(const u64* __restrict__ err, int errstride, PairCtl* ctl, int* iters, int B,
                             int total, int inner, int median_on, double thr_q, int level, int warp, int nlev, int warps)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const u64* e = err + (size_t)b * errstride;
    int n_it = total;
    for (int j = 0; j < total; ++j)
        if (!((double)e[j] > thr_q)) { n_it = j + 1; break; }
    const int n_out = n_it > 0 ? (n_it - 1) / inner + 1 : 0;
    const int passes = (n_it + 2) / 3;
    PairCtl c = ctl[b];
    c.ubase = (c.ubase + passes + (median_on ? n_out : 0)) & 1;
    c.pbase = (c.pbase + passes) & 1;
    ctl[b] = c;
    int* o = iters + (((size_t)b * nlev + level) * warps + warp) * 2;
    o[0] = n_it; o[1] = n_out;
}
