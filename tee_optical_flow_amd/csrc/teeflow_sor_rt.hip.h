// teeflow_sor_rt.hip.h -- red-black SOR of cv::VariationalRefinement with the whole region in REGISTERS (gfx950).
//
// What it replaces: the inner solver loop of cv2.optflow.createOptFlow_DeepFlow().calc()
// (/root/reference/optical_flow/calculate_optical_flow.py:568, 631 -> upstream VariationalRefinement's red-black SOR),
// restated in oracle/deepflow_oracle.c.  Same arithmetic contract as every kernel here: IEEE single precision, the
// oracle's operation order, no contraction -> bit-identical results.
//
// Why registers.  The LDS-tile form (k_df_sor_fused) is VALU-bound at ~200 executed lane-instructions per useful pixel
// update: 14 LDS reads with their address arithmetic, flag tests, two full IEEE divisions, and every halo pixel updated in
// every half-sweep.  An update itself is 2 x (4 mul + 3 add) + 2 x (3 + division + 3).  Here a wave owns 128 columns x R rows:
// lane l holds columns 2l, 2l+1 of R rows -- per pixel du, dv, the two diagonals pre-scaled by 2^64 with their refined
// reciprocals, a12, b1, b2 and the two edge weights -- so
//   * the vertical neighbours of a pixel are the thread's own registers, the horizontal ones are its own register or the
//     adjacent lane's (DPP wave_shr / wave_shl: no LDS, no address arithmetic);
//   * a missing neighbour is a zero edge weight times a zero value: no flags, no selects, no exec masking in the loop;
//   * each division is the 7-instruction tail of the hardware's correctly rounded sequence (the set-up -- scale, v_rcp, one
//     Newton step -- is done once per pixel and launch; see div1s in teeflow_deepflow.hip.h);
//   * row k of a thread holds one pixel of each colour, so every lane works in every half-sweep.
// NB waves stack their bands vertically to a 128 x (NB*R) region; only a band's first and last row are exchanged with the
// neighbouring bands, through LDS (one 8-byte write and read per edge row and half-sweep), with one barrier per half-sweep.
//
// Tiling (hl = 2 * sweeps per launch; hl = 0: the region is the whole level and any number of sweeps runs in one launch):
// region origins step by (128 - 2 hl, NB*R - 2 hl); a region edge that is the image border needs no halo, every other
// edge invalidates one more ring of pixels per half-sweep, so after `nsw` sweeps everything at least hl pixels away from
// such an edge equals what one-colour-per-launch SOR computes.  Only that core is written -- to (du2, dv2), because the
// neighbouring regions still read (du, dv) as their halo.
//
// Three launch forms share the band code below:
//   k_df_sor_rt<R,NB>            tiled: `nsw` sweeps per launch (hl = 2 nsw), or a whole level in one region (hl = 0, any number of sweeps)
//   k_df_sor_rt<R,NB,HALF>       whole levels at most 62 px wide: two bands per wave
//   k_df_sor_rt_coop<R,NB>       (end of this file) all sweeps of a fixed-point iteration in one launch: the regions are resident together
//                                and trade (du, dv) with the regions they overlap every S sweeps -- the default wherever it applies
#pragma once
#include "teeflow_deepflow.hip.h"

__device__ __forceinline__ float rt_from_left(float v)      // lane l receives lane l-1's value, lane 0 receives 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float rt_from_right(float v)     // lane l receives lane l+1's value, lane 63 receives 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// a / b for a diagonal prepared as bs = b * 2^64, rs = refined 1 / bs (FAST), or bs = b itself (plain IEEE division)
template <bool FAST>
__device__ __forceinline__ float rt_div(float a, float bs, float rs)
{
    if constexpr (FAST) {
        const float as = a * 0x1p64f;
        float q = as * rs;
        float e = __builtin_fmaf(-bs, q, as);
        q = __builtin_fmaf(e, rs, q);
        e = __builtin_fmaf(-bs, q, as);
        q = __builtin_fmaf(e, rs, q);
        return __builtin_amdgcn_div_fixupf(q, bs, as);
    } else {
        return a / bs;
    }
}

template <int R, int NB>
struct SorRtState {
    float du[R][2], dv[R][2];
    float s11[R][2], r11[R][2], s22[R][2], r22[R][2];
    float a12[R][2], b1[R][2], b2[R][2];
    float wx[R][2], wy[R][2];      // weight of the edge to the right / lower neighbour (0 if that neighbour is not in the image)
    float wxl[R];                  // wx of the pixel left of the thread's even column (the previous lane's odd column)
    float wyu[2];                  // wy of the pixels above the band's first row
};

// Issue the loads of a band (rows gy0 .. gy0+R-1, columns gx, gx+1) straight into the state registers, RAW: a11 / a22 land in
// s11 / s22, the pixel weights in wx, the weights of the row above the band in wyu.  Nothing is waited for here, so a band
// can be fetched while other bands are being swept; rt_band_setup() finishes the job.
template <int R, int NB, int K0 = 0, int N = R, bool WITH_DUV = true>
__device__ __forceinline__ void rt_band_issue_loads(SorRtState<R, NB>& t, const DfBufs& d, size_t po, int gy0, int gx, int W, int H, int pitch, bool has_above)
{
    // No branch here: a lane (or row) outside the image loads from the nearest 8-byte slot inside it and rt_band_setup_raw()
    // discards what it got -- 33 independent loads in flight per thread, no exec juggling.  Rows are padded to a multiple of 32
    // floats, so the slot at an odd W's last column stays inside the row.
    const unsigned gxc = (unsigned)(gx < W ? gx : (W - 1) & ~1);
#pragma unroll
    for (int k = K0; k < K0 + N; ++k) {
        const int gy = gy0 + k < H ? gy0 + k : H - 1;
        // uniform plane base (SGPRs) + 32-bit lane offset
        const unsigned i = (unsigned)gy * (unsigned)pitch + gxc;
        if constexpr (WITH_DUV) {
            const float2 vdu = *reinterpret_cast<const float2*>(d.du + po + i), vdv = *reinterpret_cast<const float2*>(d.dv + po + i);
            t.du[k][0] = vdu.x; t.du[k][1] = vdu.y; t.dv[k][0] = vdv.x; t.dv[k][1] = vdv.y;
        }
        const float2 vw = *reinterpret_cast<const float2*>(d.wg + po + i);
        const float2 v11 = *reinterpret_cast<const float2*>(d.A11 + po + i), v12 = *reinterpret_cast<const float2*>(d.A12 + po + i);
        const float2 v22 = *reinterpret_cast<const float2*>(d.A22 + po + i);
        const float2 vb1 = *reinterpret_cast<const float2*>(d.b1 + po + i), vb2 = *reinterpret_cast<const float2*>(d.b2 + po + i);
        t.s11[k][0] = v11.x; t.s11[k][1] = v11.y; t.s22[k][0] = v22.x; t.s22[k][1] = v22.y;
        t.a12[k][0] = v12.x; t.a12[k][1] = v12.y; t.b1[k][0] = vb1.x; t.b1[k][1] = vb1.y; t.b2[k][0] = vb2.x; t.b2[k][1] = vb2.y;
        t.wx[k][0] = vw.x; t.wx[k][1] = vw.y;
    }
    if (K0 == 0) {        // the weights of the row above the band (row 0 of the image when there is none: discarded)
        const int gy = has_above ? (gy0 - 1 < H ? gy0 - 1 : H - 1) : 0;
        const float2 vw = *reinterpret_cast<const float2*>(d.wg + po + ((unsigned)gy * (unsigned)pitch + gxc));
        t.wyu[0] = vw.x; t.wyu[1] = vw.y;
    }
}

// Turn the raw band into the sweep's form: pixels outside the image become inert (du = dv = 0, unit diagonal, zero weights),
// the weights become edge weights, and -- unless a diagonal leaves the range the pre-scaled division is exact for -- the
// diagonals are scaled by 2^64 and get their refined reciprocals.  Returns "take the plain-division path" (caller makes it uniform).
template <int R, int NB>
__device__ __forceinline__ int rt_band_setup_raw(SorRtState<R, NB>& t, int gy0, int gx, int W, int H, bool has_above)
{
    // Selects, not 0 / 1 factors: the odd column of an odd W's last lane was loaded from the row's padding, which may hold anything
    // (NaN included -- the planes are reused across pyramid levels with different pitches and start out uninitialised).
    int bad = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int gy = gy0 + k;
        const bool in[2] = {gy < H && gx < W, gy < H && gx + 1 < W};
        const float mdn = gy + 1 < H ? 1.f : 0.f, mx2 = gx + 2 < W ? 1.f : 0.f, m1 = in[1] ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            t.du[k][s] = in[s] ? t.du[k][s] : 0.f; t.dv[k][s] = in[s] ? t.dv[k][s] : 0.f;
            t.s11[k][s] = in[s] ? t.s11[k][s] : 1.f; t.s22[k][s] = in[s] ? t.s22[k][s] : 1.f;
            t.a12[k][s] = in[s] ? t.a12[k][s] : 0.f; t.b1[k][s] = in[s] ? t.b1[k][s] : 0.f; t.b2[k][s] = in[s] ? t.b2[k][s] : 0.f;
            t.wx[k][s] = in[s] ? t.wx[k][s] : 0.f;
        }
        const float w0 = t.wx[k][0], w1 = t.wx[k][1];
        t.wy[k][0] = w0 * mdn; t.wy[k][1] = w1 * mdn;
        t.wx[k][0] = w0 * m1;                                       // the even column's right neighbour is the odd column
        t.wx[k][1] = w1 * mx2;
        t.wxl[k] = rt_from_left(t.wx[k][1]);
        // the pre-scaled division is the exact quotient for 2^-24 < |b| < 2^60; stay well inside (pixels outside the image carry 1)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            bad |= !((int)(fabsf(t.s11[k][s]) > 0x1p-20f) & (int)(fabsf(t.s11[k][s]) < 0x1p50f) & (int)(fabsf(t.s22[k][s]) > 0x1p-20f) & (int)(fabsf(t.s22[k][s]) < 0x1p50f));     // '&': no short-circuit branches
    }
    const bool ab = has_above && gy0 < H;
    t.wyu[0] = ab && gx < W ? t.wyu[0] : 0.f; t.wyu[1] = ab && gx + 1 < W ? t.wyu[1] : 0.f;
    return bad;
}
template <int R, int NB, int K0 = 0, int N = R>
__device__ __forceinline__ void rt_band_scale(SorRtState<R, NB>& t)
{
#pragma unroll
    for (int k = K0; k < K0 + N; ++k)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            t.s11[k][s] *= 0x1p64f; t.r11[k][s] = rcp1s(t.s11[k][s]);
            t.s22[k][s] *= 0x1p64f; t.r22[k][s] = rcp1s(t.s22[k][s]);
        }
}

// one pixel update: row K of the band, column parity S; (upx, upy) / (dnx, dny) = (du, dv) of the pixel above the band's first /
// below its last row (used only by K == 0 / K == R - 1)
template <int R, int NB, int K, int S, bool FAST>
__device__ __forceinline__ void rt_update(SorRtState<R, NB>& t, float upx, float upy, float dnx, float dny, float omega)
{
    float dul, dvl, wl, dur, dvr;
    if constexpr (S == 1) { dul = t.du[K][0]; dvl = t.dv[K][0]; wl = t.wx[K][0]; dur = rt_from_right(t.du[K][0]); dvr = rt_from_right(t.dv[K][0]); }
    else { dul = rt_from_left(t.du[K][1]); dvl = rt_from_left(t.dv[K][1]); wl = t.wxl[K]; dur = t.du[K][1]; dvr = t.dv[K][1]; }
    const float wr = t.wx[K][S], wd = t.wy[K][S];
    constexpr int KU = K > 0 ? K - 1 : 0, KD = K < R - 1 ? K + 1 : K;
    const float wu_ = K > 0 ? t.wy[KU][S] : t.wyu[S];
    const float duu = K > 0 ? t.du[KU][S] : upx, dvu = K > 0 ? t.dv[KU][S] : upy;
    const float dud = K < R - 1 ? t.du[KD][S] : dnx, dvd = K < R - 1 ? t.dv[KD][S] : dny;
    const float sigmaU = wl * dul + wr * dur + wu_ * duu + wd * dud;
    const float sigmaV = wl * dvl + wr * dvr + wu_ * dvu + wd * dvd;
    float du = t.du[K][S], dv = t.dv[K][S];
    du += omega * (rt_div<FAST>(sigmaU + t.b1[K][S] - dv * t.a12[K][S], t.s11[K][S], t.r11[K][S]) - du);
    dv += omega * (rt_div<FAST>(sigmaV + t.b2[K][S] - du * t.a12[K][S], t.s22[K][S], t.r22[K][S]) - dv);
    t.du[K][S] = du; t.dv[K][S] = dv;
}

// rows K0 .. K0+N-1 of the band, colour C (the active pixel of row k has column parity (k + C) & 1)
template <int R, int NB, int K0, int N, int C, bool FAST>
__device__ __forceinline__ void rt_update_rows(SorRtState<R, NB>& t, float2 up, float2 dn, float omega)
{
    if constexpr (N > 0) {
        rt_update<R, NB, K0, (K0 + C) & 1, FAST>(t, up.x, up.y, dn.x, dn.y, omega);
        rt_update_rows<R, NB, K0 + 1, N - 1, C, FAST>(t, up, dn, omega);
    }
}

// write the band's rows inside [ylo, yhi) x [xlo, xhi) to (du2, dv2)
template <int R, int NB>
__device__ __forceinline__ void rt_band_store(const SorRtState<R, NB>& t, const DfBufs& d, size_t po, int gy0, int gx, int W, int H, int pitch, int xlo, int xhi, int ylo, int yhi)
{
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int gy = gy0 + k;
        if (gy < ylo || gy >= yhi || gy >= H) continue;
        const bool w0 = gx >= xlo && gx < xhi && gx < W, w1 = gx + 1 >= xlo && gx + 1 < xhi && gx + 1 < W;
        const unsigned i = (unsigned)gy * (unsigned)pitch + (unsigned)gx;
        float* pu = d.du2 + po; float* pv = d.dv2 + po;
        if (w0 && w1) {
            *reinterpret_cast<float2*>(pu + i) = make_float2(t.du[k][0], t.du[k][1]);
            *reinterpret_cast<float2*>(pv + i) = make_float2(t.dv[k][0], t.dv[k][1]);
        } else if (w0) { pu[i] = t.du[k][0]; pv[i] = t.dv[k][0]; }
        else if (w1) { pu[i + 1] = t.du[k][1]; pv[i + 1] = t.dv[k][1]; }
    }
}
// `nsw` sweeps.  Half-sweep j only has to be right within 2 nsw - 1 - j rows of the core [ylo, yhi) (the rest of the halo has
// done its job: what it feeds is one row nearer per half-sweep), so a band that lies wholly outside that range sits the
// half-sweep out -- ~1/8 of the updates of a 64-row region at nsw = 5.  (No such luck in x: the halo columns are lanes.)
// wv: the band of this lane (wave-uniform unless the wave carries two bands), gy0 / ROWS: first row and row count of the WAVE
template <int R, int NB, bool FAST, int ROWS = R>
__device__ __forceinline__ void sor_rt_sweeps(SorRtState<R, NB>& t, float2 (*exT)[2][64], float2 (*exB)[2][64], int wv, int ln, float omega, int nsw, bool live,
                                              int gy0, int ylo, int yhi)
{
#pragma unroll 1
    for (int sw = 0; sw < nsw; ++sw) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int st = c, sb = (R - 1 + c) & 1;            // column parity of the active pixel in the band's first / last row
            const int m = 2 * (nsw - sw) - 1 - c;
            if (live && gy0 + ROWS > ylo - m && gy0 < yhi + m) {
                const float2 up = exB[wv][st][ln], dn = exT[wv + 1][sb][ln];
                // Order within the half-sweep (the pixels of one colour do not depend on each other): an inner row first -- its 40
                // instructions cover the latency of the two LDS reads every wave of the block issues right behind the barrier --, then the
                // two edge rows and their publication, then the other inner rows, which cover the LDS writes before the next barrier.
                // (sched_barrier: the machine scheduler would otherwise put the edge rows first and the LDS writes last again.)
                if constexpr (R >= 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == 0) rt_update_rows<R, NB, 1, 1, 0, FAST>(t, up, dn, omega);
                    else rt_update_rows<R, NB, 1, 1, 1, FAST>(t, up, dn, omega);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == 0) { rt_update_rows<R, NB, 0, 1, 0, FAST>(t, up, dn, omega); rt_update_rows<R, NB, R - 1, 1, 0, FAST>(t, up, dn, omega); }
                    else { rt_update_rows<R, NB, 0, 1, 1, FAST>(t, up, dn, omega); rt_update_rows<R, NB, R - 1, 1, 1, FAST>(t, up, dn, omega); }
                    exT[wv][st][ln] = make_float2(t.du[0][st], t.dv[0][st]);
                    exB[wv + 1][sb][ln] = make_float2(t.du[R - 1][sb], t.dv[R - 1][sb]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == 0) rt_update_rows<R, NB, 2, R - 3, 0, FAST>(t, up, dn, omega);
                    else rt_update_rows<R, NB, 2, R - 3, 1, FAST>(t, up, dn, omega);
                } else {
                    if (c == 0) rt_update_rows<R, NB, 0, R, 0, FAST>(t, up, dn, omega);
                    else rt_update_rows<R, NB, 0, R, 1, FAST>(t, up, dn, omega);
                    exT[wv][st][ln] = make_float2(t.du[0][st], t.dv[0][st]);
                    exB[wv + 1][sb][ln] = make_float2(t.du[R - 1][sb], t.dv[R - 1][sb]);
                }
            }
            __syncthreads();
        }
    }
}

// grid (tiles x, tiles y, pairs), 64 * NB threads.  R even (row parity of a thread's rows must not depend on the band).
// HALF: a level at most 62 px wide uses at most 31 of a wave's 64 lanes, so each wave carries TWO bands: lanes 0-31 hold band 2w, lanes
// 32-63 band 2w + 1 (columns 2 (lane & 31) ..).  Nothing else changes: lane 31's columns (62, 63) lie outside the image, so it carries zero
// values and zero weights, and that is all lane 32 sees of it through the DPP shift (and vice versa).  Whole-level launches only (hl = 0,
// grid 1 x 1): half the waves for the 25 sweeps of the small pyramid levels, which are bound by the VALU throughput of ONE CU.
template <int R, int NB, bool HALF = false>
__global__ __launch_bounds__(64 * NB) void k_df_sor_rt(DfBufs d, Geom g, float omega, int nsw, int hl, int plain_div)
{
    static_assert(R % 2 == 0, "rows per band must be even");
    constexpr int RW = 128, NBANDS = HALF ? 2 * NB : NB, RH = R * NBANDS;
    __shared__ float2 exT[NBANDS + 1][2][64];   // [band]: first row of the band (read by the band above); [NBANDS] stays zero
    __shared__ float2 exB[NBANDS + 1][2][64];   // [band + 1]: last row of the band (read by the band below); [0] stays zero
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.z;
    const int band = HALF ? 2 * wv + (int)((threadIdx.x >> 5) & 1) : wv, ln = HALF ? threadIdx.x & 31 : threadIdx.x & 63;
    const int W = g.w, H = g.h, pitch = g.pitch;
    const int x0 = blockIdx.x * (RW - 2 * hl), y0 = blockIdx.y * (RH - 2 * hl);          // both even
    const int gx = x0 + 2 * ln, gy0 = y0 + band * R;
    const size_t po = (size_t)b * g.splane;
    SorRtState<R, NB> t;
    rt_band_issue_loads(t, d, po, gy0, gx, W, H, pitch, band > 0);
    const int bad = rt_band_setup_raw(t, gy0, gx, W, H, band > 0) | plain_div;    // plain_div: tests force the plain-division path
    exT[band][0][ln] = make_float2(t.du[0][0], t.dv[0][0]); exT[band][1][ln] = make_float2(t.du[0][1], t.dv[0][1]);
    exB[band + 1][0][ln] = make_float2(t.du[R - 1][0], t.dv[R - 1][0]); exB[band + 1][1][ln] = make_float2(t.du[R - 1][1], t.dv[R - 1][1]);
    if (wv == 0) {
        exT[NBANDS][0][ln] = exT[NBANDS][1][ln] = make_float2(0, 0);
        exB[0][0][ln] = exB[0][1][ln] = make_float2(0, 0);
    }
    const bool slow = __builtin_amdgcn_readfirstlane(__syncthreads_or(bad)) != 0;      // block-uniform; also the barrier behind the LDS fill
    const int gyw = y0 + (HALF ? 2 * wv : wv) * R;                                      // first row of the wave (HALF: of its upper band)
    const bool live = gyw < H;                                                          // wave-uniform: nothing to update below the image
    // the core: at least hl pixels away from every region edge that is not the image border
    const int xlo = blockIdx.x == 0 ? 0 : x0 + hl, xhi = blockIdx.x == gridDim.x - 1 ? W : x0 + RW - hl;
    const int ylo = blockIdx.y == 0 ? 0 : y0 + hl, yhi = blockIdx.y == gridDim.y - 1 ? H : y0 + RH - hl;
    if (!slow) {
        rt_band_scale(t);
        sor_rt_sweeps<R, NB, true, HALF ? 2 * R : R>(t, exT, exB, band, ln, omega, nsw, live, gyw, ylo, yhi);
    } else {
        sor_rt_sweeps<R, NB, false, HALF ? 2 * R : R>(t, exT, exB, band, ln, omega, nsw, live, gyw, ylo, yhi);
    }
    rt_band_store(t, d, po, gy0, gx, W, H, pitch, xlo, xhi, ylo, yhi);
}

// =================================================================================================
// All sweeps of a fixed-point iteration in ONE launch: co-resident regions that trade du, dv through memory.
//
// The tiled form above ends the kernel every S sweeps only because a region's halo has gone stale; the next launch loads all 8 planes
// again although 6 of them (the linear system) have not changed.  Here every region of `nb` pairs is a block that is resident at the
// same time (the host sizes the grid to the CUs it may use: one 1024-thread block per CU), keeps the system in registers for all the
// sweeps and, every S sweeps ("phase"), publishes the core of (du, dv) and re-reads its region -- 8 bytes per pixel instead of 40, and
// the wait is for the at most 8 regions it overlaps, not for the grid:
//   * phase p reads buffer (p-1)&1 and writes its core to buffer p&1 ((du,dv) = buffer 0, (du2,dv2) = buffer 1; the cores partition
//     the level, so a buffer is complete again after every phase).  A region overwrites buffer p&1 only after its neighbours have
//     raised flag p-1, i.e. after they have finished reading that buffer at the start of phase p-1: no write-after-read hazard;
//   * du, dv go through memory with agent-scope (sc1) accesses: the stores are written through, the loads do not hit lines another XCD's
//     L2 may hold stale, and a wave's stores are acknowledged (s_waitcnt vmcnt(0)) before the block raises its flag -- no L2 write-back /
//     invalidate (an agent-scope fence per wave costs ~200 us per meeting on this chip, tools/microbench/gridsync_probe.hip; this
//     exchange ~1.5 us on top of the sweeps);
//   * flags: one 128-byte line per block holding `base + phase` (base grows from launch to launch, the lines are never cleared);
//   * every wait is bounded: a block that has polled ~0.1 s raises the launch's abort word and leaves, every other block sees the
//     word in its own wait and leaves too; the host then repeats the solve with the tiled form.  (Cannot happen while the grid fits
//     the CUs this handle was given; a second process on the same GPU can make it happen.)
// Bit-identical to the tiled form by construction: same regions, same halo, same sweeps.
// The pointers reach these helpers through a run-time choice between the two (du, dv) buffers, which makes hipcc forget that they are
// global memory and emit flat_load / flat_store.  The hand-off this exchange relies on is documented for global_ / buffer_ `sc1` accesses
// only (MI355X_MICROARCH.md, inter-workgroup visibility: "never flat_"), so the address space is stated.
typedef __attribute__((address_space(1))) unsigned long long rt_gu64;
typedef __attribute__((address_space(1))) float rt_gf32;
typedef __attribute__((address_space(1))) unsigned rt_gu32;
__device__ __forceinline__ float2 rt_ld_sc1(const float* p)
{
    const unsigned long long v = __hip_atomic_load((const rt_gu64*)(const void*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void rt_st_sc1(float* p, float a, float b)
{
    __hip_atomic_store((rt_gu64*)(void*)p, (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rt_st_sc1(float* p, float a) { __hip_atomic_store((rt_gf32*)(void*)p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rt_ld_flag(const unsigned* p) { return __hip_atomic_load((const rt_gu32*)(const void*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rt_st_flag(unsigned* p, unsigned v) { __hip_atomic_store((rt_gu32*)(void*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef TF_COOP_TIMING
__device__ unsigned long long g_coop_t[4][64];      // [block sample][event]: s_memrealtime (100 MHz) stamps of wave 0
#define COOP_T(i) do { if (wv == 0 && ln == 0 && tslot >= 0 && (i) < 64) g_coop_t[tslot][(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define COOP_TE() do { COOP_T(te); ++te; } while (0)
#else
#define COOP_T(i) do { } while (0)
#define COOP_TE() do { } while (0)
#endif
constexpr int RT_COOP_POLL_LIMIT = 1 << 16;      // x (64-clock sleep + one load round trip) ~ 0.1 s

// the phases of one block (see k_df_sor_rt_coop); one instantiation per division form so that the loop carries ONE form of the state
template <int R, int NB, bool FAST>
__device__ __forceinline__ void sor_rt_coop_phases(SorRtState<R, NB>& t, float2 (*exT)[2][64], float2 (*exB)[2][64], int* gone, float* const (&bufu)[2], float* const (&bufv)[2],
                                                   int wv, int ln, int gx, unsigned gxc, int gy0, int W, int H, int pitch, float omega, int nsw, int S,
                                                   int xlo, int xhi, int ylo, int yhi, unsigned* __restrict__ flags, unsigned base, unsigned* __restrict__ abort_word, bool mute)
{
    const bool live = gy0 < H;
    const int me = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#ifdef TF_COOP_TIMING
    const int tslot = me == 0 ? 0 : (me == (int)(gridDim.x * gridDim.y) / 2 ? 1 : (me == (int)(gridDim.x * gridDim.y * gridDim.z) - 1 ? 2 : -1));
    int te = 2;
#endif
    int left = nsw;
#pragma unroll 1
    for (int phase = 1;; ++phase) {
        const int n = left < S ? left : S;
        exT[wv][0][ln] = make_float2(t.du[0][0], t.dv[0][0]); exT[wv][1][ln] = make_float2(t.du[0][1], t.dv[0][1]);
        exB[wv + 1][0][ln] = make_float2(t.du[R - 1][0], t.dv[R - 1][0]); exB[wv + 1][1][ln] = make_float2(t.du[R - 1][1], t.dv[R - 1][1]);
        __syncthreads();
        COOP_TE();
        sor_rt_sweeps<R, NB, FAST>(t, exT, exB, wv, ln, omega, n, live, gy0, ylo, yhi);
        COOP_TE();
        // the core goes to buffer phase & 1
        float* pu = bufu[phase & 1]; float* pv = bufv[phase & 1];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int gy = gy0 + k;
            if (gy < ylo || gy >= yhi || gy >= H) continue;
            const bool w0 = gx >= xlo && gx < xhi && gx < W, w1 = gx + 1 >= xlo && gx + 1 < xhi && gx + 1 < W;
            const unsigned i = (unsigned)gy * (unsigned)pitch + (unsigned)gx;
            if (w0 && w1) { rt_st_sc1(pu + i, t.du[k][0], t.du[k][1]); rt_st_sc1(pv + i, t.dv[k][0], t.dv[k][1]); }
            else if (w0) { rt_st_sc1(pu + i, t.du[k][0]); rt_st_sc1(pv + i, t.dv[k][0]); }
            else if (w1) { rt_st_sc1(pu + i + 1, t.du[k][1]); rt_st_sc1(pv + i + 1, t.dv[k][1]); }
        }
        left -= n;
        if (left <= 0) break;
        // meet the regions this one overlaps
        __builtin_amdgcn_s_waitcnt(0);                  // this wave's stores are acknowledged
        COOP_TE();
        __syncthreads();
        COOP_TE();
        if (wv == 0) {
            const unsigned target = base + (unsigned)phase;
            if (ln == 0 && !(mute && me == 0)) rt_st_flag(flags + (size_t)me * 32, target);
            const int q = ln < 9 ? ln : 4;
            const int nx = (int)blockIdx.x + q % 3 - 1, ny = (int)blockIdx.y + q / 3 - 1;
            const bool ex = nx >= 0 && nx < (int)gridDim.x && ny >= 0 && ny < (int)gridDim.y;
            const unsigned* f = flags + (size_t)(ex ? (blockIdx.z * gridDim.y + ny) * gridDim.x + nx : me) * 32;
            int polls = 0;
            bool ok;
            unsigned ab;
            while (true) {
                const unsigned v = rt_ld_flag(f);
                ab = rt_ld_flag(abort_word);
                ok = __all((int)(v - target) >= 0) != 0;
                if (ok || ab != 0 || ++polls > RT_COOP_POLL_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && ln == 0) {
                if (ab == 0) rt_st_flag(abort_word, 1u);
                *gone = 1;
            }
        }
        COOP_TE();
        __syncthreads();
        if (*gone) return;                                // block-uniform; the host repeats the solve with the tiled form
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int gy = gy0 + k < H ? gy0 + k : H - 1;
            const unsigned i = (unsigned)gy * (unsigned)pitch + gxc;
            const float2 vu = rt_ld_sc1(bufu[phase & 1] + i), vv = rt_ld_sc1(bufv[phase & 1] + i);
            const bool in0 = gy0 + k < H && gx < W, in1 = gy0 + k < H && gx + 1 < W;
            t.du[k][0] = in0 ? vu.x : 0.f; t.dv[k][0] = in0 ? vv.x : 0.f;
            t.du[k][1] = in1 ? vu.y : 0.f; t.dv[k][1] = in1 ? vv.y : 0.f;
        }
#ifdef TF_COOP_TIMING
        if (wv == 0 && ln == 0 && tslot >= 0 && te < 64) g_coop_t[tslot][te] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(__float_as_uint(t.du[0][0]) & 0);   // after the reload has landed
        ++te;
#endif
    }
}

template <int R, int NB>
__global__ __launch_bounds__(64 * NB, 4) void k_df_sor_rt_coop(DfBufs d, Geom g, float omega, int nsw, int S, int plain_div, int pair0,
                                                            unsigned* __restrict__ flags, unsigned base, unsigned* __restrict__ abort_word)
{
    static_assert(R % 2 == 0, "rows per band must be even");
    constexpr int RW = 128, RH = R * NB;
    __shared__ float2 exT[NB + 1][2][64];
    __shared__ float2 exB[NB + 1][2][64];
    __shared__ int gone;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = threadIdx.x & 63, b = pair0 + blockIdx.z;
    const int W = g.w, H = g.h, pitch = g.pitch, hl = 2 * S;
    const int x0 = blockIdx.x * (RW - 2 * hl), y0 = blockIdx.y * (RH - 2 * hl);
    const int gx = x0 + 2 * ln, gy0 = y0 + wv * R;
    const size_t po = (size_t)b * g.splane;
#ifdef TF_COOP_TIMING
    const int me_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int tslot = me_ == 0 ? 0 : (me_ == (int)(gridDim.x * gridDim.y) / 2 ? 1 : (me_ == (int)(gridDim.x * gridDim.y * gridDim.z) - 1 ? 2 : -1));
    COOP_T(0);
#endif
    SorRtState<R, NB> t;
    rt_band_issue_loads<R, NB, 0, R, false>(t, d, po, gy0, gx, W, H, pitch, wv > 0);         // the system; every access to du, dv is sc1
    const unsigned gxc = (unsigned)(gx < W ? gx : (W - 1) & ~1);
    float* const bufu[2] = {d.du + po, d.du2 + po};
    float* const bufv[2] = {d.dv + po, d.dv2 + po};
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int gy = gy0 + k < H ? gy0 + k : H - 1;
        const unsigned i = (unsigned)gy * (unsigned)pitch + gxc;
        const float2 vu = rt_ld_sc1(bufu[0] + i), vv = rt_ld_sc1(bufv[0] + i);
        t.du[k][0] = vu.x; t.du[k][1] = vu.y; t.dv[k][0] = vv.x; t.dv[k][1] = vv.y;
    }
    const int bad = rt_band_setup_raw(t, gy0, gx, W, H, wv > 0) | (plain_div & 1);     // plain_div bit 1 (tests): block 0 never raises its flag
    if (threadIdx.x == 0) gone = 0;
    if (wv == 0) {
        exT[NB][0][ln] = exT[NB][1][ln] = make_float2(0, 0);
        exB[0][0][ln] = exB[0][1][ln] = make_float2(0, 0);
    }
    const bool slow = __builtin_amdgcn_readfirstlane(__syncthreads_or(bad)) != 0;
    COOP_T(1);
    const int xlo = blockIdx.x == 0 ? 0 : x0 + hl, xhi = blockIdx.x == gridDim.x - 1 ? W : x0 + RW - hl;
    const int ylo = blockIdx.y == 0 ? 0 : y0 + hl, yhi = blockIdx.y == gridDim.y - 1 ? H : y0 + RH - hl;
    if (!slow) {
        rt_band_scale(t);
        sor_rt_coop_phases<R, NB, true>(t, exT, exB, &gone, bufu, bufv, wv, ln, gx, gxc, gy0, W, H, pitch, omega, nsw, S, xlo, xhi, ylo, yhi, flags, base, abort_word, (plain_div & 2) != 0);
    } else {
        sor_rt_coop_phases<R, NB, false>(t, exT, exB, &gone, bufu, bufv, wv, ln, gx, gxc, gy0, W, H, pitch, omega, nsw, S, xlo, xhi, ylo, yhi, flags, base, abort_word, (plain_div & 2) != 0);
    }
}
