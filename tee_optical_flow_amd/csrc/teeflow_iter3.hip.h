// tvl1_iter, THREE fused inner iterations per launch on full-width row strips (iter_variant 3).
//
// Same arithmetic, row march and NORMAL / REPLAY / EXIT protocol as k_iter2_rows (teeflow_kernels.hip.h) with one more
// pipeline stage: per pass a pixel's 9 input planes are read and its 6 state planes written once per THREE iterations
// (20 B per iteration instead of 30), and a stage takes a third fewer dependent launches.  Price: a second set of rolling
// dual rows and a third pair of iterate planes in LDS (two resident blocks per CU instead of three), 5 halo rows per strip
// instead of 3, and a stop on the first or second iteration of a pass is repaired by a REPLAY of one or two iterations.
//
//   step s:  stage 1  group s    primal 1 (from global)                     -> U1
//            stage 2  group s-1  dual 1 (U1)   [REPLAY 1: store]  primal 2  -> U2      (dual rows of iterate 1 in B*)
//            stage 3  group s-2  dual 2 (U2)   [REPLAY 2: store]  primal 3  -> U3      (dual rows of iterate 2 in C*)
//            stage 4  group s-3  dual 3 (U3), store
//
// Rows (strip = y0 .. y0+R-1):  primal 1 on y0-2 .. y0+R+2 | dual 1 .. y0+R+1 | primal 2 y0-1 .. y0+R+1 | dual 2 .. y0+R |
// primal 3 y0 .. y0+R | dual 3 + store y0 .. y0+R-1.  Each primal update adds its convergence term for rows y0 .. y0+R-1.
// Used when inner_iterations is a multiple of 3 (the median cadence stays on a pass boundary); other stages keep the
// two-iteration kernels.  Lock-step launches only (the free-running scheduler keeps k_iter2_q).
#pragma once

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

struct Iter3Args {
    IterArgs a;                           // a.it = first iteration of the pass (multiple of 3), a.utog/ptog/pzero for it
    int utog_prev, ptog_prev, pzero_prev; // the same three for the previous pass (used by REPLAY blocks)
    int total;                            // inner*outer
};

// strips of the three-iteration march: 5 halo rows each -> at least 6 steps per strip
TF_HD inline void strip_rule3(int n, int H, int RY, int slots, int* R, int* S)
{
    if (n < 1) n = 1;
    const int k = (n + slots - 1) / slots;
    int s = (int)(((long long)k * slots) / n);
    const int smax = H / (6 * RY) > 0 ? H / (6 * RY) : 1;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    int r = (H + s - 1) / s;
    *R = r;
    *S = (H + r - 1) / r;
}

__global__ __launch_bounds__(512) void k_iter3_rows(Iter3Args A, int R, int QX, int RY, int slots)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const IterArgs& a = A.a;
    u64* sred = reinterpret_cast<u64*>(smem);      // 3 x 8 x u64 = 192 B (one per wave and error sum)
    int nrep_dummy;
    if (a.host_slot && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) {
        int c = 0;
        for (int b2 = threadIdx.x; b2 < a.B; b2 += 64)
            c += pair_mode3(a.err + (size_t)b2 * a.errstride, a.it, A.total, a.thr_q, &nrep_dummy) != M_EXIT ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (threadIdx.x == 0) __hip_atomic_store(a.host_slot, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int b = blockIdx.z, strip = blockIdx.x;
    if (slots > 0) {
        const int nchunk = (a.B + 63) >> 6, nw = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
        for (int c = wv; c < nchunk; c += nw) {
            const int pb = c * 64 + ln;
            int nr;
            const bool on = pb < a.B && pair_mode3(a.err + (size_t)pb * a.errstride, a.it, A.total, a.thr_q, &nr) != M_EXIT;
            const u64 m = __ballot(on);
            if (ln == 0) sred[c] = m;
        }
        __syncthreads();
        int nact = 0;
        for (int c = 0; c < nchunk; ++c) nact += __popcll(sred[c]);
        int S;
        strip_rule3(nact, a.g.h, RY, slots, &R, &S);
        const int item = blockIdx.x;
        if (item >= nact * S) return;                      // block-uniform
        int k = item / S;
        strip = item - k * S;
        int c = 0;
        u64 m = sred[0];
        while (k >= __popcll(m)) { k -= __popcll(m); m = sred[++c]; }
        for (; k > 0; --k) m &= m - 1;
        b = c * 64 + (__ffsll((long long)m) - 1);
        __syncthreads();                                    // sred is reused for the error sums below
    }
    u64* errb = a.err + (size_t)b * a.errstride;
    int nrep;
    const int mode = pair_mode3(errb, a.it, A.total, a.thr_q, &nrep);     // block-uniform
    if (mode == M_EXIT) return;
    const bool replay = mode == M_REPLAY;
    const PairCtl cc = a.ctl[b];
    const int utog = replay ? A.utog_prev : a.utog, ptog = replay ? A.ptog_prev : a.ptog;
    const bool pzero = (replay ? A.pzero_prev : a.pzero) != 0;
    const int uc = (cc.ubase ^ utog) & 1, pc = (cc.pbase ^ ptog) & 1;

    const int LW = QX * 4 + 4;
    float* U1a = smem + 48;                        // [2][RY][LW] each: iterate 1 / 2 / 3, planes u1 (a) and u2 (b)
    float* U1b = U1a + 2 * RY * LW;
    float* U2a = U1b + 2 * RY * LW;
    float* U2b = U2a + 2 * RY * LW;
    float* U3a = U2b + 2 * RY * LW;
    float* U3b = U3a + 2 * RY * LW;
    float* B12 = U3b + 2 * RY * LW;                // [RY+1][LW] rolling rows of p1_12 / p1_22
    float* B22 = B12 + (RY + 1) * LW;
    float* C12 = B22 + (RY + 1) * LW;              // ... of p2_12 / p2_22
    float* C22 = C12 + (RY + 1) * LW;
    float* B11w = C22 + (RY + 1) * LW;             // [RY][QX] last element of each quad of p1_11 / p1_21
    float* B21w = B11w + RY * QX;
    float* C11w = B21w + RY * QX;                  // ... of p2_11 / p2_21
    float* C21w = C11w + RY * QX;
    const int tid = threadIdx.x;
    const int ty = tid / QX, tx = tid - ty * QX;
    const bool lane_on = ty < RY;
    const int W = a.g.w, H = a.g.h, pitch = a.g.pitch;
    const int x = tx * 4;
    const int y0 = strip * R;
    const int ngroups = (R + 5 + RY - 1) / RY;             // primal 1 covers rows y0-2 .. y0+R+2, groups start at y0-2
    const size_t po = (size_t)b * (size_t)a.g.splane;
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

    const int yout_hi = y0 + R - 1;

    // pipeline registers (written and read under the same validity predicates)
    float s1_wx[4], s1_wy[4], s1_r[4], s1_11[4], s1_12[4], s1_21[4], s1_22[4];     // stage 1 -> 2: constants, p0
    float s2_wx[4], s2_wy[4], s2_r[4], s2_11[4], s2_12[4], s2_21[4], s2_22[4];     // stage 2 -> 3: constants, p1
    float s3_11[4], s3_12[4], s3_21[4], s3_22[4];                                  // stage 3 -> 4: p2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s1_wx[i] = s1_wy[i] = s1_r[i] = s1_11[i] = s1_12[i] = s1_21[i] = s1_22[i] = 0.f;
        s2_wx[i] = s2_wy[i] = s2_r[i] = s2_11[i] = s2_12[i] = s2_21[i] = s2_22[i] = 0.f;
        s3_11[i] = s3_12[i] = s3_21[i] = s3_22[i] = 0.f;
    }
    bool s1_valid = false, s2_valid = false, s3_valid = false;
    unsigned inw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) inw[j] = opaque_u(x + j < W ? ~0u : 0u);
    int ring2 = (ty + 1) % RB;                            // (r mod RB) for r = (s-1)*RY + ty
    int ring3 = (ty + 2) % RB;                            // ... for r = (s-2)*RY + ty
    u64 qA = 0, qB = 0, qC = 0;
    double accA = 0.0, accB = 0.0, accC = 0.0;

    // dual update of one quad from an iterate held in LDS planes (Ua, Ub): own row in buffer `bc`, the row below either the
    // next row of the same buffer or row 0 of buffer bc^1 (the following group); returns the own quad in (w1, w2)
    auto dual = [&](const float* Ua, const float* Ub, int bc, int yrow, float (&w1)[4], float (&w2)[4],
                    const float (&i11)[4], const float (&i12)[4], const float (&i21)[4], const float (&i22)[4],
                    float (&q11)[4], float (&q12)[4], float (&q21)[4], float (&q22)[4]) {
        const bool lastrow = yrow >= H - 1;
        float4 dn1 = make_float4(0, 0, 0, 0), dn2 = dn1;
        if (!lastrow) {
            const float* d1 = ty < RY - 1 ? Ua + (bc * RY + ty + 1) * LW + x : Ua + ((bc ^ 1) * RY) * LW + x;
            const float* d2 = ty < RY - 1 ? Ub + (bc * RY + ty + 1) * LW + x : Ub + ((bc ^ 1) * RY) * LW + x;
            dn1 = ld4(d1); dn2 = ld4(d2);
        }
        float rr1 = 0.f, rr2 = 0.f;
        if (x + 4 < W) { rr1 = Ua[(bc * RY + ty) * LW + x + 4]; rr2 = Ub[(bc * RY + ty) * LW + x + 4]; }
        const unsigned mnl = opaque_u(lastrow ? 0u : ~0u);
        float dv1[4], dv2[4], u1x[4], u1y[4], u2x[4], u2y[4];
        UNPACK4(dv1, dn1) UNPACK4(dv2, dn2)
        {
            const float4 o1 = ld4(Ua + (bc * RY + ty) * LW + x), o2 = ld4(Ub + (bc * RY + ty) * LW + x);
            UNPACK4(w1, o1) UNPACK4(w2, o2)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e1 = i < 3 ? w1[i + 1] : rr1, e2 = i < 3 ? w2[i + 1] : rr2;
            u1x[i] = mask_f(e1 - w1[i], inw[i + 1]);          // 0 in the last column and beyond
            u2x[i] = mask_f(e2 - w2[i], inw[i + 1]);
            u1y[i] = mask_f(dv1[i] - w1[i], mnl);             // 0 in the last row
            u2y[i] = mask_f(dv2[i] - w2[i], mnl);
        }
        tv_p_quad_pk(a.taut, u1x, u1y, u2x, u2y, i11, i12, i21, i22, q11, q12, q21, q22);
    };
    // primal update of one quad from registers (own iterate, own dual quad, constants) + the dual row above / pixel to the left
    // from the rolling LDS rows (R12, R22, R11w, R21w); adds the convergence term for output rows
    auto primal = [&](const float* R12, const float* R22, const float* R11w, const float* R21w, int ring, int yrow,
                      const float (&w1)[4], const float (&w2)[4], const float (&cwx)[4], const float (&cwy)[4], const float (&cr)[4],
                      const float (&q11)[4], const float (&q12)[4], const float (&q21)[4], const float (&q22)[4],
                      float (&n1)[4], float (&n2)[4], double& acc) {
        float4 up12 = make_float4(0, 0, 0, 0), up22 = up12;
        if (yrow > 0) {
            const int ri = ring > 0 ? ring - 1 : RB - 1;
            up12 = ld4(R12 + ri * LW + x); up22 = ld4(R22 + ri * LW + x);
        }
        float l11 = 0.f, l21 = 0.f;
        if (tx > 0) { l11 = R11w[ty * QX + tx - 1]; l21 = R21w[ty * QX + tx - 1]; }
        QuadU qu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qu.u1k[i] = w1[i]; qu.u2k[i] = w2[i]; qu.wx[i] = cwx[i]; qu.wy[i] = cwy[i]; qu.r[i] = cr[i];
            qu.p11[i] = q11[i]; qu.p12[i] = q12[i]; qu.p21[i] = q21[i]; qu.p22[i] = q22[i];
        }
        UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
        qu.l11 = l11; qu.l21 = l21;
        const bool isout = yrow >= y0 && yrow <= yout_hi;
        tv_u_quad_pk(a.l_t, a.theta, qu, yrow == 0, x == 0, n1, n2);
        const unsigned mrow = opaque_u(isout ? ~0u : 0u);
        const unsigned keep[4] = {inw[0] & mrow, inw[1] & mrow, inw[2] & mrow, inw[3] & mrow};
        acc += tv_err_quad_pk(n1, w1, n2, w2, keep);
    };
    auto store_state = [&](int yrow, const float (&w1)[4], const float (&w2)[4],
                           const float (&q11)[4], const float (&q12)[4], const float (&q21)[4], const float (&q22)[4]) {
        const size_t prow = (size_t)yrow * pitch + x;
        st4(ou1 + prow, PACK4(w1)); st4(ou2 + prow, PACK4(w2));
        st4(o11 + prow, PACK4(q11)); st4(o12 + prow, PACK4(q12));
        st4(o21 + prow, PACK4(q21)); st4(o22 + prow, PACK4(q22));
    };

    for (int s = 0; s < ngroups + 3; ++s) {
        // ================= stage 1: group s, primal 1 from global memory =================
        const int y = y0 - 2 + s * RY + ty;
        const bool v1 = lane_on && s < ngroups && y >= 0 && y < H && y <= y0 + R + 2;
        const size_t row = (size_t)y * pitch + x;
        float4 u1q, u2q, wxq, wyq, rq, a11, a12, a21, a22;
        u1q = u2q = wxq = wyq = rq = a11 = a12 = a21 = a22 = make_float4(0, 0, 0, 0);
        if (v1) {
            u1q = ld4(gu1 + row); u2q = ld4(gu2 + row);
            wxq = ld4(gwx + row); wyq = ld4(gwy + row); rq = ld4(grh + row);
            if (!pzero) { a11 = ld4(g11 + row); a12 = ld4(g12 + row); a21 = ld4(g21 + row); a22 = ld4(g22 + row); }
        }
        float c11[4], c12[4], c21[4], c22[4], wxv[4], wyv[4], rv[4];
        UNPACK4(c11, a11) UNPACK4(c12, a12) UNPACK4(c21, a21) UNPACK4(c22, a22) UNPACK4(wxv, wxq) UNPACK4(wyv, wyq) UNPACK4(rv, rq)
        if (v1) {
            float4 up12 = make_float4(0, 0, 0, 0), up22 = up12;
            if (y > 0 && !pzero) { up12 = ld4(g12 + row - pitch); up22 = ld4(g22 + row - pitch); }
            float l11 = 0.f, l21 = 0.f;
            if (tx > 0 && !pzero) { l11 = g11[row - 1]; l21 = g21[row - 1]; }
            QuadU qu;
            UNPACK4(qu.u1k, u1q) UNPACK4(qu.u2k, u2q) UNPACK4(qu.wx, wxq) UNPACK4(qu.wy, wyq) UNPACK4(qu.r, rq)
            UNPACK4(qu.p11, a11) UNPACK4(qu.p12, a12) UNPACK4(qu.p21, a21) UNPACK4(qu.p22, a22)
            UNPACK4(qu.p12u, up12) UNPACK4(qu.p22u, up22)
            qu.l11 = l11; qu.l21 = l21;
            float n_u1[4], n_u2[4];
            const bool isout = y >= y0 && y <= yout_hi;
            tv_u_quad_pk(a.l_t, a.theta, qu, y == 0, x == 0, n_u1, n_u2);
            const unsigned mrow = opaque_u(!replay && isout ? ~0u : 0u);
            const unsigned keep[4] = {inw[0] & mrow, inw[1] & mrow, inw[2] & mrow, inw[3] & mrow};
            accA += tv_err_quad_pk(n_u1, qu.u1k, n_u2, qu.u2k, keep);
            st4(U1a + ((s & 1) * RY + ty) * LW + x, PACK4(n_u1));
            st4(U1b + ((s & 1) * RY + ty) * LW + x, PACK4(n_u2));
        }
        __syncthreads();
        // ================= stage 2: group s-1: dual 1, then primal 2 =================
        const int yb = y - RY;
        const bool v2 = s1_valid && yb <= y0 + R + 1;
        float p1_11[4] = {0, 0, 0, 0}, p1_12[4] = {0, 0, 0, 0}, p1_21[4] = {0, 0, 0, 0}, p1_22[4] = {0, 0, 0, 0};
        float w1_1[4] = {0, 0, 0, 0}, w1_2[4] = {0, 0, 0, 0};               // own quad of iterate 1
        if (v2) {
            dual(U1a, U1b, (s - 1) & 1, yb, w1_1, w1_2, s1_11, s1_12, s1_21, s1_22, p1_11, p1_12, p1_21, p1_22);
            if (nrep == 1) {
                if (yb >= y0 && yb <= yout_hi) store_state(yb, w1_1, w1_2, p1_11, p1_12, p1_21, p1_22);
            } else {
                st4(B12 + ring2 * LW + x, PACK4(p1_12));
                st4(B22 + ring2 * LW + x, PACK4(p1_22));
                B11w[ty * QX + tx] = p1_11[3];
                B21w[ty * QX + tx] = p1_21[3];
            }
        }
        bool v2u = false, v3u = false;
        float p2_11[4] = {0, 0, 0, 0}, p2_12[4] = {0, 0, 0, 0}, p2_21[4] = {0, 0, 0, 0}, p2_22[4] = {0, 0, 0, 0};
        if (nrep != 1) {
            __syncthreads();
            v2u = v2 && yb >= y0 - 1;                         // rows y0-1 .. y0+R+1 get the second primal update
            if (v2u) {
                float m_u1[4], m_u2[4];
                double acc = 0.0;
                primal(B12, B22, B11w, B21w, ring2, yb, w1_1, w1_2, s1_wx, s1_wy, s1_r, p1_11, p1_12, p1_21, p1_22, m_u1, m_u2, acc);
                if (!replay) accB += acc;
                st4(U2a + (((s - 1) & 1) * RY + ty) * LW + x, PACK4(m_u1));
                st4(U2b + (((s - 1) & 1) * RY + ty) * LW + x, PACK4(m_u2));
            }
            __syncthreads();
            // ================= stage 3: group s-2: dual 2, then primal 3 =================
            const int yc = y - 2 * RY;
            const bool v3 = s2_valid && yc <= y0 + R;
            float w2_1[4] = {0, 0, 0, 0}, w2_2[4] = {0, 0, 0, 0};           // own quad of iterate 2
            if (v3) {
                dual(U2a, U2b, s & 1, yc, w2_1, w2_2, s2_11, s2_12, s2_21, s2_22, p2_11, p2_12, p2_21, p2_22);
                if (nrep == 2) {
                    if (yc >= y0 && yc <= yout_hi) store_state(yc, w2_1, w2_2, p2_11, p2_12, p2_21, p2_22);
                } else {
                    st4(C12 + ring3 * LW + x, PACK4(p2_12));
                    st4(C22 + ring3 * LW + x, PACK4(p2_22));
                    C11w[ty * QX + tx] = p2_11[3];
                    C21w[ty * QX + tx] = p2_21[3];
                }
            }
            if (nrep == 0) {
                __syncthreads();
                v3u = v3 && yc >= y0;                         // rows y0 .. y0+R get the third primal update
                if (v3u) {
                    float m_u1[4], m_u2[4];
                    primal(C12, C22, C11w, C21w, ring3, yc, w2_1, w2_2, s2_wx, s2_wy, s2_r, p2_11, p2_12, p2_21, p2_22, m_u1, m_u2, accC);
                    st4(U3a + ((s & 1) * RY + ty) * LW + x, PACK4(m_u1));
                    st4(U3b + ((s & 1) * RY + ty) * LW + x, PACK4(m_u2));
                }
                __syncthreads();
                // ================= stage 4: group s-3: dual 3, store =================
                const int yd = y - 3 * RY;
                if (s3_valid && yd <= yout_hi) {
                    float w3_1[4], w3_2[4], r11[4], r12[4], r21[4], r22[4];
                    dual(U3a, U3b, (s - 1) & 1, yd, w3_1, w3_2, s3_11, s3_12, s3_21, s3_22, r11, r12, r21, r22);
                    store_state(yd, w3_1, w3_2, r11, r12, r21, r22);
                }
            }
        }
        else __syncthreads();       // REPLAY 1 has no other barrier between this step's LDS reads and the next step's stage-1 writes
        // ================= rotate the pipeline registers =================
        s3_valid = v3u;
        s2_valid = v2u;
        s1_valid = v1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s3_11[i] = p2_11[i]; s3_12[i] = p2_12[i]; s3_21[i] = p2_21[i]; s3_22[i] = p2_22[i];
            s2_11[i] = p1_11[i]; s2_12[i] = p1_12[i]; s2_21[i] = p1_21[i]; s2_22[i] = p1_22[i];
            s2_wx[i] = s1_wx[i]; s2_wy[i] = s1_wy[i]; s2_r[i] = s1_r[i];
            s1_wx[i] = wxv[i]; s1_wy[i] = wyv[i]; s1_r[i] = rv[i];
            s1_11[i] = c11[i]; s1_12[i] = c12[i]; s1_21[i] = c21[i]; s1_22[i] = c22[i];
        }
        ring3 = ring2;
        ring2 += RY; ring2 = ring2 >= RB ? ring2 - RB : ring2;
        if ((s & 255) == 255) { qA += (u64)accA; qB += (u64)accB; qC += (u64)accC; accA = accB = accC = 0.0; }   // keep the double sums exact
    }
    if (!replay) {
        qA += (u64)accA; qB += (u64)accB; qC += (u64)accC;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { qA += __shfl_down(qA, off, 64); qB += __shfl_down(qB, off, 64); qC += __shfl_down(qC, off, 64); }
        __syncthreads();
        if ((tid & 63) == 0) { sred[tid >> 6] = qA; sred[8 + (tid >> 6)] = qB; sred[16 + (tid >> 6)] = qC; }
        __syncthreads();
        if (tid == 0) {
            u64 ta = 0, tb = 0, tc = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ta += sred[w]; tb += sred[8 + w]; tc += sred[16 + w]; }
            atomicAdd(&errb[a.it], ta);
            atomicAdd(&errb[a.it + 1], tb);
            atomicAdd(&errb[a.it + 2], tc);
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

// stage end for the three-iterations-per-launch schedule: a pair took part in ceil(n_it/3) passes (a REPLAY pass writes
// the half its overshoot pass wrote, so it does not count)
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
