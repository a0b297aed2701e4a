// teeflow_sched.hip.h -- free-running pair scheduler of the DualTVL1 engine (gfx950).
//
// The lock-step driver (run_stage in teeflow.hip) moves a whole batch through one (level, warp) stage at a time: a stage
// lasts as long as its slowest pair, and its last launches run on a handful of pairs.  Frame pairs are independent
// (reference loop /root/reference/optical_flow/calculate_optical_flow.py:584-597; useInitialFlow is never set), so here
// every pair walks through its own stages: the device keeps one small state machine per pair, and one SUPER-STEP
//     k_sched   (1 block)   advance every pair by the phase it has just finished, size the strips, build the work lists
//     k_misc_q              warp / 5x5 median / flow upsample / output tiles of the pairs that are in those phases
//     k_iter2_q             two fused inner iterations (or one REPLAY iteration) for the pairs that iterate
// advances every pair by exactly one phase, whatever level, warp or iteration it is at.  A launch therefore carries the
// work of ALL pairs that are not finished, the machine stays full until the last pairs drain, and the number of launches
// is the longest pair's chain instead of the sum over stages of the slowest pair.  The arithmetic is the same device code
// as in the lock-step kernels (iter2_rows_body, warp_px, median_block, resize_px): results do not change by a bit.
//
// Per-pair stop logic is the one of k_iter2_rows / k_stage_end2 (NORMAL / REPLAY / EXIT from the exact error sums),
// evaluated once per pair and super-step in k_sched instead of by every block.
#pragma once
#include "teeflow_kernels.hip.h"

#define SC_MAXLEV 8
enum { PH_WARP = 0, PH_MEDIAN = 1, PH_ITER = 2, PH_REPLAY = 3, PH_UP = 4, PH_OUT = 5, PH_DONE = 6 };
enum { SC_N_ITER_ITEMS = 0, SC_N_MISC_ITEMS = 1, SC_N_MISC_PAIRS = 2, SC_N_NOT_DONE = 3, SC_CNT_WORDS = 8 };

struct PairSt {
    int level, warp;     // pyramid level (0 = full size) and warp of the stage the pair is in
    int it;              // ITER: first of the two iterations to run; REPLAY: the iteration to re-run is it - 2
    int phase;           // what the coming super-step does for this pair (PH_*)
    int ucur, pcur;      // ping-pong halves that hold the pair's current u / p
    int strips, rows;    // ITER / REPLAY: strip count and rows per strip chosen for the coming launch
};

struct SchedLevel {
    int w, h, pitch, qx, ry, smax;
    long long plane;          // floats per frame of this level's pyramid buffer
    double thr_q;             // epsilon^2 * area * 2^30
    double up_sx, up_sy;      // resize scale from this level to the next finer one (unused at level 0)
};

struct SchedArgs {
    PairSt* st; int B;
    u64* err; int errstride;
    int* iters;               // [B][nlev][warps][2] executed (inner, outer) counts
    int nlev, warps, inner, total, median_on;
    int slots;                // resident k_iter2_q blocks on the device: the strips of a launch fill one round of them
    int first;                // first super-step of a solve: initialise the states
    int* host_slot;           // host-mapped word: pairs that are not DONE after this super-step's transitions
    int* cnt;                 // [SC_CNT_WORDS]
    int* iter_items; int iter_cap;   // pair | strip << 16
    int* misc_pair;           // [B]   pairs with a warp / median / up / out phase, in pair order
    int* misc_pref;           // [B+1] cumulative 64x16-tile counts
    double* work;             // [max super-steps] pixel-iterations the super-step's k_iter2_q launch executes
    int ss;
    SchedLevel lv[SC_MAXLEV];
};

// block-uniform values that come out of global memory: tell the compiler so (they then live in SGPRs, and so does
// everything derived from them -- geometry, strides, plane pointers)
__device__ __forceinline__ int sc_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ PairSt sc_uni(const PairSt& s)
{
    PairSt r;
    r.level = sc_uni(s.level); r.warp = sc_uni(s.warp); r.it = sc_uni(s.it); r.phase = sc_uni(s.phase);
    r.ucur = sc_uni(s.ucur); r.pcur = sc_uni(s.pcur); r.strips = sc_uni(s.strips); r.rows = sc_uni(s.rows);
    return r;
}

__device__ __forceinline__ int sc_tiles(int w, int h) { return ((w + 63) >> 6) * ((h + 15) >> 4); }

// exclusive scan over the block (blockDim <= 1024) + block total
__device__ __forceinline__ int sc_block_scan(int v, int* lds16, int* total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)((blockDim.x + 63) >> 6);
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) lds16[wv] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) { const int t = lds16[i]; if (i < wv) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(1024) void k_sched(SchedArgs A)
{
    __shared__ int lds16[16];
    const int b = threadIdx.x;
    const bool on = b < A.B;
    PairSt s;
    s.level = s.warp = s.it = s.ucur = s.pcur = s.strips = s.rows = 0; s.phase = PH_DONE;
    if (on) {
        if (A.first) { s.level = A.nlev - 1; s.phase = PH_WARP; }
        else {
            s = A.st[b];
            u64* e = A.err + (size_t)b * A.errstride;
            bool stage_end = false;
            int n_it = 0;
            switch (s.phase) {
                case PH_WARP: s.it = 0; s.phase = A.median_on ? PH_MEDIAN : PH_ITER; break;
                case PH_MEDIAN: s.ucur ^= 1; s.phase = PH_ITER; break;
                case PH_ITER: {
                    s.ucur ^= 1; s.pcur ^= 1; s.it += 2;
                    const double thr = A.lv[s.level].thr_q;
                    const bool a2 = (double)e[s.it - 2] > thr, a1 = (double)e[s.it - 1] > thr;
                    if (s.it < A.total && a2 && a1) s.phase = (A.median_on && s.it % A.inner == 0) ? PH_MEDIAN : PH_ITER;
                    else if (!a2) s.phase = PH_REPLAY;                     // stopped at the first of the two: it - 2 is re-run alone
                    else { stage_end = true; n_it = a1 ? A.total : s.it; }
                    break;
                }
                case PH_REPLAY: stage_end = true; n_it = s.it - 1; break;
                case PH_UP: s.level -= 1; s.warp = 0; s.ucur ^= 1; s.pcur = 0; s.phase = PH_WARP; break;
                case PH_OUT: s.phase = PH_DONE; break;
                default: break;
            }
            if (stage_end) {
                int* o = A.iters + (((size_t)b * A.nlev + s.level) * A.warps + s.warp) * 2;
                o[0] = n_it; o[1] = n_it > 0 ? (n_it - 1) / A.inner + 1 : 0;
                const int used = s.it < A.total ? s.it : A.total;             // slots a launch of this stage may have written
                for (int j = 0; j < used; ++j) e[j] = 0;
                if (s.warp + 1 < A.warps) { s.warp += 1; s.phase = PH_WARP; }
                else s.phase = s.level > 0 ? PH_UP : PH_OUT;
            }
        }
    }
    // ---- strips of the iterating pairs ----
    // A block's time is its number of pipeline steps (a step = RY rows of QX quads on 256 threads, whatever the level), so
    // the launch is balanced when every block marches the same number of steps T: a strip of R rows takes
    // ceil((R + 3) / RY) + 2 steps (3 halo rows, 2 steps of pipeline fill).  T is the smallest step count for which the
    // strips of all iterating pairs fit one round of `slots` resident blocks (two rounds if there are more pairs than slots).
    const bool iter = on && (s.phase == PH_ITER || s.phase == PH_REPLAY);
    const SchedLevel L = A.lv[on && s.level >= 0 ? s.level : 0];
    const double px = iter ? (double)L.w * (double)L.h : 0.0;
    __shared__ double dsum[16], dwork[16];
    __shared__ int nsum[16], lsum[16], ssum[16];
    const int nwv = (int)((blockDim.x + 63) >> 6);
    double rowsteps = iter ? (double)L.h / (double)L.ry : 0.0, wwork = iter ? px * (s.phase == PH_ITER ? 2.0 : 1.0) : 0.0;
    int n_iter = iter ? 1 : 0, n_live = on && s.phase != PH_DONE ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rowsteps += __shfl_down(rowsteps, o, 64); wwork += __shfl_down(wwork, o, 64);
        n_iter += __shfl_down(n_iter, o, 64); n_live += __shfl_down(n_live, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { dsum[threadIdx.x >> 6] = rowsteps; dwork[threadIdx.x >> 6] = wwork; nsum[threadIdx.x >> 6] = n_iter; lsum[threadIdx.x >> 6] = n_live; }
    __syncthreads();
    double tot_steps = 0.0, tot_work = 0.0; int tot_iter = 0, tot_live = 0;
    for (int i = 0; i < nwv; ++i) { tot_steps += dsum[i]; tot_work += dwork[i]; tot_iter += nsum[i]; tot_live += lsum[i]; }
    int S = 0;
    if (tot_iter > 0) {                                                        // block-uniform
        const int rounds = (tot_iter + A.slots - 1) / A.slots;
        const int budget = rounds * A.slots;
        int T = (int)(tot_steps / (double)budget) + 3;                         // lower bound: no strip can be shorter than that
        if (T < 6) T = 6;                                                      // at least 4 useful steps per strip
        for (int tries = 0; tries < 64; ++tries) {
            int R = L.ry * (T - 2) - 3;
            if (R < 1) R = 1;
            int Sb = iter ? (L.h + R - 1) / R : 0;
            if (Sb > L.smax && iter) Sb = L.smax;
            int sum = Sb;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = sum;
            __syncthreads();
            int tot = 0;
            for (int i = 0; i < nwv; ++i) tot += ssum[i];
            S = Sb;
            if (tot <= budget) break;
            // too many strips: lengthen them; jump by the ratio when far off, by one step when close
            const int Tn = 2 + (int)((double)(T - 2) * (double)tot / (double)budget);
            T = Tn > T ? Tn : T + 1;
        }
        if (iter) {
            const int R = (L.h + S - 1) / S;
            S = (L.h + R - 1) / R;
            s.strips = S; s.rows = R;
        }
    }
    int n_items = 0;
    const int off = sc_block_scan(S, lds16, &n_items);
    if (iter)
        for (int k = 0; k < S; ++k)
            if (off + k < A.iter_cap) A.iter_items[off + k] = b | (k << 16);
    // ---- tile work of the other phases ----
    int tiles = 0;
    if (on) {
        if (s.phase == PH_WARP) tiles = sc_tiles(L.w, L.h);
        else if (s.phase == PH_MEDIAN) tiles = 2 * sc_tiles(L.w, L.h);
        else if (s.phase == PH_UP) tiles = sc_tiles(A.lv[s.level - 1].w, A.lv[s.level - 1].h);
        else if (s.phase == PH_OUT) tiles = sc_tiles(L.w, L.h);
    }
    int n_mpairs = 0, n_mitems = 0;
    const int midx = sc_block_scan(tiles > 0 ? 1 : 0, lds16, &n_mpairs);
    const int moff = sc_block_scan(tiles, lds16, &n_mitems);
    if (tiles > 0) { A.misc_pair[midx] = b; A.misc_pref[midx] = moff; }
    if (on) A.st[b] = s;
    if (threadIdx.x == 0) {
        A.misc_pref[n_mpairs] = n_mitems;
        A.cnt[SC_N_ITER_ITEMS] = n_items < A.iter_cap ? n_items : A.iter_cap;
        A.cnt[SC_N_MISC_ITEMS] = n_mitems;
        A.cnt[SC_N_MISC_PAIRS] = n_mpairs;
        A.cnt[SC_N_NOT_DONE] = tot_live;
        if (A.work) A.work[A.ss] = tot_work;
        if (A.host_slot) __hip_atomic_store(A.host_slot, tot_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct MiscArgs {
    const PairSt* st; const int* cnt; const int* misc_pair; const int* misc_pref;
    StateBufs sb;
    const float* pyr[SC_MAXLEV];
    int off0, off1;
    const float* tab;
    float *wx, *wy, *rho;
    long long splane;
    float up_mul, out_scale;
    float* out;
    SchedLevel lv[SC_MAXLEV];
};

// 64 x 16-pixel tiles of the pairs in a warp / median / upsample / output phase; every block takes a contiguous run of
// tiles (the grid is fixed; runs are sized on the device from the item count k_sched published)
template <int KS>
__global__ __launch_bounds__(256) void k_misc_q(MiscArgs A)
{
    constexpr int MR = KS / 2;
    __shared__ __attribute__((aligned(16))) float t[16 + 2 * MR][64 + 2 * MR];
    __shared__ float stab[128];
    __shared__ int spref[1025];
    __shared__ short spair[1024];
    const int total = sc_uni(A.cnt[SC_N_MISC_ITEMS]), np = sc_uni(A.cnt[SC_N_MISC_PAIRS]);
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    int i = (int)blockIdx.x * chunk;
    const int i1 = i + chunk < total ? i + chunk : total;
    if (i >= i1) return;                                             // block-uniform
    if (threadIdx.x < 128) stab[threadIdx.x] = A.tab[threadIdx.x];
    for (int k = threadIdx.x; k <= np; k += 256) spref[k] = A.misc_pref[k];
    for (int k = threadIdx.x; k < np; k += 256) spair[k] = (short)A.misc_pair[k];
    __syncthreads();
    int j = 0;
    { int lo = 0, hi = np - 1; while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (spref[mid] <= i) lo = mid; else hi = mid - 1; } j = lo; }
    for (; i < i1; ++i) {
        while (i >= spref[j + 1]) ++j;
        j = sc_uni(j);
        const int b = sc_uni((int)spair[j]), li = sc_uni(i - spref[j]);
        const PairSt s = sc_uni(A.st[b]);
        const SchedLevel L = A.lv[s.level];
        const size_t po = (size_t)b * (size_t)A.splane;
        const int lx = threadIdx.x & 63, lr = threadIdx.x >> 6;
        if (s.phase == PH_WARP) {
            const int tw = (L.w + 63) >> 6, tx = li % tw, ty = li / tw;
            const float* I0 = A.pyr[s.level] + (size_t)(A.off0 + b) * (size_t)L.plane;
            const float* I1 = A.pyr[s.level] + (size_t)(A.off1 + b) * (size_t)L.plane;
            const int x = tx * 64 + lx;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int y = ty * 16 + r * 4 + lr;
                if (x < L.w && y < L.h)
                    warp_px(stab, I0, I1, A.sb.u1[s.ucur] + po, A.sb.u2[s.ucur] + po, A.wx + po, A.wy + po, A.rho + po, L.w, L.h, L.pitch, x, y);
            }
        } else if (s.phase == PH_MEDIAN) {
            if constexpr (KS > 1) {
                const int tw = (L.w + 63) >> 6, per = tw * ((L.h + 15) >> 4);
                const int plane = li / per, rem = li - plane * per, tx = rem % tw, ty = rem / tw;
                const float* src = (plane ? A.sb.u2[s.ucur] : A.sb.u1[s.ucur]) + po;
                float* dst = (plane ? A.sb.u2[s.ucur ^ 1] : A.sb.u1[s.ucur ^ 1]) + po;
                median_block<KS>(t, src, dst, tx * 64, ty * 16, L.w, L.h, L.pitch);
                __syncthreads();                                       // the tile is staged again by the next item
            }
        } else if (s.phase == PH_UP) {
            const SchedLevel D = A.lv[s.level - 1];
            const int tw = (D.w + 63) >> 6, tx = li % tw, ty = li / tw;
            const int dx = tx * 64 + lx;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int dy = ty * 16 + r * 4 + lr;
                if (dx < D.w && dy < D.h) {
                    const size_t di = po + (size_t)dy * D.pitch + dx;
                    A.sb.u1[s.ucur ^ 1][di] = resize_px(A.sb.u1[s.ucur] + po, L.w, L.h, L.pitch, dx, dy, L.up_sx, L.up_sy) * A.up_mul;
                    A.sb.u2[s.ucur ^ 1][di] = resize_px(A.sb.u2[s.ucur] + po, L.w, L.h, L.pitch, dx, dy, L.up_sx, L.up_sy) * A.up_mul;
                }
            }
        } else if (s.phase == PH_OUT) {
            const int tw = (L.w + 63) >> 6, tx = li % tw, ty = li / tw;
            const int x = tx * 64 + lx;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int y = ty * 16 + r * 4 + lr;
                if (x < L.w && y < L.h) {
                    const size_t idx = po + (size_t)y * L.pitch + x;
                    reinterpret_cast<float2*>(A.out)[((size_t)b * L.h + y) * L.w + x] =
                        make_float2(A.sb.u1[s.ucur][idx] * A.out_scale, A.sb.u2[s.ucur][idx] * A.out_scale);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct IterQArgs {
    IterArgs a;               // planes, parameters, error slots (a.g / a.it / a.utog... are set per block below)
    const PairSt* st; const int* cnt; const int* items;
    long long splane;
    SchedLevel lv[SC_MAXLEV];
};

// one (pair, strip) work item per block: the row march of k_iter2_rows with the pair's own level, iteration and mode
__global__ __launch_bounds__(512) void k_iter2_q(IterQArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= sc_uni(A.cnt[SC_N_ITER_ITEMS])) return;
    const int item = sc_uni(A.items[blockIdx.x]);
    const int b = item & 0xFFFF, strip = item >> 16;
    const PairSt s = sc_uni(A.st[b]);
    const SchedLevel L = A.lv[s.level];
    const IterArgs& a = A.a;
    Iter2Blk blk;
    blk.W = L.w; blk.H = L.h; blk.pitch = L.pitch; blk.splane = A.splane;
    blk.b = b; blk.strip = strip; blk.R = s.rows; blk.QX = L.qx; blk.RY = L.ry;
    blk.replay = s.phase == PH_REPLAY;
    blk.it = blk.replay ? s.it - 2 : s.it;
    blk.pzero = s.warp == 0 && blk.it == 0;
    blk.uc = blk.replay ? s.ucur ^ 1 : s.ucur;           // REPLAY re-reads what the overshooting launch read and overwrites what it wrote
    blk.pc = blk.replay ? s.pcur ^ 1 : s.pcur;
    blk.errb = a.err + (size_t)b * a.errstride;
    iter2_rows_body(a, blk, smem);
}
