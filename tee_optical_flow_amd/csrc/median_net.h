// median_net.h -- min/max selection networks for the 3x3 and 5x5 float median (cv::medianBlur on
// CV_32F uses the same idea; the median VALUE is unique, so any correct network gives upstream's
// result).  Shared between the HIP kernels and tests/csrc/verify_median_net.c, which proves both
// networks with the zero-one principle (all 2^9 / 2^25 binary inputs).
#ifndef TEEFLOW_MEDIAN_NET_H
#define TEEFLOW_MEDIAN_NET_H

#include <type_traits>

#ifndef TF_HD
#define TF_HD
#endif

#define TF_CE(a, b) { const T lo_ = tf_min(p[a], p[b]); const T hi_ = tf_max(p[a], p[b]); p[a] = lo_; p[b] = hi_; }

// On the GPU a float min/max is ONE v_min_f32 / v_max_f32 instead of compare + select (no NaNs reach the median and
// +-0 compare equal downstream); the generic form serves the host-side zero-one proof.
template <typename T> TF_HD inline T tf_min(T a, T b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fminf(a, b);
#endif
    return b < a ? b : a;
}
template <typename T> TF_HD inline T tf_max(T a, T b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fmaxf(a, b);
#endif
    return b < a ? a : b;
}

// 19 compare-exchanges; median lands in p[4]
template <typename T> TF_HD inline T tf_median9(T* p)
{
    TF_CE(1, 2) TF_CE(4, 5) TF_CE(7, 8) TF_CE(0, 1) TF_CE(3, 4) TF_CE(6, 7)
    TF_CE(1, 2) TF_CE(4, 5) TF_CE(7, 8) TF_CE(0, 3) TF_CE(5, 8) TF_CE(4, 7)
    TF_CE(3, 6) TF_CE(1, 4) TF_CE(2, 5) TF_CE(4, 7) TF_CE(4, 2) TF_CE(6, 4)
    TF_CE(4, 2)
    return p[4];
}

// 99 compare-exchanges; median lands in p[12]
template <typename T> TF_HD inline T tf_median25(T* p)
{
    TF_CE(0, 1) TF_CE(3, 4) TF_CE(2, 4) TF_CE(2, 3) TF_CE(6, 7) TF_CE(5, 7) TF_CE(5, 6) TF_CE(9, 10)
    TF_CE(8, 10) TF_CE(8, 9) TF_CE(12, 13) TF_CE(11, 13) TF_CE(11, 12) TF_CE(15, 16) TF_CE(14, 16)
    TF_CE(14, 15) TF_CE(18, 19) TF_CE(17, 19) TF_CE(17, 18) TF_CE(21, 22) TF_CE(20, 22) TF_CE(20, 21)
    TF_CE(23, 24) TF_CE(2, 5) TF_CE(3, 6) TF_CE(0, 6) TF_CE(0, 3) TF_CE(4, 7) TF_CE(1, 7) TF_CE(1, 4)
    TF_CE(11, 14) TF_CE(8, 14) TF_CE(8, 11) TF_CE(12, 15) TF_CE(9, 15) TF_CE(9, 12) TF_CE(13, 16)
    TF_CE(10, 16) TF_CE(10, 13) TF_CE(20, 23) TF_CE(17, 23) TF_CE(17, 20) TF_CE(21, 24) TF_CE(18, 24)
    TF_CE(18, 21) TF_CE(19, 22) TF_CE(8, 17) TF_CE(9, 18) TF_CE(0, 18) TF_CE(0, 9) TF_CE(10, 19)
    TF_CE(1, 19) TF_CE(1, 10) TF_CE(11, 20) TF_CE(2, 20) TF_CE(2, 11) TF_CE(12, 21) TF_CE(3, 21)
    TF_CE(3, 12) TF_CE(13, 22) TF_CE(4, 22) TF_CE(4, 13) TF_CE(14, 23) TF_CE(5, 23) TF_CE(5, 14)
    TF_CE(15, 24) TF_CE(6, 24) TF_CE(6, 15) TF_CE(7, 16) TF_CE(7, 19) TF_CE(13, 21) TF_CE(15, 23)
    TF_CE(7, 13) TF_CE(7, 15) TF_CE(1, 9) TF_CE(3, 11) TF_CE(5, 17) TF_CE(11, 17) TF_CE(9, 17)
    TF_CE(4, 10) TF_CE(6, 12) TF_CE(7, 14) TF_CE(4, 6) TF_CE(4, 7) TF_CE(12, 14) TF_CE(10, 14)
    TF_CE(6, 7) TF_CE(10, 12) TF_CE(6, 10) TF_CE(6, 17) TF_CE(12, 17) TF_CE(7, 17) TF_CE(7, 10)
    TF_CE(12, 18) TF_CE(7, 12) TF_CE(10, 18) TF_CE(12, 20) TF_CE(10, 20) TF_CE(10, 12)
    return p[12];
}


// ---- 5x5 median of FOUR horizontally adjacent outputs from their 5x8 window -------------------------------------------
// 306 min/max/med3 operations for four outputs (76 each) instead of 4 x 198:
//   1. the 8 columns are sorted once (each is used by up to four outputs);
//   2. neighbouring sorted columns are merged: M(1,2), M(3,4), M(5,6) (sorted 10-lists; M(3,4) serves both output pairs);
//   3. outputs (0,1) share columns 1..4 and outputs (2,3) columns 3..6.  Of such 20 shared values the 7 smallest have 13
//      larger ones next to them and can be the median of neither 25-window, likewise the 7 largest: only ranks 8..13
//      matter, taken from the two sorted 10-lists by one bitonic split and two partial sorts;
//   4. each output is then the 6th smallest of those six (sorted) and its own fifth column (sorted): min over i+j=6 of
//      max(Z_i, C_j).
// Everything is built from monotone operations, so the zero-one principle applies: tests/csrc/verify_median_net.cpp checks
// all 2^25 binary windows of every output.
template <typename T> TF_HD inline T tf_min3(T a, T b, T c) { return tf_min(tf_min(a, b), c); }
template <typename T> TF_HD inline T tf_max3(T a, T b, T c) { return tf_max(tf_max(a, b), c); }
template <typename T> TF_HD inline T tf_med3(T a, T b, T c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_amdgcn_fmed3f(a, b, c);
#endif
    return tf_max(tf_min(a, b), tf_min(tf_max(a, b), c));
}

// ascending sort of 5 values in 12 operations
template <typename T> TF_HD inline void tf_sort5(const T* v, T* s)
{
    const T a = tf_min3(v[0], v[1], v[2]), b = tf_med3(v[0], v[1], v[2]), c = tf_max3(v[0], v[1], v[2]);
    const T d = tf_min(v[3], v[4]), e = tf_max(v[3], v[4]);
    s[0] = tf_min(a, d); s[4] = tf_max(c, e);
    const T x = tf_max(a, d), z = tf_min(c, e);            // {x, b, z} are the three values in between
    s[1] = tf_min3(x, b, z); s[2] = tf_med3(x, b, z); s[3] = tf_max3(x, b, z);
}

#define TF_CE2(lo, hi) { const T l_ = tf_min(lo, hi); const T h_ = tf_max(lo, hi); lo = l_; hi = h_; }

// Batcher odd-even merge of two ascending 5-lists into an ascending 10-list (13 compare-exchanges)
template <typename T> TF_HD inline void tf_merge5(const T* a, const T* b, T* z)
{
    // odd-indexed elements (a0 a2 a4 | b0 b2 b4) -> c[0..5]
    T c[6], d[4];
    {
        // merge (a0, a4 | b0, b4) -> o[0..3];  (a2 | b2) -> e[0..1]
        T o0 = tf_min(a[0], b[0]), o1 = tf_max(a[0], b[0]);
        T o2 = tf_min(a[4], b[4]), o3 = tf_max(a[4], b[4]);
        TF_CE2(o1, o2)
        T e0 = tf_min(a[2], b[2]), e1 = tf_max(a[2], b[2]);
        c[0] = o0; c[5] = o3;
        c[1] = tf_min(e0, o1); c[2] = tf_max(e0, o1);
        c[3] = tf_min(e1, o2); c[4] = tf_max(e1, o2);
    }
    // even-indexed elements (a1 a3 | b1 b3) -> d[0..3]
    {
        d[0] = tf_min(a[1], b[1]); T t1 = tf_max(a[1], b[1]);
        T t2 = tf_min(a[3], b[3]); d[3] = tf_max(a[3], b[3]);
        d[1] = tf_min(t1, t2); d[2] = tf_max(t1, t2);
    }
    z[0] = c[0]; z[9] = c[5];
    z[1] = tf_min(d[0], c[1]); z[2] = tf_max(d[0], c[1]);
    z[3] = tf_min(d[1], c[2]); z[4] = tf_max(d[1], c[2]);
    z[5] = tf_min(d[2], c[3]); z[6] = tf_max(d[2], c[3]);
    z[7] = tf_min(d[3], c[4]); z[8] = tf_max(d[3], c[4]);
}

// ranks 8..13 (of 20, ascending, 1-based) of the union of two ascending 10-lists -> m[0..5]
template <typename T> TF_HD inline void tf_middle6(const T* x, const T* y, T* m)
{
    T lo[10], hi[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { lo[i] = tf_min(x[i], y[9 - i]); hi[i] = tf_max(x[i], y[9 - i]); }   // 10 smallest | 10 largest
    // lo is bitonic: the pairwise maxima at distance 5 hold its 5 largest; hi likewise with minima
    T t[5], u[5], s[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { t[i] = tf_max(lo[i], lo[i + 5]); u[i] = tf_min(hi[i], hi[i + 5]); }
    tf_sort5(t, s);
    m[0] = s[2]; m[1] = s[3]; m[2] = s[4];
    tf_sort5(u, s);
    m[3] = s[0]; m[4] = s[1]; m[5] = s[2];
}

// 6th smallest of an ascending 6-list z and an ascending 5-list c
template <typename T> TF_HD inline T tf_select6of11(const T* z, const T* c)
{
    const T t1 = tf_max(z[0], c[4]), t2 = tf_max(z[1], c[3]), t3 = tf_max(z[2], c[2]), t4 = tf_max(z[3], c[1]), t5 = tf_max(z[4], c[0]);
    return tf_min(tf_min3(t1, t2, t3), tf_min3(t4, t5, z[5]));
}

// col[j][r]: value at column j (0..7), row r (0..4) of the 5x8 window; out[o] = median of columns o..o+4
template <typename T> TF_HD inline void tf_median25_row4(const T (*col)[5], T* out)
{
    T s[8][5];
#pragma unroll
    for (int j = 0; j < 8; ++j) tf_sort5(col[j], s[j]);
    T m12[10], m34[10], m56[10], mid[6];
    tf_merge5(s[1], s[2], m12);
    tf_merge5(s[3], s[4], m34);
    tf_merge5(s[5], s[6], m56);
    tf_middle6(m12, m34, mid);
    out[0] = tf_select6of11(mid, s[0]);
    out[1] = tf_select6of11(mid, s[5]);
    tf_middle6(m34, m56, mid);
    out[2] = tf_select6of11(mid, s[2]);
    out[3] = tf_select6of11(mid, s[7]);
}

#endif
